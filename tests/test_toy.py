"""S1 (BASELINE configs[0]): the toy hybrid experiment -- 8 Gaussians -> two moons, N = 1, Dc = 2, V = 8, the notebook's
small MLP (notebooks/data_toys.ipynb cell 3).

The golden (tests/golden/model_toy.npz, tools/make_golden.py: golden_toy) was produced by executing the notebook's
MultiModalMLP class inside the reference's MultiModalBridgeMatching (its bridges, loss_fn, MultiModeLoss, solvers and
simulate_dynamics).  CPU tests pin the oracle restatement against it; the gpu tests compare the HIP path (through the
C ABI) with it: heads and losses rtol 1e-4, every parameter gradient, 10- and 100-step generation with injected
Poisson counts (k bit-exact, x rtol 1e-4).  The toy samplers (NGaussians / TwoMoons) are checked at distribution level.
"""
import copy
import math

import pytest
import torch

from conftest import GOLDEN
from oracle import encoder as OE, model as OM, toys as OT


def close(a, b, rtol=1e-4, atol=1e-5):
    torch.testing.assert_close(a.detach().cpu().float(), b.detach().cpu().float(), rtol=rtol, atol=atol)


# ------------------------------------------------------------------ CPU: oracle vs the reference golden
def test_oracle_mlp_forward_loss_grads(golden):
    g = golden("toy")
    t = g.t
    p = {n: v.clone().requires_grad_(True) for n, v in g.params.items()}
    target = {"continuous": t["x1"], "discrete": t["k1"], "mask": t["mask"]}
    source = {"continuous": t["x0"], "discrete": t["k0"]}
    st = OM.sample_bridges(g.flat, target, source, {"t_u": t["t_u"], "z": t["z"], "u_k": t["u_k"].reshape(-1, 1)})
    close(st["time"], t["time"], rtol=1e-6, atol=1e-7)
    close(st["continuous"], t["xt"], rtol=1e-5, atol=1e-6)
    ok = t["u_k_margin"] > 1e-5
    assert torch.equal(st["discrete"].squeeze(-1)[ok], t["kt"].squeeze(-1)[ok])
    st["discrete"] = t["kt"]
    hc, hd = OE.forward(p, st["time"], st["continuous"], st["discrete"], st["mask"], "MultiModalMLP")
    close(hc, t["head_continuous"])
    close(hd, t["head_discrete"])
    lc, ld = OM.loss_fn(hc, hd, OM.drift_target(g.flat, st, source, target), t["k1"], st["mask"])
    loss = OM.multimode_loss([lc, ld], g.flat["loss_weights"])
    close(lc, t["loss_continuous"])
    close(ld, t["loss_discrete"])
    close(loss, t["loss"])
    loss.backward()
    for n, gref in g.grads().items():
        close(p[n].grad, gref, rtol=1e-4, atol=1e-6)


def test_oracle_generation(golden):
    g = golden("toy")
    t = g.t
    for steps in (10, 100):
        tag = f"gen{steps}"
        source = {"continuous": t[tag + ":src_x"], "discrete": t[tag + ":src_k"], "mask": t[tag + ":src_mask"]}
        with torch.no_grad():
            out = OM.simulate_dynamics(g.params, dict(g.flat, num_timesteps=steps), source,
                                       {"delta_n": t[tag + ":delta_n"]})
        close(out["continuous"], t[tag + ":final_x"], rtol=1e-4, atol=1e-5)
        assert torch.equal(out["discrete"], t[tag + ":final_k"])


def test_oracle_toy_distributions():
    g = torch.Generator().manual_seed(5)
    x, k = OT.n_gaussians(80000, generator=g)
    assert x.shape == (80000, 1, 2) and k.shape == (80000, 1, 1) and k.dtype == torch.int64
    for v in range(8):
        sel = x[k.view(-1) == v, 0]
        assert sel.shape[0] == 10000                                   # balanced blocks, toys.py:17
        a = 2 * math.pi * v / 8
        close(sel.mean(0), torch.tensor([5 * math.cos(a), 5 * math.sin(a)]), rtol=0, atol=0.03)
        close(sel.var(0), torch.full((2,), math.sqrt(0.1)), rtol=0.05, atol=0)
    x, k = OT.two_moons(20000, generator=g)
    assert set(k.view(-1).tolist()) == {0, 1} and int(k.sum()) == 10000


# ------------------------------------------------------------------ GPU: HIP path vs the reference golden
def build(g, **model_updates):
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    cfg = copy.deepcopy(g.cfg)
    cfg["data"]["target_path"] = GOLDEN
    cfg["model"].update(model_updates)
    model = MultiModalBridgeMatching(ExperimentConfigs(cfg))
    model.load_state_dict(g.params, strict=True)         # the notebook class's parameter names
    return model.cuda().eval()


def batch_of(g):
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    t = g.t
    c = lambda a: a.cuda()
    return DataCoupling(TensorMultiModal(None, c(t["x0"]), c(t["k0"]), c(t["mask"])),
                        TensorMultiModal(None, c(t["x1"]), c(t["k1"]), c(t["mask"])), None)


@pytest.mark.gpu
def test_hip_bridges_heads_losses(golden):
    from tensorclass import TensorMultiModal
    g = golden("toy")
    t = g.t
    model = build(g)
    st = model.sample_bridges(batch_of(g), noise={"t_u": t["t_u"].cuda(), "u_k": t["u_k"].cuda(), "z": t["z"].cuda()})
    close(st.time, t["time"], rtol=1e-6, atol=1e-7)
    close(st.continuous, t["xt"], rtol=1e-5, atol=1e-6)
    ok = t["u_k_margin"] > 1e-5
    assert torch.equal(st.discrete.cpu().squeeze(-1)[ok], t["kt"].squeeze(-1)[ok])
    c = lambda a: a.cuda()
    state = TensorMultiModal(c(t["time"]), c(t["xt"]), c(t["kt"]), c(t["mask"]))
    batch = batch_of(g)
    heads = model(state, batch)
    assert heads.continuous.shape == (t["mask"].shape[0], 1, 2) and heads.discrete.shape == (t["mask"].shape[0], 1, 8)
    close(heads.continuous, t["head_continuous"], rtol=1e-4, atol=2e-5)
    close(heads.discrete, t["head_discrete"], rtol=1e-4, atol=2e-5)
    lc, ld = model.loss_fn(heads, state, batch)
    close(lc, t["loss_continuous"], rtol=1e-4, atol=1e-6)
    close(ld, t["loss_discrete"], rtol=1e-4, atol=1e-6)
    close(model.loss_multimode([lc, ld]), t["loss"], rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_hip_training_step_gradients(golden, margins):
    from tensorclass import TensorMultiModal
    g = golden("toy")
    t = g.t
    model = build(g).train()
    c = lambda a: a.cuda()
    state = TensorMultiModal(c(t["time"]), c(t["xt"]), c(t["kt"]), c(t["mask"]))
    out = model.step_from_state(state, batch_of(g), "train", True)
    close(out["loss"], t["loss"], rtol=1e-4, atol=1e-6)
    model.zero_grad(set_to_none=True)
    out["loss"].backward()
    params = dict(model.named_parameters())
    assert set(params) == set(g.grads())
    from conftest import grads_close, GRAD_TOL
    assert all(params[n].grad is not None for n in params)
    grads_close({n: params[n].grad for n in params}, g.grads(), "toy", "layered")
    # a second backward without zeroing accumulates (torch.autograd semantics)
    out2 = model.step_from_state(state, batch_of(g), "train", True)
    out2["loss"].backward()
    n = "encoder.body.2.weight"
    grads_close({n: params[n].grad}, {n: 2 * g.grads()[n]}, "toy", "layered_accumulated")


@pytest.mark.gpu
@pytest.mark.parametrize("steps", [10, 100])
def test_hip_generation_injected(golden, steps):
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    g = golden("toy")
    t = g.t
    tag = f"gen{steps}"
    model = build(g, num_timesteps=steps)
    c = lambda a: a.cuda()
    src = TensorMultiModal(None, c(t[tag + ":src_x"]), c(t[tag + ":src_k"]), c(t[tag + ":src_mask"]))
    batch = DataCoupling(src, TensorMultiModal(None, None, None, src.mask), None)
    state = TensorMultiModal(torch.zeros_like(src.mask), src.continuous.clone(), src.discrete.clone(), src.mask)
    paths = model.simulate_dynamics(state, batch, noise={"delta_n": t[tag + ":delta_n"].cuda()})
    close(paths.continuous[1], t[tag + ":final_x"], rtol=1e-4, atol=2e-5)
    assert torch.equal(paths.discrete[1].cpu(), t[tag + ":final_k"])


@pytest.mark.gpu
def test_hip_toy_samplers_and_predict_step(golden):
    """NGaussians / TwoMoons on device: the laws of distributions/toys.py (moments per label), and a 100-step
    predict_step from a sampled source with in-kernel noise."""
    from distributions.toys import NGaussians, TwoMoons
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    s = NGaussians(dim=2, vocab_size=8)(80003, device="cuda")
    assert s.continuous.shape == (80000, 1, 2) and s.discrete.shape == (80000, 1, 1) and s.discrete.dtype == torch.int64
    x, k = s.continuous.cpu()[:, 0], s.discrete.cpu().view(-1)
    assert not torch.equal(k[:8], torch.arange(8))                     # shuffled (toys.py:39)
    for v in range(8):
        sel = x[k == v]
        assert sel.shape[0] == 10000
        a = 2 * math.pi * v / 8
        close(sel.mean(0), torch.tensor([5 * math.cos(a), 5 * math.sin(a)]), rtol=0, atol=0.03)
        close(sel.var(0), torch.full((2,), math.sqrt(0.1)), rtol=0.05, atol=0)
    m = TwoMoons(dim=2)(20000, device="cuda")
    xm, km = m.continuous.cpu()[:, 0], m.discrete.cpu().view(-1)
    assert int(km.sum()) == 10000 and set(km.tolist()) == {0, 1}
    gen = torch.Generator().manual_seed(3)
    xo, ko = OT.two_moons(20000, generator=gen)
    for lab in (0, 1):                                                 # same law as the oracle restatement
        a, b = xm[km == lab], xo[:, 0][ko.view(-1) == lab]
        close(a.mean(0), b.mean(0), rtol=0, atol=0.05)
        close(a.var(0), b.var(0), rtol=0.06, atol=0)
    g = golden("toy")
    model = build(g, num_timesteps=100)
    B = 256
    tgt = TwoMoons(dim=2)(B, device="cuda")
    paths, target = model.predict_step(DataCoupling(TensorMultiModal(), tgt, None), 0)
    assert paths.continuous.shape == (2, B, 1, 2) and paths.discrete.shape == (2, B, 1, 1)
    assert torch.isfinite(paths.continuous).all() and int(paths.discrete.min()) >= 0 and int(paths.discrete.max()) < 8


@pytest.mark.gpu
@pytest.mark.parametrize("solver,bridge", [("euler", "UniformFlow"), ("euler_maruyama", "SchrodingerBridge")])
def test_hip_single_launch_generation_matches_layered(golden, solver, bridge):
    """mmb_toy_generate (the whole loop in one launch, activations in registers) against the layered path (16 launches
    per grid point) on the same injected noise: a ragged batch of 1000 points (not a multiple of the 32-point wave
    tile), Euler and Euler-Maruyama with a diffusion term; tokens bit-exact, x rtol 1e-4."""
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    from distributions.toys import NGaussians
    g = golden("toy")
    steps, B = 40, 1000
    model = build(g, num_timesteps=steps, solver_continuous=solver, bridge_continuous=bridge, sigma=0.3)
    gen = torch.Generator().manual_seed(11)
    src = NGaussians(dim=2, vocab_size=8)(B, device="cuda")
    dn = torch.poisson(torch.full((steps, B, 1, 8), 0.02), generator=gen).cuda()
    dw = torch.randn(steps, B, 1, 2, generator=gen).cuda() * math.sqrt(1.0 / steps)
    noise = {"delta_n": dn, "delta_w": dw}
    outs = []
    for fused in (True, False):
        model.use_fused = fused
        s = TensorMultiModal(None, src.continuous.clone(), src.discrete.clone(), src.mask)
        batch = DataCoupling(s, TensorMultiModal(None, None, None, src.mask), None)
        state = TensorMultiModal(torch.zeros_like(src.mask), s.continuous.clone(), s.discrete.clone(), s.mask)
        outs.append(model.simulate_dynamics(state, batch, noise=noise))
    assert outs[0].continuous.shape == outs[1].continuous.shape == (2, B, 1, 2)
    assert float((outs[0].continuous[1] - outs[0].continuous[0]).abs().max()) > 0.1          # it moved
    close(outs[0].continuous[1], outs[1].continuous[1], rtol=1e-4, atol=5e-5)
    same = (outs[0].discrete[1] == outs[1].discrete[1]).float().mean().item()
    assert same >= 0.999, same          # (the final argmax of near-tied rates may differ in a handful of points)
    close(outs[0].continuous[0], outs[1].continuous[0], rtol=0, atol=0)                      # the source is untouched
