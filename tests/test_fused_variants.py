"""The compiled width sets of the fused per-jet kernels beyond the two BASELINE ones (csrc/fused_common.h: DimsWide,
DimsNarrow, UniH64, UniH32) and unequal branch depths (num_blocks[0] != num_blocks[1]).

Freshly initialised models; the fused kernels (forward, single-launch generation, training forward + reverse) are
compared with (i) the CPU oracle, which is pinned against the reference goldens, and (ii) the layered HIP path: heads
rtol 1e-4, jump states bit-exact under injected Poisson counts, losses and every parameter gradient."""
import copy

import pytest
import torch

from conftest import GOLDEN, grads_close
from oracle import encoder as OE, model as OM

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-4, atol=2e-5):
    torch.testing.assert_close(a.detach().cpu().float(), b.detach().cpu().float(), rtol=rtol, atol=atol)


CASES = {
    # name: (golden whose inputs are reused, encoder overrides, expected variant id)
    "wide_2_1_2": ("mm_mid", dict(num_blocks=[2, 1, 2], dim_hidden_local=[64, 64, 128], dim_hidden_glob=[32, 32, 64],
                                  dim_emb_time=64, dim_emb_continuous=32, dim_emb_discrete=32), 2),
    "narrow_1_2_3": ("mm_mid", dict(num_blocks=[1, 2, 3], dim_hidden_local=[32, 32, 64], dim_hidden_glob=[16, 16, 32],
                                    dim_emb_time=32, dim_emb_continuous=16, dim_emb_discrete=16), 3),
    "s3_3_1_2": ("mm_mid", dict(num_blocks=[3, 1, 2], dim_hidden_local=[64, 64, 128], dim_hidden_glob=[16, 16, 32],
                                dim_emb_time=32, dim_emb_continuous=16, dim_emb_discrete=16), 0),
    "mid_1_3_2": ("mm_mid", dict(num_blocks=[1, 3, 2]), 1),
    "uni_h64": ("uni_mid", dict(num_blocks=2, dim_hidden_local=64, dim_hidden_glob=16, dim_emb_time=64,
                                dim_emb_discrete=64), 2),
    "uni_h32": ("uni_mid", dict(num_blocks=3, dim_hidden_local=32, dim_hidden_glob=16, dim_emb_time=32,
                                dim_emb_discrete=32), 3),
}


def build(golden, case, **model_updates):
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    from mmb_hip import fused
    gname, enc, variant = CASES[case]
    g = golden(gname)
    cfg = copy.deepcopy(g.cfg)
    cfg["data"]["target_path"] = GOLDEN
    cfg["encoder"].update(enc)
    cfg["model"].update(model_updates)
    torch.manual_seed(sum(map(ord, case)))
    model = MultiModalBridgeMatching(ExperimentConfigs(cfg)).cuda().eval()
    v = fused.eligible_uni(model) if gname.startswith("uni") else fused.eligible(model)
    assert v == variant, f"{case}: expected compiled variant {variant}, got {v}"
    return g, model


def state_of(g):
    from tensorclass import TensorMultiModal
    t = g.t
    c = lambda a: None if a is None else a.cuda()
    return TensorMultiModal(c(t["time"]), c(t.get("xt")), c(t["kt"]), c(t["mask"]))


@pytest.mark.parametrize("case", sorted(CASES))
def test_forward_vs_oracle_and_layered(golden, case):
    g, model = build(golden, case)
    t = g.t
    p = {n: v.detach().cpu() for n, v in model.state_dict().items()}
    with torch.no_grad():
        hc, hd = OE.forward(p, t["time"], t.get("xt"), t["kt"], t["mask"], g.flat["encoder_name"], g.flat["skip_connection"])
    model.use_fused = True
    hf = model(state_of(g), None)
    assert model._fused_cache[1] is not None, "the fused kernels did not serve this model"
    model.use_fused = False
    hl = model(state_of(g), None)
    for h in (hf, hl):
        if hc is not None:
            close(h.continuous, hc)
        close(h.discrete, hd)


@pytest.mark.parametrize("case", sorted(CASES))
def test_generation_injected_vs_oracle_and_layered(golden, case):
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    steps = 20
    g, model = build(golden, case, num_timesteps=steps)
    t = g.t
    tag = "gen10"
    gen = torch.Generator().manual_seed(5)
    B, N = t[tag + ":src_mask"].shape[:2]
    dn = torch.poisson(torch.full((steps, B, N, 8), 0.05), generator=gen)
    p = {n: v.detach().cpu() for n, v in model.state_dict().items()}
    source = {"continuous": t.get(tag + ":src_x"), "discrete": t[tag + ":src_k"], "mask": t[tag + ":src_mask"]}
    with torch.no_grad():
        want = OM.simulate_dynamics(p, dict(g.flat, num_timesteps=steps), source, {"delta_n": dn})
    c = lambda a: None if a is None else a.cuda()
    outs = {}
    for fz in (True, False):
        model.use_fused = fz
        src = TensorMultiModal(None, c(source["continuous"]), c(source["discrete"]), c(source["mask"]))
        state = TensorMultiModal(torch.zeros_like(src.mask), None if src.continuous is None else src.continuous.clone(),
                                 src.discrete.clone(), src.mask)
        paths = model.simulate_dynamics(state, DataCoupling(src, TensorMultiModal(None, None, None, src.mask), None),
                                        noise={"delta_n": dn.cuda()})
        if want["continuous"] is not None:
            close(paths.continuous[1], want["continuous"])
        assert torch.equal(paths.discrete[1].cpu(), want["discrete"]), f"{case}: fused={fz}"
        outs[fz] = paths


@pytest.mark.parametrize("case", sorted(CASES))
def test_training_step_vs_layered(golden, case):
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    g, model = build(golden, case)
    model.train()
    model.config.encoder.dropout = 0.1
    t = g.t
    c = lambda a: None if a is None else a.cuda()

    def batch():
        tgt = TensorMultiModal(None, c(t.get("x1")), c(t["k1"]), c(t["mask"]))
        src = TensorMultiModal(None, c(t.get("x0")), c(t["k0"]), torch.ones_like(c(t["mask"])))
        return DataCoupling(src, tgt, None)

    res = {}
    for fused in (True, False):
        model.use_fused_training = model.use_fused_backward = fused
        model.manual_seed(3)
        out = model.step_from_state(state_of(g), batch(), "train", True)
        model.zero_grad(set_to_none=True)
        out["loss"].backward()
        res[fused] = (out["loss"].detach().clone(), {n: q.grad.clone() for n, q in model.named_parameters()
                                                     if q.grad is not None})
    assert model._train_ws is not None, "the fused training step did not run"
    close(res[True][0], res[False][0], rtol=1e-5, atol=1e-6)
    grads_close(res[True][1], res[False][1], "variant_" + str(case), "fused_vs_layered")
