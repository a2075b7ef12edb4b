"""Size-independent properties of the generation hot path at BASELINE's full size (S3: B=1024 jets x
N=128 particles, the fused single-launch loop), where the oracle is too slow to serve as the checker:

* jet independence  -- a jet's result does not depend on which other jets share the launch;
* permutation equivariance of the EPiC encoder + solvers -- permuting the slots of every jet (state,
  mask and injected noise alike) permutes the result;
* padded-slot isolation -- the state of padded slots never leaks into real particles;
* determinism.

The jump noise is injected (Poisson counts, SURVEY 8c protocol), so the only tolerance is the fp32
order of the pooled sums (x: rtol 1e-4; k: identical except at near-ties of the final argmax)."""
import copy

import pytest
import torch

from conftest import GOLDEN, grads_close, GRAD_TOL_FULL_SIZE, GRAD_TOL_RERUN

pytestmark = pytest.mark.gpu

B, N, STEPS = 1024, 128, 12


@pytest.fixture(scope="module")
def setup(golden):
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    g = golden("S3")
    cfg = copy.deepcopy(g.cfg)
    cfg["data"]["target_path"] = GOLDEN
    cfg["model"]["num_timesteps"] = STEPS
    model = MultiModalBridgeMatching(ExperimentConfigs(cfg))
    model.load_state_dict(g.params, strict=True)
    model = model.cuda().eval()
    gen = torch.Generator().manual_seed(2024)
    n = torch.randint(1, N + 1, (B,), generator=gen)
    rank = torch.rand(B, N, generator=gen).argsort(1).argsort(1)
    mask = (rank < n[:, None]).long().unsqueeze(-1)
    x0 = torch.randn(B, N, 3, generator=gen)
    k0 = torch.randint(0, 8, (B, N, 1), generator=gen)
    dn = torch.poisson(torch.full((STEPS, B, N, 8), 0.05), generator=gen)
    return model, mask.cuda(), x0.cuda(), k0.cuda(), dn.cuda()


def run(model, mask, x0, k0, dn):
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    src = TensorMultiModal(None, x0.clone(), k0.clone(), mask)
    state = TensorMultiModal(torch.zeros_like(mask), x0.clone(), k0.clone(), mask)
    batch = DataCoupling(src, TensorMultiModal(None, None, None, mask), None)
    paths = model.simulate_dynamics(state, batch, noise={"delta_n": dn})
    return paths.continuous[1], paths.discrete[1]


def same(xa, ka, xb, kb, k_floor=0.9995):
    torch.testing.assert_close(xa, xb, rtol=1e-4, atol=2e-5)
    agree = (ka == kb).float().mean().item()
    assert agree >= k_floor, f"final-k agreement {agree}"


def test_full_size_properties(setup):
    model, mask, x0, k0, dn = setup
    assert model._fused_pack() is not None, "the full-size run must go through the fused HIP kernel"
    x, k = run(model, mask, x0, k0, dn)
    assert torch.isfinite(x).all() and int(k.min()) >= 0 and int(k.max()) < 8
    # determinism: bit-identical
    x2, k2 = run(model, mask, x0, k0, dn)
    assert torch.equal(x, x2) and torch.equal(k, k2)
    # jet independence: 16 jets alone == the same 16 jets inside the 1024-jet launch (bit-identical: per-jet workgroups)
    idx = torch.arange(0, B, 64, device="cuda")
    xs, ks = run(model, mask[idx], x0[idx], k0[idx], dn[:, idx])
    assert torch.equal(xs, x[idx]) and torch.equal(ks, k[idx])
    # permutation equivariance
    perm = torch.rand(B, N, device="cuda").argsort(1)
    gather = lambda t: torch.gather(t, 1, perm[:, :, None].expand(-1, -1, t.shape[-1]))
    dn_p = torch.gather(dn, 2, perm[None, :, :, None].expand(STEPS, -1, -1, 8))
    xp, kp = run(model, gather(mask), gather(x0), gather(k0), dn_p)
    same(xp, kp, gather(x), gather(k))
    # padded-slot isolation: scramble the state of padded slots -> real particles unchanged (bit-identical)
    pad = mask[..., 0] == 0
    x0b, k0b = x0.clone(), k0.clone()
    x0b[pad] = 37.0 * torch.randn_like(x0b[pad])
    k0b[pad] = (k0b[pad] + 3) % 8
    xb, kb = run(model, mask, x0b, k0b, dn)
    real = ~pad
    assert torch.equal(xb[real], x[real]) and torch.equal(kb[real], k[real])


def test_full_size_training_forward_jet_independence(setup):
    """Layered (taped) training forward at B=1024: heads of a jet do not depend on the rest of the batch (a 16-jet
    batch goes through the small-M kernels, whose k order differs: fp32 rounding only)."""
    from tensorclass import TensorMultiModal
    model, mask, x0, k0, _ = setup
    model.use_fused = False
    try:
        t = torch.rand(B, device="cuda") * 0.98 + 0.01
        st = TensorMultiModal(t.reshape(B, 1, 1), x0, k0, mask)
        full = model(st, None)
        idx = torch.arange(0, B, 64, device="cuda")
        sub = model(TensorMultiModal(t[idx].reshape(-1, 1, 1), x0[idx], k0[idx], mask[idx]), None)
        torch.testing.assert_close(sub.continuous, full.continuous[idx], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(sub.discrete, full.discrete[idx], rtol=1e-4, atol=1e-5)
        # fused forward agrees with the layered one at full size
        model.use_fused = True
        fz = model(st, None)
        torch.testing.assert_close(fz.continuous, full.continuous, rtol=1e-4, atol=2e-5)
        torch.testing.assert_close(fz.discrete, full.discrete, rtol=1e-4, atol=2e-5)
    finally:
        model.use_fused = True


def test_full_size_fused_training_step(setup):
    """Full-size (B=1024, N=128, ragged multiplicities, dropout 0.2) training step: the fused step (one forward
    launch, one reverse launch, multi-problem weight gradients) against the layered taped step with the same
    Philox streams -- loss and every parameter gradient; then again on the reused persistent workspace, and once
    more after a half-size step in between (a step must leave no trace in the next one)."""
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    model, mask, x0, k0, _ = setup
    gen = torch.Generator().manual_seed(99)
    x1 = torch.randn(B, N, 3, generator=gen).cuda() * mask
    k1 = torch.randint(0, 8, (B, N, 1), generator=gen).cuda() * mask
    t = (torch.rand(B, generator=gen) * 0.98 + 0.01).cuda()
    model.train()
    old_p = model.config.encoder.dropout
    model.config.encoder.dropout = 0.2

    def step(fused, xs=x0, ks=k0, ms=mask, x1s=x1, k1s=k1, ts=t):
        model.use_fused_training = model.use_fused_backward = fused
        model.manual_seed(5)
        st = TensorMultiModal(ts.reshape(-1, 1, 1), xs, ks, ms)
        src = TensorMultiModal(None, torch.zeros_like(xs), torch.zeros_like(ks), torch.ones_like(ms))
        batch = DataCoupling(src, TensorMultiModal(None, x1s, k1s, ms), None)
        out = model.step_from_state(st, batch, "train", True)
        model.zero_grad(set_to_none=True)
        out["loss"].backward()
        return out["loss"].detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    def check(ga, gb, tol=None):
        grads_close(ga, gb, "S3_full_size", "fused_vs_layered" if tol is None else "fused_rerun",
                    tol=GRAD_TOL_FULL_SIZE if tol is None else tol)

    try:
        lf, gf = step(True)
        ll, gl = step(False)
        torch.testing.assert_close(lf, ll, rtol=1e-5, atol=1e-6)
        check(gf, gl)
        lf2, gf2 = step(True)                      # same inputs on the reused workspace: bit-identical heads -> same loss
        assert torch.equal(lf, lf2)
        check(gf2, gf, tol=GRAD_TOL_RERUN)                  # (atomics in the weight gradients: order-dependent rounding only)
        # half the batch, then the full batch again: a smaller step in between must leave no trace
        h = B // 2
        step(True, x0[:h], k0[:h], mask[:h], x1[:h], k1[:h], t[:h])
        lf3, gf3 = step(True)
        assert torch.equal(lf, lf3)
        check(gf3, gf, tol=GRAD_TOL_RERUN)
    finally:
        model.config.encoder.dropout = old_p
        model.use_fused_training = model.use_fused_backward = True
        model.eval()
