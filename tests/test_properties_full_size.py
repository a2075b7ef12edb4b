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

from conftest import GOLDEN

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
