"""The fused kernels evaluate fp32 GEMMs as six bf16 x bf16 partial products of a three-plane operand split
(csrc/fused_common.h: mma_x; host side mmb_hip.fused.split_planes).  CPU properties of that arithmetic: the split
loses no bit of an fp32 value, the packed planes are laid out [fragment][plane][lane][8], and the six products the
kernel keeps reproduce the exact product to within 2^-24 (below one fp32 rounding)."""
import torch


def _planes(x):
    from mmb_hip.fused import split_planes
    F = x.numel() // 512
    p = split_planes(x.view(F, 512)).view(torch.bfloat16).view(F, 3, 512)
    return p[:, 0].reshape(-1), p[:, 1].reshape(-1), p[:, 2].reshape(-1)


def _values(n, seed):
    g = torch.Generator().manual_seed(seed)
    mag = 10.0 ** torch.empty(n).uniform_(-8, 6, generator=g)
    return torch.randn(n, generator=g) * mag


def test_three_planes_hold_every_bit():
    x = _values(512 * 64, 0)
    x[:8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 1.0 + 2 ** -23, 3.0 - 2 ** -22, 2.0 ** -60, -(2.0 ** 40)])
    hi, mid, lo = _planes(x)
    back = hi.double() + mid.double() + lo.double()
    assert torch.equal(back, x.double())
    assert torch.equal((hi.float() + mid.float()) + lo.float(), x)          # ... and in fp32, largest first


def test_six_partial_products_are_an_fp32_product():
    w, x = _values(512 * 64, 1), _values(512 * 64, 2)
    wh, wm, wl = (p.double() for p in _planes(w))
    xh, xm, xl = (p.double() for p in _planes(x))
    kept = wl * xh + wm * xm + wh * xl + wm * xh + wh * xm + wh * xh         # the order mma_x accumulates in
    exact = w.double() * x.double()
    rel = ((kept - exact).abs() / exact.abs().clamp_min(1e-300)).max().item()
    assert rel <= 2.0 ** -24, rel


def test_pack_x_covers_the_matrix_once():
    """fragment (ft, jj, ot), lane (hk, i), slot (qq, e) <-> W[32 ot + i][32 ft + 16 jj + 8 qq + 4 hk + e]"""
    from mmb_hip.fused import _Packer
    O, K = 40, 70
    W = torch.arange(O * K, dtype=torch.float32).view(O, K) + 1.0
    u = _Packer.pack_x(W)
    OT, KT = 2, 3
    assert u.shape == (KT * 2 * OT, 512)
    v = u.view(KT, 2, OT, 2, 32, 2, 4)                                       # (ft, jj, ot, hk, i, qq, e)
    for (ft, jj, ot, hk, i, qq, e) in [(0, 0, 0, 0, 0, 0, 0), (2, 0, 1, 1, 7, 1, 3), (1, 1, 0, 0, 31, 1, 2), (2, 1, 1, 1, 31, 1, 3)]:
        o, k = 32 * ot + i, 32 * ft + 16 * jj + 8 * qq + 4 * hk + e
        want = W[o, k].item() if (o < O and k < K) else 0.0
        assert v[ft, jj, ot, hk, i, qq, e].item() == want
    assert sorted(u[u != 0].tolist()) == sorted(W.reshape(-1).tolist())
