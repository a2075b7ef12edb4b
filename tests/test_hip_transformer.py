"""GPU parity of the transformer set encoder's own kernels (masked attention, LayerNorm; forward and reverse)
against a plain PyTorch fp32 statement of the same op -- the model-level comparison with the reference's goldens
(tr_tiny, tr_odd, S4) is in test_hip_model.py.  Through the C ABI (mmb_hip.ops -> ctypes)."""
import math

import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-4, atol=1e-5):
    torch.testing.assert_close(a.detach().cpu().float(), b.detach().cpu().float(), rtol=rtol, atol=atol)


def ref_attention(qkv, mask, nheads, valid_is_zero):
    """qkv (B,N,3H) float64 -> (B,N,H); keys restricted to mask == 0 (or != 0); no valid key -> zeros."""
    B, N, H3 = qkv.shape
    H = H3 // 3
    dh = H // nheads
    q, k, v = [z.reshape(B, N, nheads, dh).transpose(1, 2) for z in qkv.split(H, dim=-1)]
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    valid = (mask == 0) if valid_is_zero else (mask != 0)
    s = s.masked_fill(~valid[:, None, None, :], float("-inf"))
    m = s.max(-1, keepdim=True)[0]
    m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    e = torch.exp(s - m)
    den = e.sum(-1, keepdim=True)
    a = torch.where(den > 0, e / den.clamp(min=1e-300), torch.zeros_like(e))
    return (a @ v).transpose(1, 2).reshape(B, N, H)


def make_mask(B, N, g):
    n = torch.randint(0, N + 1, (B,), generator=g)
    n[0], n[1 % B] = N, 0                     # one jet without a padded slot, one without a real particle
    r = torch.rand(B, N, generator=g).argsort(1).argsort(1)
    return (r < n[:, None]).float()


@pytest.mark.parametrize("B,N,H,nh", [(5, 16, 8, 2), (3, 11, 12, 2), (4, 150, 64, 4), (3, 150, 128, 4), (2, 200, 32, 8),
                                      (2, 70, 128, 2)])
@pytest.mark.parametrize("valid_is_zero", [True, False])
@pytest.mark.parametrize("path", ["mfma", "valu"])
def test_attention_forward_backward(B, N, H, nh, valid_is_zero, path, monkeypatch):
    """path = mfma: the matrix-core kernels where they apply (head width <= 32, N <= 256), valu: the
    one-query-per-thread kernels for every shape (what head width 64 always runs)."""
    from mmb_hip import ops
    monkeypatch.setenv("MMB_ATTN_PATH", path)
    g = torch.Generator().manual_seed(B * 1000 + N + H)
    qkv = torch.randn(B, N, 3 * H, generator=g)
    mask = make_mask(B, N, g)
    dout = torch.randn(B, N, H, generator=g)
    q64 = qkv.double().requires_grad_(True)
    want = ref_attention(q64, mask, nh, valid_is_zero)
    (want * dout.double()).sum().backward()
    mi = ops.mask_prepare(mask.cuda())
    qd = qkv.reshape(B * N, 3 * H).cuda()
    out, lse = ops.attention_fwd(qd, mi, nh, valid_is_zero=valid_is_zero, want_lse=True)
    close(out.view(B, N, H), want, rtol=1e-4, atol=2e-5)
    dq = ops.attention_bwd(qd, mi, nh, out, dout.reshape(B * N, H).cuda().contiguous(), lse, valid_is_zero=valid_is_zero)
    scale = float(q64.grad.abs().max())
    close(dq.view(B, N, 3 * H), q64.grad, rtol=1e-4, atol=2e-5 * scale)
    # jets with no valid key: zero output, zero gradient
    nvalid = ((mask == 0) if valid_is_zero else (mask != 0)).sum(1)
    empty = nvalid == 0
    assert empty.any()
    assert float(out.view(B, N, H)[empty.cuda()].abs().max()) == 0.0
    assert float(dq.view(B, N, 3 * H)[empty.cuda()].abs().max()) == 0.0


def test_attention_rejects_bad_shapes():
    from mmb_hip import ops, MMBError
    mi = ops.mask_prepare(torch.ones(2, 8, device="cuda"))
    with pytest.raises(MMBError):
        ops.attention_fwd(torch.zeros(16, 3 * 10, device="cuda"), mi, 4)            # 10 % 4 != 0
    with pytest.raises(MMBError):
        ops.attention_fwd(torch.zeros(16, 3 * 512, device="cuda"), mi, 2)           # head width 256
    mi = ops.mask_prepare(torch.ones(1, 2000, device="cuda"))
    with pytest.raises(MMBError):
        ops.attention_fwd(torch.zeros(2000, 3 * 64, device="cuda"), mi, 1)          # LDS budget
    with pytest.raises(MMBError):
        ops.attention_fwd(torch.zeros(16, 24), mi, 2)                               # CPU tensor: no fallback


@pytest.mark.parametrize("M,D", [(7, 8), (1000, 64), (19200, 128), (33, 200), (5, 512), (301, 256), (4099, 64)])
def test_layernorm_forward_backward(M, D):
    from mmb_hip import ops
    g = torch.Generator().manual_seed(M + D)
    x = (torch.randn(M, D, generator=g) * 2 + 0.5)
    gamma, beta = torch.randn(D, generator=g), torch.randn(D, generator=g)
    dy = torch.randn(M, D, generator=g)
    x64 = x.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    want = torch.nn.functional.layer_norm(x64, (D,), g64, b64, 1e-5)
    (want * dy.double()).sum().backward()
    xd = x.cuda()
    y, stats = ops.layernorm_fwd(xd, gamma.cuda(), beta.cuda(), 1e-5, want_stats=True)
    close(y, want, rtol=1e-4, atol=2e-5)
    dx = torch.empty_like(xd)
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    ops.layernorm_bwd(xd, gamma.cuda(), stats, dy.cuda(), dx, False, dg, db)
    close(dx, x64.grad, rtol=1e-4, atol=2e-5 * float(x64.grad.abs().max()))
    close(dg, g64.grad, rtol=1e-4, atol=2e-5 * float(g64.grad.abs().max()) + 1e-6)
    close(db, b64.grad, rtol=1e-4, atol=2e-5 * float(b64.grad.abs().max()) + 1e-6)
    # accumulate mode and a strided (column-slice) output
    wide = torch.zeros(M, D + 8, device="cuda")
    ops.layernorm_fwd(xd, gamma.cuda(), beta.cuda(), 1e-5, out=wide[:, 4:4 + D])
    close(wide[:, 4:4 + D], want, rtol=1e-4, atol=2e-5)
    assert float(wide[:, :4].abs().max()) == 0.0 and float(wide[:, 4 + D:].abs().max()) == 0.0
    ops.layernorm_bwd(xd, gamma.cuda(), stats, dy.cuda(), dx, True, dg, db)
    close(dx, 2 * x64.grad, rtol=1e-4, atol=4e-5 * float(x64.grad.abs().max()))


def test_sum_replicas():
    from mmb_hip import ops
    src = torch.randn(8, 1000, device="cuda")
    wide = torch.randn(8, 1200, device="cuda")
    dst = torch.empty(1000, device="cuda")
    ops.sum_replicas(src, dst)
    close(dst, src.sum(0), rtol=1e-6, atol=1e-6)
    ops.sum_replicas(wide[:, :1000], dst, accumulate=True)
    close(dst, src.sum(0) + wide[:, :1000].sum(0), rtol=1e-6, atol=1e-5)


def test_reference_style_block_call(golden):
    """TransformerEncoderBlock.forward(t_emb, x, mask) with materialised (B,N,*) tensors, against the golden capture."""
    import copy
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    g = golden("tr_tiny")
    cfg = copy.deepcopy(g.cfg)
    cfg["data"]["target_path"] = GOLDEN
    model = MultiModalBridgeMatching(ExperimentConfigs(cfg))
    model.load_state_dict(g.params, strict=True)
    model = model.cuda().eval()
    t = g.t
    h = model.encoder.continuous_encoder(t["cap:emb.time"].cuda(), t["cap:emb.continuous"].cuda(), t["mask"].cuda())
    close(h, t["cap:continuous_encoder"], rtol=1e-4, atol=2e-5)


def test_transformer_refuses_dropout_in_training(golden):
    import copy
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    from tensorclass import TensorMultiModal
    g = golden("tr_tiny")
    cfg = copy.deepcopy(g.cfg)
    cfg["data"]["target_path"] = GOLDEN
    cfg["encoder"]["dropout"] = 0.1
    model = MultiModalBridgeMatching(ExperimentConfigs(cfg)).cuda().train()
    t = g.t
    state = TensorMultiModal(t["time"].cuda(), t["xt"].cuda(), t["kt"].cuda(), t["mask"].cuda())
    with pytest.raises(NotImplementedError):
        model(state, None)


@pytest.mark.parametrize("M,K,Nout", [(6000, 64, 64), (6000, 256, 64), (4100, 128, 128), (6000, 256, 128),
                                      (300, 64, 64), (6000, 64, 96), (5000, 48, 32)])
@pytest.mark.parametrize("path", ["x6", "tiled"])
def test_linear_cat_ln(M, K, Nout, path, monkeypatch):
    """mmb_linear_cat_ln = LayerNorm(act(x W^T + b) + residual) in one call: normalised in the epilogue of the
    row-streaming kernel where a wave owns whole rows (the first three shapes and the last under x6), by the linear
    kernel + mmb_layernorm_fwd otherwise (weight block beyond LDS, few rows, ragged width, MMB_LINEAR_PATH=tiled) --
    the same function either way, against fp64."""
    from mmb_hip import ops
    monkeypatch.setenv("MMB_LINEAR_PATH", path)
    g = torch.Generator().manual_seed(M + K + Nout)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(Nout, K, generator=g) / math.sqrt(K)
    b, gamma, beta = torch.randn(Nout, generator=g), torch.randn(Nout, generator=g), torch.randn(Nout, generator=g)
    res = torch.randn(M, Nout, generator=g)
    z = torch.nn.functional.linear(x.double(), W.double(), b.double()).relu() + res.double()
    want = torch.nn.functional.layer_norm(z, (Nout,), gamma.double(), beta.double(), 1e-5)
    segs = [(x.cuda(), ops.SEG_ROW)]
    y1, y, stats = ops.linear_cat_ln(segs, W.cuda(), b.cuda(), 50, gamma.cuda(), beta.cuda(), 1e-5, act="relu",
                                     residual=res.cuda(), want_y1=True, want_stats=True)
    close(y1, z, rtol=1e-4, atol=2e-5)
    close(y, want, rtol=1e-4, atol=5e-5)
    close(stats[:, 0], z.mean(1), rtol=1e-4, atol=1e-5)
    close(stats[:, 1], 1.0 / torch.sqrt(z.var(1, unbiased=False) + 1e-5), rtol=1e-4, atol=1e-5)
    # without y1 / stats, into a column slice of a wider buffer
    wide = torch.zeros(M, Nout + 8, device="cuda")
    n1, y2, s2 = ops.linear_cat_ln(segs, W.cuda(), b.cuda(), 50, gamma.cuda(), beta.cuda(), 1e-5, act="relu",
                                   residual=res.cuda(), out=wide[:, 4:4 + Nout])
    assert n1 is None and s2 is None
    close(wide[:, 4:4 + Nout], want, rtol=1e-4, atol=5e-5)
    assert float(wide[:, :4].abs().max()) == 0.0 and float(wide[:, 4 + Nout:].abs().max()) == 0.0
    # the backward's LayerNorm entry takes (y1, stats) as mmb_layernorm_fwd would have produced them
    ref_y, ref_stats = ops.layernorm_fwd(y1, gamma.cuda(), beta.cuda(), 1e-5, want_stats=True)
    close(y, ref_y, rtol=1e-5, atol=1e-5)
    close(stats, ref_stats, rtol=1e-5, atol=1e-6)
