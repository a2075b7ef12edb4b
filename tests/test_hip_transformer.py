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


def ref_attention(qkv, mask, nheads, valid_is_zero, keep=None):
    """qkv (B,N,3H) float64 -> (B,N,H); keys restricted to mask == 0 (or != 0); no valid key -> zeros.
    keep (B,nheads,N,N) | None: dropout factors on the attention probabilities."""
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
    if keep is not None:
        a = a * keep
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


@pytest.mark.parametrize("B,N,H,nh", [(5, 16, 8, 2), (3, 11, 12, 2), (4, 150, 64, 4), (3, 150, 128, 4), (2, 70, 128, 2)])
@pytest.mark.parametrize("path", ["mfma", "valu"])
def test_attention_dropout_forward_backward(B, N, H, nh, path, monkeypatch):
    """Dropout on the attention probabilities (training mode of nn.MultiheadAttention inside
    nn.TransformerEncoderLayer(dropout=p), particle_transformer.py:29-35): the Philox keep factors are extracted
    (mmb_attention_dropout_mask) and the dropped attention rebuilt in float64 with autograd -- forward, log-sum-exp
    and all three input gradients; the factors are 0 or 1/(1-p), kept at the rate 1-p, and change with the offset."""
    from mmb_hip import ops
    monkeypatch.setenv("MMB_ATTN_PATH", path)
    p, seed, off = 0.3, 1234567, 5
    g = torch.Generator().manual_seed(B * 1000 + N + H + 1)
    qkv = torch.randn(B, N, 3 * H, generator=g)
    mask = make_mask(B, N, g)
    dout = torch.randn(B, N, H, generator=g)
    mi = ops.mask_prepare(mask.cuda())
    qd = qkv.reshape(B * N, 3 * H).cuda()
    keep = ops.attention_dropout_mask(mi, nh, p, seed, off).cpu()
    valid = (mask == 0)[:, None, None, :].expand(B, nh, N, N)
    assert float(keep[~valid].abs().max() if (~valid).any() else 0.0) == 0.0
    kv = keep[valid]
    assert bool(((kv == 0) | ((kv - 1.0 / (1.0 - p)).abs() < 1e-6)).all())
    n = kv.numel()
    if n > 2000:
        rate = float((kv != 0).float().mean())
        assert abs(rate - (1.0 - p)) < 4.0 * math.sqrt(p * (1.0 - p) / n) + 1e-3, rate
    other = ops.attention_dropout_mask(mi, nh, p, seed, off + 1).cpu()
    if n > 200:
        assert not torch.equal(other, keep)
    q64 = qkv.double().requires_grad_(True)
    want = ref_attention(q64, mask, nh, True, keep=keep.double())
    (want * dout.double()).sum().backward()
    out, lse, bits = ops.attention_fwd(qd, mi, nh, valid_is_zero=True, want_lse=True, drop_p=p, seed=seed, offset=off)
    out0, lse0 = ops.attention_fwd(qd, mi, nh, valid_is_zero=True, want_lse=True)
    scale = float(want.detach().abs().max())
    close(out.view(B, N, H), want, rtol=1e-4, atol=2e-5 * scale)
    assert torch.equal(lse, lse0)                              # the normaliser is that of the undropped scores
    out_again = ops.attention_fwd(qd, mi, nh, valid_is_zero=True, drop_p=p, seed=seed, offset=off)
    assert torch.equal(out_again, out)
    dq = ops.attention_bwd(qd, mi, nh, out, dout.reshape(B * N, H).cuda(), lse, valid_is_zero=True, drop_p=p, seed=seed,
                           offset=off)
    gs = float(q64.grad.abs().max())
    close(dq.view(B, N, 3 * H), q64.grad, rtol=1e-4, atol=2e-5 * gs)
    # the reverse pass reading the forward's stored keep bits (matrix-core kernels) instead of regenerating them
    dq2 = ops.attention_bwd(qd, mi, nh, out, dout.reshape(B * N, H).cuda(), lse, valid_is_zero=True, drop_p=p, seed=seed,
                            offset=off, keep_bits=bits)
    assert torch.equal(dq2, dq)


def _dropout_factors(M, Nout, p, seed, off):
    """keep factors of the second-stage dropout of mmb_linear_cat for an (M, Nout) output: run it on a layer whose
    output is 1 everywhere"""
    from mmb_hip import ops
    x = torch.zeros(M, 4, device="cuda")
    W = torch.zeros(Nout, 4, device="cuda")
    b = torch.ones(Nout, device="cuda")
    _, y2 = ops.linear_cat([(x, ops.SEG_ROW)], W, b, 1, want_y1=False, out2=torch.empty(M, Nout, device="cuda"),
                           drop_p=p, seed=seed, offset=off)
    return y2


@pytest.mark.parametrize("name", ["tr_tiny", "tr_odd"])
def test_transformer_block_with_dropout_matches_float64_with_extracted_masks(golden, name):
    """One TransformerEncoderBlock in training mode with dropout 0.25: the four Philox masks of every layer (attention
    probabilities, dropout1 after out_proj, the FFN's inner dropout, dropout2 after linear2) are extracted and
    nn.TransformerEncoderLayer's arithmetic restated in float64 with them."""
    import copy
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    from encoders.epic import ExecCtx
    from mmb_hip import ops
    F = torch.nn.functional
    g = golden(name)
    cfg = copy.deepcopy(g.cfg)
    cfg["data"]["target_path"] = GOLDEN
    model = MultiModalBridgeMatching(ExperimentConfigs(cfg))
    model.load_state_dict(g.params, strict=True)
    model = model.cuda().train()
    blk = model.encoder.continuous_encoder
    t = g.t
    mask = t["mask"]
    B, N = mask.shape[:2]
    p, seed = 0.25, 4242
    mi = ops.mask_prepare(mask.cuda())
    ctx = ExecCtx(mi, tape=None, drop_p=p, seed=seed, offset0=0)
    temb = t["cap:emb.time"].cuda().float().contiguous()                   # (B,N,Et), already masked
    x = t["cap:emb.continuous"].cuda().float().reshape(B * N, -1).contiguous()
    got = blk.run(ctx, temb, [(x, ops.SEG_ROW)]).view(B, N, -1)
    # float64 restatement
    d = lambda a: a.detach().double().cpu()
    h = F.linear(d(x), d(blk.input_proj.weight), d(blk.input_proj.bias))
    tp = blk.time_proj
    te = d(temb).reshape(B * N, -1)
    h = h + (F.linear(te, d(tp.weight), d(tp.bias)) if isinstance(tp, torch.nn.Linear) else te)
    D = h.shape[1]
    for i, l in enumerate(blk.transformer.layers):
        o = 4 * i
        qkv = F.linear(h, d(l.self_attn.in_proj_weight), d(l.self_attn.in_proj_bias))
        keep = ops.attention_dropout_mask(mi, blk.n_heads, p, seed, o + 1).double().cpu()
        a = ref_attention(qkv.view(B, N, -1), mask.squeeze(-1), blk.n_heads, True, keep=keep).reshape(B * N, D)
        m1 = _dropout_factors(B * N, D, p, seed, o + 2).double().cpu()
        h = F.layer_norm(h + F.linear(a, d(l.self_attn.out_proj.weight), d(l.self_attn.out_proj.bias)) * m1, (D,),
                         d(l.norm1.weight), d(l.norm1.bias), l.norm1.eps)
        ff = l.linear1.out_features
        m2 = _dropout_factors(B * N, ff, p, seed, o + 3).double().cpu()
        f = F.linear(h, d(l.linear1.weight), d(l.linear1.bias)).relu() * m2
        m3 = _dropout_factors(B * N, D, p, seed, o + 4).double().cpu()
        h = F.layer_norm(h + F.linear(f, d(l.linear2.weight), d(l.linear2.bias)) * m3, (D,), d(l.norm2.weight),
                         d(l.norm2.bias), l.norm2.eps)
        for m in (m1, m2, m3):
            assert bool(((m == 0) | ((m - 1.0 / (1.0 - p)).abs() < 1e-6)).all())
    close(got, h.view(B, N, D), rtol=1e-4, atol=2e-5)
    # eval mode: no dropout, the golden capture
    model.eval()
    h_eval = blk(t["cap:emb.time"].cuda(), t["cap:emb.continuous"].cuda(), t["mask"].cuda())
    close(h_eval, t["cap:continuous_encoder"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("name", ["tr_tiny", "S4"])
def test_transformer_training_step_with_dropout(golden, name):
    """The reference's S4 block trains with dropout 0.2.  Training steps with dropout: deterministic for a seed,
    different for another, and the reverse pass (which regenerates every mask) agrees with central finite differences
    of the loss along the gradient direction, the masks being the same at both points."""
    import copy
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    g = golden(name)
    cfg = copy.deepcopy(g.cfg)
    cfg["data"]["target_path"] = GOLDEN
    cfg["encoder"]["dropout"] = 0.2
    model = MultiModalBridgeMatching(ExperimentConfigs(cfg))
    model.load_state_dict(g.params, strict=True)
    model = model.cuda().train()
    t = g.t
    c = lambda a: None if a is None else a.cuda()

    def loss_and_grads(seed, want_grads=True):
        model.manual_seed(seed)
        state = TensorMultiModal(c(t["time"]), c(t.get("xt")), c(t["kt"]), c(t["mask"]))
        src = TensorMultiModal(None, c(t.get("x0")), c(t["k0"]), torch.ones_like(c(t["mask"])))
        batch = DataCoupling(src, TensorMultiModal(None, c(t.get("x1")), c(t["k1"]), c(t["mask"])), None)
        out = model.step_from_state(state, batch, "train", True)
        if not want_grads:
            return out["loss"].detach().double().item(), None
        model.zero_grad(set_to_none=True)
        out["loss"].backward()
        return out["loss"].detach().double().item(), {n: q.grad.detach().clone() for n, q in model.named_parameters()
                                                      if q.grad is not None}

    l1, g1 = loss_and_grads(5)
    l1b, g1b = loss_and_grads(5)
    l2, _ = loss_and_grads(6, want_grads=False)
    assert l1 == l1b and l1 != l2
    assert abs(l1 - float(t["loss"])) > 1e-6 * abs(l1)         # not the dropout-free golden loss
    for n in g1:
        close(g1b[n], g1[n], rtol=1e-4, atol=1e-5 * float(g1[n].abs().max()) + 1e-9)
    # directional derivative along the gradient (encoder parameters only: where the dropout sits)
    names = [n for n in g1 if ".transformer.layers." in n or "input_proj" in n]
    gn = math.sqrt(sum(float((g1[n].double() ** 2).sum()) for n in names))
    params = dict(model.named_parameters())
    eps = 2e-3 / gn * max(abs(l1), 1.0) if name == "tr_tiny" else 5e-4 / gn * max(abs(l1), 1.0)
    def shifted(sign):
        with torch.no_grad():
            for n in names:
                params[n].add_(g1[n], alpha=sign * eps / gn)
        val, _ = loss_and_grads(5, want_grads=False)
        with torch.no_grad():
            for n in names:
                params[n].sub_(g1[n], alpha=sign * eps / gn)
        return val
    fd = (shifted(+1.0) - shifted(-1.0)) / (2.0 * eps)
    assert abs(fd - gn) <= 0.03 * gn, (fd, gn)


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


@pytest.mark.parametrize("d,ff,M", [(64, 256, 1000), (128, 256, 4133), (64, 512, 129), (128, 1024, 300)])
@pytest.mark.parametrize("with_next", [True, False])
def test_transformer_tail_kernel(d, ff, M, with_next):
    """mmb_transformer_tail (out_proj + residual + LayerNorm1 + FFN + residual + LayerNorm2 + the next layer's in_proj in
    one launch, activations in registers) against nn.TransformerEncoderLayer's own arithmetic in float64; the output
    goes into a column slice of a wider buffer, M is not a multiple of the 128-row tile."""
    from mmb_hip import fused
    F = torch.nn.functional
    torch.manual_seed(d + ff + M)
    mk = lambda: torch.nn.TransformerEncoderLayer(d_model=d, nhead=4, dim_feedforward=ff, dropout=0.0, batch_first=True)
    l0, l1 = mk(), mk()
    for l in (l0, l1):
        for n, q in l.named_parameters():                      # LayerNorm weights off their 1 / 0 init
            if "norm" in n:
                q.data.add_(0.3 * torch.randn_like(q))
    a, h = torch.randn(M, d), torch.randn(M, d) * 1.5
    D = lambda t: t.detach().double()
    h1 = F.layer_norm(D(h) + F.linear(D(a), D(l0.self_attn.out_proj.weight), D(l0.self_attn.out_proj.bias)), (d,),
                      D(l0.norm1.weight), D(l0.norm1.bias), l0.norm1.eps)
    f = F.linear(h1, D(l0.linear1.weight), D(l0.linear1.bias)).relu()
    want = F.layer_norm(h1 + F.linear(f, D(l0.linear2.weight), D(l0.linear2.bias)), (d,), D(l0.norm2.weight),
                        D(l0.norm2.bias), l0.norm2.eps)
    want_qkv = F.linear(want, D(l1.self_attn.in_proj_weight), D(l1.self_attn.in_proj_bias))
    l0, l1 = l0.cuda(), l1.cuda()
    pack = fused.pack_transformer_tail(l0, l1 if with_next else None)
    wide = torch.full((M, d + 8), 7.0, device="cuda")
    out, qkv = fused.transformer_tail(pack, a.cuda(), h.cuda(), out=wide[:, 8:], want_qkv=True)
    close(out, want, rtol=1e-4, atol=2e-5)
    assert float((wide[:, :8] - 7.0).abs().max()) == 0.0
    if with_next:
        close(qkv, want_qkv, rtol=1e-4, atol=2e-5 * float(want_qkv.abs().max()))
    else:
        assert qkv is None
    # gathered attention rows: row r reads a_src[idx[r]] out of a wider buffer (pitch 3d), zeros where idx < 0 --
    # bit-identical to the same call on the materialised rows
    R = 37
    a_src = torch.randn(R, 3 * d, device="cuda")
    idx = torch.randint(-1, R, (M,), dtype=torch.int32, device="cuda")
    a_rows = torch.where((idx >= 0)[:, None], a_src[:, 2 * d:][idx.clamp(min=0).long()], torch.zeros(M, d, device="cuda"))
    o1, q1 = fused.transformer_tail(pack, a_rows.contiguous(), h.cuda())
    o2, q2 = fused.transformer_tail(pack, a_src[:, 2 * d:], h.cuda(), a_idx=idx)
    assert torch.equal(o1, o2) and (q1 is None or torch.equal(q1, q2))


@pytest.mark.parametrize("B,N,H,nh", [(5, 16, 8, 2), (4, 150, 64, 4), (3, 150, 128, 4), (2, 70, 128, 2)])
@pytest.mark.parametrize("path", ["mfma", "valu"])
def test_attention_uniform_keys(B, N, H, nh, path, monkeypatch):
    """What the reference's model feeds the attention in eval mode: every valid key (= padded slot, quirk Q4) of a jet
    carries the same row.  The forward kernels detect that on the staged rows and write the one V row out for every
    query; the result must be the general algorithm's (float64 restatement), also for the jets whose keys are NOT all
    alike in the same launch (jet 0 gets one perturbed key) and for a jet without any valid key."""
    from mmb_hip import ops
    monkeypatch.setenv("MMB_ATTN_PATH", path)
    g = torch.Generator().manual_seed(B * 77 + N + H)
    qkv = torch.randn(B, N, 3 * H, generator=g)
    mask = make_mask(B, N, g)
    pad = mask == 0                                            # the valid keys
    row = torch.randn(B, 1, 3 * H, generator=g)
    qkv = torch.where(pad[..., None], row.expand(B, N, 3 * H), qkv)
    if int(pad[0].sum()) >= 2:                                 # one odd key in jet 0: that jet takes the general path
        j = int(pad[0].nonzero()[-1])
        qkv[0, j, H:] += 0.5
    want = ref_attention(qkv.double(), mask, nh, True)
    mi = ops.mask_prepare(mask.cuda())
    out = ops.attention_fwd(qkv.reshape(B * N, 3 * H).cuda(), mi, nh, valid_is_zero=True)
    close(out.view(B, N, H), want, rtol=1e-5, atol=1e-6 * float(want.abs().max()))
    out2, lse = ops.attention_fwd(qkv.reshape(B * N, 3 * H).cuda(), mi, nh, valid_is_zero=True, want_lse=True)
    close(out2, out, rtol=1e-5, atol=1e-6 * float(want.abs().max()))      # (with lse: the general path)


@pytest.mark.parametrize("name", ["tr_tiny", "tr_odd", "S4"])
def test_transformer_row_compaction_is_bit_identical(golden, name, monkeypatch):
    """Inference path of MultiModalParticleTransformer: real particles + one padded slot per jet go through the encoder,
    the representative's heads are copied to the other padded slots (the padded slots' rows are identical by
    construction, quirk Q4 + the embedder's masking).  Must give the bits of the uncompacted path -- every kernel on
    it is row-wise -- for ragged masks, a jet without padded slots and a jet without real particles."""
    import copy
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    from tensorclass import TensorMultiModal
    g = golden(name)
    cfg = copy.deepcopy(g.cfg)
    cfg["data"]["target_path"] = GOLDEN
    model = MultiModalBridgeMatching(ExperimentConfigs(cfg))
    model.load_state_dict(g.params, strict=True)
    model = model.cuda().eval()
    N = g.t["mask"].shape[1]
    B = 9
    gen = torch.Generator().manual_seed(N)
    n = torch.randint(1, N, (B,), generator=gen)
    n[0], n[1] = N, 0
    mask = (torch.arange(N)[None, :] < n[:, None]).long().unsqueeze(-1)
    if name == "tr_odd":                                        # padded slots anywhere, not only at the end
        mask = mask[:, torch.randperm(N, generator=gen)]
    x = torch.randn(B, N, 3, generator=gen)                     # padded slots carry values too (K3)
    k = torch.randint(0, 8, (B, N, 1), generator=gen)
    t = torch.rand(B, generator=gen).reshape(B, 1, 1)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MMB_TR_COMPACT", flag)
        state = TensorMultiModal(t.cuda(), x.cuda(), k.cuda(), mask.cuda())
        heads = model(state, None)
        res[flag] = (heads.continuous.clone(), heads.discrete.clone())
    assert torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][1], res["0"][1])
    # padded slots of a jet all get the same heads; real particles do not
    hd = res["1"][1]
    pad = mask[2, :, 0] == 0
    assert int(pad.sum()) >= 2 and torch.equal(hd[2][pad][0], hd[2][pad][-1])
