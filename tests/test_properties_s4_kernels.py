"""Size-independent properties of the S4 (transformer) kernels at BASELINE's full size -- 1024 jets x 150 particles,
where an fp64 reference of every element is too slow for the test suite:

* attention: a query's output is a convex combination of the valid keys' V rows (V = ones -> ones; within the min/max
  of V per dim); it does not depend on the values stored in ignored key slots; jets are independent; permuting the
  keys of a jet (K, V and mask alike) leaves every query's output unchanged; the backward of sum(out * dout) w.r.t. V
  sums, per (jet, head), to sum(dout) over the queries (softmax rows sum to 1);
* mmb_linear_cat (row-streaming bf16x6 kernel): linearity in the input and agreement with the tiled fp32-MFMA kernel;
  a 4096-row slice against fp64;
* LayerNorm: rows have zero mean / unit variance before the affine map; shift invariance.
Through the C ABI (mmb_hip.ops)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

B, N = 1024, 150


def close(a, b, rtol=1e-4, atol=2e-5):
    torch.testing.assert_close(a.float().cpu(), b.float().cpu(), rtol=rtol, atol=atol)


@pytest.fixture(scope="module")
def jets():
    from mmb_hip import ops
    g = torch.Generator().manual_seed(150)
    n = torch.randint(0, N + 1, (B,), generator=g)
    n[0], n[1] = N, 0
    rank = torch.rand(B, N, generator=g).argsort(1).argsort(1)
    mask = (rank < n[:, None]).float().cuda()
    return ops, g, mask, ops.mask_prepare(mask)


@pytest.mark.parametrize("H,nh", [(64, 4), (128, 4)])
def test_attention_properties_full_size(jets, H, nh):
    ops, g, mask, mi = jets
    dh = H // nh
    qkv = torch.randn(B * N, 3 * H, generator=g).cuda()
    out, lse = ops.attention_fwd(qkv, mi, nh, want_lse=True)
    valid = (mask == 0)                                     # quirk Q4: the keys are the padded slots
    has_key = valid.any(1)
    o = out.view(B, N, H)
    assert float(o[~has_key].abs().max()) == 0.0            # jet without a key: zeros
    # convexity: V = 1 gives 1; outputs stay inside the range of the valid V rows
    q1 = qkv.clone()
    q1[:, 2 * H:] = 1.0
    o1 = ops.attention_fwd(q1, mi, nh).view(B, N, H)
    close(o1[has_key], torch.ones_like(o1[has_key]), rtol=0, atol=2e-6)
    v = qkv[:, 2 * H:].view(B, N, H)
    big = torch.finfo(torch.float32).max
    vmax = torch.where(valid[:, :, None], v, torch.full_like(v, -big)).max(1).values
    vmin = torch.where(valid[:, :, None], v, torch.full_like(v, big)).min(1).values
    ok = has_key[:, None, None].expand_as(o)
    assert bool(((o <= vmax[:, None] + 1e-5) | ~ok).all()) and bool(((o >= vmin[:, None] - 1e-5) | ~ok).all())
    # ignored key slots do not matter
    q2 = qkv.clone().view(B, N, 3 * H)
    q2[:, :, H:] = torch.where(valid[:, :, None], q2[:, :, H:], torch.full_like(q2[:, :, H:], 1e6))
    assert torch.equal(ops.attention_fwd(q2.view(B * N, 3 * H), mi, nh), out)
    # jet independence: the first 8 jets alone
    mi8 = ops.mask_prepare(mask[:8].contiguous())
    assert torch.equal(ops.attention_fwd(qkv[:8 * N].contiguous(), mi8, nh), out[:8 * N])
    # key permutation (K, V and mask permuted together; Q left in place)
    perm = torch.rand(B, N, generator=g).argsort(1).cuda()
    q3 = qkv.view(B, N, 3 * H).clone()
    q3[:, :, H:] = torch.gather(q3[:, :, H:], 1, perm[:, :, None].expand(B, N, 2 * H))
    mi3 = ops.mask_prepare(torch.gather(mask, 1, perm))
    close(ops.attention_fwd(q3.view(B * N, 3 * H).contiguous(), mi3, nh), out, rtol=1e-4, atol=1e-5)
    # backward: dV summed over the keys = dout summed over the queries, per (jet, head dim)
    dout = torch.randn(B * N, H, generator=g).cuda()
    dq = ops.attention_bwd(qkv, mi, nh, out, dout, lse).view(B, N, 3 * H)
    lhs = dq[:, :, 2 * H:].sum(1)
    rhs = dout.view(B, N, H).sum(1) * has_key[:, None]
    close(lhs, rhs, rtol=1e-4, atol=1e-3)
    assert float(dq[:, :, H:][~valid].abs().max()) == 0.0   # ignored keys get no gradient
    # and the valu kernels agree with the matrix-core kernels at this size
    import os
    os.environ["MMB_ATTN_PATH"] = "valu"
    try:
        close(ops.attention_fwd(qkv, mi, nh), out, rtol=1e-4, atol=1e-5)
        close(ops.attention_bwd(qkv, mi, nh, out, dout, lse), dq.view(B * N, 3 * H), rtol=1e-4, atol=2e-4)
    finally:
        del os.environ["MMB_ATTN_PATH"]


@pytest.mark.parametrize("K,Nout,act", [(128, 384, None), (64, 256, "relu"), (256, 128, None), (192, 128, "gelu")])
def test_linear_stream_properties_full_size(K, Nout, act):
    import os
    from mmb_hip import ops
    g = torch.Generator().manual_seed(K + Nout)
    M = B * N
    x, y = torch.randn(M, K, generator=g).cuda(), torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(Nout, K, generator=g) / math.sqrt(K)).cuda()
    b = torch.randn(Nout, generator=g).cuda()
    res = torch.randn(M, Nout, generator=g).cuda()
    run = lambda inp, bias=None, r=None, a=None: ops.linear_cat([(inp, ops.SEG_ROW)], W, bias, N, act=a, residual=r)
    # linearity of the GEMM part (no bias, no act): f(2x - 3y) = 2 f(x) - 3 f(y)
    fx, fy = run(x), run(y)
    close(run(2 * x - 3 * y), 2 * fx - 3 * fy, rtol=1e-4, atol=5e-5)
    full = run(x, b, res, act)
    # a slice against fp64
    sl = slice(70000, 74096)
    ref = torch.nn.functional.linear(x[sl].double(), W.double(), b.double())
    ref = {None: ref, "relu": ref.relu(), "gelu": torch.nn.functional.gelu(ref)}[act] + res[sl].double()
    close(full[sl], ref, rtol=1e-4, atol=2e-5)
    # the tiled fp32-MFMA kernel computes the same function
    os.environ["MMB_LINEAR_PATH"] = "tiled"
    try:
        close(run(x, b, res, act), full, rtol=1e-4, atol=2e-5)
    finally:
        del os.environ["MMB_LINEAR_PATH"]
    # rows are independent: the last 1000 rows alone (another tile assignment)
    tail = ops.linear_cat([(x[-1000:].contiguous(), ops.SEG_ROW)], W, b, N, act=act, residual=res[-1000:].contiguous())
    close(tail, full[-1000:], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("D", [64, 128])
def test_layernorm_properties_full_size(D):
    from mmb_hip import ops
    g = torch.Generator().manual_seed(D)
    M = B * N
    x = (torch.randn(M, D, generator=g) * 3 + 1).cuda()
    one, zero = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    y, stats = ops.layernorm_fwd(x, one, zero, 1e-5, want_stats=True)
    assert float(y.mean(1).abs().max()) < 1e-5
    close(y.var(1, unbiased=False), torch.ones(M), rtol=1e-3, atol=1e-3)
    close(ops.layernorm_fwd(x + 7.5, one, zero, 1e-5), y, rtol=1e-4, atol=2e-5)            # shift invariance
    close(stats[:, 0], x.mean(1), rtol=1e-5, atol=1e-5)
    # backward: dx is orthogonal to the constant vector and to xhat (rows of the Jacobian's null space)
    dy = torch.randn(M, D, generator=g).cuda()
    dx = torch.empty_like(x)
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    ops.layernorm_bwd(x, one, stats, dy, dx, False, dg, db)
    scale = float(dx.abs().max())
    assert float(dx.sum(1).abs().max()) < 1e-4 * scale * D
    assert float((dx * y).sum(1).abs().max()) < 2e-4 * scale * D
    close(db, dy.sum(0), rtol=1e-4, atol=1e-2)
    close(dg, (dy * y).sum(0), rtol=1e-4, atol=1e-2)
