"""Tensor-level wrappers over the C ABI: validate device/dtype/contiguity, allocate outputs
with torch (plumbing only), enqueue on torch's current HIP stream.  No host syncs."""
import ctypes as C
import os

import torch

import mmb_hip as _m
from . import call, MMBError

THERMOSTATS = {"ConstantThermostat": 0, "InverseThermostat": 1, "LinearThermostat": 2,
               "InverseSquareThermostat": 3, "SigmoidThermostat": 4}
BRIDGES = {"UniformFlow": 0, "SchrodingerBridge": 1}
ACT = {None: 0, "none": 0, "gelu": 1, "lrelu": 2, "relu": 3, "selu": 4}
SEG_ROW, SEG_JET, SEG_JET_MASK = 0, 1, 2


def _chk(t, dtype, name):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise MMBError(f"{name}: expected a tensor, got {type(t)}")
    if not t.is_cuda:
        raise MMBError(f"{name}: tensor lives on {t.device}; the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise MMBError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise MMBError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


def f32(t, name="tensor"):
    return _chk(t, torch.float32, name)


def i64(t, name="tensor"):
    return _chk(t, torch.int64, name)


def i32(t, name="tensor"):
    return _chk(t, torch.int32, name)


def stream():
    return torch.cuda.current_stream().cuda_stream


def _fl(v):
    return C.c_float(float(v))


# ------------------------------------------------------------------------ masks / source
class MaskInfo:
    """Canonical per-batch mask data: maskf (B,N) f32, counts (B) i32, total (1) i32."""
    __slots__ = ("maskf", "counts", "total", "B", "N", "_compact", "_order", "_long")

    def __init__(self, maskf, counts, total):
        self.maskf, self.counts, self.total = maskf, counts, total
        self.B, self.N = maskf.shape
        self._compact = None                       # row compaction of the transformer's inference path, built on demand
        self._order = None                         # launch order of the jets of a fused generation call, built on demand
        self._long = None                          # jets of 128+ particles (fused generation at 128 < N <= 160), on demand


def mask_prepare(mask):
    """mask: (B,N,1) or (B,N), int64 or float32."""
    if mask.dim() == 3:
        mask = mask.squeeze(-1)
    B, N = mask.shape
    if mask.dtype not in (torch.int64, torch.float32):
        mask = mask.to(torch.int64)
    mask = mask.contiguous()
    dev = mask.device
    maskf = torch.empty(B, N, dtype=torch.float32, device=dev)
    counts = torch.empty(B, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    is64 = mask.dtype == torch.int64
    ptr = i64(mask, "mask") if is64 else f32(mask, "mask")
    call("mmb_mask_prepare", B, N, ptr, int(is64), f32(maskf), i32(counts), i32(total), stream())
    return MaskInfo(maskf, counts, total)


def source_uniform(B, N, Dc, V, device, pmf=None, sample_masks=False, seed=0, offset=0,
                   want_x=True, want_k=True):
    x0 = torch.empty(B, N, Dc, dtype=torch.float32, device=device) if want_x else None
    k0 = torch.empty(B, N, 1, dtype=torch.int64, device=device) if want_k else None
    mask = torch.empty(B, N, 1, dtype=torch.int64, device=device)
    maskf = torch.empty(B, N, dtype=torch.float32, device=device)
    counts = torch.empty(B, dtype=torch.int32, device=device)
    npmf = 0 if pmf is None else pmf.numel()
    call("mmb_source_uniform", B, N, Dc, V, int(bool(sample_masks)), f32(pmf, "pmf"), npmf, seed, offset,
         f32(x0), i64(k0), i64(mask), f32(maskf), i32(counts), stream())
    return x0, k0, mask, maskf, counts


# ------------------------------------------------------------------------ K-A
def bridge_sample(t_u, eps, x0, x1, z, k0, k1, u, bridge_kind, sigma, thermo_kind, gamma, V):
    ref = x1 if x1 is not None else k1
    B, N = ref.shape[0], ref.shape[1]
    Dc = x1.shape[2] if x1 is not None else 0
    dev = ref.device
    t_out = torch.empty(B, dtype=torch.float32, device=dev)
    xt = torch.empty_like(x1) if x1 is not None else None
    kt = torch.empty_like(k1) if k1 is not None else None
    call("mmb_bridge_sample", B, N, Dc, V, f32(t_u, "t_u"), _fl(eps), f32(x0, "x0"), f32(x1, "x1"), f32(z, "z"),
         i64(k0, "k0"), i64(k1, "k1"), f32(u, "u"), bridge_kind, _fl(sigma or 0.0), thermo_kind, _fl(gamma or 0.0),
         f32(t_out), f32(xt), i64(kt), stream())
    return t_out, xt, kt


def bridge_sample_rng(eps, x1, k1, bridge_kind, sigma, thermo_kind, gamma, V, seed, offset):
    ref = x1 if x1 is not None else k1
    B, N = ref.shape[0], ref.shape[1]
    Dc = x1.shape[2] if x1 is not None else 0
    dev = ref.device
    t_out = torch.empty(B, dtype=torch.float32, device=dev)
    x0 = torch.empty_like(x1) if x1 is not None else None
    xt = torch.empty_like(x1) if x1 is not None else None
    k0 = torch.empty_like(k1) if k1 is not None else None
    kt = torch.empty_like(k1) if k1 is not None else None
    call("mmb_bridge_sample_rng", B, N, Dc, V, _fl(eps), f32(x1, "x1"), i64(k1, "k1"), bridge_kind,
         _fl(sigma or 0.0), thermo_kind, _fl(gamma or 0.0), seed, offset, f32(t_out), f32(x0), i64(k0), f32(xt),
         i64(kt), stream())
    return t_out, x0, k0, xt, kt


def telegraph_probs(t, k0, k1, thermo_kind, gamma, V):
    B, N = k0.shape[0], k0.shape[1]
    probs = torch.empty(B, N, V, dtype=torch.float32, device=k0.device)
    call("mmb_telegraph_probs", B, N, V, f32(t, "t"), i64(k0, "k0"), i64(k1, "k1"), thermo_kind, _fl(gamma),
         f32(probs), stream())
    return probs


def bridge_drift(t, xt, x0, x1, bridge_kind):
    B, N, Dc = xt.shape
    out = torch.empty_like(xt)
    call("mmb_bridge_drift", B, N, Dc, f32(t, "t"), f32(xt, "xt"), f32(x0, "x0"), f32(x1, "x1"), bridge_kind,
         f32(out), stream())
    return out


# ------------------------------------------------------------------------ K-B
def euler_step_(x, v, dt):
    call("mmb_euler_step", x.numel(), f32(x, "x"), f32(v, "v"), _fl(dt), stream())
    return x


def euler_maruyama_step_(x, v, dt, t, t_scalar, sigma, dW=None, seed=0, offset=0):
    B, N, Dc = x.shape
    call("mmb_euler_maruyama_step", B, N, Dc, f32(x, "x"), f32(v, "v"), _fl(dt), f32(t, "t"), _fl(t_scalar),
         _fl(sigma), f32(dW, "dW"), seed, offset, stream())
    return x


# ------------------------------------------------------------------------ K-C
def telegraph_rate(logits, k, t, t_scalar, thermo_kind, gamma):
    B, N, V = logits.shape
    rates = torch.empty_like(logits)
    call("mmb_telegraph_rate", B, N, V, f32(logits, "logits"), i64(k, "k"), f32(t, "t"), _fl(t_scalar), thermo_kind,
         _fl(gamma), f32(rates), stream())
    return rates


def rate_tauleap_(logits, k, t, t_scalar, dt, thermo_kind, gamma, delta_n=None, seed=0, offset=0,
                  overflow="wrap", final_argmax=False, rates_out=None):
    B, N, V = logits.shape
    call("mmb_rate_tauleap", B, N, V, f32(logits, "logits"), i64(k, "k"), f32(t, "t"), _fl(t_scalar), _fl(dt),
         thermo_kind, _fl(gamma), f32(delta_n, "delta_n"), seed, offset, int(overflow == "clamp"),
         int(bool(final_argmax)), f32(rates_out, "rates_out"), stream())
    return k


def rate_euler_jump_(logits, k, t, t_scalar, dt, thermo_kind, gamma, u=None, seed=0, offset=0,
                     final_argmax=False, rates_out=None):
    B, N, V = logits.shape
    call("mmb_rate_euler_jump", B, N, V, f32(logits, "logits"), i64(k, "k"), f32(t, "t"), _fl(t_scalar), _fl(dt),
         thermo_kind, _fl(gamma), f32(u, "u"), seed, offset, int(bool(final_argmax)), f32(rates_out, "rates_out"),
         stream())
    return k


# ------------------------------------------------------------------------ K-F
LOSS_WS = 2 * 2048


def loss_fwd_bwd(head_c, head_d, t, xt, x0, x1, bridge_kind, k1, mi, vocab_freq, coef, want_grad=True):
    ref = head_c if head_c is not None else head_d
    B, N = ref.shape[0], ref.shape[1]
    dev = ref.device
    Dc = head_c.shape[2] if head_c is not None else 0
    V = head_d.shape[2] if head_d is not None else 0
    partials = torch.empty(LOSS_WS, dtype=torch.float32, device=dev)
    losses = torch.empty(2, dtype=torch.float32, device=dev)
    dc = torch.empty_like(head_c) if (want_grad and head_c is not None) else None
    dd = torch.empty_like(head_d) if (want_grad and head_d is not None) else None
    call("mmb_loss_fwd_bwd", B, N, Dc, V, f32(head_c, "head_c"), f32(head_d, "head_d"), f32(t, "t"), f32(xt, "xt"),
         f32(x0, "x0"), f32(x1, "x1"), bridge_kind, i64(k1, "k1"), f32(mi.maskf), i32(mi.total),
         f32(vocab_freq, "vocab_freq"), f32(coef, "coef"), f32(partials), f32(losses), f32(dc), f32(dd), stream())
    return losses, dc, dd


def _host2(w):
    if w is None:
        return None
    return (C.c_float * 2)(float(w[0]), float(w[1]))


def multimode_coef(s, fixed_w, device):
    coef = torch.empty(2, dtype=torch.float32, device=device)
    call("mmb_multimode_coef", f32(s, "loss_weights"), _host2(fixed_w), f32(coef), stream())
    return coef


def multimode_loss(losses, s, fixed_w):
    loss = torch.empty(1, dtype=torch.float32, device=losses.device)
    d_s = torch.empty(2, dtype=torch.float32, device=losses.device) if s is not None else None
    call("mmb_multimode_loss", f32(losses), f32(s, "loss_weights"), _host2(fixed_w), f32(loss), f32(d_s), stream())
    return loss, d_s


# ------------------------------------------------------------------------ K-D / K-E
def weight_norm(row_off, row_len, v_flat, g_flat, want_inv=False):
    W = torch.empty_like(v_flat)
    inv = torch.empty_like(g_flat) if want_inv else None
    call("mmb_weight_norm", g_flat.numel(), i32(row_off), i32(row_len), f32(v_flat), f32(g_flat), f32(W), f32(inv),
         stream())
    return W, inv


def weight_norm_bwd(row_off, row_len, v_flat, g_flat, inv, dW):
    dv = torch.empty_like(v_flat)
    dg = torch.empty_like(g_flat)
    call("mmb_weight_norm_bwd", g_flat.numel(), i32(row_off), i32(row_len), f32(v_flat), f32(g_flat), f32(inv),
         f32(dW), f32(dv), f32(dg), stream())
    return dv, dg


def linear_cat(segs, W, bias, rows_per_jet, maskf=None, act=None, residual=None, row_scale=None, out=None,
               want_pre=False, post=None, res2=None, out2=None, want_y1=True, drop_p=0.0, seed=0, offset=0):
    """Y1 = act(cat(segs) @ W^T + b) + residual ; optional Y2 = dropout(post(Y1)) + res2.
    segs: list of (tensor2d, kind).  ROW tensors are (M,Ks) [row stride may exceed Ks];
    JET tensors are (B,Ks).  W: (Nout,K), row stride may exceed K (column slice).
    Returns Y1, or (Y1, Y2) when a second stage is requested, with `pre` appended if want_pre."""
    n = len(segs)
    M = None
    ptrs = (C.c_void_p * n)()
    widths = (C.c_int * n)()
    kinds = (C.c_int * n)()
    lds = (C.c_int * n)()
    for i, (t, kind) in enumerate(segs):
        if t.dim() != 2 or t.stride(1) != 1:
            raise MMBError("linear_cat: segments must be 2-D with unit inner stride")
        if not t.is_cuda or t.dtype != torch.float32:
            raise MMBError("linear_cat: segments must be float32 HIP tensors (no CPU fallback)")
        ptrs[i], widths[i], kinds[i], lds[i] = t.data_ptr(), t.shape[1], kind, t.stride(0)
        if kind == SEG_ROW:
            M = t.shape[0] if M is None else M
            if t.shape[0] != M:
                raise MMBError("linear_cat: ROW segments disagree on M")
    if M is None:
        raise MMBError("linear_cat: at least one ROW segment is required")
    if W.dim() != 2 or W.stride(1) != 1 or not W.is_cuda or W.dtype != torch.float32:
        raise MMBError("linear_cat: W must be a 2-D float32 HIP tensor with unit inner stride")
    Nout = W.shape[0]
    dev = segs[0][0].device
    two = post is not None or res2 is not None or out2 is not None
    if out is None and (want_y1 or not two):
        out = torch.empty(M, Nout, dtype=torch.float32, device=dev)
    if two and out2 is None:
        out2 = torch.empty(M, Nout, dtype=torch.float32, device=dev)
    pre = torch.empty(M, Nout, dtype=torch.float32, device=dev) if want_pre else None

    def p2(t):
        return (t.data_ptr(), t.stride(0)) if t is not None else (None, 0)

    rp, rl = p2(residual)
    op, ol = p2(out)
    r2p, r2l = p2(res2)
    o2p, o2l = p2(out2)
    if _m.PROFILE is not None:
        K = sum(widths)
        _m.PROFILE_META = {"M": M, "K": K, "Nout": Nout, "flops": 2.0 * M * K * Nout}
    call("mmb_linear_cat", M, rows_per_jet, n, ptrs, widths, kinds, lds, f32(maskf, "maskf"),
         W.data_ptr(), W.stride(0), f32(bias, "bias"), Nout, ACT[act], rp, rl, f32(row_scale, "row_scale"),
         op, ol, f32(pre), ACT[post], _fl(drop_p), seed, offset, r2p, r2l, o2p, o2l, stream())
    _m.PROFILE_META = None
    res = (out, out2) if two else out
    if want_pre:
        res = (res + (pre,)) if two else (res, pre)
    return res


def linear_cat_gate(segs, W, rows_per_jet, gate, out, maskf=None, act="relu"):
    """out = (cat(segs) @ W^T) * act'(gate): the gradient of an activated layer's pre-activation from the gradient of
    its consumer's (one GEMM, no mmb_act_bwd round trip).  gate = the pre-activation (gelu), anything with its sign
    (relu, lrelu) or the output (selu); gate, out: (M, Nout) float32, unit inner stride."""
    if act not in ("relu", "lrelu", "gelu", "selu"):
        raise MMBError(f"linear_cat_gate: no derivative for activation {act!r}")
    for t, _ in list(segs) + [(gate, 0), (out, 0), (W, 0)]:
        if t.dim() != 2 or t.stride(1) != 1 or not t.is_cuda or t.dtype != torch.float32:
            raise MMBError("linear_cat_gate: operands must be 2-D float32 HIP tensors with unit inner stride")
    n, ptrs, widths, kinds, lds, M = _seg_arrays(segs)
    Nout = W.shape[0]
    if gate.shape != (M, Nout) or out.shape != (M, Nout):
        raise MMBError("linear_cat_gate: gate / out must be (M, Nout)")
    if _m.PROFILE is not None:
        _m.PROFILE_META = {"M": M, "K": sum(widths), "Nout": Nout, "flops": 2.0 * M * sum(widths) * Nout}
    call("mmb_linear_cat_gate", M, rows_per_jet, n, ptrs, widths, kinds, lds, f32(maskf, "maskf"), W.data_ptr(), W.stride(0),
         Nout, ACT[act], gate.data_ptr(), gate.stride(0), out.data_ptr(), out.stride(0), stream())
    _m.PROFILE_META = None
    return out


def linear_cat_ln(segs, W, bias, rows_per_jet, gamma, beta, eps, maskf=None, act=None, residual=None, want_y1=False,
                  want_stats=False, out=None):
    """y1 = act(cat(segs) @ W^T + b) + residual ; y = LayerNorm(y1) * gamma + beta  (one call: the normalisation runs
    in the linear kernel's epilogue where a wave owns whole rows).  Returns (y1 | None, y, stats | None)."""
    for t, _ in segs:
        if t.dim() != 2 or t.stride(1) != 1 or not t.is_cuda or t.dtype != torch.float32:
            raise MMBError("linear_cat_ln: segments must be 2-D float32 HIP tensors with unit inner stride (no CPU fallback)")
    n, ptrs, widths, kinds, lds, M = _seg_arrays(segs)
    if M is None:
        raise MMBError("linear_cat_ln: at least one ROW segment is required")
    if W.dim() != 2 or W.stride(1) != 1 or not W.is_cuda or W.dtype != torch.float32:
        raise MMBError("linear_cat_ln: W must be a 2-D float32 HIP tensor with unit inner stride")
    Nout = W.shape[0]
    dev = W.device
    y1 = torch.empty(M, Nout, dtype=torch.float32, device=dev) if want_y1 else None
    y = torch.empty(M, Nout, dtype=torch.float32, device=dev) if out is None else out
    stats = torch.empty(M, 2, dtype=torch.float32, device=dev) if want_stats else None
    rp, rl = (residual.data_ptr(), residual.stride(0)) if residual is not None else (None, 0)
    if _m.PROFILE is not None:
        _m.PROFILE_META = {"M": M, "K": sum(widths), "Nout": Nout, "flops": 2.0 * M * sum(widths) * Nout}
    call("mmb_linear_cat_ln", M, rows_per_jet, n, ptrs, widths, kinds, lds, f32(maskf, "maskf"), W.data_ptr(), W.stride(0),
         f32(bias, "bias"), Nout, ACT[act], rp, rl, f32(y1), Nout if y1 is not None else 0, f32(gamma, "gamma"),
         f32(beta, "beta"), _fl(eps), y.data_ptr(), y.stride(0), f32(stats), stream())
    _m.PROFILE_META = None
    return y1, y, stats


def meansum_pool(x, mi, out=None):
    """x: (B*N, H) rows (row stride may exceed H) -> pooled (B, 2H) = [mean, 0.01*sum]."""
    B, N = mi.B, mi.N
    H = x.shape[1]
    if out is None:
        out = torch.empty(B, 2 * H, dtype=torch.float32, device=x.device)
    call("mmb_meansum_pool", B, N, H, x.data_ptr(), x.stride(0), f32(mi.maskf), i32(mi.counts), out.data_ptr(),
         out.stride(0), stream())
    return out


def embed(t, t_scalar, x, k, mi, fourier_W, time_W, time_b, cont_W, cont_b, table, want_feat=False):
    B, N = mi.B, mi.N
    dev = mi.maskf.device
    Et = time_W.shape[0]
    Dc = x.shape[-1] if x is not None else 0
    Ec = cont_W.shape[0] if (x is not None) else 0
    V, Ed = (table.shape if (k is not None) else (0, 0))
    temb = torch.empty(B, Et, dtype=torch.float32, device=dev)
    cont = torch.empty(B * N, Ec, dtype=torch.float32, device=dev) if x is not None else None
    disc = torch.empty(B * N, Ed, dtype=torch.float32, device=dev) if k is not None else None
    feat = torch.empty(B, Et, dtype=torch.float32, device=dev) if want_feat else None
    call("mmb_embed", B, N, Dc, V, Et, Ec, Ed, f32(t, "t"), _fl(t_scalar), f32(x, "x"), i64(k, "k"), f32(mi.maskf),
         f32(fourier_W), f32(time_W), f32(time_b), f32(cont_W if x is not None else None),
         f32(cont_b if x is not None else None), f32(table if k is not None else None), f32(temb), f32(cont),
         f32(disc), f32(feat), stream())
    return (temb, cont, disc, feat) if want_feat else (temb, cont, disc)


def time_embed(t, fourier_W, time_W, time_b):
    """Per-row time embedding Linear(cat[sin, cos](2 pi t W)) for a vector t (n,) -> (n, Et)."""
    n = t.numel()
    Et = time_W.shape[0]
    temb = torch.empty(n, Et, dtype=torch.float32, device=t.device)
    ones = torch.ones(n, dtype=torch.float32, device=t.device)
    call("mmb_embed", n, 1, 0, 0, Et, 0, 0, f32(t.contiguous(), "t"), _fl(0.0), None, None, f32(ones),
         f32(fourier_W), f32(time_W), f32(time_b), None, None, None, f32(temb), None, None, None, stream())
    return temb


# ------------------------------------------------------------------------ backward
def _pl(t):
    return (t.data_ptr(), t.stride(0)) if t is not None else (None, 0)


def _seg_arrays(segs):
    n = len(segs)
    ptrs = (C.c_void_p * n)()
    widths = (C.c_int * n)()
    kinds = (C.c_int * n)()
    lds = (C.c_int * n)()
    M = None
    for i, (t, kind) in enumerate(segs):
        ptrs[i], widths[i], kinds[i], lds[i] = t.data_ptr(), t.shape[1], kind, t.stride(0)
        if kind == SEG_ROW:
            M = t.shape[0]
    return n, ptrs, widths, kinds, lds, M


def act_bwd(M, Nout, gY1, gY2, y1, y2, res2, pre, act, post, drop_p=0.0, seed=0, offset=0):
    ref = gY1 if gY1 is not None else gY2
    dZ = torch.empty(M, Nout, dtype=torch.float32, device=ref.device)
    call("mmb_act_bwd", M, Nout, *_pl(gY1), *_pl(gY2), *_pl(y1), *_pl(y2), *_pl(res2), *_pl(pre), ACT[act], ACT[post],
         _fl(drop_p), seed, offset, dZ.data_ptr(), dZ.stride(0), stream())
    return dZ


def jet_sum(x, B, N, maskf=None, out=None, accumulate=False):
    H = x.shape[1]
    if out is None:
        out = torch.empty(B, H, dtype=torch.float32, device=x.device)
        accumulate = False
    call("mmb_jet_sum", B, N, H, x.data_ptr(), x.stride(0), f32(maskf), out.data_ptr(), out.stride(0),
         int(accumulate), stream())
    return out


def pool_bwd(dP, mi, dX, accumulate):
    H = dX.shape[1]
    call("mmb_pool_bwd", mi.B, mi.N, H, dP.data_ptr(), dP.stride(0), f32(mi.maskf), i32(mi.counts), dX.data_ptr(),
         dX.stride(0), int(accumulate), stream())
    return dX


def grad_weight(segs, rows_per_jet, maskf, dZ, dW, db):
    """dW (Nout,K) += dZ^T @ cat(segs); db (Nout) += colsum(dZ).  Buffers must be pre-zeroed."""
    n, ptrs, widths, kinds, lds, M = _seg_arrays(segs)
    Nout = dZ.shape[1]
    call("mmb_grad_weight", M, rows_per_jet, n, ptrs, widths, kinds, lds, f32(maskf), dZ.data_ptr(), dZ.stride(0),
         Nout, dW.data_ptr(), dW.stride(0), db.data_ptr() if db is not None else None, stream())


def grad_weight_rows(X, dZ, dW, db=None, replicas=1, rep_stride=0, db_stride=0):
    """dW (Nout,K) += dZ^T @ X for one row-major input X (M,K) [row stride may exceed K]; db += colsum(dZ).
    dW may be a column slice of a wider weight gradient.  Buffers must be pre-zeroed.  With replicas > 1
    the accumulation is spread over copies dW + r*rep_stride / db + r*db_stride (floats) that the caller sums."""
    for t, name in ((X, "X"), (dZ, "dZ"), (dW, "dW")):
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
            raise MMBError(f"grad_weight_rows: {name} must be a 2-D float32 HIP tensor with unit inner stride")
    if X.shape[0] != dZ.shape[0] or dW.shape != (dZ.shape[1], X.shape[1]):
        raise MMBError("grad_weight_rows: shape mismatch")
    call("mmb_grad_weight_rows", X.shape[0], X.data_ptr(), X.stride(0), X.shape[1], dZ.data_ptr(), dZ.stride(0),
         dZ.shape[1], dW.data_ptr(), dW.stride(0), db.data_ptr() if db is not None else None, replicas, rep_stride,
         db_stride, stream())


def weight_norm_bwd_rep(row_off, row_len, v_flat, g_flat, inv, dW_rep, db_rep, out=None):
    """dW_rep (R, Nout*K) and db_rep (R, Nout) replicas -> (dv, dg, db); out = (dv, dg, db) flat contiguous
    buffers to overwrite (e.g. views of the flat gradient bucket), any of them None to allocate."""
    dv, dg, db = out if out is not None else (None, None, None)
    dv = torch.empty_like(v_flat) if dv is None else dv
    dg = torch.empty_like(g_flat) if dg is None else dg
    db = torch.empty(g_flat.numel(), dtype=torch.float32, device=v_flat.device) if db is None else db
    call("mmb_weight_norm_bwd_rep", g_flat.numel(), i32(row_off), i32(row_len), f32(v_flat), f32(g_flat), f32(inv),
         f32(dW_rep), dW_rep.shape[0], dW_rep.stride(0), f32(db_rep), db_rep.stride(0), f32(dv), f32(dg), f32(db),
         stream())
    return dv, dg, db


def postprocess_(x, k, maskf, scale=None, shift=None, log_pt=0, N=None):
    """In-place inverse data transform + apply_mask on a generated sample (x (..,N,Dc) | None, k (..,N,1) | None).
    log_pt: 0 | 1 (feature 0) | 2 (slot 0 of every jet: the reference's index expression on stacked 4-D paths)."""
    P = maskf.numel()
    Dc = x.shape[-1] if x is not None else 0
    if N is None:
        N = (x if x is not None else k).shape[-2]
    call("mmb_postprocess", P, int(N), Dc, f32(x, "x"), i64(k, "k"), f32(maskf, "maskf"), f32(scale, "scale"),
         f32(shift, "shift"), int(log_pt), stream())


def postprocess_onehot(x, V, maskf, scale=None, shift=None, log_pt=0):
    """'onehot' samples: x (..,N,Dc+V) -> (x (..,N,Dc) | None, k (..,N,1) int64) after the inverse transform, argmax
    over the last V channels and apply_mask."""
    P, D, N = maskf.numel(), x.shape[-1], x.shape[-2]
    Dc = D - V
    xout = torch.empty(*x.shape[:-1], Dc, dtype=torch.float32, device=x.device) if Dc > 0 else None
    kout = torch.empty(*x.shape[:-1], 1, dtype=torch.int64, device=x.device)
    call("mmb_postprocess_onehot", P, int(N), D, int(V), f32(x, "x"), f32(xout, "xout"), i64(kout, "kout"),
         f32(maskf, "maskf"), f32(scale, "scale"), f32(shift, "shift"), int(log_pt), stream())
    return xout, kout


# ---- multi-problem launches (device tables of 8-byte fields, built once per training workspace) ----
def gw_class(K, Nout):
    """Tile class (OTb, KTb) mmb_grad_weight_rows picks for a (Nout, K) weight gradient."""
    ot_all, kt_all = (Nout + 31) // 32, (K + 31) // 32
    OTb = 4 if ot_all >= 4 else (2 if ot_all >= 2 else 1)
    KTb = 4 if kt_all >= 4 else (2 if kt_all >= 2 else 1)
    max_tiles = int(os.environ.get("MMB_GW_TILES_MULTI", "4"))       # 8 tiles leave one wave per SIMD: S2 +11 %, S3 unchanged
    while OTb * KTb > max_tiles:
        if KTb >= OTb:
            KTb //= 2
        else:
            OTb //= 2
    return OTb, KTb


def gw_fast(X, dZ):
    """True when mmb_grad_weight_rows would take its row-streaming path for these operands."""
    K, Nout = X.shape[1], dZ.shape[1]
    return (K % 4 == 0 and Nout % 4 == 0 and X.stride(0) % 4 == 0 and dZ.stride(0) % 4 == 0
            and X.data_ptr() % 16 == 0 and dZ.data_ptr() % 16 == 0 and X.stride(1) == 1 and dZ.stride(1) == 1)


def gw_rows_per_block(problems):
    """Rows per block for the big (M > 16384) problems of ONE multi launch: every block ends with an LDS reduction and
    16*OT*KT atomics per lane, so blocks should be as long as the launch can afford while still filling the chip
    (>= 2 x 256 CUs x 3 resident blocks).  problems: iterable of (M, K, Nout)."""
    env = os.environ.get("MMB_GW_RPB")
    if env:
        return int(env)
    for rpb in (2048, 1024):
        blocks = 0
        for M, K, Nout in problems:
            OTb, KTb = gw_class(K, Nout)
            r = 64 if M <= 16384 else rpb
            blocks += ((M + r - 1) // r) * (((K + 31) // 32 + KTb - 1) // KTb) * (((Nout + 31) // 32 + OTb - 1) // OTb)
        if blocks >= 1536:
            return rpb
    return 512


def mask_rows(mi, rows):
    """rows (B*N) int32 <- the flat rows of the real particles, ascending (the first mi.total of them are valid): the
    reduction range of the per-particle weight-gradient problems (csrc/backward_kernels.hip: mask_rows_kernel)."""
    offs = torch.cumsum(mi.counts, 0, dtype=torch.int32) - mi.counts
    call("mmb_mask_rows", mi.B, mi.N, f32(mi.maskf), i32(offs), i32(rows), stream())
    return rows


def gw_record(X, dZ, dW, db, replicas, rep_stride, db_stride, block_begin, rpb_big=512, rows=None, n_rows=None):
    """One GwProb record (list of 20 ints) + the number of blocks it takes.  rows / n_rows: device row list and its
    length (mask_rows) for a per-particle problem, or None."""
    M, K = X.shape
    Nout = dZ.shape[1]
    OTb, KTb = gw_class(K, Nout)
    rpb = 64 if M <= 16384 else rpb_big
    nbx = (M + rpb - 1) // rpb
    nby = ((K + 31) // 32 + KTb - 1) // KTb
    nbz = ((Nout + 31) // 32 + OTb - 1) // OTb
    n = nbx * nby * nbz
    rec = [X.data_ptr(), dZ.data_ptr(), dW.data_ptr(), 0 if db is None else db.data_ptr(), X.stride(0), dZ.stride(0),
           dW.stride(0), rep_stride, db_stride, M, K, Nout, rpb, replicas, nbx, nby, block_begin, block_begin + n,
           0 if rows is None else rows.data_ptr(), 0 if n_rows is None else n_rows.data_ptr()]
    return rec, (n + 7) // 8 * 8          # ranges start on multiples of 8 (XCD-aware block order); gap blocks idle


def grad_weight_rows_multi(cls, table, n_probs, total_blocks):
    call("mmb_grad_weight_rows_multi", cls[0], cls[1], C.c_void_p(table.data_ptr()), n_probs, total_blocks, stream())


def weight_norm_multi(table, n_rows):
    call("mmb_weight_norm_multi", n_rows, C.c_void_p(table.data_ptr()), stream())


def weight_norm_bwd_multi(table, n_rows):
    call("mmb_weight_norm_bwd_multi", n_rows, C.c_void_p(table.data_ptr()), stream())


OBSERVABLES = ("numParticles", "px", "py", "pz", "e", "pt", "m", "eta", "phi", "numPhotons", "numNeutralHadrons",
               "numNegativeHadrons", "numPositiveHadrons", "numElectrons", "numPositrons", "numMuons", "numAntiMuons",
               "charge", "jet_charge")


def jet_observables(x, k, mask):
    """(B,19) jet-level observables (columns: OBSERVABLES) of a post-processed sample, one wave per jet."""
    B, N = mask.shape[0], mask.shape[1]
    maskf = mask.reshape(B, N).float().contiguous()
    out = torch.empty(B, 19, dtype=torch.float32, device=maskf.device)
    Dc = x.shape[-1] if x is not None else 0
    call("mmb_jet_observables", B, N, Dc, f32(None if x is None else x.contiguous(), "x"),
         i64(None if k is None else k.contiguous(), "k"), f32(maskf, "maskf"), f32(out), stream())
    return out


def wasserstein_1d(a, b):
    """scipy.stats.wasserstein_distance for two equally weighted 1-D samples, on device (utils.py:531-534):
    the integral of |F_a - F_b| over the merged support."""
    a, b = a.reshape(-1).double().sort().values, b.reshape(-1).double().sort().values
    allv = torch.cat([a, b]).sort().values
    deltas = allv[1:] - allv[:-1]
    ca = torch.searchsorted(a, allv[:-1], right=True).double() / a.numel()
    cb = torch.searchsorted(b, allv[:-1], right=True).double() / b.numel()
    return ((ca - cb).abs() * deltas).sum()


def grad_weight_smallk(X, dZ, dW, db=None):
    """dW (Nout,K) += dZ^T @ X, db += colsum(dZ) for K <= 4, Nout <= 32 (rows may be strided)."""
    for t, name in ((X, "X"), (dZ, "dZ"), (dW, "dW")):
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
            raise MMBError(f"grad_weight_smallk: {name} must be a 2-D float32 HIP tensor with unit inner stride")
    if X.shape[0] != dZ.shape[0] or dW.shape != (dZ.shape[1], X.shape[1]):
        raise MMBError("grad_weight_smallk: shape mismatch")
    call("mmb_grad_weight_smallk", X.shape[0], X.data_ptr(), X.stride(0), X.shape[1], dZ.data_ptr(), dZ.stride(0),
         dZ.shape[1], dW.data_ptr(), dW.stride(0), db.data_ptr() if db is not None else None, stream())


# ------------------------------------------------------------------------ transformer set encoder
def attention_fwd(qkv, mi, nheads, valid_is_zero=True, want_lse=False, drop_p=0.0, seed=0, offset=0):
    """qkv (B*N, 3H) [q|k|v] -> out (B*N, H) (+ lse (B, nheads, N) for the backward).  drop_p > 0: dropout on the
    attention probabilities (training mode of nn.MultiheadAttention), Philox keyed by (seed, offset); with want_lse
    the result is (out, lse, keep_bits): hand keep_bits to attention_bwd."""
    B, N = mi.B, mi.N
    H = qkv.shape[1] // 3
    if qkv.dim() != 2 or qkv.shape[0] != B * N or qkv.shape[1] != 3 * H or qkv.stride(1) != 1:
        raise MMBError("attention_fwd: qkv must be (B*N, 3H) with unit inner stride")
    f32(qkv.new_empty(0), "qkv")                              # device / dtype check
    out = torch.empty(B * N, H, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(B, nheads, N, dtype=torch.float32, device=qkv.device) if want_lse else None
    if drop_p > 0.0:
        bits = torch.empty(B * nheads * N * ((N + 31) // 32), dtype=torch.int32, device=qkv.device) if want_lse else None
        call("mmb_attention_dropout_fwd", B, N, H, nheads, qkv.data_ptr(), qkv.stride(0), f32(mi.maskf),
             int(bool(valid_is_zero)), f32(out), out.stride(0), f32(lse), _fl(drop_p), seed, offset, i32(bits), stream())
        if want_lse:
            return out, lse, bits
    else:
        call("mmb_attention_fwd", B, N, H, nheads, qkv.data_ptr(), qkv.stride(0), f32(mi.maskf),
             int(bool(valid_is_zero)), f32(out), out.stride(0), f32(lse), stream())
    return (out, lse) if want_lse else out


def attention_bwd(qkv, mi, nheads, out, dout, lse, valid_is_zero=True, drop_p=0.0, seed=0, offset=0, keep_bits=None):
    B, N = mi.B, mi.N
    H = qkv.shape[1] // 3
    if dout.stride(1) != 1 or out.stride(1) != 1:
        raise MMBError("attention_bwd: unit inner stride required")
    dqkv = torch.empty(B * N, 3 * H, dtype=torch.float32, device=qkv.device)
    if drop_p > 0.0:
        call("mmb_attention_dropout_bwd", B, N, H, nheads, qkv.data_ptr(), qkv.stride(0), f32(mi.maskf),
             int(bool(valid_is_zero)), out.data_ptr(), out.stride(0), dout.data_ptr(), dout.stride(0), f32(lse), f32(dqkv),
             dqkv.stride(0), _fl(drop_p), seed, offset, i32(keep_bits), stream())
    else:
        call("mmb_attention_bwd", B, N, H, nheads, qkv.data_ptr(), qkv.stride(0), f32(mi.maskf),
             int(bool(valid_is_zero)), out.data_ptr(), out.stride(0), dout.data_ptr(), dout.stride(0), f32(lse), f32(dqkv),
             dqkv.stride(0), stream())
    return dqkv


def attention_dropout_mask(mi, nheads, drop_p, seed, offset, valid_is_zero=True):
    """(B, nheads, N, N) keep factors [b,h,q,n] of attention_fwd(..., drop_p, seed, offset): 1/(1-p) or 0."""
    B, N = mi.B, mi.N
    out = torch.empty(B, nheads, N, N, dtype=torch.float32, device=mi.maskf.device)
    call("mmb_attention_dropout_mask", B, N, nheads, f32(mi.maskf), int(bool(valid_is_zero)), _fl(drop_p), seed, offset,
         f32(out), stream())
    return out


def layernorm_fwd(x, gamma, beta, eps=1e-5, want_stats=False, out=None):
    M, D = x.shape
    if x.stride(1) != 1 or not x.is_cuda or x.dtype != torch.float32:
        raise MMBError("layernorm_fwd: x must be a 2-D float32 HIP tensor with unit inner stride")
    y = torch.empty(M, D, dtype=torch.float32, device=x.device) if out is None else out
    stats = torch.empty(M, 2, dtype=torch.float32, device=x.device) if want_stats else None
    call("mmb_layernorm_fwd", M, D, x.data_ptr(), x.stride(0), f32(gamma, "gamma"), f32(beta, "beta"), _fl(eps),
         y.data_ptr(), y.stride(0), f32(stats), stream())
    return (y, stats) if want_stats else y


def layernorm_bwd(x, gamma, stats, dy, dx, accumulate, dgamma, dbeta):
    """dx (M,D) overwritten or accumulated; dgamma/dbeta (D) accumulated with atomics (pre-zeroed)."""
    M, D = x.shape
    call("mmb_layernorm_bwd", M, D, x.data_ptr(), x.stride(0), f32(gamma, "gamma"), f32(stats), dy.data_ptr(),
         dy.stride(0), dx.data_ptr(), dx.stride(0), int(bool(accumulate)), f32(dgamma), f32(dbeta), stream())


def sum_replicas(src, dst, accumulate=False):
    """src (R, n) replicas (row stride may exceed n) -> dst (n) contiguous."""
    R, n = src.shape
    call("mmb_sum_replicas", n, R, src.stride(0), src.data_ptr(), f32(dst.view(-1)), int(bool(accumulate)), stream())
    return dst


# ------------------------------------------------------------------------ optimizer step
ADAMW_WS = 1024


def adamw_clip_step_(params, grads, exp_avg, exp_avg_sq, n_active, lr, betas, eps, weight_decay, step, max_norm,
                     partials, norm_out=None, write_clipped=False):
    """Global-norm clip + AdamW on flat fp32 buffers, in place, two launches, no host sync."""
    call("mmb_adamw_clip_step", params.numel(), int(n_active), f32(params, "params"), f32(grads, "grads"),
         f32(exp_avg, "exp_avg"), f32(exp_avg_sq, "exp_avg_sq"), _fl(lr), _fl(betas[0]), _fl(betas[1]), _fl(eps),
         _fl(weight_decay), int(step), _fl(max_norm or 0.0), f32(partials, "partials"), f32(norm_out, "norm_out"),
         int(bool(write_clipped)), stream())


# ------------------------------------------------------------------------ toy config (S1)
def toy_rows(t, x, k, table, N):
    """rows (P, 1+Dc+E) = [t(jet), x, table[k]]: the input of the toy MLP (data_toys.ipynb cell 3).
    t: (B,) per jet or a 1-element tensor for every jet; x (P,Dc); k (P,) int64; table (V,E)."""
    P, Dc = x.shape
    V, E = table.shape
    rows = torch.empty(P, 1 + Dc + E, dtype=torch.float32, device=x.device)
    call("mmb_toy_rows", P, N, Dc, E, V, f32(t, "t"), 0 if t.numel() == 1 else 1, f32(x, "x"), i64(k, "k"),
         f32(table, "table"), f32(rows), stream())
    return rows


def toy_embedding_bwd(k, d_rows, col0, d_table):
    """d_table[k[p], :] += d_rows[p, col0:col0+E]"""
    V, E = d_table.shape
    if d_rows.stride(1) != 1:
        raise MMBError("toy_embedding_bwd: d_rows must have unit inner stride")
    call("mmb_toy_embedding_bwd", k.numel(), E, V, i64(k, "k"), d_rows.data_ptr(), d_rows.stride(0), col0,
         f32(d_table, "d_table"), stream())
    return d_table


def toy_sample(kind, n, device, V=8, noise=0.2, seed=0, offset=0):
    """kind 0: NGaussians, 1: TwoMoons (distributions/toys.py) -> x (n,2) f32, k (n,) int64, in index order."""
    x = torch.empty(n, 2, dtype=torch.float32, device=device)
    k = torch.empty(n, dtype=torch.int64, device=device)
    call("mmb_toy_sample", kind, n, V, _fl(noise), seed, offset, f32(x), i64(k), stream())
    return x, k


def toy_generate(mlp, mi, x, k, times, dt, thermo_kind, gamma, delta_n=None, seed=0, offset=0, em_sigma=None,
                 delta_w=None):
    """In-place generation loop of the toy MultiModalMLP on x (B,N,2) f32 and k (B,N,1) int64: one launch."""
    ws = [mlp.body[0].weight, mlp.body[0].bias, mlp.body[2].weight, mlp.body[2].bias, mlp.body[4].weight,
          mlp.body[4].bias, mlp.body[6].weight, mlp.body[6].bias, mlp.continous_head.weight, mlp.continous_head.bias,
          mlp.discrete_head.weight, mlp.discrete_head.bias]
    ptrs = (C.c_void_p * 12)()
    keep = []
    for i, w in enumerate(ws):
        t = w.detach()
        if not t.is_cuda or t.dtype != torch.float32:
            raise MMBError("toy_generate: parameters must be float32 HIP tensors (no CPU fallback)")
        t = t.contiguous()
        keep.append(t)
        ptrs[i] = t.data_ptr()
    table = mlp.embed.weight.detach().contiguous()
    call("mmb_toy_generate", mi.B, mi.N, mlp.dim_hidden, mlp.dim, mlp.vocab_size, f32(table, "table"), ptrs,
         f32(mi.maskf), f32(x, "x"), i64(k, "k"), times.numel(), f32(times, "times"), _fl(dt), thermo_kind, _fl(gamma),
         f32(delta_n, "delta_n"), seed, offset, 0 if em_sigma is None else 1, _fl(0.0 if em_sigma is None else em_sigma),
         f32(delta_w, "delta_w"), stream())


# ------------------------------------------------------------------------ AOJ reader
def aoj_features(pfcands):
    """raw PFCands (J,M,F) f32 on device -> (feats (J,M,3) = pt, eta_rel, phi_rel; tokens (J,M,1) i64; mask (J,M,1) i64)"""
    J, M, F = pfcands.shape
    feats = torch.empty(J, M, 3, dtype=torch.float32, device=pfcands.device)
    tokens = torch.empty(J, M, 1, dtype=torch.int64, device=pfcands.device)
    mask = torch.empty(J, M, 1, dtype=torch.int64, device=pfcands.device)
    call("mmb_aoj_features", J, M, F, f32(pfcands, "pfcands"), f32(feats), i64(tokens), i64(mask), stream())
    return feats, tokens, mask
