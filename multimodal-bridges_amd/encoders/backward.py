"""Reverse pass of the taped HIP forward (training).

The forward (encoders/epic.py, embedder.py) records one entry per launched op.  This module
replays the tape backwards with the hand-written gradient kernels of libmmb_hip.so:

  linear : mmb_act_bwd (epilogue/activation) -> mmb_grad_weight (dW, db; fp32 MFMA) ->
           mmb_weight_norm_bwd (dv, dg) -> per segment mmb_linear_cat with W^T (dX) or
           mmb_jet_sum + a per-jet mmb_linear_cat (gradient of the broadcast segments)
  pool   : mmb_pool_bwd
  embed  : mmb_grad_weight on the Fourier features / the raw kinematics

Gradients of activations live in a GradStore keyed by the activation's storage, so a column
slice of a buffer (the zero-copy concat of the two branch outputs) automatically gets the
matching slice of the buffer's gradient.  Parameters that take no part in the forward (the
truncated embedding, quirk Q1; the never-executed last fused layer, K2) keep grad None, as
in the reference.
"""
import os

import torch

from mmb_hip import ops
from mmb_hip.ops import SEG_ROW, SEG_JET, SEG_JET_MASK


GW_REPLICAS = 8


class GradStore:
    def __init__(self):
        self.bufs = {}
        self.relu_gates, self.gated = {}, set()         # (see _relu_gates)

    @staticmethod
    def _key(t):
        return t.untyped_storage().data_ptr()

    @staticmethod
    def _owns_storage(t):
        return t.is_contiguous() and t.storage_offset() == 0 and t.untyped_storage().nbytes() == t.numel() * 4

    def seed(self, t, g):
        """Gradient of a tensor that owns its whole storage."""
        self.bufs[self._key(t)] = (g.contiguous().reshape(-1), t)

    def view(self, t):
        key = self._key(t)
        if key not in self.bufs:
            n = t.untyped_storage().nbytes() // 4
            self.bufs[key] = (torch.zeros(n, dtype=torch.float32, device=t.device), t)
        return self.bufs[key][0].as_strided(t.shape, t.stride(), t.storage_offset())

    def get(self, t):
        if t is None:
            return None
        key = self._key(t)
        if key not in self.bufs:
            return None
        return self.bufs[key][0].as_strided(t.shape, t.stride(), t.storage_offset())

    def target(self, t):
        """(gradient view, fresh): fresh=True means the buffer was just created UNINITIALISED (t owns its
        whole storage), so the caller must overwrite instead of accumulate -- saves a zero-fill and an add."""
        key = self._key(t)
        if key not in self.bufs and self._owns_storage(t):
            self.bufs[key] = (torch.empty(t.numel(), dtype=torch.float32, device=t.device), t)
            return self.bufs[key][0].view(t.shape), True
        return self.view(t), False

    def add(self, t, g):
        """grad(t) += g.  When t has no gradient yet and owns its storage, g's buffer is ADOPTED (no copy):
        callers pass temporaries / gradients that nothing reads afterwards."""
        key = self._key(t)
        if key not in self.bufs and self._owns_storage(t) and g.is_contiguous() and g.numel() == t.numel() \
                and g.untyped_storage().nbytes() == g.numel() * 4:
            self.bufs[key] = (g.reshape(-1), t)
        else:
            self.view(t).add_(g)


class Arena:
    """One zero-filled flat buffer per backward pass for every dW/db accumulator (the weight-gradient
    kernel accumulates with atomics) instead of two fill launches per layer."""

    def __init__(self, n, device):
        self.buf = torch.zeros(n, dtype=torch.float32, device=device)
        self.off = 0

    def take(self, shape):
        n = 1
        for d in shape:
            n *= d
        pad = (-n) % 4
        if self.off + n > self.buf.numel():
            return torch.zeros(shape, dtype=torch.float32, device=self.buf.device)
        out = self.buf[self.off:self.off + n].view(shape)
        self.off += n + pad
        return out


def _acc_param(p, g, shared=False):
    """p.grad += g (torch.autograd semantics).  shared: g lives in memory that is re-used (the arena, a slice of a
    wider buffer), so it is copied when it has to BECOME the gradient -- and only then."""
    g = g.reshape(p.shape) if g.is_contiguous() else g
    if p.grad is None:
        p.grad = g.clone() if shared or g.untyped_storage().data_ptr() == p.untyped_storage().data_ptr() else g
    else:
        p.grad.add_(g)


def _produced(tape):
    keys = set()
    k = GradStore._key
    for e in tape:
        if e[0] in ("linear", "plinear"):
            for t in (e[9], e[10]):
                if t is not None:
                    keys.add(k(t))
        elif e[0] in ("attn", "ln"):
            keys.add(k(e[2]))
        elif e[0] == "pool":
            keys.add(k(e[2]))
        elif e[0] == "cat":
            keys.add(k(e[2]))
        elif e[0] == "toyin":
            keys.add(k(e[3]))
        elif e[0] == "embed":
            keys.add(k(e[4]))
            if e[5] is not None:
                keys.add(k(e[5]))
    return keys


def _relu_gates(tape):
    """{key of an activated layer's output: (act, reference tensor)} for the layers whose reverse pass can be folded
    into their consumer's input-gradient GEMM (mmb_linear_cat_gate): plain first stage (no residual, no second stage),
    read exactly once on the tape, and that by a Linear as a per-particle ROW segment."""
    if os.environ.get("MMB_RELU_GATE") == "0":         # A/B switch: mmb_act_bwd for every activation
        return {}
    k = GradStore._key
    uses, row_uses = {}, {}

    def use(t, row=False):
        if t is not None:
            uses[k(t)] = uses.get(k(t), 0) + 1
            if row:
                row_uses[k(t)] = row_uses.get(k(t), 0) + 1

    for e in tape:
        if e[0] in ("linear", "plinear"):
            for t, kind in e[3]:
                use(t, kind == SEG_ROW)
            use(e[5])
            use(e[7])
        elif e[0] in ("attn", "ln", "pool"):
            use(e[1])
        elif e[0] == "cat":
            for part in e[1]:
                use(part)
    gates = {}
    for e in tape:
        if e[0] in ("linear", "plinear") and e[4] in ("relu", "lrelu", "selu", "gelu") and e[5] is None \
                and e[6] is None and e[10] is None and e[9] is not None and GradStore._owns_storage(e[9]):
            ref = e[8] if e[4] == "gelu" else e[9]          # GELU' needs the pre-activation, the others y1's sign / value
            key = k(e[9])
            if ref is not None and uses.get(key, 0) == 1 and row_uses.get(key, 0) == 1:
                gates[key] = (e[4], ref, 1.0)
        # relu followed by a plain dropout (the transformer's FFN): y2 = keep * relu(z) / (1 - p) is positive exactly where
        # the gradient passes, so dZ = (consumer's input gradient) * [y2 > 0] / (1 - p) -- no Philox, no mmb_act_bwd pass
        elif e[0] in ("linear", "plinear") and e[4] == "relu" and e[5] is None and e[6] is None and e[7] is None \
                and e[9] is None and e[10] is not None and e[11] > 0.0 and GradStore._owns_storage(e[10]):
            key = k(e[10])
            if uses.get(key, 0) == 1 and row_uses.get(key, 0) == 1:
                gates[key] = ("relu", e[10], 1.0 / (1.0 - e[11]))
    return gates


def _direct(p, touched):
    """Always None: gradients ACCUMULATE into an existing .grad (torch.autograd semantics -- a second backward()
    without zeroing, or micro-batch accumulation, must sum).  An earlier version overwrote a pre-zeroed bucket here
    to save one add per parameter."""
    return None


def param_grads(layer, segs, dZ, S, Sm, arena, touched):
    """dW, db of one weight-normed Linear from its dZ (M,Nout) and, for the broadcast segments, the per-jet column
    sums of dZ (S unmasked for SEG_JET, Sm masked for SEG_JET_MASK), then through the weight norm into
    weight_v / weight_g / bias.  Returns the effective weight."""
    M, Nout = dZ.shape
    W, inv = layer.weight(want_inv=True)
    R = GW_REPLICAS if M >= 16384 else 1                # same-address atomics serialise: spread them
    Kall = W.shape[1]
    dWr = arena.take((R, Nout * Kall))
    dbr = arena.take((R, Nout))
    dW = dWr[0].view(Nout, Kall)
    db = dbr[0]
    rep = dict(replicas=R, rep_stride=dWr.stride(0), db_stride=dbr.stride(0))
    koff = 0
    first = True
    for t, kind in segs:                                # per-particle inputs stream; broadcast inputs go
        Ks = t.shape[1]                                 # through the per-jet sums of dZ (M = B rows)
        dWs = dW[:, koff:koff + Ks]
        if kind == SEG_ROW:
            ops.grad_weight_rows(t, dZ, dWs, db if first else None, **rep)
            first = False
        elif kind == SEG_JET:
            ops.grad_weight_rows(t, S, dWs, None)
        else:
            ops.grad_weight_rows(t, Sm, dWs, None)
        koff += Ks
    roff, rlen = layer._row_tables(W.device)
    prm = (layer.weight_v, layer.weight_g, layer.bias)
    outs = tuple(_direct(q, touched) for q in prm)
    res = ops.weight_norm_bwd_rep(roff, rlen, layer.weight_v.detach().reshape(-1),
                                  layer.weight_g.detach().reshape(-1), inv, dWr, dbr, out=outs)
    for q, o, r in zip(prm, outs, res):
        if o is None:                                   # no pre-zeroed bucket (or a second use): accumulate
            _acc_param(q, r)
    return W


def plain_param_grads(spec, segs, dZ, S, Sm, arena):
    """The same for a PlainLinear (encoders/particle_transformer.py): dW of the concatenated weight through the
    row-streaming kernel, replicas folded by mmb_sum_replicas, then split over the nn.Linear parameters (an
    identity part has none).  Returns the effective (concatenated) weight."""
    M, Nout = dZ.shape
    W = spec.weight()
    R = GW_REPLICAS if M >= 16384 else 1
    Kall = W.shape[1]
    dWr = arena.take((R, Nout * Kall))
    dbr = arena.take((R, Nout))
    dW = dWr[0].view(Nout, Kall)
    rep = dict(replicas=R, rep_stride=dWr.stride(0), db_stride=dbr.stride(0))
    koff = 0
    first = True
    for t, kind in segs:
        Ks = t.shape[1]
        dWs = dW[:, koff:koff + Ks]
        if kind == SEG_ROW:
            ops.grad_weight_rows(t, dZ, dWs, dbr[0] if first else None, **rep)
            first = False
        elif kind == SEG_JET:
            ops.grad_weight_rows(t, S, dWs, None)
        else:
            ops.grad_weight_rows(t, Sm, dWs, None)
        koff += Ks
    def direct(q):                                      # an existing contiguous .grad takes the replica sum itself
        return q is not None and q.grad is not None and q.grad.is_contiguous()

    own = False
    if R > 1:
        w0 = spec.parts[0][0]
        if len(spec.parts) == 1 and direct(w0):
            ops.sum_replicas(dWr, w0.grad.view(-1), accumulate=True)
            dW = None
        else:
            dW = ops.sum_replicas(dWr, torch.empty(Nout * Kall, dtype=torch.float32, device=dZ.device)).view(Nout, Kall)
            own = len(spec.parts) == 1
        if len(spec.biases) == 1 and direct(spec.biases[0]):
            ops.sum_replicas(dbr, spec.biases[0].grad.view(-1), accumulate=True)
            db = None
        else:
            db = ops.sum_replicas(dbr, torch.empty(Nout, dtype=torch.float32, device=dZ.device))
    else:
        db = dbr[0]
    koff = 0
    if dW is not None:
        for w, k in spec.parts:
            if w is not None:
                _acc_param(w, dW[:, koff:koff + k] if len(spec.parts) > 1 else dW, shared=not own)
            koff += k
    if db is not None:
        for i, b in enumerate(spec.biases):
            _acc_param(b, db, shared=True)
    return W


def _linear_bwd(e, store, produced, mi, seed, arena, touched):
    kind_, layer, N, segs, act, res1, post, res2, pre, y1, y2, p, off = e
    gY1, gY2 = store.get(y1), store.get(y2)
    if gY1 is None and gY2 is None:
        return
    ref = y1 if y1 is not None else y2
    M, Nout = ref.shape
    if act is None and gY2 is None:
        dZ = gY1                        # identity first stage, no gradient through the second: nothing to compute
    elif y1 is not None and GradStore._key(y1) in store.gated:
        dZ = gY1                        # its consumer's GEMM already applied relu' (mmb_linear_cat_gate)
    elif y1 is None and y2 is not None and GradStore._key(y2) in store.gated:
        dZ = gY2                        # ... and, for relu + dropout, the keep factor with it
    else:
        dZ = ops.act_bwd(M, Nout, gY1, gY2, y1, y2, res2, pre, act, post, p, seed, off)
    if res2 is not None and gY2 is not None and GradStore._key(res2) in produced:
        store.add(res2, gY2)            # y2's gradient is final here: nothing reads it after its producer
    if res1 is not None and GradStore._key(res1) in produced:
        if any(GradStore._key(t) == GradStore._key(res1) for t, _ in segs):
            store.view(res1).add_(dZ)   # (the input gradient below is written into the same buffer)
        else:
            store.add(res1, dZ)         # dZ is only READ below
    # ---- parameters
    maskf = mi.maskf
    B = mi.B
    kinds = {k for _, k in segs}
    S = ops.jet_sum(dZ, B, N) if SEG_JET in kinds else None
    Sm = ops.jet_sum(dZ, B, N, maskf=maskf) if SEG_JET_MASK in kinds else None
    if kind_ == "plinear":
        W = plain_param_grads(layer, segs, dZ, S, Sm, arena)
    else:
        W = param_grads(layer, segs, dZ, S, Sm, arena, touched)
    # ---- inputs
    need = [GradStore._key(t) in produced for t, _ in segs]
    if not any(need):
        return
    WT = W.t().contiguous()
    koff = 0
    for (t, kind), nd in zip(segs, need):
        Ks = t.shape[1]
        if nd:
            gx, fresh = store.target(t)
            res = None if fresh else gx
            Wseg = WT[koff:koff + Ks]
            if kind == SEG_ROW and fresh and GradStore._key(t) in store.relu_gates:
                g_act, g_ref, g_scale = store.relu_gates[GradStore._key(t)]
                Wg = Wseg if g_scale == 1.0 else (Wseg * g_scale).contiguous()         # (a d x ff matrix: one tiny launch)
                ops.linear_cat_gate([(dZ, SEG_ROW)], Wg, N, g_ref, gx, act=g_act)      # dZ of the activated layer
                store.gated.add(GradStore._key(t))
            elif kind == SEG_ROW:
                ops.linear_cat([(dZ, SEG_ROW)], Wseg, None, N, residual=res, out=gx)
            else:
                red = S if kind == SEG_JET else Sm
                ops.linear_cat([(red, SEG_ROW)], Wseg, None, 1, residual=res, out=gx)
        koff += Ks


def _attn_bwd(e, store, mi, seed=0):
    _, qkv, a, lse, nheads = e[:5]
    p, off = (e[5], e[6]) if len(e) > 5 else (0.0, 0)
    bits = e[7] if len(e) > 7 else None
    g = store.get(a)
    if g is not None:
        store.add(qkv, ops.attention_bwd(qkv, mi, nheads, a, g, lse, valid_is_zero=True, drop_p=p, seed=seed, offset=off,
                                         keep_bits=bits))


def _ln_bwd(e, store, produced, arena):
    _, x, y, stats, norm = e
    g = store.get(y)
    if g is None:
        return
    D = x.shape[1]
    gw, gb = norm.weight.grad, norm.bias.grad
    direct = gw is not None and gb is not None and gw.is_contiguous() and gb.is_contiguous()
    if direct:                                   # the kernel ACCUMULATES (atomics): straight into the existing .grad
        dgamma, dbeta = gw, gb
    else:
        dgamma, dbeta = arena.take((D,)), arena.take((D,))
    if GradStore._key(x) in produced:
        gx, fresh = store.target(x)
    else:
        gx, fresh = torch.empty_like(x), True
    ops.layernorm_bwd(x, norm.weight.detach(), stats, g, gx, not fresh, dgamma, dbeta)
    if not direct:
        _acc_param(norm.weight, dgamma, shared=True)
        _acc_param(norm.bias, dbeta, shared=True)


def _embed_bwd(e, store, mi):
    _, emb, x, mi_, temb, cont, feat = e
    g_t = store.get(temb)
    lin = emb.time_embedding[1]
    if g_t is not None:
        dW = torch.zeros_like(lin.weight)
        db = torch.zeros_like(lin.bias)
        ops.grad_weight([(feat, SEG_ROW)], 1, None, g_t.contiguous(), dW, db)
        _acc_param(lin.weight, dW)
        _acc_param(lin.bias, db)
    g_c = store.get(cont) if cont is not None else None
    if g_c is not None:
        ce = emb.continuous_embedding
        d_pre = g_c * mi_.maskf.reshape(-1, 1)
        dW = torch.zeros_like(ce.weight)
        db = torch.zeros_like(ce.bias)
        xr = x.reshape(-1, x.shape[-1])
        if xr.shape[1] <= 4 and d_pre.shape[1] <= 32:
            ops.grad_weight_smallk(xr, d_pre, dW, db)
        else:
            ops.grad_weight([(xr, SEG_ROW)], mi_.N, None, d_pre, dW, db)
        _acc_param(ce.weight, dW)
        _acc_param(ce.bias, db)


def run_backward(model, tape, heads, d_hc, d_hd, d_s=None, scale=None):
    """Fill .grad of every parameter that took part in the taped forward."""
    plan = next((e[1] for e in tape if e[0] == "fused_plan"), None)
    if plan is not None and getattr(model, "use_fused_backward", True):
        from encoders import fused_train
        if scale is not None:
            d_hc = None if d_hc is None else d_hc * scale
            d_hd = None if d_hd is None else d_hd * scale
            d_s = None if d_s is None else d_s * scale
        fused_train.backward(model, plan, d_hc, d_hd)
        lw = getattr(model.loss_multimode, "loss_weights", None)
        if d_s is not None and isinstance(lw, torch.nn.Parameter):
            _acc_param(lw, d_s)
        return
    store = GradStore()
    if scale is not None:
        d_hc = None if d_hc is None else d_hc * scale
        d_hd = None if d_hd is None else d_hd * scale
        d_s = None if d_s is None else d_s * scale
    if d_hc is not None:
        store.seed(heads.continuous, d_hc.reshape(-1, d_hc.shape[-1]))
    if d_hd is not None:
        store.seed(heads.discrete, d_hd.reshape(-1, d_hd.shape[-1]))
    produced = _produced(tape)
    store.relu_gates, store.gated = _relu_gates(tape), set()
    mi = next(e[3] if e[0] == "embed" else e[2] for e in tape if e[0] in ("embed", "toyin"))
    seed = model.seed
    n_acc = sum(GW_REPLICAS * (e[1].weight_v.numel() + e[1].weight_v.shape[0] + 8) for e in tape if e[0] == "linear")
    n_acc += sum(GW_REPLICAS * (e[1].dim_out * (e[1].K + 1) + 8) for e in tape if e[0] == "plinear")
    n_acc += sum(2 * e[1].shape[1] + 8 for e in tape if e[0] == "ln")
    arena = Arena(n_acc, mi.maskf.device)
    touched = set()
    for e in reversed(tape):
        kind = e[0]
        if kind in ("linear", "plinear"):
            _linear_bwd(e, store, produced, mi, seed, arena, touched)
        elif kind == "attn":
            _attn_bwd(e, store, mi, seed)
        elif kind == "ln":
            _ln_bwd(e, store, produced, arena)
        elif kind == "pool":
            gP = store.get(e[2])
            if gP is not None and GradStore._key(e[1]) in produced:
                gx, fresh = store.target(e[1])
                ops.pool_bwd(gP, mi, gx, not fresh)
        elif kind == "cat":
            g = store.get(e[2])
            if g is not None:
                o = 0
                for part in e[1]:
                    w = part.shape[1]
                    if GradStore._key(part) in produced:
                        store.view(part).add_(g[:, o:o + w])
                    o += w
        elif kind == "embed":
            _embed_bwd(e, store, mi)
        elif kind == "toyin":                           # input rows [t, x, embed(k)] of the toy MLP: nn.Embedding's gradient
            _, enc, _mi, rows, k_rows, col0 = e
            g = store.get(rows)
            if g is not None:
                d_table = torch.zeros_like(enc.embed.weight)
                ops.toy_embedding_bwd(k_rows, g, col0, d_table)
                _acc_param(enc.embed.weight, d_table)
    lw = getattr(model.loss_multimode, "loss_weights", None)
    if d_s is not None and isinstance(lw, torch.nn.Parameter):
        _acc_param(lw, d_s)


class _TapeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, loss_val, holder):
        ctx.holder = holder
        return loss_val.clone()

    @staticmethod
    def backward(ctx, g):
        h = ctx.holder
        ctx.holder = None
        run_backward(h["model"], h["tape"], h["heads"], h["d_hc"], h["d_hd"], h["d_s"], scale=g)
        return None, None, None


def attach_backward(model, loss, tape, heads, d_hc, d_hd, d_s):
    """Return the scalar loss wired into autograd so that `loss.backward()` replays the tape."""
    anchor = torch.zeros(1, device=loss.device, requires_grad=True)
    holder = {"model": model, "tape": tape, "heads": heads, "d_hc": d_hc, "d_hd": d_hd, "d_s": d_s}
    return _TapeFn.apply(anchor, loss[0].detach(), holder)
