"""Training step of MultiModalEPiC through the fused per-jet kernels (csrc/fused_epic.hip,
csrc/fused_epic_bwd.hip).

forward():  ONE launch (mmb_fused_epic_train_forward) computes both heads and writes every activation the
            reverse pass reads.  This module lays out the destinations ("slots", in the kernel's execution
            order) and records the SAME tape the layered forward (encoders/epic.py, multimodal_epic.py)
            would have produced, so the layered reverse pass (encoders/backward.py) still applies.
backward(): ONE launch (mmb_fused_epic_train_backward) runs the reverse pass of the whole encoder with the
            activation gradients in registers and leaves, per weight-normed Linear, its dZ and the per-jet
            column sums of dZ; the weight-gradient / weight-norm kernels then turn those into parameter
            gradients exactly as the layered reverse pass does.

Reference: encoders/multimodal_epic.py:158-194, encoders/epic.py:71-107,145-159,187-208 and their autograd."""
import os

import torch

from mmb_hip import fused as fused_epic
from mmb_hip.ops import SEG_ROW, SEG_JET, SEG_JET_MASK
from tensorclass import TensorMultiModal


def _slot(t):
    return (t.data_ptr(), t.stride(0), t.untyped_storage().nbytes() // 4 - t.storage_offset())


class _Plan:
    def __init__(self, B, N, dev, tape, drop_p, offset0):
        self.B, self.N, self.M, self.dev, self.tape = B, N, B * N, dev, tape
        self.slots = []
        self.drop_p, self.offset0, self.off = drop_p, offset0, offset0
        self.enc = []                # per encoder (continuous, discrete, fused): dict(proj, layers, out)
        self.heads = []
        self.sign_words = 2 * 64 * (4 if N <= 128 else 5)     # floats: the kernels run 4 or 5 waves per jet

    def slot(self, t):
        self.slots.append(_slot(t))
        return t

    def rows(self, width):                       # one row per particle
        # zeros, not empty: the kernels leave the rows of padded slots unwritten (whole idle tiles and the padded slots
        # of working tiles); whoever reads them multiplies by gradient rows that are exactly zero -- they must be finite
        return self.slot(torch.zeros(self.M, width, dtype=torch.float32, device=self.dev))

    def jets(self, width):                       # one row per jet
        return self.slot(torch.empty(self.B, width, dtype=torch.float32, device=self.dev))

    def next_offset(self):
        if self.drop_p > 0.0:
            self.off += 1
            return self.drop_p, self.off
        return 0.0, 0

    def linear(self, layer, n, segs, act=None, res1=None, post=None, res2=None, pre=None, y1=None, y2=None, p=0.0,
               off=0):
        self.tape.append(("linear", layer, n, segs, act, res1, post, res2, pre, y1, y2, p, off))
        return {"layer": layer, "segs": segs, "pre": pre, "y1": y1, "y2": y2}


def _encoder(pl, enc, tseg, x_local, x_global, out, out_slot=True):
    """x_local: a (rows, width) tensor, or the list of segments the projection concatenates after the time"""
    N = pl.N
    rec = {"proj": None, "layers": [], "skip": None}
    if hasattr(enc, "epic_proj"):
        pr = enc.epic_proj
        H, G = pr.mlp_local["0"].dim_out, pr.mlp_global["2"].dim_out
        pre, h = pl.rows(H), pl.rows(H)
        loc = x_local if isinstance(x_local, list) else [(x_local, SEG_ROW)]
        p0 = pl.linear(pr.mlp_local["0"], N, [tseg] + loc, act="gelu", pre=pre, y1=h)
        pre, h2 = pl.rows(H), pl.rows(H)
        p2 = pl.linear(pr.mlp_local["2"], N, [(h, SEG_ROW)], act="gelu", pre=pre, y1=h2)
        pooled = pl.jets(2 * H)
        pl.tape.append(("pool", h2, pooled))
        pre, g = pl.jets(H), pl.jets(H)
        q0 = pl.linear(pr.mlp_global["0"], 1, [(pooled, SEG_ROW), (x_global, SEG_ROW)], act="gelu", pre=pre, y1=g)
        pre, g2 = pl.jets(G), pl.jets(G)
        q2 = pl.linear(pr.mlp_global["2"], 1, [(g, SEG_ROW)], act="gelu", pre=pre, y1=g2)
        rec["proj"] = {"p0": p0, "p2": p2, "q0": q0, "q2": q2}
        x_local, x_global = h2, g2
    skip_l = x_local if enc.use_skip_connection else None
    skip_g = x_global if enc.use_skip_connection else None
    rec["skip"] = skip_l
    for i in range(enc.num_blocks):
        ly = enc.epic_layers[i]
        H, G = ly.fc_loc2.dim_out, ly.fc_glob2.dim_out
        pooled = pl.jets(2 * x_local.shape[1])
        pl.tape.append(("pool", x_local, pooled))
        gh = pl.jets(ly.fc_glob1.dim_out)
        g1 = pl.linear(ly.fc_glob1, 1, [(pooled, SEG_ROW), (x_global, SEG_ROW)], act="lrelu", y1=gh)
        g_new, g_out = pl.jets(G), pl.jets(G)
        p, off = pl.next_offset()
        g2 = pl.linear(ly.fc_glob2, 1, [(gh, SEG_ROW)], res1=x_global, post="lrelu", res2=skip_g, y1=g_new, y2=g_out,
                       p=p, off=off)
        sgn_lh = pl.jets(pl.sign_words)          # one 64-bit word per thread of the jet's workgroup: sign bits for the reverse kernel
        lh = pl.rows(ly.fc_loc1.dim_out)
        l1 = pl.linear(ly.fc_loc1, N, [tseg, (x_local, SEG_ROW), (g_new, SEG_JET)], act="lrelu", y1=lh)
        sgn_y1 = pl.jets(pl.sign_words)
        l_out = pl.rows(H)
        p, off = pl.next_offset()
        l2 = pl.linear(ly.fc_loc2, N, [(lh, SEG_ROW)], res1=x_local, post="lrelu", res2=skip_l, y2=l_out, p=p, off=off)
        rec["layers"].append({"g1": g1, "g2": g2, "l1": l1, "l2": l2, "sgn_lh": sgn_lh, "sgn_y1": sgn_y1})
        x_local, x_global = l_out, g_out
    if out_slot:
        pl.slot(out)
    rec["out"] = pl.linear(enc.output_layer, N, [tseg, (x_local, SEG_ROW), (x_global, SEG_JET)], y1=out)
    pl.enc.append(rec)
    return out, x_global


def _plan_forward(model, pl, temb, cont, disc, hc, hd):
    """Lay out every activation of one training forward (slots in kernel order, tape records in layered order)."""
    enc = model.encoder
    dev = pl.dev
    pl.kind = "uni" if hasattr(enc, "epic") else "multi"
    if pl.kind == "uni":                                   # UniModalEPiC: cat[t, t, disc] -> one EPiC -> logits
        tseg = (temb, SEG_JET_MASK)
        if hd is None:
            hd = torch.empty(pl.M, model.config.data.vocab_size, dtype=torch.float32, device=dev)
        _encoder(pl, enc.epic, tseg, [tseg, (disc, SEG_ROW)], temb, hd, out_slot=False)
        pl.table = torch.tensor(pl.slots, dtype=torch.int64).to(dev, non_blocking=True)
        pl.keep = ()
        return
    H0, H1, H2, half = enc.dims
    tseg = (temb, SEG_JET_MASK)
    both = torch.empty(pl.M, H0 + H1, dtype=torch.float32, device=dev)
    h1, g1 = _encoder(pl, enc.continuous_encoder, tseg, cont, temb, both[:, :H0])
    h2, g2 = _encoder(pl, enc.discrete_encoder, tseg, disc, temb, both[:, H0:])
    g = pl.jets(g1.shape[1] + g2.shape[1])
    pl.tape.append(("cat", [g1, g2], g))
    f = torch.empty(pl.M, H2, dtype=torch.float32, device=dev)
    _encoder(pl, enc.fused_encoder, tseg, both, g, f)
    split = (H2 + 1) // 2
    f1, f2 = f[:, :split], f[:, split:]
    for head, fi, hi, y in ((enc.continuous_head, f1, h1, hc), (enc.discrete_head, f2, h2, hd)):
        pre, h = pl.rows(head["0"].dim_out), pl.rows(head["0"].dim_out)
        r0 = pl.linear(head["0"], pl.N, [tseg, (fi, SEG_ROW), (hi, SEG_ROW)], act="gelu", pre=pre, y1=h)
        r2 = pl.linear(head["2"], pl.N, [(h, SEG_ROW)], y1=y)
        pl.heads.append({"0": r0, "2": r2})
    pl.both_width = H0 + H1
    pl.table = torch.tensor(pl.slots, dtype=torch.int64).to(dev, non_blocking=True)
    pl.keep = (both, f, g)


def forward(model, pack, loc, x, k, tape, seed, offset0):
    """-> heads.  loc: the LocalState of embedder.embed (its tape entry is already recorded); x, k: the
    raw bridge state the kernel embeds itself."""
    mi = loc.mi
    B, N = mi.B, mi.N
    dev = k.device
    drop_p = float(model.config.encoder.dropout or 0.0) if model.training else 0.0
    pl = _Plan(B, N, dev, tape, drop_p, offset0)
    uni = pack["kind"] == "uni"
    hc = None if uni else torch.empty(B * N, 3, dtype=torch.float32, device=dev)
    hd = torch.empty(B * N, 8, dtype=torch.float32, device=dev)
    _plan_forward(model, pl, loc.temb, loc.cont_rows, loc.disc_rows, hc, hd)
    if uni:
        fused_epic.train_forward_uni(pack, mi, k, loc.temb, pl.table, len(pl.slots), drop_p, seed, offset0, hd)
    else:
        fused_epic.train_forward(pack, mi, x, k, loc.temb, pl.table, len(pl.slots), drop_p, seed, offset0, hc, hd)
    pl.pack, pl.mi, pl.seed = pack, mi, seed
    pl.embed = next(e for e in tape if e[0] == "embed")
    tape.append(("fused_plan", pl))
    return TensorMultiModal(None, None if uni else hc.view(B, N, -1), hd.view(B, N, -1), loc.mask)


class _Grads:
    """GradStore look-alike for _embed_bwd: gradients by tensor identity."""

    def __init__(self):
        self.by_id = {}

    def put(self, t, g):
        self.by_id[id(t)] = g

    def get(self, t):
        return None if t is None else self.by_id.get(id(t))


def _layers_bwd(rec, use, rows, jets, jobs):
    """slots + jobs of the executed EPiC layers of one encoder, last layer first (the kernel's order)"""
    for lr in reversed(rec["layers"]):
        H = lr["l2"]["y2"].shape[1]
        G = lr["g2"]["y1"].shape[1]
        use(lr["sgn_y1"])
        dz_l2 = rows(H)
        use(lr["sgn_lh"])
        dz_l1, S_l1 = rows(H), jets(H)
        use(lr["g2"]["y1"])
        dz_g2 = jets(G)
        use(lr["g1"]["y1"])
        dz_g1 = jets(lr["g1"]["y1"].shape[1])
        jobs.extend([(lr["l2"], dz_l2, None), (lr["l1"], dz_l1, S_l1), (lr["g2"], dz_g2, None),
                     (lr["g1"], dz_g1, None)])


def _plan_backward(pl, d_heads, cont_width):
    """Slots of the backward kernel (its walking order) + the weight-gradient jobs (record, dZ, column sums)."""
    B, N, M, dev = pl.B, pl.N, pl.M, pl.dev
    slots, jobs = [], []

    def rows(w):
        t = torch.empty(M, w, dtype=torch.float32, device=dev)
        slots.append(_slot(t))
        return t

    def jets(w):
        t = torch.empty(B, w, dtype=torch.float32, device=dev)
        slots.append(_slot(t))
        return t

    def use(t):
        slots.append(_slot(t))
        return t

    Et = pl.enc[0]["out"]["segs"][0][0].shape[1]

    def proj_bwd(rec, Hb):
        pr = rec["proj"]
        use(pr["q2"]["pre"])
        dz_q2 = jets(pr["q2"]["pre"].shape[1])
        use(pr["q0"]["pre"])
        dz_q0 = jets(pr["q0"]["pre"].shape[1])
        use(pr["p2"]["pre"])
        dz_p2 = rows(Hb)
        use(pr["p0"]["pre"])
        dz_p0, S_p0 = rows(Hb), jets(Hb)
        jobs.extend([(pr["q2"], dz_q2, None), (pr["q0"], dz_q0, None), (pr["p2"], dz_p2, None), (pr["p0"], dz_p0, S_p0)])

    if pl.kind == "uni":
        d_temb = jets(Et)
        rec = pl.enc[0]
        d_hd = use(d_heads[-1])
        jobs.append((rec["out"], d_hd, jets(d_hd.shape[1])))
        _layers_bwd(rec, use, rows, jets, jobs)
        proj_bwd(rec, rec["proj"]["p2"]["pre"].shape[1])
        table = torch.tensor(slots, dtype=torch.int64).to(dev, non_blocking=True)
        return {"table": table, "n": len(slots), "jobs": jobs, "d_temb": d_temb, "d_cont": None, "keep": ()}

    W2 = pl.both_width
    d_f, d_hb, d_both, d_temb = rows(W2), rows(W2), rows(W2), jets(Et)
    for hr, dh in zip(pl.heads, d_heads):
        use(dh)
        use(hr["0"]["pre"])
        Hh = hr["0"]["pre"].shape[1]
        dz, S = rows(Hh), jets(Hh)
        jobs.append((hr["2"], dh, None))
        jobs.append((hr["0"], dz, S))

    def layers_bwd(rec):
        _layers_bwd(rec, use, rows, jets, jobs)

    fe = pl.enc[2]
    jobs.append((fe["out"], d_f, jets(W2)))
    layers_bwd(fe)
    d_cont = None
    for br in (0, 1):
        rec = pl.enc[br]
        Hb = rec["out"]["y1"].shape[1]
        jobs.append((rec["out"], d_both[:, br * Hb:(br + 1) * Hb], jets(Hb)))
        layers_bwd(rec)
        proj_bwd(rec, Hb)
        if br == 0:
            d_cont = rows(cont_width)
    table = torch.tensor(slots, dtype=torch.int64).to(dev, non_blocking=True)
    return {"table": table, "n": len(slots), "jobs": jobs, "d_temb": d_temb, "d_cont": d_cont,
            "keep": (d_f, d_hb, d_both)}


def _embed_grads(pl_embed, mi, d_temb, d_cont):
    from encoders import backward as bw
    gs = _Grads()
    gs.put(pl_embed[4], d_temb)
    if pl_embed[5] is not None:
        gs.put(pl_embed[5], d_cont)
    bw._embed_bwd(pl_embed, gs, mi)


def backward(model, pl, d_hc, d_hd):
    """Parameter gradients of the fused training forward `pl` given the head gradients (B,N,3), (B,N,8)."""
    if isinstance(pl, WorkspaceStep):
        return pl.ws.backward(model, pl, d_hc, d_hd)
    from encoders import backward as bw
    M, dev = pl.M, pl.dev
    pack, mi = pl.pack, pl.mi
    d_heads = tuple(d.reshape(M, -1).contiguous() for d in (d_hc, d_hd) if d is not None)
    bp = _plan_backward(pl, d_heads, 0 if pl.embed[5] is None else pl.embed[5].shape[1])
    pbwd = fused_epic.packed_bwd(pack, fused_epic.params_version(model))
    fused_epic.train_backward(pack, pbwd, mi, bp["table"], bp["n"], pl.drop_p, pl.seed, pl.offset0)
    # ---- parameter gradients: dW = dZ^T X per segment (broadcast segments: column sums, M = B rows), weight norm
    jobs = bp["jobs"]
    n_acc = sum(bw.GW_REPLICAS * (r["layer"].weight_v.numel() + r["layer"].weight_v.shape[0] + 8) for r, _, _ in jobs)
    arena = bw.Arena(n_acc, dev)
    touched = set()
    for r, dZ, S in jobs:
        bw.param_grads(r["layer"], r["segs"], dZ, S, S, arena, touched)
    _embed_grads(pl.embed, mi, bp["d_temb"], bp["d_cont"])


# ------------------------------------------------------------------------------------------------------
class WorkspaceStep:
    """What one training step leaves on the tape for the reverse pass."""

    def __init__(self, ws, mi, embed, drop_p, seed, offset0):
        self.ws, self.mi, self.embed, self.drop_p, self.seed, self.offset0 = ws, mi, embed, drop_p, seed, offset0


class Workspace:
    """Persistent buffers and device tables of the fused training step for one (B, N): the activations of the
    forward, every dZ of the backward, the weight-gradient accumulators, and the tables of the multi-problem
    launches (weight gradients per tile class, weight-norm backward).  A step is then ~25 launches and no
    allocation: embed, 3 copies, forward, loss, 2 scalings, backward, one memset, <= 8 + 2 weight-gradient
    launches, one weight-norm backward, the embedding gradients.  The step's activations are overwritten by
    the next step: the reverse pass of a step must run before the next forward (the training loop's order)."""

    def __init__(self, model, pack, B, N, dev):
        from encoders import backward as bw
        from mmb_hip import ops
        e = model.config.encoder
        M = B * N
        self.key = (B, N, str(dev))
        self.uni = pack["kind"] == "uni"
        self.temb = torch.empty(B, e.dim_emb_time, dtype=torch.float32, device=dev)
        self.cont = None if self.uni else torch.empty(M, e.dim_emb_continuous, dtype=torch.float32, device=dev)
        self.disc = torch.empty(M, e.dim_emb_discrete, dtype=torch.float32, device=dev)
        # (B*N, 4) with a zero 4th column: 16-byte rows put the continuous head's weight gradient on the row-streaming
        # kernel (a 12-byte pitch falls back to the generic one)
        self.d_hc = None if self.uni else torch.zeros(M, 4, dtype=torch.float32, device=dev)
        self.d_hd = torch.empty(M, 8, dtype=torch.float32, device=dev)
        pl = _Plan(B, N, dev, [], 0.0, 0)
        _plan_forward(model, pl, self.temb, self.cont, self.disc, None, self.d_hd if self.uni else None)
        self.pl = pl
        d_heads = (self.d_hd,) if self.uni else (self.d_hc, self.d_hd)
        self.bp = _plan_backward(pl, d_heads, 0 if self.uni else self.cont.shape[1])
        # ---- weight-gradient problems and their accumulators
        jobs = self.bp["jobs"]
        sizes = []
        for r, dZ, S in jobs:
            Mr, Nout = dZ.shape
            R = bw.GW_REPLICAS if Mr >= 16384 else 1
            Kall = r["layer"].dim_in
            sizes.append((R, Nout, Kall))
        pad4 = lambda n: n + (-n) % 4
        total = sum(R * pad4(Nout * Kall) + R * pad4(Nout) for R, Nout, Kall in sizes)
        self.arena = torch.zeros(total, dtype=torch.float32, device=dev)
        probs, self.generic, self.layers = {}, [], []
        off = 0
        for (r, dZ, S), (R, Nout, Kall) in zip(jobs, sizes):
            dWr = self.arena[off:off + R * pad4(Nout * Kall)].view(R, pad4(Nout * Kall))
            off += R * pad4(Nout * Kall)
            dbr = self.arena[off:off + R * pad4(Nout)].view(R, pad4(Nout))
            off += R * pad4(Nout)
            dW = dWr[0, :Nout * Kall].view(Nout, Kall)
            koff, first = 0, True
            for t, kind in r["segs"]:
                Ks = t.shape[1]
                dWs = dW[:, koff:koff + Ks]
                row_seg = kind == SEG_ROW
                Z = dZ if row_seg else S
                db = dbr[0] if (row_seg and first) else None
                reps = R if row_seg else 1
                if row_seg:
                    first = False
                if ops.gw_fast(t, Z):
                    probs.setdefault(ops.gw_class(Ks, Nout), []).append((t, Z, dWs, db, reps, dWr.stride(0),
                                                                         dbr.stride(0)))
                else:
                    self.generic.append((t, Z, dWs, db, reps, dWr.stride(0), dbr.stride(0)))
                koff += Ks
            self.layers.append((r["layer"], dWr, dbr, R))
        # per-particle problems reduce over the rows of real particles only (their dZ rows are the only non-zero ones):
        # the list is refilled from the mask of every step (backward), the table points at these two buffers
        self.rows = torch.zeros(M, dtype=torch.int32, device=dev)
        self.n_rows = torch.zeros(1, dtype=torch.int32, device=dev)
        self.use_rows = os.environ.get("MMB_GW_ROWS", "1") != "0"
        self.gw = []
        for cls, plist in sorted(probs.items()):
            rpb = ops.gw_rows_per_block([(q[0].shape[0], q[0].shape[1], q[1].shape[1]) for q in plist])
            recs, nblk = [], 0
            for q in plist:
                per_particle = self.use_rows and q[0].shape[0] == M and M > B
                rec, n = ops.gw_record(*q, nblk, rpb_big=rpb, rows=self.rows if per_particle else None,
                                       n_rows=self.n_rows if per_particle else None)
                recs.append(rec)
                nblk += n
            self.gw.append((cls, torch.tensor(recs, dtype=torch.int64).to(dev), len(recs), nblk))
        # When every reader of the kernel-written per-particle dZ slots goes through the row list, the reverse kernel's
        # idle tiles (padded slots only) need not store their zero rows (MMB_BWD_DEAD_ROWS=0: developer switch).  The
        # head gradients come from the loss kernel, which writes every row; a generic launch reads every row of its Z.
        outside = {t.data_ptr() for t in d_heads}
        self.real_rows_only = (self.use_rows and M > B and os.environ.get("MMB_BWD_DEAD_ROWS", "1") != "0"
                               and all(q[1].shape[0] != M or q[1].data_ptr() in outside for q in self.generic))
        self.pack = pack
        self.grad_ptrs = None
        self.own_grads = None
        self.own_views = {}
        self.inv_base = None

    # ---- weight-norm backward table: rebuilt whenever a parameter's .grad buffer or storage moved
    def _params(self):
        for layer, _, _, _ in self.layers:
            yield layer.weight_v
            yield layer.weight_g
            yield layer.bias

    def _wn_table(self, model):
        own = self.own_views
        back = [p for p in self._params() if p.grad is None and id(p) in own]
        if back:                                           # zero_grad(set_to_none=True) loops: hand the buffers back,
            whole = (len(back) == len(own) and self.own_grads is not None       # zeroed: the kernel ACCUMULATES
                     and self.own_grads.numel() == sum(v.numel() for v in own.values()))   # into them (None == 0)
            if whole:
                self.own_grads.zero_()
            for p in back:
                p.grad = own[id(p)]
                if not whole:
                    p.grad.zero_()
        ptrs = tuple((p.data_ptr(), -1 if p.grad is None else p.grad.data_ptr()) for p in self._params())
        if ptrs == self.grad_ptrs:
            return
        missing = [p for p in self._params() if p.grad is None or not p.grad.is_contiguous()]
        if missing:                                        # no flat bucket: own gradient buffers
            flat = torch.zeros(sum(p.numel() for p in missing), dtype=torch.float32, device=self.arena.device)
            o = 0
            for p in missing:
                p.grad = own[id(p)] = flat[o:o + p.numel()].view_as(p)
                o += p.numel()
            self.own_grads = flat
        ar = lambda n: torch.arange(n, dtype=torch.int64)
        recs = []
        for layer, dWr, dbr, R in self.layers:
            W, inv = layer.weight(want_inv=True)
            O, K = layer.dim_out, layer.dim_in
            r = torch.empty(O, 12, dtype=torch.int64)
            r[:, 0] = layer.weight_v.data_ptr() + 4 * K * ar(O)
            r[:, 1] = layer.weight_g.data_ptr() + 4 * ar(O)
            r[:, 2] = inv.data_ptr() + 4 * ar(O)
            r[:, 3] = dWr.data_ptr() + 4 * K * ar(O)
            r[:, 4] = dbr.data_ptr() + 4 * ar(O)
            r[:, 5] = layer.weight_v.grad.data_ptr() + 4 * K * ar(O)
            r[:, 6] = layer.weight_g.grad.data_ptr() + 4 * ar(O)
            r[:, 7] = layer.bias.grad.data_ptr() + 4 * ar(O)
            r[:, 8], r[:, 9], r[:, 10], r[:, 11] = K, R, dWr.stride(0), dbr.stride(0)
            recs.append(r)
        self.wn_rows = torch.cat(recs).to(self.arena.device)
        self.inv_base = self.pack["inv"].data_ptr()      # (every layer's inv is a slice of it after refresh_flat)
        self.grad_ptrs = tuple((p.data_ptr(), p.grad.data_ptr()) for p in self._params())

    def forward(self, model, loc, x, k, tape, seed, offset0):
        mi = loc.mi
        B, N = mi.B, mi.N
        dev = k.device
        drop_p = float(model.config.encoder.dropout or 0.0) if model.training else 0.0
        self.temb.copy_(loc.temb)
        self.disc.copy_(loc.disc_rows)
        hd = torch.empty(B * N, 8, dtype=torch.float32, device=dev)
        pl = self.pl
        embed = next(e for e in tape if e[0] == "embed")
        tape.append(("fused_plan", WorkspaceStep(self, mi, embed, drop_p, seed, offset0)))
        if self.uni:
            fused_epic.train_forward_uni(self.pack, mi, k, self.temb, pl.table, len(pl.slots), drop_p, seed, offset0, hd)
            return TensorMultiModal(None, None, hd.view(B, N, -1), loc.mask)
        self.cont.copy_(loc.cont_rows)
        hc = torch.empty(B * N, 3, dtype=torch.float32, device=dev)
        fused_epic.train_forward(self.pack, mi, x, k, self.temb, pl.table, len(pl.slots), drop_p, seed, offset0, hc, hd)
        return TensorMultiModal(None, hc.view(B, N, -1), hd.view(B, N, -1), loc.mask)

    def backward(self, model, st, d_hc, d_hd):
        from mmb_hip import ops
        M = self.pl.M
        if not self.uni:
            self.d_hc[:, :3].copy_(d_hc.reshape(M, -1))
        self.d_hd.copy_(d_hd.reshape(M, -1))
        pack = self.pack
        pbwd = fused_epic.packed_bwd(pack, fused_epic.params_version(model))
        bp = self.bp
        fused_epic.train_backward(pack, pbwd, st.mi, bp["table"], bp["n"], st.drop_p, st.seed, st.offset0,
                                  real_rows_only=self.real_rows_only)
        self.arena.zero_()
        if self.use_rows:
            ops.mask_rows(st.mi, self.rows)
            self.n_rows.copy_(st.mi.total)
        for cls, table, n, nblk in self.gw:
            ops.grad_weight_rows_multi(cls, table, n, nblk)
        for X, Z, dWs, db, reps, rs, ds in self.generic:
            ops.grad_weight_rows(X, Z, dWs, db, replicas=reps, rep_stride=rs, db_stride=ds)
        if self.inv_base != pack["inv"].data_ptr():
            self.grad_ptrs = None
        self._wn_table(model)
        ops.weight_norm_bwd_multi(self.wn_rows, self.wn_rows.shape[0])
        # the embed entry references this step's own (B,Et)/(B*N,Ec) tensors: gradients go by identity
        _embed_grads(st.embed, st.mi, bp["d_temb"], bp["d_cont"])
