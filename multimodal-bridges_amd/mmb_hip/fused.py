"""Host side of the fused per-jet EPiC kernels (csrc/fused_epic.hip): eligibility check,
parameter packing into MFMA-fragment order, and the launch wrappers.

Packing (per weight-normed Linear, effective weight W = g*v/||v|| from the HIP weight-norm
kernel):
  x-part  W[:, x-columns]  -> zero-padded to (32*OT, 32*KT), cut into fragments (ft, jj, ot) of 16 k-values x 32
          outputs in the A-operand order of v_mfma_f32_32x32x16_bf16 (_Packer.pack_x), each value split exactly into
          three bf16 planes hi + mid + lo (split_planes): [KT][2][OT][3 planes][64 lanes][8 bf16] = 3 KiB per
          fragment, stored after the fp32 part of the buffer (the kernels add the six significant partial
          products in fp32: csrc/fused_common.h, mma_x)
  t-part / g-part / per-jet matrices -> transposed [K][O] so the matvec threads read coalesced
  bias    -> zero-padded to a multiple of 32
Every chunk starts on a 16-byte boundary.  The layer order is the order the kernel walks.
"""
import ctypes as C
import os

import torch

from . import call
import mmb_hip as _m
from .ops import f32, i32, i64, stream, _fl

VARIANTS = {  # (Et, Ec, Ed, Hb, Gb) -> variant id (csrc/fused_common.h: DimsS3 / DimsMid / DimsWide / DimsNarrow)
    (32, 16, 16, 64, 16): 0,
    (16, 8, 8, 32, 8): 1,
    (64, 32, 32, 64, 32): 2,
    (32, 16, 16, 32, 16): 3,
}


UNI_VARIANTS = {  # (Et, Ed, H, G) -> variant id (csrc/fused_common.h: UniS2 / UniMid / UniH64 / UniH32)
    (128, 128, 128, 16): 0,
    (16, 16, 32, 8): 1,
    (64, 64, 64, 16): 2,
    (32, 32, 32, 16): 3,
}
MAX_TIME_LAYERS = 64


MAX_PARTICLES = 160          # five particle tiles per jet (N > 128: the five-wave compilation of every fused kernel)
MAX_PARTICLES_TRAIN = 160


def eligible_uni(model):
    """Variant id if the model is a discrete-only UniModalEPiC of a compiled width set."""
    cfg = model.config
    e, d = cfg.encoder, cfg.data
    if e.name != "UniModalEPiC" or d.modality != "discrete":
        return None
    if d.vocab_size != 8 or d.dim_discrete != 1 or (d.max_num_particles or 0) > MAX_PARTICLES:
        return None
    if e.dim_emb_context_continuous or e.dim_emb_context_discrete:
        return None
    hl, hg = e.dim_hidden_local, e.dim_hidden_glob
    if isinstance(hl, (list, tuple)) or isinstance(hg, (list, tuple)):
        return None
    return UNI_VARIANTS.get((e.dim_emb_time, e.dim_emb_discrete, hl, hg))


def eligible(model):
    """Variant id if the model is a branched+fused MultiModalEPiC of a compiled width set."""
    cfg = model.config
    e, d = cfg.encoder, cfg.data
    if e.name != "MultiModalEPiC" or d.modality != "multi-modal":
        return None
    enc = model.encoder
    if not (getattr(enc, "mode_branched", False) and getattr(enc, "mode_fused", False)):
        return None
    if d.dim_continuous != 3 or d.vocab_size != 8 or (d.max_num_particles or 0) > MAX_PARTICLES:
        return None
    H0, H1, H2, half = enc.dims
    hl, hg = e.dim_hidden_local, e.dim_hidden_glob
    if H0 != H1 or H2 != H0 + H1 or hg[0] != hg[1] or hg[2] != hg[0] + hg[1]:
        return None
    if e.dim_emb_context_continuous or e.dim_emb_context_discrete:
        return None
    return VARIANTS.get((e.dim_emb_time, e.dim_emb_continuous, e.dim_emb_discrete, H0, hg[0]))


class _IdxLin:
    """Stands in for a WNLinear while the gather map of the packed buffer is built (see _IndexSource)."""

    def __init__(self, w, b):
        self._w, self.bias = w, b

    def weight(self):
        return self._w


class _RealSource:
    index = False

    def lin(self, l):
        return l

    def tensor(self, getter, obj=None):
        return getter()


class _IndexSource:
    """Feeds the packer tensors whose VALUES are 1 + their position in the flat concatenation of all
    sources (0 = padding), so that the packed result is a gather map: training re-packs the changed
    parameters with one torch.cat + one index_select instead of ~10 small launches per layer."""
    index = True

    def __init__(self, device):
        self.n, self.getters, self.device = 0, [], device
        self.items = []               # (kind, object, offset in the flat buffer): what refresh_flat() must produce

    def _idx(self, shape, getter, kind, obj):
        numel = 1
        for d in shape:
            numel *= d
        t = (torch.arange(numel, device=self.device, dtype=torch.float32) + float(self.n + 1)).view(shape)
        self.items.append((kind, obj, self.n))
        self.n += numel
        assert self.n < (1 << 24), "gather map indices must stay exact in fp32"
        self.getters.append(getter)
        return t

    def lin(self, l):
        return _IdxLin(self._idx((l.dim_out, l.dim_in), l.weight, "wn", l),
                       self._idx((l.dim_out,), l.bias.detach, "copy", l.bias))

    def tensor(self, getter, obj=None):
        """obj: the nn.Parameter (plain copy) or weight-normed module (rows normalised) behind the getter"""
        kind = "copy" if isinstance(obj, torch.nn.Parameter) else "wn"
        return self._idx(tuple(getter().shape), getter, kind, obj)


class _Packer:
    def __init__(self, device):
        self.chunks, self.n, self.desc, self.device = [], 0, [], device
        self.time_parts = []          # W[:, :Et] of every layer that consumes the time embedding, kernel order
        self.time_lins = []
        self.second = []              # index mode: (offset, indices) of blocks that are ADDED on top (time_twice)
        self.mfma_flops = []          # 2*K_pad*O_pad of every per-particle layer (what the MFMAs execute)
        self.xchunks, self.nfrag = [], 0   # MFMA operands: fragments of 64 lanes x 8 values, split into bf16 planes last

    def add(self, t):
        t = t.contiguous().reshape(-1).float()
        off = self.n
        pad = (-t.numel()) % 4
        self.chunks.append(t)
        if pad:
            self.chunks.append(torch.zeros(pad, device=self.device))
        self.n += t.numel() + pad
        return off

    @staticmethod
    def pack_x(W):
        """(O,Kx) -> (fragments, 512) in MFMA operand order, zero padded to multiples of 32: fragment (ft, jj, ot)
        holds, for lane (hk, i) and slot j = (qq, e), W[32 ot + i][32 ft + 16 jj + 8 qq + 4 hk + e] -- the A operand of
        one v_mfma_f32_32x32x16_bf16 whose B operand is registers 8jj..8jj+7 of input tile ft (fused_common.h)."""
        O, K = W.shape
        OT, KT = (O + 31) // 32, (K + 31) // 32
        Wp = torch.zeros(OT * 32, KT * 32, device=W.device)
        Wp[:O, :K] = W
        v = Wp.view(OT, 32, KT, 2, 2, 2, 4)              # (ot, i, ft, jj, qq, hk, e)
        return v.permute(2, 3, 0, 5, 1, 4, 6).reshape(KT * 2 * OT, 512)

    def add_x(self, W):
        """-> marker resolved by finish(): float offset of this matrix's first fragment in the packed buffer"""
        u = self.pack_x(W)
        start = self.nfrag
        self.xchunks.append(u)
        self.nfrag += u.shape[0]
        self.mfma_flops.append(2 * u.numel())
        return ("x", start)

    def finish(self):
        """-> (main floats, (F,512) MFMA operand values, int32 descriptors with the fragment markers resolved)"""
        main = torch.cat(self.chunks)
        xs = torch.cat(self.xchunks)
        n_main = main.numel()
        assert n_main % 4 == 0
        rows = [[n_main + v[1] * FRAG_FLOATS if isinstance(v, tuple) else v for v in row] for row in self.desc]
        return main, xs, torch.tensor(rows, dtype=torch.int32, device=self.device).reshape(-1)

    def bias(self, b):
        O = b.numel()
        bp = torch.zeros((O + 31) // 32 * 32, device=b.device)
        bp[:O] = b
        return self.add(bp)

    def particle_layer(self, lin, n_t, n_x, n_g, time_twice=False):
        """columns = [t (n_t) | x (n_x) | g (n_g)]; time_twice: [t | t | x | g], both time blocks act
        on the same embedding and are summed into one"""
        W = lin.weight()
        Wt = None
        if n_t:
            Wt = W[:, :n_t]
            if time_twice:
                if isinstance(lin, _IdxLin):                 # gather map: the second block goes to a second map
                    self.second.append((self.n, W[:, n_t:2 * n_t].t().contiguous().reshape(-1)))
                else:
                    Wt = (Wt + W[:, n_t:2 * n_t]).contiguous()
                W = W[:, n_t:]
        wt = self.add(Wt.t()) if n_t else -1
        if n_t:
            self.time_parts.append(Wt)
            self.time_lins.append((lin, n_t, time_twice))
        wx = self.add_x(W[:, n_t:n_t + n_x])
        wg = self.add(W[:, n_t + n_x:].t()) if n_g else -1
        self.desc.append([wx, wt, wg, self.bias(lin.bias.detach())])

    def jet_layer(self, lin):
        W = lin.weight()
        self.desc.append([-1, self.add(W.t()), -1, self.add(lin.bias.detach())])

    # ---- backward pack (csrc/fused_epic_bwd.hip, BDesc): transposed x-part in fragment order, the time and
    # global-vector column blocks row-major (O,Kseg) -- the [K][O] layout of the kernel's matvec for W^T v
    def particle_layer_bwd(self, lin, n_t, n_x, n_g, time_twice=False):
        W = lin.weight()
        O = W.shape[0]
        Op = (O + 31) // 32 * 32                              # the kernel's matvecs run over whole 32-row tiles

        def rows(blk):
            if Op == O:
                return blk
            out = torch.zeros(Op, blk.shape[1], device=blk.device)
            out[:O] = blk
            return out

        Wt = W[:, :n_t] if n_t else None
        if time_twice:
            if isinstance(lin, _IdxLin):
                self.second.append((None, rows(W[:, n_t:2 * n_t]).contiguous().reshape(-1)))
            else:
                Wt = Wt + W[:, n_t:2 * n_t]
            W = W[:, n_t:]
        a = self.add_x(W[:, n_t:n_t + n_x].t())
        if time_twice and isinstance(lin, _IdxLin):
            self.second[-1] = (self.n, self.second[-1][1])
        b = self.add(rows(Wt)) if n_t else -1
        c = self.add(rows(W[:, n_t + n_x:])) if n_g else -1
        self.desc.append([a, b, c])

    def jet_layer_bwd(self, lin, k0):
        W = lin.weight()
        a = self.add(W[:, :k0])
        b = self.add(W[:, k0:]) if k0 < W.shape[1] else -1
        self.desc.append([a, b, -1])


def _pack_uni(model, variant, src):
    cfg = model.config
    enc, emb = model.encoder.epic, model.embedder
    Et, Ed = cfg.encoder.dim_emb_time, cfg.encoder.dim_emb_discrete
    H, G = cfg.encoder.dim_hidden_local, cfg.encoder.dim_hidden_glob
    dev = next(model.parameters()).device
    pk = _Packer(dev)
    L = src.lin
    pr = enc.epic_proj
    pk.particle_layer(L(pr.mlp_local["0"]), Et, Ed, 0, time_twice=True)
    pk.particle_layer(L(pr.mlp_local["2"]), 0, H, 0)
    pk.jet_layer(L(pr.mlp_global["0"]))
    pk.jet_layer(L(pr.mlp_global["2"]))
    for i in range(enc.num_blocks):
        ly = enc.epic_layers[i]
        pk.jet_layer(L(ly.fc_glob1))
        pk.jet_layer(L(ly.fc_glob2))
        pk.particle_layer(L(ly.fc_loc1), Et, H, G)
        pk.particle_layer(L(ly.fc_loc2), 0, H, 0)
    pk.particle_layer(L(enc.output_layer), Et, H, G)
    off_table = pk.add(src.tensor(emb.discrete_embedding.table, emb.discrete_embedding))
    main, xs, desc = pk.finish()
    return {"kind": "uni", "main": main, "xs": xs, "desc": desc, "variant": variant, "num_blocks": enc.num_blocks,
            "nb_branch": enc.num_blocks, "nb_fused_exec": 0,
            "use_skip": int(bool(cfg.encoder.skip_connection)), "off_table": off_table, "time_parts": pk.time_parts,
            "time_lins": pk.time_lins, "second": pk.second, "mfma_flops_per_forward": sum(pk.mfma_flops)}


def _pack_uni_bwd(model, src):
    cfg = model.config
    enc = model.encoder.epic
    Et, Ed = cfg.encoder.dim_emb_time, cfg.encoder.dim_emb_discrete
    H, G = cfg.encoder.dim_hidden_local, cfg.encoder.dim_hidden_glob
    dev = next(model.parameters()).device
    pk = _Packer(dev)
    L = src.lin
    pr = enc.epic_proj
    pk.particle_layer_bwd(L(pr.mlp_local["0"]), Et, Ed, 0, time_twice=True)
    pk.particle_layer_bwd(L(pr.mlp_local["2"]), 0, H, 0)
    pk.jet_layer_bwd(L(pr.mlp_global["0"]), 2 * H)
    pk.jet_layer_bwd(L(pr.mlp_global["2"]), H)
    for i in range(enc.num_blocks):
        ly = enc.epic_layers[i]
        pk.jet_layer_bwd(L(ly.fc_glob1), 2 * H)
        pk.jet_layer_bwd(L(ly.fc_glob2), H)
        pk.particle_layer_bwd(L(ly.fc_loc1), Et, H, G)
        pk.particle_layer_bwd(L(ly.fc_loc2), 0, H, 0)
    pk.particle_layer_bwd(L(enc.output_layer), Et, H, G)
    return pk.finish() + (pk.second,)


def _gather_maps(imap, second, n_flat):
    """index map (1 + flat position, 0 = padding) -> (gather, gather2 | None); gather2 adds the time_twice blocks"""
    idx = imap.long() - 1
    gather = torch.where(idx < 0, torch.full_like(idx, n_flat), idx)           # padding -> the trailing zero
    gather2 = None
    if second:
        gather2 = torch.full_like(idx, n_flat)
        for off, ind in second:
            i2 = ind.long() - 1
            gather2[off:off + i2.numel()] = torch.where(i2 < 0, torch.full_like(i2, n_flat), i2)
    return gather, gather2


def pack_model(model):
    """-> dict(kind, packed, desc, variant, ...) for the fused kernels, or None when the architecture
    is not one of the compiled width sets."""
    uv = eligible_uni(model)
    if uv is not None:
        variant, fwd, bwd = uv, _pack_uni, _pack_uni_bwd
    else:
        variant = eligible(model)
        if variant is None:
            return None
        fwd = _pack_multi
        bwd = lambda m, sb: _pack_multi_bwd(m, sb) + ([],)
    pack = fwd(model, variant, _RealSource())
    dev = pack["main"].device
    pack["n_main"] = pack["main"].numel()
    pack["packed"] = torch.cat([pack.pop("main"), split_planes(pack.pop("xs"))])
    src = _IndexSource(dev)
    im = fwd(model, variant, src)
    assert im["main"].numel() == pack["n_main"]
    pack["gather"], pack["gather2"] = _gather_maps(im["main"], im.get("second"), src.n)
    pack["gather_x"] = _gather_maps(im["xs"].reshape(-1), None, src.n)[0]
    pack["getters"] = src.getters
    pack["items"], pack["n_flat"] = src.items, src.n
    pack["flat"] = None                                    # persistent [all sources | 0], see refresh_flat
    pack["zero"] = torch.zeros(1, dtype=torch.float32, device=dev)
    # backward weights (training): same sources, so the same flat buffer feeds a second set of gathers
    srcb = _IndexSource(dev)
    bmain, bxs, pack["desc_bwd"], second_b = bwd(model, srcb)
    assert srcb.n == src.n - _tail_sources(model)
    pack["n_main_bwd"] = bmain.numel()
    pack["gather_bwd"], pack["gather2_bwd"] = _gather_maps(bmain, second_b, src.n)
    pack["gather_x_bwd"] = _gather_maps(bxs.reshape(-1), None, src.n)[0]
    pack["packed_bwd"] = None                              # built on first training use
    pack["bwd_version"] = None
    return pack


FRAG_FLOATS = 768      # one MFMA operand fragment in the packed buffer: 3 bf16 planes x 64 lanes x 8 values = 3072 B


def split_planes(xs):
    """(F, 512) f32 operand values -> flat f32 view of (F, 3, 64 lanes, 8) bf16: hi + mid + lo = the value's 24-bit
    significand (round-to-nearest splits; every residual is exact in f32)."""
    hi = xs.bfloat16()
    r = xs - hi.float()
    mid = r.bfloat16()
    lo = (r - mid.float()).bfloat16()
    return torch.stack([hi, mid, lo], dim=1).contiguous().view(torch.float32).reshape(-1)


def pack_transformer_tail(layer, next_layer=None):
    """Parameter block of mmb_transformer_tail for one nn.TransformerEncoderLayer (+ the in_proj of the layer after it):
    -> dict(packed, n_vec, wx_o, wx_1, wx_2, wx_in, d, ff, eps1, eps2, has_in)."""
    sa = layer.self_attn
    Wo, W1, W2 = sa.out_proj.weight.detach().float(), layer.linear1.weight.detach().float(), layer.linear2.weight.detach().float()
    d, ff = Wo.shape[0], W1.shape[0]
    dev = Wo.device
    if next_layer is not None:
        Win, b_in = next_layer.self_attn.in_proj_weight.detach().float(), next_layer.self_attn.in_proj_bias.detach().float()
    else:
        Win, b_in = None, torch.zeros(3 * d, device=dev)
    vec = torch.cat([sa.out_proj.bias.detach().float(), layer.norm1.weight.detach().float(), layer.norm1.bias.detach().float(),
                     layer.linear2.bias.detach().float(), layer.norm2.weight.detach().float(), layer.norm2.bias.detach().float(),
                     layer.linear1.bias.detach().float(), b_in])
    blocks, offs, n = [], {}, 0

    def add(name, W):
        nonlocal n
        u = _Packer.pack_x(W)
        offs.setdefault(name, n * FRAG_FLOATS)
        blocks.append(u)
        n += u.shape[0]

    add("wx_o", Wo)
    for c in range(ff // 64):
        add("wx_1", W1[64 * c:64 * c + 64, :])
    for c in range(ff // 64):
        add("wx_2", W2[:, 64 * c:64 * c + 64])
    if Win is not None:
        for j in range(3):
            add("wx_in", Win[d * j:d * (j + 1), :])
    packed = torch.cat([vec, split_planes(torch.cat(blocks))]).contiguous()
    return dict(packed=packed, n_vec=vec.numel(), d=d, ff=ff, eps1=float(layer.norm1.eps), eps2=float(layer.norm2.eps),
                has_in=Win is not None, wx_in=offs.get("wx_in", 0), **{k: offs[k] for k in ("wx_o", "wx_1", "wx_2")})


def transformer_tail(pack, a, h, out=None, want_qkv=True, a_idx=None):
    """-> (out (M,d), qkv (M,3d) | None); see include/mmb_hip.h: mmb_transformer_tail.  a_idx (M,) int32: row r reads
    a[a_idx[r]] (zeros where negative); a may then have any number of rows and a row pitch > d."""
    M, d = h.shape
    if out is None:
        out = torch.empty(M, d, dtype=torch.float32, device=a.device)
    qkv = torch.empty(M, 3 * d, dtype=torch.float32, device=a.device) if (want_qkv and pack["has_in"]) else None
    if _m.PROFILE is not None:
        macs = d * d + 2 * d * pack["ff"] + (3 * d * d if qkv is not None else 0)
        _m.PROFILE_META = {"M": M, "flops": 2.0 * macs * M, "bytes": 4.0 * M * (3 * d + (3 * d if qkv is not None else 0))}
    call("mmb_transformer_tail", M, d, pack["ff"], f32(pack["packed"]), pack["n_vec"], pack["wx_o"], pack["wx_1"],
         pack["wx_2"], pack["wx_in"], a.data_ptr(), a.stride(0), f32(h, "h"), h.stride(0), out.data_ptr(), out.stride(0),
         f32(qkv), qkv.stride(0) if qkv is not None else 0, _fl(pack["eps1"]), _fl(pack["eps2"]), i32(a_idx, "a_idx"),
         stream())
    _m.PROFILE_META = None
    return out, qkv


def _fill_packed(dst, n_main, flat, gather, gather2, gather_x):
    """dst <- the parameter block for the values now in `flat`: one launch (mmb_pack_params).  The torch lines below
    are the same arithmetic for host tensors: only the CPU tests of the index maps get there."""
    if dst.is_cuda:
        call("mmb_pack_params", f32(flat, "flat"), i64(gather, "gather"), i64(gather2, "gather2"), n_main,
             i64(gather_x, "gather_x"), gather_x.numel(), f32(dst, "dst"), stream())
        return
    torch.index_select(flat, 0, gather, out=dst[:n_main])
    if gather2 is not None:
        dst[:n_main].add_(flat.index_select(0, gather2))
    dst[n_main:] = split_planes(flat.index_select(0, gather_x).view(-1, 512))


def _tail_sources(model):
    """floats of the sources the forward pack appends after the layers (embedding parameters)"""
    emb = model.embedder
    n = emb.discrete_embedding.weight_v.numel()
    if hasattr(emb, "continuous_embedding"):
        n += emb.continuous_embedding.weight.numel() + emb.continuous_embedding.bias.numel()
    return n


def packed_bwd(pack, version):
    """Backward parameter buffer for the current parameter version (one cat + one index_select)."""
    if pack["bwd_version"] != version:
        flat = refresh_flat(pack, version)
        if pack["packed_bwd"] is None:
            n = pack["n_main_bwd"] + pack["gather_x_bwd"].numel() * 3 // 2
            pack["packed_bwd"] = torch.empty(n, dtype=torch.float32, device=flat.device)
        _fill_packed(pack["packed_bwd"], pack["n_main_bwd"], flat, pack["gather_bwd"], pack["gather2_bwd"],
                     pack["gather_x_bwd"])
        pack["bwd_version"] = version
    return pack["packed_bwd"]


def _pack_multi_bwd(model, src):
    """Index map of the backward parameter buffer + its descriptors (layer order = the forward pack's)."""
    cfg = model.config
    enc = model.encoder
    Et = cfg.encoder.dim_emb_time
    H0, H1, H2, half = enc.dims
    G0 = cfg.encoder.dim_hidden_glob[0]
    dev = next(model.parameters()).device
    pk = _Packer(dev)
    L = src.lin
    for branch, n_in in ((enc.continuous_encoder, cfg.encoder.dim_emb_continuous),
                         (enc.discrete_encoder, cfg.encoder.dim_emb_discrete)):
        pr = branch.epic_proj
        pk.particle_layer_bwd(L(pr.mlp_local["0"]), Et, n_in, 0)
        pk.particle_layer_bwd(L(pr.mlp_local["2"]), 0, H0, 0)
        pk.jet_layer_bwd(L(pr.mlp_global["0"]), 2 * H0)
        pk.jet_layer_bwd(L(pr.mlp_global["2"]), H0)
        for i in range(branch.num_blocks):
            ly = branch.epic_layers[i]
            pk.jet_layer_bwd(L(ly.fc_glob1), 2 * H0)
            pk.jet_layer_bwd(L(ly.fc_glob2), H0)
            pk.particle_layer_bwd(L(ly.fc_loc1), Et, H0, G0)
            pk.particle_layer_bwd(L(ly.fc_loc2), 0, H0, 0)
        pk.particle_layer_bwd(L(branch.output_layer), Et, H0, G0)
    fe = enc.fused_encoder
    for i in range(fe.num_blocks):
        ly = fe.epic_layers[i]
        pk.jet_layer_bwd(L(ly.fc_glob1), 2 * H2)
        pk.jet_layer_bwd(L(ly.fc_glob2), H2)
        pk.particle_layer_bwd(L(ly.fc_loc1), Et, H2, 2 * G0)
        pk.particle_layer_bwd(L(ly.fc_loc2), 0, H2, 0)
    pk.particle_layer_bwd(L(fe.output_layer), Et, H2, 2 * G0)
    for head in (enc.continuous_head, enc.discrete_head):
        pk.particle_layer_bwd(L(head["0"]), Et, half + H0, 0)
        pk.particle_layer_bwd(L(head["2"]), 0, H0, 0)
    return pk.finish()


def _flat_rows(pack):
    """Device table of mmb_weight_norm_multi: one record per row of every source -> its place in pack["flat"]."""
    dev = pack["packed"].device
    flat, inv = pack["flat"], pack["inv"]
    recs, ioff, ptrs = [], 0, []
    ar = lambda n: torch.arange(n, dtype=torch.int64)
    for kind, obj, off in pack["items"]:
        if kind == "copy":
            p = obj.detach()
            recs.append(torch.tensor([[p.data_ptr(), 0, flat.data_ptr() + 4 * off, 0, p.numel()]], dtype=torch.int64))
            ptrs.append(p.data_ptr())
            continue
        v, g = obj.weight_v.detach(), obj.weight_g.detach()
        O, K = v.shape
        r = torch.empty(O, 5, dtype=torch.int64)
        r[:, 0] = v.data_ptr() + 4 * K * ar(O)
        r[:, 1] = g.data_ptr() + 4 * ar(O)
        r[:, 2] = flat.data_ptr() + 4 * (off + K * ar(O))
        r[:, 3] = inv.data_ptr() + 4 * (ioff + ar(O))
        r[:, 4] = K
        recs.append(r)
        ptrs += [v.data_ptr(), g.data_ptr()]
        ioff += O
    pack["flat_rows"] = torch.cat(recs).to(dev)
    pack["flat_ptrs"] = tuple(ptrs)


def _source_ptrs(pack):
    out = []
    for kind, obj, off in pack["items"]:
        if kind == "copy":
            out.append(obj.data_ptr())
        else:
            out += [obj.weight_v.data_ptr(), obj.weight_g.data_ptr()]
    return tuple(out)


def refresh_flat(pack, version):
    """pack["flat"] = [effective weights, biases, embedding parameters | 0] for the current parameter values: ONE
    launch (mmb_weight_norm_multi) instead of one weight-norm launch per layer; every layer's weight() cache is
    pointed at its slice, so nothing else re-normalises this parameter version."""
    if pack.get("flat_version") == version:
        return pack["flat"]
    from . import ops
    dev = pack["packed"].device
    if pack["flat"] is None:
        pack["flat"] = torch.zeros(pack["n_flat"] + 1, dtype=torch.float32, device=dev)
        n_inv = sum(obj.weight_v.shape[0] for kind, obj, _ in pack["items"] if kind == "wn")
        pack["inv"] = torch.empty(n_inv, dtype=torch.float32, device=dev)
        pack["flat_ptrs"] = None
    if pack["flat_ptrs"] != _source_ptrs(pack):            # first use, or a parameter's storage moved
        _flat_rows(pack)
    ops.weight_norm_multi(pack["flat_rows"], pack["flat_rows"].shape[0])
    flat, inv, ioff = pack["flat"], pack["inv"], 0
    for kind, obj, off in pack["items"]:
        if kind != "wn":
            continue
        v, g = obj.weight_v, obj.weight_g
        O, K = v.shape
        W = flat[off:off + O * K].view(O, K)
        if hasattr(obj, "dim_in"):                         # WNLinear
            obj._cache = ((v._version, g._version, v.data_ptr(), g.data_ptr()), W, inv[ioff:ioff + O])
        else:                                              # WNEmbedding
            obj._cache = ((v._version, g._version, v.data_ptr()), W)
        ioff += O
    pack["flat_version"] = version
    return flat


def repack(pack, version):
    """Refresh the packed parameter buffer after an optimizer step (same architecture, new values)."""
    flat = refresh_flat(pack, version)
    pack.pop("grid_key", None)                               # cached time-grid tables were built from the old weights
    pack.pop("grid", None)
    _fill_packed(pack["packed"], pack["n_main"], flat, pack["gather"], pack["gather2"], pack["gather_x"])
    pack["time_parts"] = None                                # rebuilt by the next generation (time_vector_table)
    return pack


def _pack_multi(model, variant, src):
    cfg = model.config
    enc, emb = model.encoder, model.embedder
    Et = cfg.encoder.dim_emb_time
    H0, H1, H2, half = enc.dims
    G0 = cfg.encoder.dim_hidden_glob[0]
    dev = next(model.parameters()).device
    pk = _Packer(dev)

    def L(lin):
        return src.lin(lin)

    for branch, n_in in ((enc.continuous_encoder, cfg.encoder.dim_emb_continuous),
                         (enc.discrete_encoder, cfg.encoder.dim_emb_discrete)):
        pr = branch.epic_proj
        pk.particle_layer(L(pr.mlp_local["0"]), Et, n_in, 0)
        pk.particle_layer(L(pr.mlp_local["2"]), 0, H0, 0)
        pk.jet_layer(L(pr.mlp_global["0"]))
        pk.jet_layer(L(pr.mlp_global["2"]))
        for i in range(branch.num_blocks):
            ly = branch.epic_layers[i]
            pk.jet_layer(L(ly.fc_glob1))
            pk.jet_layer(L(ly.fc_glob2))
            pk.particle_layer(L(ly.fc_loc1), Et, H0, G0)
            pk.particle_layer(L(ly.fc_loc2), 0, H0, 0)
        pk.particle_layer(L(branch.output_layer), Et, H0, G0)
    fe = enc.fused_encoder
    for i in range(fe.num_blocks):                        # executed layers only (K2)
        ly = fe.epic_layers[i]
        pk.jet_layer(L(ly.fc_glob1))
        pk.jet_layer(L(ly.fc_glob2))
        pk.particle_layer(L(ly.fc_loc1), Et, H2, 2 * G0)
        pk.particle_layer(L(ly.fc_loc2), 0, H2, 0)
    pk.particle_layer(L(fe.output_layer), Et, H2, 2 * G0)
    for head in (enc.continuous_head, enc.discrete_head):
        pk.particle_layer(L(head["0"]), Et, half + H0, 0)
        pk.particle_layer(L(head["2"]), 0, H0, 0)
    off_cw = pk.add(src.tensor(emb.continuous_embedding.weight.detach, emb.continuous_embedding.weight))
    off_cb = pk.add(src.tensor(emb.continuous_embedding.bias.detach, emb.continuous_embedding.bias))
    off_table = pk.add(src.tensor(emb.discrete_embedding.table, emb.discrete_embedding))
    main, xs, desc = pk.finish()
    return {"kind": "multi", "main": main, "xs": xs, "time_lins": pk.time_lins, "desc": desc, "variant": variant, "nb_branch": enc.continuous_encoder.num_blocks,
            "nb_branch1": enc.discrete_encoder.num_blocks,
            "nb_fused_exec": fe.num_blocks, "use_skip": int(bool(cfg.encoder.skip_connection)),
            "off_cw": off_cw, "off_cb": off_cb, "off_table": off_table, "time_parts": pk.time_parts,
            # a generation forward evaluates ONE head (the last four particle layers are head_c.0/.2, head_d.0/.2)
            "mfma_flops_per_forward": sum(pk.mfma_flops) - sum(pk.mfma_flops[-2:])}


def time_parts(pack):
    """W[:, time columns] of every layer that consumes the time embedding (kernel order) for the current weights;
    dropped by repack() and rebuilt here, because a training step repacks without ever needing them"""
    if pack["time_parts"] is None:
        pack["time_parts"] = [(l.weight()[:, :n_t] + l.weight()[:, n_t:2 * n_t]) if twice else l.weight()[:, :n_t]
                              for l, n_t, twice in pack["time_lins"]]
    return pack["time_parts"]


def time_vector_table(pack, temb_steps):
    """(steps, n_ctab, 128): c = W_t temb for every time-consuming layer at every grid point."""
    from . import ops
    steps = temb_steps.shape[0]
    parts = time_parts(pack)
    tab = torch.zeros(steps, len(parts), 128, dtype=torch.float32, device=temb_steps.device)
    for i, Wt in enumerate(parts):
        ops.linear_cat([(temb_steps, ops.SEG_ROW)], Wt, None, 1, out=tab[:, i, :Wt.shape[0]])
    return tab


def params_version(model):
    return tuple(p._version for p in model.parameters()) + (next(model.parameters()).data_ptr(),)


def forward(pack, mi, x, k, temb, which, temb_stride=None):
    """-> (head_c (B,N,3) | None, head_d (B,N,8) | None)"""
    B, N = mi.B, mi.N
    dev = x.device
    hc = torch.empty(B, N, 3, dtype=torch.float32, device=dev) if which != 1 else None
    hd = torch.empty(B, N, 8, dtype=torch.float32, device=dev) if which != 0 else None
    stride = temb.stride(0) if temb_stride is None else temb_stride
    call("mmb_fused_epic_forward", pack["variant"], B, N, f32(pack["packed"]), i32(pack["desc"]), pack["nb_branch"],
         pack["nb_branch1"], pack["nb_fused_exec"], pack["use_skip"], pack["off_cw"], pack["off_cb"], pack["off_table"],
         f32(mi.maskf), i32(mi.counts), f32(x, "x"), i64(k, "k"), f32(temb, "temb"), stride, which, f32(hc), f32(hd), stream())
    return hc, hd


def train_forward(pack, mi, x, k, temb, slots, n_slots, drop_p, seed, offset0, hc=None, hd=None):
    """Training forward: both heads + every activation of the tape (encoders/fused_train.py) -> HBM.
    slots: device int64 tensor (n_slots, 3) of (pointer, row pitch, floats available)."""
    B, N = mi.B, mi.N
    dev = x.device
    hc = torch.empty(B * N, 3, dtype=torch.float32, device=dev) if hc is None else hc
    hd = torch.empty(B * N, 8, dtype=torch.float32, device=dev) if hd is None else hd
    call("mmb_fused_epic_train_forward", pack["variant"], B, N, f32(pack["packed"]), i32(pack["desc"]),
         pack["nb_branch"], pack["nb_branch1"], pack["nb_fused_exec"], pack["use_skip"], pack["off_cw"], pack["off_cb"],
         pack["off_table"], f32(mi.maskf), i32(mi.counts), f32(x, "x"), i64(k, "k"), f32(temb, "temb"), temb.stride(0),
         C.c_void_p(slots.data_ptr()), n_slots, _fl(drop_p), seed, offset0, f32(hc), f32(hd), i32(jet_schedule(mi)),
         stream())
    return hc, hd


def train_forward_uni(pack, mi, k, temb, slots, n_slots, drop_p, seed, offset0, hd):
    """Training forward of the discrete-only model: logits -> hd (B*N,8), activations -> slots."""
    call("mmb_fused_uni_train_forward", pack["variant"], mi.B, mi.N, f32(pack["packed"]), i32(pack["desc"]),
         pack["num_blocks"], pack["use_skip"], pack["off_table"], f32(mi.maskf), i32(mi.counts), i64(k, "k"),
         f32(temb, "temb"), temb.stride(0), C.c_void_p(slots.data_ptr()), n_slots, _fl(drop_p), seed, offset0, f32(hd),
         i32(jet_schedule(mi)), stream())
    return hd


def train_backward(pack, pbwd, mi, slots, n_slots, drop_p, seed, offset0, real_rows_only=False):
    """Reverse pass of the whole encoder for every jet in one launch (csrc/fused_epic_bwd.hip): fills the dZ / column-sum
    slots the weight-gradient kernel consumes.  slots: device int64 (n_slots, 3).  real_rows_only: the caller reads
    the per-particle dZ slots at the rows of real particles only (mmb_mask_rows lists) -- tiles of padded slots then
    leave their rows UNWRITTEN (bit 1 of the entry points' use_skip); the continuous-input gradient keeps every row."""
    skip = int(pack["use_skip"]) | (2 if real_rows_only else 0)
    if pack["kind"] == "uni":
        call("mmb_fused_uni_train_backward", pack["variant"], mi.B, mi.N, f32(pbwd), i32(pack["desc_bwd"]),
             pack["num_blocks"], skip, f32(mi.maskf), i32(mi.counts), C.c_void_p(slots.data_ptr()), n_slots,
             _fl(drop_p), seed, offset0, i32(jet_schedule(mi)), stream())
        return
    call("mmb_fused_epic_train_backward", pack["variant"], mi.B, mi.N, f32(pbwd), i32(pack["desc_bwd"]), pack["nb_branch"],
         pack["nb_branch1"], pack["nb_fused_exec"], skip, f32(mi.maskf), i32(mi.counts),
         C.c_void_p(slots.data_ptr()), n_slots,
         _fl(drop_p), seed, offset0, i32(jet_schedule(mi)), stream())


def jet_schedule(mi):
    """(B) int32 launch order of the jets of a fused generation or training call: heaviest and lightest jets in
    alternating rounds (csrc/fused_epic.hip: jet_schedule_kernel); kept with the mask it was computed from."""
    order = getattr(mi, "_order", None)
    if order is None:
        order = torch.empty(mi.B, dtype=torch.int32, device=mi.maskf.device)
        call("mmb_jet_schedule", mi.B, i32(mi.counts), i32(order), stream())
        mi._order = order
    return order


def working_tiles(counts, N):
    """(B) number of 32-particle tiles of each jet that do per-particle work in the generation kernels (TilePlan in
    csrc/fused_epic.hip): the tiles its real particles fill, plus one when no padded slot sits next to a real one."""
    import os
    n = counts.to(torch.int64)
    all_tiles = (N + 31) // 32
    if not (int(os.environ.get("MMB_GEN_TILES", "3")) & 1):
        return torch.full_like(n, all_tiles)
    t = (n + 31) // 32
    t = t + ((n % 32 == 0) & (n < N)).to(torch.int64)
    return t.clamp(max=all_tiles)


def long_jets(mi):
    """(list, count) of the jets with 128 or more particles, for a generation call at 128 < N <= 160: those need the
    five-wave kernel, all others run on four waves with shadow slots (csrc/fused_epic.hip).  The count is read back --
    one small synchronisation per mask, against a generation loop of tens of milliseconds -- because the remainder launch
    is sized by it (and skipped when it is zero).  (None, -1) when N <= 128: not a split call."""
    if mi.N <= 128 or os.environ.get("MMB_GEN_LONG_UNKNOWN"):      # (developer switch: the C ABI's "caller does not know" path)
        return None, -1
    if mi._long is None:
        lst = torch.nonzero(mi.counts >= 128).flatten().to(torch.int32)
        mi._long = (lst if lst.numel() else None, int(lst.numel()))
    return mi._long


def generate(pack, mi, x, k, times, temb_steps, dt, thermo_kind, gamma, delta_n=None, seed=0, offset=0, ctab=None,
             em_sigma=None, delta_w=None):
    """In-place generation loop on x (B,N,3) and k (B,N,1).  em_sigma is None -> Euler, else
    Euler-Maruyama with sigma_t = em_sigma*sqrt(t(1-t)) and increments delta_w (steps,B,N,3) | Philox."""
    B, N = mi.B, mi.N
    if ctab is None:                       # the kernels take the per-step time vectors from this table only
        ctab = time_vector_table(pack, temb_steps)
    n_ctab = ctab.shape[1]
    lj, n_lj = long_jets(mi)
    call("mmb_fused_epic_generate", pack["variant"], B, N, f32(pack["packed"]), i32(pack["desc"]), pack["nb_branch"],
         pack["nb_branch1"], pack["nb_fused_exec"], pack["use_skip"], pack["off_cw"], pack["off_cb"], pack["off_table"],
         f32(mi.maskf), i32(mi.counts), f32(x, "x"), i64(k, "k"), times.numel(), f32(times, "times"), f32(temb_steps, "temb_steps"),
         f32(ctab, "ctab"), n_ctab, _fl(dt), thermo_kind, _fl(gamma), f32(delta_n, "delta_n"), seed, offset,
         0 if em_sigma is None else 1, _fl(0.0 if em_sigma is None else em_sigma), f32(delta_w, "delta_w"),
         i32(jet_schedule(mi)), i32(lj), n_lj, stream())


def forward_uni(pack, mi, k, temb, temb_stride=None):
    """-> logits (B,N,8) of the discrete-only model"""
    B, N = mi.B, mi.N
    hd = torch.empty(B, N, 8, dtype=torch.float32, device=k.device)
    stride = temb.stride(0) if temb_stride is None else temb_stride
    call("mmb_fused_uni_forward", pack["variant"], B, N, f32(pack["packed"]), i32(pack["desc"]), pack["num_blocks"],
         pack["use_skip"], pack["off_table"], f32(mi.maskf), i32(mi.counts), i64(k, "k"), f32(temb, "temb"), stride,
         f32(hd), stream())
    return hd


def generate_uni(pack, mi, k, times, temb_steps, dt, thermo_kind, gamma, delta_n=None, seed=0, offset=0, ctab=None):
    """In-place discrete-only generation loop on k (B,N,1)."""
    B, N = mi.B, mi.N
    if ctab is None:
        ctab = time_vector_table(pack, temb_steps)
    n_ctab = ctab.shape[1]
    lj, n_lj = long_jets(mi)
    call("mmb_fused_uni_generate", pack["variant"], B, N, f32(pack["packed"]), i32(pack["desc"]), pack["num_blocks"],
         pack["use_skip"], pack["off_table"], f32(mi.maskf), i32(mi.counts), i64(k, "k"), times.numel(),
         f32(times, "times"), f32(temb_steps, "temb_steps"), f32(ctab, "ctab"), n_ctab, _fl(dt), thermo_kind,
         _fl(gamma), f32(delta_n, "delta_n"), seed, offset, i32(jet_schedule(mi)), i32(lj), n_lj, stream())


def algorithmic_flops_per_particle(model):
    """GEMM flops per particle per encoder forward AS THE REFERENCE EXECUTES IT (2*K*N over every
    weight-normed Linear on the per-particle path, concatenated inputs included, both heads; the
    continuous embedding adds 2*Dc*Ec).  SURVEY.md 8(d): 574 944 for S3, 1 159 424 for S2."""
    from encoders.epic import EPiCEncoder, WNLinear

    def enc_flops(e):
        f = 0
        if hasattr(e, "epic_proj"):
            f += sum(2 * l.dim_in * l.dim_out for l in e.epic_proj.mlp_local.values())
        for i in range(e.num_blocks):
            ly = e.epic_layers[i]
            f += 2 * ly.fc_loc1.dim_in * ly.fc_loc1.dim_out + 2 * ly.fc_loc2.dim_in * ly.fc_loc2.dim_out
        return f + 2 * e.output_layer.dim_in * e.output_layer.dim_out

    total = 0
    for m in model.encoder.children():
        if isinstance(m, EPiCEncoder):
            total += enc_flops(m)
        elif isinstance(m, torch.nn.ModuleDict):            # heads
            total += sum(2 * l.dim_in * l.dim_out for l in m.values() if isinstance(l, WNLinear))
    ce = getattr(model.embedder, "continuous_embedding", None)
    if ce is not None:
        total += 2 * ce.in_features * ce.out_features
    return total
