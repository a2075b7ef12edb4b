"""MultiModalParticleTransformer on the HIP path (reference: encoders/particle_transformer.py:8-167).

Same constructor (config), module tree and parameter names as the reference -- the encoder layers ARE
torch's nn.TransformerEncoderLayer objects, kept as parameter containers only (so `transformer.layers.{i}.
self_attn.in_proj_weight`, `.linear1.weight`, `.norm1.weight` ... load from a reference checkpoint); their
forward is never called.  What runs instead, per TransformerEncoderBlock:

  h = input_proj(x) + time_proj(t * mask)      ONE mmb_linear_cat over [x rows | per-jet time embedding x mask]
  per layer (post-LN, ReLU FFN, as built at particle_transformer.py:29-35):
      qkv = in_proj(h)                         mmb_linear_cat
      a   = attention(qkv)                     mmb_attention_fwd -- keys restricted to the slots with mask == 0:
                                               the reference passes src_key_padding_mask = mask > 0 (:50, quirk Q4),
                                               so every slot attends to the PADDED slots only; a jet without a
                                               padded slot gets zeros (torch's safe softmax), i.e. out_proj.bias
      h   = LayerNorm1(h + out_proj(a))        mmb_linear_cat (+ residual epilogue), mmb_layernorm_fwd
      h   = LayerNorm2(h + linear2(relu(linear1(h))))

Deviation (documented in DESIGN.md): under torch.no_grad() + eval() the reference runs torch's fused / nested-tensor
inference kernels, which return NaN for a jet without a padded slot and zeros for the real particles when every
jet's padded slots come first; this implementation always follows the training-path arithmetic (identical
everywhere else: the goldens' generation runs agree).  Dropout inside the encoder layers is not implemented:
the path refuses dropout > 0 in training mode instead of silently dropping it.
"""
import torch
from torch import nn

from tensorclass import TensorMultiModal
from encoders.epic import WNLinear, ExecCtx, _time_seg
from encoders.multimodal_epic import _unpack, _Head
from mmb_hip import ops
from mmb_hip.ops import SEG_ROW


class PlainLinear:
    """cat_s[X_s] @ cat_s[W_s]^T + sum_s b_s for plain (not weight-normed) nn.Linear parameters; a part with
    weight None is an identity (the reference's `time_proj = nn.Identity()` when d_time == d_model)."""

    def __init__(self, dim_out, parts, biases):
        self.dim_out = dim_out
        self.parts = parts              # [(weight Parameter (dim_out, k) | None, k)]
        self.biases = biases            # [bias Parameter (dim_out)]
        self.K = sum(k for _, k in parts)
        self._cache = None

    def weight(self):
        ws = [w for w, _ in self.parts if w is not None]
        key = tuple((w._version, w.data_ptr()) for w in ws) + tuple((b._version, b.data_ptr()) for b in self.biases)
        if self._cache is None or self._cache[0] != key:
            dev = self.biases[0].device
            if len(self.parts) == 1:
                W = self.parts[0][0].detach()
            else:
                W = torch.cat([w.detach() if w is not None else torch.eye(self.dim_out, k, device=dev)
                               for w, k in self.parts], dim=1).contiguous()
            b = self.biases[0].detach() if len(self.biases) == 1 else sum(b.detach() for b in self.biases)
            self._cache = (key, W, b.contiguous())
        return self._cache[1]

    def bias(self):
        self.weight()
        return self._cache[2]

    def run(self, ctx, segs, act=None, res1=None, out=None, norm=None):
        """y = act(cat(segs) @ W^T + b) + res1;  with norm (an nn.LayerNorm): LayerNorm(y), in the same call
        (mmb_linear_cat_ln: the post-LN sublayer of nn.TransformerEncoderLayer)."""
        W = self.weight()
        if norm is None:
            y = ops.linear_cat(segs, W, self.bias(), ctx.N, maskf=ctx.mi.maskf, act=act, residual=res1, out=out)
            if ctx.recording:
                ctx.tape.append(("plinear", self, ctx.N, segs, act, res1, None, None, None, y, None, 0.0, 0))
            return y
        rec = ctx.recording
        y1, y, stats = ops.linear_cat_ln(segs, W, self.bias(), ctx.N, norm.weight.detach(), norm.bias.detach(), norm.eps,
                                         maskf=ctx.mi.maskf, act=act, residual=res1, want_y1=rec, want_stats=rec, out=out)
        if rec:                                           # the same two tape entries as the unfused pair
            ctx.tape.append(("plinear", self, ctx.N, segs, act, res1, None, None, None, y1, None, 0.0, 0))
            ctx.tape.append(("ln", y1, y, stats, norm))
        return y


class TransformerEncoderBlock(nn.Module):
    def __init__(self, d_input, d_model, d_time, n_layers, n_heads, dim_feedforward, dropout):
        super().__init__()
        self.input_proj = nn.Linear(d_input, d_model)
        self.time_proj = nn.Linear(d_time, d_model) if d_time != d_model else nn.Identity()
        layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=n_heads, dim_feedforward=dim_feedforward,
                                           dropout=dropout, batch_first=True)
        self.transformer = nn.TransformerEncoder(layer, num_layers=n_layers, enable_nested_tensor=False)
        self.n_heads, self.d_model, self.d_time, self.p_drop = n_heads, d_model, d_time, dropout
        self._specs = None

    def _plan(self):
        if self._specs is None:
            tp = self.time_proj
            t_w, t_b = (tp.weight, [tp.bias]) if isinstance(tp, nn.Linear) else (None, [])
            first = PlainLinear(self.d_model, [(self.input_proj.weight, self.input_proj.in_features),
                                               (t_w, self.d_time)], [self.input_proj.bias] + t_b)
            layers = []
            for l in self.transformer.layers:
                sa = l.self_attn
                layers.append((PlainLinear(3 * self.d_model, [(sa.in_proj_weight, self.d_model)], [sa.in_proj_bias]),
                               PlainLinear(self.d_model, [(sa.out_proj.weight, self.d_model)], [sa.out_proj.bias]),
                               PlainLinear(l.linear1.out_features, [(l.linear1.weight, self.d_model)], [l.linear1.bias]),
                               PlainLinear(self.d_model, [(l.linear2.weight, l.linear2.in_features)], [l.linear2.bias]),
                               l.norm1, l.norm2))
            self._specs = (first, layers)
        return self._specs

    def run(self, ctx, time, x_segs, out=None):
        """x_segs: list of (rows, SEG_ROW) whose widths add up to d_input.  Returns (B*N, d_model) rows (written into
        `out`, which may be a column slice, when given)."""
        if ctx.drop_p > 0.0:
            raise NotImplementedError("dropout inside the transformer encoder layers is not implemented on the HIP "
                                      "path: set encoder.dropout to 0 (DESIGN.md, deviations)")
        first, layers = self._plan()
        h = first.run(ctx, list(x_segs) + [_time_seg(time)])
        for i, (in_proj, out_proj, lin1, lin2, norm1, norm2) in enumerate(layers):
            qkv = in_proj.run(ctx, [(h, SEG_ROW)])
            rec = ctx.recording
            r = ops.attention_fwd(qkv, ctx.mi, self.n_heads, valid_is_zero=True, want_lse=rec)      # quirk Q4
            a = r[0] if rec else r
            if rec:
                ctx.tape.append(("attn", qkv, a, r[1], self.n_heads))
            h1 = out_proj.run(ctx, [(a, SEG_ROW)], res1=h, norm=norm1)
            f = lin1.run(ctx, [(h1, SEG_ROW)], act="relu")
            h = lin2.run(ctx, [(f, SEG_ROW)], res1=h1, norm=norm2, out=out if i == len(layers) - 1 else None)
        if not layers and out is not None:
            out.copy_(h)
            h = out
        return h

    def forward(self, t_emb, x, mask):
        """Reference-style call: t_emb (B,N,d_time) per-particle (already masked), x (B,N,d_input), mask (B,N,1)."""
        ctx = ExecCtx(ops.mask_prepare(mask))
        rows = x.float().reshape(-1, x.shape[-1]).contiguous()
        h = self.run(ctx, t_emb.float().contiguous(), [(rows, SEG_ROW)])
        return h.view(ctx.B, ctx.N, -1)


class MultiModalParticleTransformer(nn.Module):
    """Two branch transformers (continuous / discrete features), an optional fusion transformer over cat[h1,h2]
    and two GELU MLP heads on cat[t, h_i, f_i] (particle_transformer.py:52-167)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        e, d = config.encoder, config.data
        Et = e.dim_emb_time
        dim_cont, dim_disc = e.dim_emb_continuous, e.dim_emb_discrete * d.dim_discrete
        H0, H1 = e.dim_hidden_local[0], e.dim_hidden_local[1]
        nb = e.num_blocks
        ff = 4 * H0                                                   # particle_transformer.py:77
        kw = dict(d_time=Et, n_heads=e.num_heads, dim_feedforward=ff, dropout=e.dropout)
        self.continuous_encoder = TransformerEncoderBlock(d_input=dim_cont, d_model=H0, n_layers=nb[0], **kw)
        self.discrete_encoder = TransformerEncoderBlock(d_input=dim_disc, d_model=H1, n_layers=nb[1], **kw)
        self.mode_fusion = nb[2] > 0
        half = 0
        if self.mode_fusion:
            H2 = e.dim_hidden_local[2]
            self.fused_encoder = TransformerEncoderBlock(d_input=H0 + H1, d_model=H2, n_layers=nb[2], **kw)
            half = H2 // 2
        self.dims = (H0, H1, half)
        self.continuous_head = _Head(Et + H0 + half, H0, d.dim_continuous)
        self.discrete_head = _Head(Et + H1 + half, H1, d.vocab_size * d.dim_discrete)

    def run(self, ctx, time, cont, disc):
        H0, H1, half = self.dims
        M = ctx.B * ctx.N
        tseg = _time_seg(time)
        both = torch.empty(M, H0 + H1, dtype=torch.float32, device=cont.device)
        h1 = self.continuous_encoder.run(ctx, time, [(cont, SEG_ROW)], out=both[:, :H0])
        h2 = self.discrete_encoder.run(ctx, time, [(disc, SEG_ROW)], out=both[:, H0:])
        if self.mode_fusion:
            f = self.fused_encoder.run(ctx, time, [(both, SEG_ROW)])
            split = (f.shape[1] + 1) // 2                             # torch.chunk(fused, 2, dim=-1)
            hc = self.continuous_head.run(ctx, [tseg, (h1, SEG_ROW), (f[:, :split], SEG_ROW)])
            hd = self.discrete_head.run(ctx, [tseg, (h2, SEG_ROW), (f[:, split:], SEG_ROW)])
        else:
            hc = self.continuous_head.run(ctx, [tseg, (h1, SEG_ROW)])
            hd = self.discrete_head.run(ctx, [tseg, (h2, SEG_ROW)])
        return hc, hd

    def forward(self, state_local, state_global=None, tape=None, seed=0, offset0=0) -> TensorMultiModal:
        glob = state_global if state_global is not None else TensorMultiModal(time=state_local.time)
        ctx, time, cont, disc, _ = _unpack(state_local, glob, self.training, self.config.encoder.dropout, tape, seed,
                                           offset0)
        hc, hd = self.run(ctx, time, cont, disc)
        B, N = ctx.B, ctx.N
        return TensorMultiModal(None, hc.view(B, N, -1), hd.view(B, N, -1), state_local.mask)
