"""UniModalEPiC / MultiModalEPiC on the HIP path (reference: encoders/multimodal_epic.py).

Same constructor (config), parameter names and forward(state_local, state_global) ->
TensorMultiModal(heads) as the reference.  The two branch encoders write their outputs
into the two column halves of ONE (B*N, H0+H1) buffer, so torch.cat([h1,h2]) and the
tensor_split of the fused features are views, not copies; the head inputs
cat[t, f_i, h_i] are three segments of one mmb_linear_cat launch.
"""
import torch
from torch import nn

from tensorclass import TensorMultiModal
from encoders.epic import EPiCEncoder, WNLinear, ExecCtx, TimeSeg, _time_seg
from encoders.embedder import LocalState
from mmb_hip import ops
from mmb_hip.ops import SEG_ROW


def _unpack(state_local, state_global, training, drop_p, tape, seed, offset0):
    """-> (ctx, time, cont_rows, disc_rows, glob) from either a LocalState (factored, fast) or a
    plain TensorMultiModal of materialised (B,N,*) tensors (reference-style call)."""
    if isinstance(state_local, LocalState):
        mi = state_local.mi
        time = TimeSeg(state_local.temb)
        cont, disc = state_local.cont_rows, state_local.disc_rows
    else:
        mi = ops.mask_prepare(state_local.mask)
        time = state_local.time.float().contiguous()
        rows = lambda a: None if a is None else a.float().reshape(-1, a.shape[-1]).contiguous()
        cont, disc = rows(state_local.continuous), rows(state_local.discrete)
    ctx = ExecCtx(mi, tape=tape, drop_p=drop_p if training else 0.0, seed=seed, offset0=offset0)
    parts = [getattr(state_global, m) for m in state_global.available_modes()]
    glob = parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1)
    return ctx, time, cont, disc, glob.contiguous()


class _Head(nn.ModuleDict):
    def __init__(self, dim_in, dim_hid, dim_out):
        super().__init__({"0": WNLinear(dim_in, dim_hid), "2": WNLinear(dim_hid, dim_out)})

    def run(self, ctx, segs):
        h, _ = self["0"].run(ctx, segs, act="gelu")
        y, _ = self["2"].run(ctx, [(h, SEG_ROW)])
        return y


class UniModalEPiC(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        e, d = config.encoder, config.data
        dim_input = e.dim_emb_time + e.dim_emb_continuous + e.dim_emb_discrete * d.dim_discrete
        dim_output = d.dim_continuous + d.vocab_size * d.dim_discrete
        dim_context = e.dim_emb_time + e.dim_emb_context_continuous + e.dim_emb_context_discrete * d.dim_context_discrete
        self.epic = EPiCEncoder(dim_time=e.dim_emb_time, dim_input_loc=dim_input, dim_input_glob=dim_context,
                                dim_output_loc=dim_output, dim_hid_loc=e.dim_hidden_local,
                                dim_hid_glob=e.dim_hidden_glob, num_blocks=e.num_blocks,
                                use_skip_connection=e.skip_connection, dropout=e.dropout)

    def run(self, ctx, time, cont, disc, glob):
        # local input = cat[time, continuous?, discrete?]; the projection prepends time again
        local = [_time_seg(time)] + [(m, SEG_ROW) for m in (cont, disc) if m is not None]
        h, _ = self.epic.run(ctx, time, local, glob)
        return h

    def forward(self, state_local, state_global, tape=None, seed=0, offset0=0) -> TensorMultiModal:
        ctx, time, cont, disc, glob = _unpack(state_local, state_global, self.training, self.config.encoder.dropout,
                                              tape, seed, offset0)
        h = self.run(ctx, time, cont, disc, glob).view(ctx.B, ctx.N, -1)
        mask = state_local.mask
        if self.config.data.modality == "continuous":
            return TensorMultiModal(continuous=h, mask=mask)
        if self.config.data.modality == "discrete":
            return TensorMultiModal(discrete=h, mask=mask)


class MultiModalEPiC(nn.Module):
    """Permutation-equivariant encoder for hybrid continuous-discrete clouds: two branch
    EPiCs, one fused EPiC, two GELU MLP heads."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        e, d = config.encoder, config.data
        Et = e.dim_emb_time
        dim_cont = e.dim_emb_continuous
        dim_disc = e.dim_emb_discrete * d.dim_discrete
        hid_loc, hid_glob = e.dim_hidden_local, e.dim_hidden_glob       # lists, mutated below like upstream
        nb = e.num_blocks
        self.mode_fused = nb[2] > 0
        self.mode_branched = nb[0] > 0
        dim_context = Et + e.dim_emb_context_continuous + e.dim_emb_context_discrete * d.dim_context_discrete
        kw = dict(use_skip_connection=e.skip_connection, dropout=e.dropout)
        if self.mode_branched:
            self.continuous_encoder = EPiCEncoder(dim_time=Et, dim_input_loc=dim_cont, dim_input_glob=dim_context,
                                                  dim_output_loc=hid_loc[0], dim_hid_loc=hid_loc[0],
                                                  dim_hid_glob=hid_glob[0], num_blocks=nb[0], **kw)
            self.discrete_encoder = EPiCEncoder(dim_time=Et, dim_input_loc=dim_disc, dim_input_glob=dim_context,
                                                dim_output_loc=hid_loc[1], dim_hid_loc=hid_loc[1],
                                                dim_hid_glob=hid_glob[1], num_blocks=nb[1], **kw)
        else:
            hid_loc[0], hid_loc[1] = dim_cont, dim_disc
            hid_glob[0], hid_glob[1] = dim_context, 0
        if self.mode_fused:
            self.fused_encoder = EPiCEncoder(dim_time=Et, dim_input_loc=hid_loc[0] + hid_loc[1],
                                             dim_input_glob=hid_glob[0] + hid_glob[1], dim_output_loc=hid_loc[2],
                                             dim_hid_loc=hid_loc[2], dim_hid_glob=hid_glob[2], num_blocks=nb[2],
                                             project_input=not self.mode_branched, **kw)
        half = hid_loc[2] // 2 if self.mode_fused else 0
        self.dims = (hid_loc[0], hid_loc[1], hid_loc[2], half)
        self.continuous_head = _Head(Et + hid_loc[0] + half, hid_loc[0], d.dim_continuous)
        self.discrete_head = _Head(Et + hid_loc[1] + half, hid_loc[1], d.vocab_size * d.dim_discrete)

    def run(self, ctx, time, cont, disc, glob):
        H0, H1, H2, half = self.dims
        M = ctx.B * ctx.N
        dev = glob.device
        tseg = _time_seg(time)
        if self.mode_branched:
            both = torch.empty(M, H0 + H1, dtype=torch.float32, device=dev)
            h1, g1 = self.continuous_encoder.run(ctx, time, cont, glob, out=both[:, :H0])
            h2, g2 = self.discrete_encoder.run(ctx, time, disc, glob, out=both[:, H0:])
            g = torch.cat([g1, g2], dim=-1)
            if ctx.recording:
                ctx.tape.append(("cat", [g1, g2], g))
        else:
            h1, h2, g, both = cont, disc, glob, None
        if self.mode_fused:
            x_in = both if both is not None else [h1, h2]
            f, _ = self.fused_encoder.run(ctx, time, x_in, g)
            split = (f.shape[1] + 1) // 2                               # torch.tensor_split(f, 2)
            f1, f2 = f[:, :split], f[:, split:]
            hc = self.continuous_head.run(ctx, [tseg, (f1, SEG_ROW), (h1, SEG_ROW)])
            hd = self.discrete_head.run(ctx, [tseg, (f2, SEG_ROW), (h2, SEG_ROW)])
        else:
            hc = self.continuous_head.run(ctx, [tseg, (h1, SEG_ROW)])
            hd = self.discrete_head.run(ctx, [tseg, (h2, SEG_ROW)])
        return hc, hd

    def forward(self, state_local, state_global, tape=None, seed=0, offset0=0) -> TensorMultiModal:
        ctx, time, cont, disc, glob = _unpack(state_local, state_global, self.training, self.config.encoder.dropout,
                                              tape, seed, offset0)
        hc, hd = self.run(ctx, time, cont, disc, glob)
        B, N = ctx.B, ctx.N
        return TensorMultiModal(None, hc.view(B, N, -1), hd.view(B, N, -1), state_local.mask)
