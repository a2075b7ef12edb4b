"""MultiModalMLP: the small per-particle MLP of the toy hybrid experiment (BASELINE configs[0]) on the
HIP path.  Reference: notebooks/data_toys.ipynb cell 3 (the class lives in the notebook, not in the package).

Same module tree and parameter names as the notebook class, so its state_dict loads:
    embed.weight (V, H/4)                           nn.Embedding
    body.{0,2,4,6}.{weight,bias}                    Linear(1+Dc+H/4, H), 3 x Linear(H, H), SELU between them
    continous_head.{weight,bias} (sic)              Linear(H, Dc)
    discrete_head.{weight,bias}                     Linear(H, V)
forward: h = body(cat[time, continuous, embed(discrete)]); heads = (continous_head(h), discrete_head(h)).

The notebook class takes the RAW state (time (B,1,1), continuous (B,1,Dc), tokens (B,1,1)), not the output of
MultiModalEmbedder, and so does this one: `raw_state = True` tells MultiModalBridgeMatching to hand the bridge state
over as it is.  It is written for any N (the MLP is per particle; the toy has N = 1).  What runs: one mmb_toy_rows
launch builds the input rows, the six Linears are mmb_linear_cat launches (SELU in the epilogue); the reverse pass is
the plain-Linear tape of encoders/backward.py plus mmb_toy_embedding_bwd.
"""
import torch
from torch import nn

from tensorclass import TensorMultiModal
from encoders.epic import ExecCtx
from encoders.particle_transformer import PlainLinear
from mmb_hip import ops
from mmb_hip.ops import SEG_ROW


class MultiModalMLP(nn.Module):
    raw_state = True

    def __init__(self, config=None, dim=2, vocab_size=8, dim_hidden=64):
        super().__init__()
        if config is not None and hasattr(config, "data"):
            dim = config.data.dim_continuous
            vocab_size = config.data.vocab_size
            dim_hidden = getattr(config.encoder, "dim_hidden_local", None) or dim_hidden
            if isinstance(dim_hidden, (list, tuple)):
                dim_hidden = dim_hidden[0]
        elif isinstance(config, int):                     # notebook-style positional call MultiModalMLP(dim)
            dim = config
        self.config = config
        self.dim, self.vocab_size, self.dim_hidden = dim, vocab_size, dim_hidden
        self.embed = nn.Embedding(vocab_size, dim_hidden // 4)
        self.body = nn.Sequential(nn.Linear(1 + dim + dim_hidden // 4, dim_hidden), nn.SELU(),
                                  nn.Linear(dim_hidden, dim_hidden), nn.SELU(),
                                  nn.Linear(dim_hidden, dim_hidden), nn.SELU(),
                                  nn.Linear(dim_hidden, dim_hidden))
        self.continous_head = nn.Linear(dim_hidden, dim)
        self.discrete_head = nn.Linear(dim_hidden, vocab_size)
        self._specs = None

    def _plan(self):
        if self._specs is None:
            mk = lambda lin: PlainLinear(lin.out_features, [(lin.weight, lin.in_features)], [lin.bias])
            self._specs = ([mk(self.body[i]) for i in (0, 2, 4, 6)], mk(self.continous_head), mk(self.discrete_head))
        return self._specs

    def run(self, ctx, t, x_rows, k_rows):
        """t: (B,) per jet or 1 element; x_rows (P,Dc) f32; k_rows (P,) int64 -> (head_c (P,Dc), head_d (P,V))"""
        body, head_c, head_d = self._plan()
        table = self.embed.weight.detach()
        rows = ops.toy_rows(t, x_rows, k_rows, table, ctx.N)
        if ctx.recording:
            ctx.tape.append(("toyin", self, ctx.mi, rows, k_rows, 1 + x_rows.shape[1]))
        h = rows
        for i, spec in enumerate(body):
            h = spec.run(ctx, [(h, SEG_ROW)], act="selu" if i < 3 else None)
        return head_c.run(ctx, [(h, SEG_ROW)]), head_d.run(ctx, [(h, SEG_ROW)])

    def forward(self, state, state_global=None, tape=None, seed=0, offset0=0) -> TensorMultiModal:
        mask = state.mask
        if mask is None:                                  # the notebook's states carry no mask
            mask = torch.ones(*state.discrete.shape[:2], 1, dtype=torch.int64, device=state.discrete.device)
        mi = getattr(state, "_mask_info", None) or ops.mask_prepare(mask)
        B, N = mi.B, mi.N
        ctx = ExecCtx(mi, tape=tape)
        ts = getattr(state, "_t_scalar", None)
        if ts is not None:
            t = torch.full((1,), float(ts), dtype=torch.float32, device=mask.device)
        else:
            t = state.time.float().reshape(B, -1)[:, 0].contiguous()
        x = state.continuous.float().reshape(B * N, -1).contiguous()
        k = state.discrete.reshape(B * N).contiguous()
        hc, hd = self.run(ctx, t, x, k)
        return TensorMultiModal(None, hc.view(B, N, -1), hd.view(B, N, -1), mask)
