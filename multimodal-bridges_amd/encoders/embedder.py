"""MultiModalEmbedder on the HIP path (reference: encoders/embedder.py:11-97).

Parameter names match the reference: time_embedding.0.W (frozen Gaussian Fourier
frequencies, scale 30), time_embedding.1.{weight,bias}, continuous_embedding.{weight,bias},
discrete_embedding.{weight_g,weight_v} (weight-normed nn.Embedding).

One mmb_embed call produces the per-jet time embedding (B,Et), the masked continuous
features and the masked + TRUNCATED discrete features (quirk Q1: the reference's apply_mask
casts the embedded tokens to int64).  The (B,N,Et) repeat of the time embedding is never
written: LocalState keeps it factored and only materialises `.time` / the int64 `.discrete`
if somebody reads them.
"""
import torch
from torch import nn

from tensorclass import TensorMultiModal
from mmb_hip import ops


class LocalState(TensorMultiModal):
    """state_loc of the reference with lazily materialised `time` (B,N,Et) and `discrete`
    (B,N,Ed) int64.  The encoders read the factored fields instead:
    temb (B,Et), cont_rows (B*N,Ec) f32, disc_rows (B*N,Ed) f32 (already truncated), mi."""

    def __init__(self, temb, cont_rows, disc_rows, mask, mi):
        self.temb, self.cont_rows, self.disc_rows, self.mi = temb, cont_rows, disc_rows, mi
        self.mask = mask
        self._time = None
        self._disc = None
        B, N = mi.B, mi.N
        self.continuous = None if cont_rows is None else cont_rows.view(B, N, -1)

    @property
    def time(self):
        if self._time is None:
            self._time = self.temb[:, None, :] * self.mi.maskf[:, :, None]
        return self._time

    @time.setter
    def time(self, v):
        self._time = v

    @property
    def discrete(self):
        if self._disc is None and self.disc_rows is not None:
            self._disc = self.disc_rows.view(self.mi.B, self.mi.N, -1).long()
        return self._disc

    @discrete.setter
    def discrete(self, v):
        self._disc = v

    def available_modes(self):
        modes = ["time"]
        if self.cont_rows is not None:
            modes.append("continuous")
        if self.disc_rows is not None:
            modes.append("discrete")
        return modes

    @property
    def ndim(self):
        return 3

    @property
    def shape(self):
        return torch.Size((self.mi.B, self.mi.N))

    def __len__(self):
        return self.mi.B


class GaussianFourierProjection(nn.Module):
    """Frozen random Fourier frequencies W ~ N(0, scale^2); sin/cos features."""

    def __init__(self, embed_dim, scale=1.0):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embed_dim // 2) * scale, requires_grad=False)


class WNEmbedding(nn.Module):
    """weight-normed nn.Embedding (norm per row): table = g * v/||v||.  Never receives a
    gradient on this path because its output is truncated to integers (Q1)."""

    def __init__(self, num, dim):
        super().__init__()
        v = torch.randn(num, dim)
        self.weight_g = nn.Parameter(v.norm(dim=1, keepdim=True))
        self.weight_v = nn.Parameter(v)
        self._cache = None

    def table(self):
        v, g = self.weight_v, self.weight_g
        key = (v._version, g._version, v.data_ptr())
        if self._cache is None or self._cache[0] != key:
            n, d = v.shape
            off = torch.arange(n, dtype=torch.int32, device=v.device) * d
            ln = torch.full((n,), d, dtype=torch.int32, device=v.device)
            W, _ = ops.weight_norm(off, ln, v.detach().reshape(-1), g.detach().reshape(-1))
            self._cache = (key, W.view(n, d))
        return self._cache[1]


class MultiModalEmbedder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        dim_x = config.data.dim_continuous
        dim_k = config.data.vocab_size
        Et = config.encoder.dim_emb_time
        self.time_embedding = nn.Sequential(GaussianFourierProjection(embed_dim=Et, scale=30.0), nn.Linear(Et, Et))
        if config.data.modality in ("continuous", "multi-modal"):
            self.continuous_embedding = nn.Linear(dim_x, config.encoder.dim_emb_continuous)
        if config.data.modality in ("discrete", "multi-modal"):
            self.discrete_embedding = WNEmbedding(dim_k, config.encoder.dim_emb_discrete)

    def embed(self, time, continuous, discrete, mask, mi=None, tape=None):
        """time: per-jet tensor with B values (any of (B,), (B,1), (B,1,1)) or a python float."""
        if mi is None:
            mi = ops.mask_prepare(mask)
        if isinstance(time, torch.Tensor):
            t = time.reshape(-1)
            if t.numel() != mi.B:
                t = time.reshape(mi.B, -1)[:, 0]
            t, ts = t.contiguous().float(), 0.0
        else:
            t, ts = None, float(time)
        te = self.time_embedding
        has_c = continuous is not None and hasattr(self, "continuous_embedding")
        has_d = discrete is not None and hasattr(self, "discrete_embedding")
        x = continuous.contiguous() if has_c else None
        k = discrete.contiguous() if has_d else None
        r = ops.embed(
            t, ts, x, k, mi, te[0].W.detach(), te[1].weight.detach(), te[1].bias.detach(),
            self.continuous_embedding.weight.detach() if has_c else None,
            self.continuous_embedding.bias.detach() if has_c else None,
            self.discrete_embedding.table() if has_d else None, want_feat=tape is not None)
        temb, cont, disc = r[0], r[1], r[2]
        if tape is not None:
            tape.append(("embed", self, x, mi, temb, cont, r[3]))
        return LocalState(temb, cont, disc, mask, mi)

    def forward(self, state: TensorMultiModal, batch=None):
        loc = self.embed(state.time, state.continuous, state.discrete, state.mask)
        glob = TensorMultiModal(time=loc.temb)
        return loc, glob
