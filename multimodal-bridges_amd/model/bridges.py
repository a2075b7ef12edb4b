"""Bridges: same class names, constructor kwargs and methods as the reference
(model/bridges.py), backed by the fused HIP kernels of libmmb_hip.so.

  UniformFlow(sigma) / SchrodingerBridge(sigma): .sample(t, batch) .drift(state, batch) .diffusion(state)
  TelegraphBridge(gamma, vocab_size, thermostat_fn): .sample(t, batch) .rate(state, heads)
        .transition_probability(t, k0, k1) .conditional_probability(t_in, t_out, k_in, k_out)

`sample` draws its noise with torch's device generator (so seeding behaves as users expect)
and hands it to the injected-noise kernel; pass noise=... to inject it explicitly.  The
training loop itself does not come through here: MultiModalBridgeMatching.sample_bridges
uses the single fused Philox kernel (mmb_bridge_sample_rng).
"""
import torch

from tensorclass import TensorMultiModal
from datamodules.datasets import DataCoupling
from mmb_hip import ops


def _flat_time(t, B):
    if not isinstance(t, torch.Tensor):
        raise TypeError("time must be a tensor of one value per jet")
    t = t.reshape(-1)
    if t.numel() != B:
        t = t.reshape(B, -1)[:, 0]
    return t.contiguous().float()


class _ContinuousBridge:
    kind = -1

    def __init__(self, sigma):
        self.sigma = sigma

    def sample(self, t, batch: DataCoupling, noise=None):
        x0 = batch.source.continuous.contiguous()
        x1 = batch.target.continuous.contiguous()
        z = torch.randn_like(x1) if noise is None else noise
        tf = _flat_time(t, x1.shape[0])
        _, xt, _ = ops.bridge_sample(tf, 0.0, x0, x1, z.contiguous(), None, None, None, self.kind, self.sigma, 0, 0.0, 0)
        return xt

    def drift(self, state: TensorMultiModal, batch: DataCoupling):
        xt = state.continuous.contiguous()
        tf = _flat_time(state.time, xt.shape[0])
        return ops.bridge_drift(tf, xt, batch.source.continuous.contiguous(), batch.target.continuous.contiguous(),
                                self.kind)


class UniformFlow(_ContinuousBridge):
    """Conditional OT flow matching: x_t = t x1 + (1-t) x0 + sigma z; drift target x1 - x0."""
    kind = ops.BRIDGES["UniformFlow"]

    def diffusion(self, state: TensorMultiModal):
        return 0.0


class SchrodingerBridge(_ContinuousBridge):
    """Brownian bridge: x_t = t x1 + (1-t) x0 + sigma sqrt(t(1-t)) z."""
    kind = ops.BRIDGES["SchrodingerBridge"]

    def diffusion(self, state: TensorMultiModal):
        # per-particle sigma_t (B,N,1).  Unlike the reference (bridges.py:106) the state's time
        # tensor is NOT mutated -- that mutation is what crashes its Euler-Maruyama path (Q5).
        t = state.time
        if t.dim() == 2:
            t = t.unsqueeze(1).repeat(1, state.shape[-1], 1)
        return self.sigma * torch.sqrt(t * (1.0 - t))


class TelegraphBridge:
    """Multivariate telegraph (Markov jump) bridge over vocab_size states."""

    def __init__(self, gamma, vocab_size, thermostat_fn):
        self.gamma = gamma
        self.vocab_size = vocab_size
        self.thermostat = thermostat_fn

    @property
    def thermo_kind(self):
        return self.thermostat.kind

    def sample(self, t, batch: DataCoupling, noise=None):
        k0 = batch.source.discrete.contiguous()
        k1 = batch.target.discrete.contiguous()
        B, N = k1.shape[0], k1.shape[1]
        u = torch.rand(B, N, device=k1.device) if noise is None else noise.reshape(B, N).contiguous()
        tf = _flat_time(t, B)
        _, _, kt = ops.bridge_sample(tf, 0.0, None, None, None, k0, k1, u, 0, 0.0, self.thermo_kind, self.gamma,
                                     self.vocab_size)
        return kt if kt.dim() == 3 else kt.unsqueeze(-1)

    def rate(self, state: TensorMultiModal, heads: TensorMultiModal):
        """rate_v = 1 + (w V/(1-w)) softmax(logits)_v + w softmax(logits)_k.  The reference's
        range assert on k (a device->host sync every step) is dropped; k is validated where it
        enters the loop."""
        logits = heads.discrete.contiguous()
        k = state.discrete.contiguous()
        tf = _flat_time(state.time, logits.shape[0])
        return ops.telegraph_rate(logits, k, tf, 0.0, self.thermo_kind, self.gamma)

    def transition_probability(self, t, k0, k1):
        B = k0.shape[0]
        k0 = k0.reshape(B, -1, 1).contiguous()
        k1 = k1.reshape(B, -1, 1).contiguous()
        return ops.telegraph_probs(_flat_time(t, B), k0, k1, self.thermo_kind, self.gamma, self.vocab_size)

    def conditional_probability(self, t_in, t_out, k_in, k_out):
        """p(k_out at t_out | k_in at t_in) = 1/V + w (delta - 1/V) (host-level helper)."""
        dev = k_in.device
        B = k_out.size(0)
        as_t = lambda t: t.to(dev) if isinstance(t, torch.Tensor) else torch.full((B,), float(t), device=dev)
        w = self.thermostat.w_ts(as_t(t_in), as_t(t_out))
        k_out = k_out if k_out.dim() == 3 else k_out[:, :, None]
        k_in = k_in if k_in.dim() == 3 else k_in[:, :, None]
        kron = (k_out == k_in).float()
        return 1.0 / self.vocab_size + w[:, None, None] * ((-1.0 / self.vocab_size) + kron)
