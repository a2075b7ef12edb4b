"""Oracle: per-step solver updates (test infrastructure; see oracle/__init__.py).

Follows model/solvers.py.  The encoder forward that the reference performs inside
each step is done by the caller; these functions take its heads as inputs.
"""
import torch

from .bridges import telegraph_rate, inverse_cdf_draw


def euler_step(x, drift, delta_t):
    """solvers.py:21-25  x += dt * v (returns a new tensor)."""
    return x + delta_t * drift


def euler_maruyama_step(x, drift, delta_t, diffusion, delta_w):
    """solvers.py:27-33  x += dt*v + sigma_t*dW -- note: NO sqrt(dt) factor (a18).
    `diffusion` is the per-jet sigma*sqrt(t(1-t)) broadcast to x (Q5-repaired: taken on a
    clone of the state so the time tensor is not mutated; documented deviation)."""
    return x + (delta_t * drift + diffusion * delta_w)


def tauleap_step(k, rates, delta_n, V, overflow="wrap"):
    """solvers.py:59-87 with the Poisson draw delta_n (B,N,V) injected.
    k (B,N,1) int64 -> new k (B,N,1) int64.  Padded slots are updated too (K3)."""
    ks = k.squeeze(-1)
    jump_ok = torch.sum(delta_n, dim=-1).type_as(ks) <= 1
    diff = torch.arange(V).view(1, 1, V) - ks[:, :, None]
    net = torch.sum(delta_n * diff, dim=-1).type_as(ks)
    if overflow == "wrap":
        out = (ks + net * jump_ok) % V
    else:
        out = torch.clamp(ks + net * jump_ok, min=0, max=V - 1)
    return out.unsqueeze(-1)


def discrete_euler_probs(k, rates, delta_t):
    """solvers.py:95-103  off-diagonal dp = min(rate*dt,1); diagonal = max(1-sum,0)."""
    ks = k.squeeze(-1)
    dp = (rates * delta_t).clamp(max=1.0)
    dp = dp.scatter(-1, ks[:, :, None], 0.0)
    dp = dp.scatter(-1, ks[:, :, None], (1.0 - dp.sum(dim=-1, keepdim=True)).clamp(min=0.0))
    return dp


def discrete_euler_step(k, rates, delta_t, u):
    """solvers.py:89-109 with Categorical.sample replaced by the injected-u inverse CDF."""
    dp = discrete_euler_probs(k, rates, delta_t)
    knew, margin = inverse_cdf_draw(dp, u.reshape(dp.shape[:-1]))
    return knew.unsqueeze(-1), margin


def rate_and_tauleap(t, k, logits, delta_t, delta_n, gamma, V, kind="ConstantThermostat",
                     overflow="wrap"):
    """tauleap_step as the reference runs it: rate() then the jump (solvers.py:60-81)."""
    rates = telegraph_rate(t, k, logits, gamma, V, kind)
    return tauleap_step(k, rates, delta_n, V, overflow), rates
