"""Oracle: continuous and discrete bridges (test infrastructure; see oracle/__init__.py).

Follows model/bridges.py.  All functions take plain tensors:
  t   : (B,1,1) or (B,1) or (B,) fp32 per-jet time
  x0,x1,xt,z : (B,N,Dc) fp32
  k0,k1,k : (B,N,1) int64
"""
import torch

from .thermostats import w_ts


# ---------------------------------------------------------------- continuous --

def uniform_flow_sample(t, x0, x1, z, sigma):
    """bridges.py:22-28  x_t = t*x1 + (1-t)*x0 + sigma*z  (z injected)."""
    xt = t * x1 + (1.0 - t) * x0
    return xt + sigma * z


def uniform_flow_drift(xt, x0, x1):
    """bridges.py:30-37  0*x_t + 1*x1 + (-1)*x0 (the literal 0*x_t term is kept)."""
    return 0.0 * xt + 1.0 * x1 + (-1.0) * x0


def schrodinger_sample(t, x0, x1, z, sigma):
    """bridges.py:87-93."""
    x = t * x1 + (1.0 - t) * x0
    std = sigma * torch.sqrt(t * (1.0 - t))
    return x + std * z


def schrodinger_drift(t, xt, x0, x1):
    """bridges.py:95-103; t broadcastable against (B,N,Dc)."""
    A = (1 - 2 * t) / (t * (1 - t))
    Bc = t ** 2 / (t * (1 - t))
    C = -1 * (1 - t) ** 2 / (t * (1 - t))
    return A * xt + Bc * x1 + C * x0


def schrodinger_diffusion(t, sigma):
    """bridges.py:105-108 (value only; the reference also mutates state.time, Q5)."""
    return sigma * torch.sqrt(t * (1.0 - t))


# ------------------------------------------------------------------ discrete --

def _cond_prob(kind, gamma, V, t_in, t_out, k_in, k_out):
    """bridges.py:193-213  1/V + w*((-1/V) + delta)."""
    w = w_ts(kind, gamma, V, t_in, t_out)
    kron = (k_out == k_in).float()
    return 1.0 / V + w[:, None, None] * ((-1.0 / V) + kron)


def telegraph_transition_probability(t, k0, k1, gamma, V, kind="ConstantThermostat"):
    """bridges.py:165-191  -> (B,N,V) fp32 (not normalised)."""
    t = t.clone().reshape(-1)
    B, N = k0.shape[0], k0.shape[1]
    k = torch.arange(0, V)[None, None, :].repeat(B, N, 1).float()
    one = torch.full((B,), 1.0)
    zero = torch.full((B,), 0.0)
    p_k_to_k1 = _cond_prob(kind, gamma, V, t, one, k, k1)
    p_k0_to_k = _cond_prob(kind, gamma, V, zero, t, k0, k)
    p_k0_to_k1 = _cond_prob(kind, gamma, V, zero, one, k0, k1)
    return (p_k_to_k1 * p_k0_to_k) / p_k0_to_k1


def inverse_cdf_draw(probs, u):
    """Noise-injection protocol: k = #{v : cdf_v <= u*cdf_{V-1}} with a
    sequential fp32 cdf.  probs (..., V), u (...) in [0,1).  Also returns the margin
    min_v |u*S - cdf_v| used to drop near-tie draws from bit-exact comparisons."""
    V = probs.shape[-1]
    cdf = torch.empty_like(probs)
    acc = torch.zeros_like(probs[..., 0])
    for v in range(V):
        acc = acc + probs[..., v]
        cdf[..., v] = acc
    thr = u * cdf[..., -1]
    k = (cdf <= thr[..., None]).sum(-1)
    k = k.clamp(max=V - 1)
    margin = (cdf - thr[..., None]).abs().min(-1)[0]
    return k.long(), margin


def telegraph_sample(t, k0, k1, u, gamma, V, kind="ConstantThermostat"):
    """bridges.py:124-135 with Categorical.sample replaced by the injected-u inverse CDF.
    Categorical normalises probs by their sum before torch.multinomial; the inverse
    CDF is scale invariant so the unnormalised probabilities are used directly."""
    p = telegraph_transition_probability(t, k0, k1, gamma, V, kind)
    k, margin = inverse_cdf_draw(p, u.reshape(p.shape[:-1]))
    return k.unsqueeze(-1), margin


def telegraph_rate(t, k, logits, gamma, V, kind="ConstantThermostat"):
    """bridges.py:137-163  rate_v = 1 + (w V/(1-w)) q_v + w q_k."""
    qx = torch.softmax(logits, dim=2)
    qy = torch.gather(qx, 2, k.long())
    t = t.reshape(-1)
    w = w_ts(kind, gamma, V, t, 1.0)
    A = 1.0
    Bc = (w * V) / (1.0 - w)
    C = w
    return A + Bc[:, None, None] * qx + C[:, None, None] * qy
