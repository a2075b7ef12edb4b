"""Oracle: losses, bridge sampling, training step and the generation loop
(test infrastructure; see oracle/__init__.py).

Follows model/multimodal_bridge_matching.py.  Every random draw is an INPUT
(noise-injection protocol): `noise` dicts carry the recorded tensors.
"""
import torch
import torch.nn.functional as F

from . import bridges as B_
from . import solvers as S_
from . import encoder as E_


def loss_fn(head_c, head_d, drift_target, k1, mask, vocab_freq=None):
    """multimodal_bridge_matching.py:165-192.  Denominator is sum(mask), NOT x Dc (a15)."""
    loss_c = torch.tensor(0.0)
    loss_d = torch.tensor(0.0)
    if head_c is not None:
        mse = (head_c - drift_target) ** 2
        mse = mse * mask
        loss_c = mse.sum() / mask.sum()
    if head_d is not None:
        w = torch.tensor(vocab_freq) if vocab_freq else None
        ce = F.cross_entropy(head_d.transpose(1, 2), k1.squeeze(-1).long(), weight=w,
                             reduction="none").unsqueeze(-1)
        ce = ce * mask
        loss_d = ce.sum() / mask.sum()
    return loss_c, loss_d


def multimode_loss(losses, weights="fixed", s=None):
    """multimodal_bridge_matching.py:268-300.  'learnable': sum 0.5*exp(-s_i)*L_i + 0.5*s_i."""
    if weights == "learnable":
        return sum(0.5 * torch.exp(-s[i]) * losses[i] + 0.5 * s[i] for i in range(len(losses)))
    w = [1.0, 1.0] if weights == "fixed" else weights
    w = torch.tensor(w)
    return sum(w[i] * losses[i] for i in range(len(losses)))


def sample_bridges(cfg, target, source, noise):
    """multimodal_bridge_matching.py:194-225 with injected noise:
    noise['t_u'] (B,) U[0,1) ; noise['z'] (B,N,Dc) ; noise['u_k'] (B,N) U[0,1).
    target/source: dicts with continuous/discrete/mask.  Returns dict state."""
    eps = cfg["time_eps"]
    t = eps + (1 - eps) * noise["t_u"]
    time = t.reshape(-1, 1, 1)
    xt = kt = margin = None
    if target.get("continuous") is not None:
        if cfg["bridge_continuous"] == "UniformFlow":
            xt = B_.uniform_flow_sample(time, source["continuous"], target["continuous"], noise["z"], cfg["sigma"])
        else:
            xt = B_.schrodinger_sample(time, source["continuous"], target["continuous"], noise["z"], cfg["sigma"])
    if target.get("discrete") is not None:
        kt, margin = B_.telegraph_sample(time, source["discrete"], target["discrete"], noise["u_k"],
                                         cfg["gamma"], cfg["vocab_size"], cfg["thermostat_fn"])
    return {"time": time, "continuous": xt, "discrete": kt, "mask": target["mask"], "margin": margin}


def drift_target(cfg, state, source, target):
    if cfg["bridge_continuous"] == "UniformFlow":
        return B_.uniform_flow_drift(state["continuous"], source["continuous"], target["continuous"])
    return B_.schrodinger_drift(state["time"], state["continuous"], source["continuous"], target["continuous"])


def training_step(p, cfg, target, source, noise):
    """multimodal_bridge_matching.py:77-94 -> (loss, loss_c, loss_d, state)."""
    st = sample_bridges(cfg, target, source, noise)
    hc, hd = E_.forward(p, st["time"], st["continuous"], st["discrete"], st["mask"],
                        cfg["encoder_name"], cfg.get("skip_connection", True), cfg.get("num_heads"))
    tgt = drift_target(cfg, st, source, target) if hc is not None else None
    lc, ld = loss_fn(hc, hd, tgt, target.get("discrete"), st["mask"], cfg.get("vocab_freq"))
    loss = multimode_loss([lc, ld], cfg.get("loss_weights", "fixed"), p.get("loss_multimode.loss_weights"))
    return loss, lc, ld, st


def time_grid(cfg):
    """multimodal_bridge_matching.py:233-236: fp32 linspace and fp32 tensor delta_t."""
    eps, steps = cfg["time_eps"], cfg["num_timesteps"]
    ts = torch.linspace(eps, 1.0 - eps, steps)
    dt = (ts[-1] - ts[0]) / (len(ts) - 1)
    return ts, dt


def simulate_dynamics(p, cfg, source, noise=None, max_steps=None):
    """multimodal_bridge_matching.py:227-259 (+ solvers.py) with injected per-step noise:
    noise['delta_n'][i] (B,N,V) for tauleap; noise['u_jump'][i] (B,N) for discrete euler;
    noise['delta_w'][i] (B,N,Dc) for euler_maruyama.
    TWO encoder forwards per step (the discrete one sees the updated x, same t); all
    `steps` grid points are stepped (a17); final k = argmax of the LAST step's rates.
    Padded slots are updated like real ones (K3); the source mask is used (K4)."""
    x = source.get("continuous")
    k = source.get("discrete")
    mask = source["mask"]
    x = None if x is None else x.clone()
    k = None if k is None else k.clone()
    Bn = mask.shape[0]
    V = cfg.get("vocab_size", 0)
    ts, dt = time_grid(cfg)
    name, skip = cfg["encoder_name"], cfg.get("skip_connection", True)
    rates = None
    margins = []
    noise = noise or {}
    for i, tv in enumerate(ts):
        if max_steps is not None and i >= max_steps:      # bounded CPU-baseline sample (bench.py)
            break
        time = torch.full((Bn, 1), tv.item())
        if x is not None:
            hc, _ = E_.forward(p, time, x, k, mask, name, skip, cfg.get("num_heads"))
            if cfg["solver_continuous"] == "euler":
                x = S_.euler_step(x, hc, dt)
            else:
                sig = B_.schrodinger_diffusion(time.unsqueeze(1).repeat(1, mask.shape[1], 1), cfg["sigma"])
                dw = noise["delta_w"][i] if "delta_w" in noise else torch.randn_like(x)   # solvers.py:31
                x = S_.euler_maruyama_step(x, hc, dt, sig, dw)
        if k is not None:
            _, hd = E_.forward(p, time, x, k, mask, name, skip, cfg.get("num_heads"))
            rates = B_.telegraph_rate(time, k, hd, cfg["gamma"], V, cfg["thermostat_fn"])
            if cfg["solver_discrete"] == "tauleap":
                dn = noise["delta_n"][i] if "delta_n" in noise else torch.poisson(rates * dt)  # solvers.py:64
                k = S_.tauleap_step(k, rates, dn, V)
            else:
                u = noise["u_jump"][i] if "u_jump" in noise else torch.rand(k.shape[:2])
                k, m = S_.discrete_euler_step(k, rates, dt, u)
                margins.append(m)
    if k is not None:
        k = torch.max(rates, dim=2)[1].unsqueeze(-1)
    return {"continuous": x, "discrete": k, "mask": mask, "rates": rates, "margins": margins}
