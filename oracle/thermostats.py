"""Oracle: thermostat schedules (test infrastructure; see oracle/__init__.py).

Follows model/thermostats.py:21-52.
"""
import torch

KINDS = ("ConstantThermostat", "InverseThermostat", "LinearThermostat",
         "InverseSquareThermostat", "SigmoidThermostat")


def _as_tensor(t, like=None):
    if isinstance(t, torch.Tensor):
        return t
    if like is not None:
        return torch.full_like(like, float(t))
    return torch.tensor(float(t))


def beta_integral(kind, t0, t1):
    """int_{t0}^{t1} beta(r) dr for the five schedules (thermostats.py:27,33,39,45,51)."""
    if kind == "ConstantThermostat":
        return t1 - t0
    if kind == "InverseThermostat":
        return torch.log(t1 / t0)
    if kind == "LinearThermostat":
        return (t1 ** 2 - t0 ** 2) / 2
    if kind == "InverseSquareThermostat":
        return (t1 - t0) / (t1 * t0)
    if kind == "SigmoidThermostat":
        t0 = _as_tensor(t0, t1 if isinstance(t1, torch.Tensor) else None)
        t1 = _as_tensor(t1, t0)
        return torch.tanh(t1 / 2) - torch.tanh(t0 / 2)
    raise KeyError(kind)


def w_ts(kind, gamma, vocab_size, t0, t1):
    """w(t0,t1) = exp(-V * gamma * integral)  (thermostats.py:21-22)."""
    integ = beta_integral(kind, t0, t1)
    if not isinstance(integ, torch.Tensor):
        integ = torch.tensor(float(integ))
    return torch.exp(-vocab_size * gamma * integ)
