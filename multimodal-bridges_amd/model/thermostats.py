"""Thermostat schedules for the telegraph bridge: w(t0,t1) = exp(-V*gamma*int beta).

Same class names / constructor / `w_ts` signature as the reference (model/thermostats.py).
Each class also carries `kind`, the enum the fused HIP kernels switch on
(include/mmb_hip.h MMB_THERMO_*), so the per-jet scalar never needs its own launch on the
hot path.  `w_ts` itself is host-level convenience on whatever device its inputs live.
"""
import math

import torch


class Thermostat:
    kind = -1

    def __init__(self, gamma, vocab_size=8):
        self.gamma = gamma
        self.vocab_size = vocab_size

    def _integral(self, t0, t1):
        raise NotImplementedError

    def w_ts(self, t0, t1):
        integ = self._integral(t0, t1)
        if not isinstance(integ, torch.Tensor):
            return math.exp(-self.vocab_size * self.gamma * integ)
        return torch.exp(-self.vocab_size * self.gamma * integ)

    def wt_0(self, t):
        w = self.w_ts(t, 1)
        return w * self.vocab_size / (1 - w)

    def wt_1(self, t):
        return self.w_ts(t, 1)


def _tanh(x):
    return torch.tanh(x) if isinstance(x, torch.Tensor) else math.tanh(x)


def _log(x):
    return torch.log(x) if isinstance(x, torch.Tensor) else math.log(x)


class ConstantThermostat(Thermostat):
    """beta(r) = 1"""
    kind = 0

    def _integral(self, t0, t1):
        return t1 - t0


class InverseThermostat(Thermostat):
    """beta(r) = 1/r"""
    kind = 1

    def _integral(self, t0, t1):
        return _log(t1 / t0)


class LinearThermostat(Thermostat):
    """beta(r) = r"""
    kind = 2

    def _integral(self, t0, t1):
        return (t1 ** 2 - t0 ** 2) / 2


class InverseSquareThermostat(Thermostat):
    """beta(r) = 1/r^2"""
    kind = 3

    def _integral(self, t0, t1):
        return (t1 - t0) / (t1 * t0)


class SigmoidThermostat(Thermostat):
    """the reference's 'sigmoid' schedule: integral = tanh(t1/2) - tanh(t0/2)"""
    kind = 4

    def _integral(self, t0, t1):
        return _tanh(t1 / 2) - _tanh(t0 / 2)
