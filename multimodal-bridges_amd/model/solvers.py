"""Per-step solvers of the generation loop (reference: model/solvers.py), on the fused HIP
update kernels.

  ContinuousSolver(config, model).fwd_step(state, batch, delta_t) -> state
      method config.model.solver_continuous in {"euler", "euler_maruyama"}
  DiscreteSolver(config, model).fwd_step(state, batch, delta_t)   -> (state, rates)
      method config.model.solver_discrete in {"tauleap", "euler"}

Each step = one encoder forward (only the head that is consumed is evaluated) + ONE fused
update kernel: K-B (x += dt*v [+ sigma_t dW]) or K-C (softmax -> telegraph rates -> jump).
Nothing synchronises with the host; padded slots are updated like real ones (quirk K3).
`midpoint` and `predictor_corrector` are not provided: the first crashes upstream, the second
is unreachable from any config.

Noise: by default drawn in-kernel (Philox, keyed by seed/step).  Tests inject it through
`solver.noise = {"delta_n": [...], "u_jump": [...], "delta_w": [...]}` indexed by step.
"""
import torch

from mmb_hip import ops


def _time_args(state):
    """(per-jet tensor | None, scalar) from state.time: generation uses one scalar per step."""
    ts = getattr(state, "_t_scalar", None)
    if ts is not None:
        return None, float(ts)
    t = state.time
    B = state.shape[0]
    return t.reshape(B, -1)[:, 0].contiguous().float(), 0.0


def philox_offset(rank, tag, stream_id, step):
    """64-bit Philox offset of one per-step kernel of the layered generation loop: rank (16 bits) | tag (4: which
    draw) | generation call (20) | grid point (24).  The rank keeps the shards of a multi-GPU generation on
    different streams (the seed is the same on every rank)."""
    return ((rank & 0xFFFF) << 48) + ((tag & 0xF) << 44) + ((stream_id & 0xFFFFF) << 24) + (step & 0xFFFFFF)


class _SolverBase:
    def __init__(self, config, model):
        self.model = model
        self.noise = None
        self.step = 0
        self.seed = int(getattr(config, "seed", 0xA0A0))
        self.stream_id = 0
        self.rank = int(getattr(model, "rank", 0) or 0)

    def _offset(self, tag):
        return philox_offset(self.rank, tag, self.stream_id, self.step)


class ContinuousSolver(_SolverBase):
    def __init__(self, config, model):
        super().__init__(config, model)
        self.method = config.model.solver_continuous

    def fwd_step(self, state, batch, delta_t):
        if not state.has_continuous:
            return state
        if self.method == "euler":
            return self.euler_step(state, batch, delta_t)
        if self.method == "euler_maruyama":
            return self.euler_maruyama_step(state, batch, delta_t)
        raise ValueError(f"unknown continuous solver {self.method!r} (expected 'euler' or 'euler_maruyama')")

    def euler_step(self, state, batch, delta_t):
        heads = self.model.forward_heads(state, batch, which="continuous")
        ops.euler_step_(state.continuous, heads.continuous, float(delta_t))
        return state

    def euler_maruyama_step(self, state, batch, delta_t):
        """x += dt*v + sigma*sqrt(t(1-t))*dW -- no sqrt(dt), as upstream (solvers.py:32).  The
        diffusion coefficient is evaluated per jet without mutating state.time (repaired Q5)."""
        heads = self.model.forward_heads(state, batch, which="continuous")
        t, ts = _time_args(state)
        sigma = float(self.model.bridge_continuous.sigma) if hasattr(self.model.bridge_continuous, "kind") and \
            self.model.bridge_continuous.kind == 1 else 0.0
        dW = self.noise["delta_w"][self.step].contiguous() if self.noise and "delta_w" in self.noise else None
        ops.euler_maruyama_step_(state.continuous, heads.continuous, float(delta_t), t, ts, sigma, dW=dW,
                                 seed=self.seed, offset=self._offset(1))
        return state


class DiscreteSolver(_SolverBase):
    def __init__(self, config, model):
        super().__init__(config, model)
        self.method = config.model.solver_discrete
        self.vocab_size = config.data.vocab_size
        self.topk = getattr(config, "topk", None) or config.model.topk        # upstream reads config.topk (Q2)
        self.final = False                  # set on the last grid point: k = argmax rates
        self.want_rates = False

    def fwd_step(self, state, batch, delta_t):
        if not state.has_discrete:
            return state, None
        if self.method == "tauleap":
            return self.tauleap_step(state, batch, delta_t)
        if self.method == "euler":
            return self.euler_step(state, batch, delta_t)
        raise ValueError(f"unknown discrete solver {self.method!r} (expected 'tauleap' or 'euler')")

    def _common(self, state, batch):
        heads = self.model.forward_heads(state, batch, which="discrete")
        br = self.model.bridge_discrete
        t, ts = _time_args(state)
        rates = torch.empty_like(heads.discrete) if self.want_rates else None
        return heads.discrete, br, t, ts, rates

    def tauleap_step(self, state, batch, delta_t, overflow="wrap"):
        logits, br, t, ts, rates = self._common(state, batch)
        dn = self.noise["delta_n"][self.step].contiguous() if self.noise and "delta_n" in self.noise else None
        ops.rate_tauleap_(logits, state.discrete, t, ts, float(delta_t), br.thermo_kind, br.gamma, delta_n=dn,
                          seed=self.seed, offset=self._offset(2), overflow=overflow, final_argmax=self.final,
                          rates_out=rates)
        return state, rates

    def euler_step(self, state, batch, delta_t):
        logits, br, t, ts, rates = self._common(state, batch)
        u = self.noise["u_jump"][self.step].contiguous() if self.noise and "u_jump" in self.noise else None
        ops.rate_euler_jump_(logits, state.discrete, t, ts, float(delta_t), br.thermo_kind, br.gamma, u=u,
                             seed=self.seed, offset=self._offset(3), final_argmax=self.final, rates_out=rates)
        return state, rates
