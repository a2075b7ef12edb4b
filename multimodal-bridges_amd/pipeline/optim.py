"""FlatAdamW: torch.optim.AdamW + global-norm gradient clipping on ONE flat parameter buffer and ONE flat gradient
buffer (the single all-reduce bucket of data-parallel training), executed by mmb_adamw_clip_step in two launches.

The reference configures `AdamW(self.parameters(), lr, weight_decay, betas, eps)` (multimodal_bridge_matching.py:
151-154) and Lightning clips the global gradient norm at `gradient_clip_val` (pipeline/experiment.py:217) -- per
step ~25 multi-tensor launches, ~40 fills and ~40 small copies in torch.  Here every parameter's `.data` and `.grad`
are views of two flat fp32 buffers; parameters that never receive a gradient upstream (their grad stays None there,
so torch's AdamW skips them entirely, weight decay included: the truncated token embedding, quirk Q1, and the fused
EPiC encoder's never-executed last layer, K2) are laid out at the END of the buffers and excluded from the update.

It is a torch.optim.Optimizer subclass so that the registry's lr schedulers drive it unchanged.
"""
import math

import torch

from mmb_hip import ops


def unused_parameter_names(model):
    """Names of the parameters whose gradient is None in the reference (Q1, K2)."""
    names = set()
    for n, _ in model.named_parameters():
        if n.startswith("embedder.discrete_embedding."):
            names.add(n)
    fe = getattr(getattr(model, "encoder", None), "fused_encoder", None)
    if fe is not None and hasattr(fe, "epic_layers") and not hasattr(fe, "epic_proj") and len(fe.epic_layers):
        last = f"encoder.fused_encoder.epic_layers.{len(fe.epic_layers) - 1}."
        names |= {n for n, _ in model.named_parameters() if n.startswith(last)}
    return names


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, named_params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False,
                 unused=(), maximize=False, foreach=None, fused=None, capturable=False, differentiable=False):
        # foreach / fused / capturable pick torch's implementation, not the update rule: accepted and ignored
        if amsgrad or maximize or differentiable:
            raise ValueError("FlatAdamW: amsgrad / maximize / differentiable are not implemented (no reference config uses them)")
        named = [(n, p) for n, p in named_params if p.requires_grad]
        self._model_order = [n for n, _ in named]
        unused = set(unused)
        order = [(n, p) for n, p in named if n not in unused] + [(n, p) for n, p in named if n in unused]
        params = [p for _, p in order]
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdamW runs on the HIP path only (no CPU fallback)")
        n = sum(p.numel() for p in params)
        self.n = n
        self.n_active = sum(p.numel() for nme, p in order if nme not in unused)
        self.names = [nme for nme, _ in order]
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)          # gradients: the all-reduce bucket
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._ws = torch.empty(ops.ADAMW_WS, dtype=torch.float32, device=dev)
        self.last_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._params = params
        self.t = 0
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + k].view_as(p)
                p.grad = self.flat[off:off + k].view_as(p)
                off += k

    # ---- the FlatGrads interface of pipeline/experiment.py
    def zero(self):
        self.flat.zero_()
        off = 0
        for p in self._params:                                   # re-attach views if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_grad(self, set_to_none=False):
        self.zero()

    def all_reduce_mean(self, dist, world):
        if dist is not None and world > 1:
            if dist.get_backend() == "gloo" and self.flat.is_cuda:            # rehearsal only
                host = self.flat.cpu()
                dist.all_reduce(host)
                self.flat.copy_(host)
            else:
                dist.all_reduce(self.flat)
            self.flat.div_(world)

    @torch.no_grad()
    def step(self, closure=None, max_norm=None):
        """clip (when max_norm) + AdamW; the norm of the unclipped gradient is left in self.last_norm (device)."""
        g = self.param_groups[0]
        self.t += 1
        ops.adamw_clip_step_(self.flat_p, self.flat, self.exp_avg, self.exp_avg_sq, self.n_active, g["lr"], g["betas"],
                             g["eps"], g["weight_decay"], self.t, max_norm, self._ws, norm_out=self.last_norm)
        torch._C._increment_version(self._params)                 # raw-pointer writes: invalidate the weight caches

    def state_dict(self):
        return {"flat_adamw": True, "t": self.t, "names": list(self.names), "exp_avg": self.exp_avg.cpu(),
                "exp_avg_sq": self.exp_avg_sq.cpu(),
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        """Takes FlatAdamW's own format, or a torch.optim.AdamW state_dict (what the reference's Lightning checkpoints
        hold under optimizer_states[0], and what earlier checkpoints of this package held): per-parameter exp_avg /
        exp_avg_sq / step in the order of model.parameters() over the parameters that had a gradient."""
        if sd.get("flat_adamw"):
            if list(sd["names"]) != list(self.names):
                raise ValueError("FlatAdamW.load_state_dict: checkpoint was written for another parameter list")
            self.t = int(sd["t"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            for g, s in zip(self.param_groups, sd["param_groups"]):
                g.update(s)
            return
        if "state" not in sd or "param_groups" not in sd:
            raise ValueError("FlatAdamW.load_state_dict: neither a FlatAdamW nor a torch.optim.AdamW state_dict")
        self._load_torch_adamw(sd)

    def _load_torch_adamw(self, sd, param_order=None):
        """param_order: names in the order the torch optimizer was given its parameters (default: model order of the
        trainable parameters, i.e. self._model_order); torch numbers them 0..n-1 across its param_groups."""
        order = list(param_order) if param_order is not None else list(self._model_order)
        ids = [i for g in sd["param_groups"] for i in g["params"]]
        if len(ids) != len(order):
            raise ValueError(f"FlatAdamW.load_state_dict: torch state has {len(ids)} parameters, the model {len(order)}")
        offs, off = {}, 0
        for n, p in zip(self.names, self._params):
            offs[n] = (off, p.numel())
            off += p.numel()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = set()
        for i, name in zip(ids, order):
            st = sd["state"].get(i)
            if st is None:                                   # never received a gradient (Q1 / K2): torch keeps no state
                continue
            o, k = offs[name]
            if st["exp_avg"].numel() != k:
                raise ValueError(f"FlatAdamW.load_state_dict: {name} has {k} elements, the checkpoint {st['exp_avg'].numel()}")
            if o >= self.n_active:
                raise ValueError(f"FlatAdamW.load_state_dict: {name} carries optimizer state but is excluded from the update here")
            self.exp_avg[o:o + k].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError(f"FlatAdamW.load_state_dict: parameters at different step counts {sorted(steps)}")
        self.t = steps.pop() if steps else 0
        g0 = sd["param_groups"][0]
        if g0.get("amsgrad") or g0.get("maximize"):
            raise ValueError("FlatAdamW.load_state_dict: amsgrad / maximize checkpoints are not supported")
        for k in ("lr", "betas", "eps", "weight_decay", "initial_lr"):
            if k in g0:
                self.param_groups[0][k] = tuple(g0[k]) if k == "betas" else g0[k]
