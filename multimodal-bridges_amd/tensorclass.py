"""TensorMultiModal: the boundary container of the hot path.

Same fields, methods and (quirky) semantics as the reference's tensorclass.py:11-255 so that
user code and YAML-driven pipelines keep working: four optional tensors (time, continuous,
discrete, mask); `shape` is the shape of the LAST available mode minus its feature axis;
an all-None instance is falsy; apply_mask truncates `discrete` to int64 (quirk Q1).
HDF5 persistence is used when h5py is importable, otherwise an .npz with the same keys.
"""
from dataclasses import dataclass
from typing import List, Optional

import torch

_FIELDS = ("time", "continuous", "discrete", "mask")
_MODES = ("time", "continuous", "discrete")


@dataclass
class TensorMultiModal:
    time: Optional[torch.Tensor] = None
    continuous: Optional[torch.Tensor] = None
    discrete: Optional[torch.Tensor] = None
    mask: Optional[torch.Tensor] = None

    # ---- introspection -------------------------------------------------------------
    def available_modes(self) -> List[str]:
        return [m for m in _MODES if getattr(self, m) is not None]

    def _last(self):
        modes = self.available_modes()
        return getattr(self, modes[-1]) if modes else None

    @property
    def ndim(self):
        last = self._last()
        return 0 if last is None else last.dim()

    @property
    def shape(self):
        last = self._last()
        return None if last is None else last.shape[:-1]

    def __len__(self):
        last = self._last()
        return 0 if last is None else len(last)

    @property
    def has_continuous(self):
        return self.continuous is not None

    @property
    def has_discrete(self):
        return self.discrete is not None

    def __getitem__(self, index) -> "TensorMultiModal":
        return TensorMultiModal(*[None if getattr(self, f) is None else getattr(self, f)[index] for f in _FIELDS])

    # ---- whole-state transforms ------------------------------------------------------
    def _map(self, fn) -> "TensorMultiModal":
        vals = []
        for f in _FIELDS:
            v = getattr(self, f)
            vals.append(fn(v) if isinstance(v, torch.Tensor) else None)
        return TensorMultiModal(*vals)

    def to(self, device):
        return self._map(lambda a: a.to(device))

    def detach(self):
        return self._map(torch.Tensor.detach)

    def cpu(self):
        return self._map(torch.Tensor.cpu)

    def clone(self):
        return self._map(torch.Tensor.clone)

    def _tensor_op(self, op, *args, mode=None, **kw) -> "TensorMultiModal":
        if mode is not None and mode not in _FIELDS:
            raise ValueError(f"Invalid mode '{mode}'. Choose from {list(_FIELDS)}")
        vals = []
        for f in _FIELDS:
            v = getattr(self, f)
            hit = v is not None and (mode is None or mode == f)
            vals.append(op(v, *args, **kw) if hit else v)
        return TensorMultiModal(*vals)

    def unsqueeze(self, dim, mode=None):
        return self._tensor_op(torch.unsqueeze, dim, mode=mode)

    def squeeze(self, dim=None, mode=None):
        return self._tensor_op(torch.squeeze, dim, mode=mode)

    def reshape(self, *shape, mode=None):
        return self._tensor_op(torch.reshape, *shape, mode=mode)

    def expand(self, *sizes, mode=None):
        return self._tensor_op(torch.Tensor.expand, *sizes, mode=mode)

    def repeat(self, *repeats, mode=None):
        return self._tensor_op(torch.Tensor.repeat, *repeats, mode=mode)

    # ---- the two mutators the hot path uses -------------------------------------------
    def broadcast_time(self):
        """(B,1) -> (B,D,1): repeat the per-jet time over the particle axis."""
        D = self.shape[-1]
        self.time = self.time.unsqueeze(1).repeat(1, D, 1)

    def apply_mask(self, condition=None):
        m = self.mask if condition is None else condition
        if self.time is not None:
            self.time *= m
        if self.continuous is not None:
            self.continuous *= m
        if self.discrete is not None:
            self.discrete = (self.discrete * m).long()      # truncation toward zero (Q1)

    # ---- combination -------------------------------------------------------------------
    @staticmethod
    def _combine(states, fn, dim):
        out = []
        for f in _FIELDS:
            parts = [getattr(s, f, None) for s in states]
            parts = [p for p in parts if p is not None]
            out.append(fn(parts, dim=dim) if parts else None)
        return TensorMultiModal(*out)

    @staticmethod
    def cat(states: List["TensorMultiModal"], dim=0) -> "TensorMultiModal":
        return TensorMultiModal._combine(states, torch.cat, dim)

    @staticmethod
    def stack(states: List["TensorMultiModal"], dim=0) -> "TensorMultiModal":
        return TensorMultiModal._combine(states, torch.stack, dim)

    @classmethod
    def sample_from(cls, distribution, shape, device=None) -> "TensorMultiModal":
        state = cls(*distribution.sample(shape))
        return state.to(device) if device else state

    # ---- persistence ---------------------------------------------------------------------
    @classmethod
    def load_from(cls, path, device=None, transform=None) -> "TensorMultiModal":
        data = dict.fromkeys(_FIELDS)
        try:
            import h5py
            with h5py.File(path, "r") as f:
                for key in _FIELDS:
                    if key in f:
                        data[key] = torch.from_numpy(f[key][:])
        except ImportError:
            import numpy as np
            with np.load(path if str(path).endswith(".npz") else str(path) + ".npz") as f:
                for key in _FIELDS:
                    if key in f.files:
                        data[key] = torch.from_numpy(f[key])
        if callable(transform):
            data = {k: (transform(v) if v is not None else None) for k, v in data.items()}
        elif isinstance(transform, dict):
            for key, fn in transform.items():
                if data.get(key) is not None and callable(fn):
                    data[key] = fn(data[key])
        state = cls(**data)
        return state.to(device) if device else state

    def save_to(self, path):
        payload = {m: getattr(self, m).detach().cpu().numpy() for m in self.available_modes()}
        payload["mask"] = self.mask.detach().cpu().numpy()
        try:
            import h5py
            with h5py.File(path, "w") as f:
                for key, arr in payload.items():
                    f.create_dataset(key, data=arr)
        except ImportError:
            import numpy as np
            np.savez(path if str(path).endswith(".npz") else str(path) + ".npz", **payload)
