"""Toy 2-D hybrid distributions on device (reference: distributions/toys.py:8-55; BASELINE configs[0]).

`NGaussians(dim=2, vocab_size=8)(num_points, device) -> TensorMultiModal(continuous (n,1,2), discrete (n,1,1))` and
`TwoMoons(dim=2)(num_points, device)`: the reference's call signatures.  Both are also constructible from an
ExperimentConfigs (the registry calls `cls(config)`) and callable as `dist(shape=(B,N), device=..., sample_masks=...)`,
the way MultiModalBridgeMatching calls its source/target samplers.  One Philox kernel (mmb_toy_sample) draws the
points in index order; the shuffle the reference applies (torch.randperm) is a device permutation here.  States carry
an all-ones mask (the reference's carry none; every consumer here wants one).

TwoMoons: upstream calls torchdyn.datasets.generate_moons, which is not installed here -- the law implemented is the
scikit-learn one (see csrc/toy_kernels.hip); parity unpinned.
"""
import torch

from tensorclass import TensorMultiModal
from mmb_hip import ops


class _Toy:
    kind = 0

    def __init__(self, dim=2, vocab_size=8, noise=0.2):
        if hasattr(dim, "data"):                          # ExperimentConfigs
            cfg = dim
            dim, vocab_size = cfg.data.dim_continuous or 2, cfg.data.vocab_size or 8
            self.seed = int(getattr(cfg, "seed", 0xA0A0))
        else:
            self.seed = 0xA0A0
        if dim != 2:
            raise ValueError("the toy distributions are two-dimensional (toys.py builds (cos, sin) centres / two moons)")
        self.dim, self.vocab_size, self.noise = dim, vocab_size, noise
        self._calls = 0

    def manual_seed(self, seed, offset=0):
        self.seed, self._calls = int(seed), int(offset)

    def __call__(self, num_points=None, device="cuda", shape=None, sample_masks=False) -> TensorMultiModal:
        if shape is not None:
            if shape[1] != 1:
                raise ValueError(f"toy states have one particle per sample, got N={shape[1]}")
            num_points = shape[0]
        elif isinstance(num_points, (tuple, list, torch.Size)):
            num_points = num_points[0]
        self._calls += 1
        n = self._count(int(num_points))
        x, k = ops.toy_sample(self.kind, n, torch.device(device), V=self.vocab_size, noise=self.noise, seed=self.seed,
                              offset=(2 << 40) + (self.kind << 36) + self._calls)
        g = torch.Generator(device="cpu").manual_seed((self.seed * 1000003 + self._calls) & 0x7FFFFFFF)
        idx = torch.randperm(n, generator=g).to(x.device)                 # toys.py:39 / :51
        x, k = x[idx], k[idx]
        mask = torch.ones(n, 1, 1, dtype=torch.int64, device=x.device)
        return TensorMultiModal(None, x.unsqueeze(1), k.view(n, 1, 1), mask)

    def _count(self, n):
        return n


class NGaussians(_Toy):
    """V Gaussians on the radius-5 circle, covariance sqrt(0.1) I, label = the Gaussian's index (toys.py:8-43).
    As upstream, num_points // V points per Gaussian: the state has V * (num_points // V) samples."""
    kind = 0

    def _count(self, n):
        per = n // self.vocab_size
        if per == 0:
            raise ValueError(f"NGaussians: num_points={n} < {self.vocab_size} Gaussians gives an empty state")
        return per * self.vocab_size


class TwoMoons(_Toy):
    """generate_moons(n, noise=0.2) * 3 - 1, labels {0, 1} (toys.py:45-55)."""
    kind = 1
