"""Oracle: the toy 2-D hybrid distributions (test infrastructure; see oracle/__init__.py).

Follows distributions/toys.py:8-55.  NGaussians is restated op for op (torch CPU generator instead of the
reference's global one).  TwoMoons calls torchdyn.datasets.generate_moons upstream; torchdyn (pinned nowhere:
requirements.txt lists `torchdyn` without a version) is not installed in this image, so `two_moons` restates the
scikit-learn law generate_moons is documented to follow -- PARITY UNPINNED for that one function: nothing in
/root/reference holds a sample of it.  Both are distribution-level checkers of the on-device Philox samplers.
"""
import math

import numpy as np
import torch


def n_gaussians(num_points, dim=2, vocab_size=8, generator=None):
    """toys.py:15-43 -> (positions (n,1,2) f32, labels (n,1,1) i64), n = V * (num_points // V)."""
    per = num_points // vocab_size
    step = 2 * np.pi / vocab_size
    pos, lab = [], []
    normal_std = math.sqrt(math.sqrt(0.1))                # MultivariateNormal(0, sqrt(0.1) I): the covariance
    for i in range(vocab_size):
        pts = torch.randn(per, dim, generator=generator) * normal_std
        pts = pts + torch.tensor([np.cos(i * step), np.sin(i * step)], dtype=torch.float32) * 5
        pos.append(pts)
        lab += [i % vocab_size] * per
    pos = torch.cat(pos, 0).float()
    lab = torch.tensor(lab)
    idx = torch.randperm(len(lab), generator=generator)
    return pos[idx].unsqueeze(1), lab[idx][:, None].long().unsqueeze(1)


def two_moons(num_points, noise=0.2, generator=None):
    """toys.py:49-55 with generate_moons restated (sklearn.datasets.make_moons law) -> positions * 3 - 1."""
    n_out = num_points // 2
    n_in = num_points - n_out
    a_out = torch.linspace(0, math.pi, n_out)
    a_in = torch.linspace(0, math.pi, n_in)
    x = torch.cat([torch.stack([torch.cos(a_out), torch.sin(a_out)], 1),
                   torch.stack([1 - torch.cos(a_in), 1 - torch.sin(a_in) - 0.5], 1)], 0)
    y = torch.cat([torch.zeros(n_out), torch.ones(n_in)]).long()
    x = x + noise * torch.randn(num_points, 2, generator=generator)
    idx = torch.randperm(num_points, generator=generator)
    return (x[idx] * 3 - 1).unsqueeze(1), y[idx][:, None].unsqueeze(1)
