"""Deterministic synthetic weights / AOJ-shaped inputs shared by tools/make_golden.py,
the tests and bench.py (test infrastructure; see oracle/__init__.py).

Weights are generated per parameter NAME with numpy's PCG64 so that the harness (which
loads them into the imported reference model) and the tests (which load them into the
oracle and the HIP path) agree without having to commit multi-MB state dicts.
"""
import json
import os
import zlib

import numpy as np
import torch

# AOJ flavour/charge token frequencies (distributions/noise.py:382-391)
AOJ_TOKEN_FREQ = [0.304, 0.0847, 0.2986, 0.3078, 0.0013, 0.0014, 0.0011, 0.001]

_HERE = os.path.dirname(os.path.abspath(__file__))
PMF_FILE = os.path.join(_HERE, "..", "tests", "golden", "metadata.json")


def _rng(seed, name):
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def synth_param(name, shape, seed=0):
    """One parameter tensor from its reference name and shape."""
    r = _rng(seed, name)
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "weight_v" or leaf == "weight":
        fan_in = shape[-1] if len(shape) > 1 else 1
        a = r.standard_normal(shape) / np.sqrt(max(fan_in, 1))
        if "discrete_embedding" in name:          # embedding rows ~ N(0,1) like nn.Embedding
            a = r.standard_normal(shape)
    elif leaf == "weight_g":
        a = r.uniform(0.6, 1.4, shape)
        if "discrete_embedding" in name:          # row norm ~ sqrt(Ed) like wn(Embedding) init
            a = a * 2.5
    elif leaf == "bias":
        a = 0.1 * r.standard_normal(shape)
    elif leaf == "W":                             # GaussianFourierProjection, scale 30
        a = 30.0 * r.standard_normal(shape)
    elif leaf == "loss_weights":
        a = r.uniform(-0.5, 0.5, shape)
    else:
        a = 0.1 * r.standard_normal(shape)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def synth_state_dict(shapes, seed=0):
    """shapes: {name: shape} -> {name: tensor}."""
    return {n: synth_param(n, s, seed) for n, s in shapes.items()}


def load_pmf(n_max=None):
    """The 151-bin AOJ multiplicity pmf (data fixture taken from the reference's
    tests/resources/metadata.json:81-233 'categorical_probs')."""
    with open(PMF_FILE) as f:
        pmf = json.load(f)["categorical_probs"]
    return pmf


def aoj_like_batch(B, N, seed=1234, Dc=3, V=8, masks="sampled", modality="multi-modal"):
    """Synthetic AOJ-shaped target clouds: multiplicity ~ pmf (clipped at N,
    never 0), ones in n random slots, x1 ~ N(0,1)*mask, k1 ~ AOJ token freqs * mask."""
    g = torch.Generator().manual_seed(seed)
    if masks == "ones":
        mask = torch.ones(B, N, 1, dtype=torch.int64)
    else:
        pmf = torch.tensor(load_pmf(), dtype=torch.float64)
        pmf[0] = 0.0
        n = torch.multinomial(pmf / pmf.sum(), B, True, generator=g).clamp(min=1, max=N)
        r = torch.rand(B, N, generator=g)
        rank = r.argsort(dim=1).argsort(dim=1)
        mask = (rank < n[:, None]).long().unsqueeze(-1)
    x = k = None
    if modality in ("continuous", "multi-modal"):
        x = torch.randn(B, N, Dc, generator=g) * mask
    if modality in ("discrete", "multi-modal"):
        f = torch.tensor(AOJ_TOKEN_FREQ[:V], dtype=torch.float64)
        k = torch.multinomial(f / f.sum(), B * N, True, generator=g).reshape(B, N, 1) * mask
    return {"continuous": x, "discrete": k, "mask": mask}
