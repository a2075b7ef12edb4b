"""Synthetic AspenOpenJets-shaped particle clouds: the real AOJ HDF5 files are
not available offline, so benchmarks and examples draw clouds with the same shape and
statistics: multiplicity ~ the 151-bin AOJ pmf (metadata.json 'categorical_probs', clipped at
N, never 0), ones in n random slots, standardised kinematics x1 ~ N(0,1)*mask and tokens
k1 ~ the AOJ flavour/charge frequencies * mask."""
import json
import os

import torch

from tensorclass import TensorMultiModal
from datamodules.datasets import DataCoupling

# AOJ flavour x charge token frequencies (photon, h0, h-, h+, e-, e+, mu-, mu+)
AOJ_TOKEN_FREQ = [0.304, 0.0847, 0.2986, 0.3078, 0.0013, 0.0014, 0.0011, 0.001]


def load_pmf(metadata_dir):
    with open(os.path.join(metadata_dir, "metadata.json")) as fh:
        return json.load(fh)["categorical_probs"]


def synthetic_target(B, N, metadata_dir, seed=1234, Dc=3, V=8, modality="multi-modal", device="cpu",
                     masks="sampled") -> TensorMultiModal:
    g = torch.Generator().manual_seed(seed)
    if masks == "ones":
        mask = torch.ones(B, N, 1, dtype=torch.int64)
    else:
        pmf = torch.tensor(load_pmf(metadata_dir), dtype=torch.float64)
        pmf[0] = 0.0
        n = torch.multinomial(pmf / pmf.sum(), B, True, generator=g).clamp(min=1, max=N)
        rank = torch.rand(B, N, generator=g).argsort(dim=1).argsort(dim=1)
        mask = (rank < n[:, None]).long().unsqueeze(-1)
    x = k = None
    if modality in ("continuous", "multi-modal"):
        x = torch.randn(B, N, Dc, generator=g) * mask
    if modality in ("discrete", "multi-modal"):
        f = torch.tensor(AOJ_TOKEN_FREQ[:V], dtype=torch.float64)
        k = torch.multinomial(f / f.sum(), B * N, True, generator=g).reshape(B, N, 1) * mask
    return TensorMultiModal(None, x, k, mask).to(device)


def synthetic_batch(B, N, metadata_dir, **kw) -> DataCoupling:
    """DataCoupling with an absent source (the model samples it) and a synthetic target."""
    return DataCoupling(TensorMultiModal(), synthetic_target(B, N, metadata_dir, **kw), None)
