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


def synthetic_target(B, N=128, metadata_dir=None, seed=1234, Dc=3, V=8, modality="multi-modal", device="cpu",
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


class SyntheticJetDataModule:
    """Stand-in for the reference's JetDataModule(config) on synthetic AOJ-shaped jets
    (config.data.num_jets, batch_size, split_ratios, max_num_particles).  The whole (small)
    dataset lives in HBM; each rank iterates its own contiguous shard of the jets."""

    def __init__(self, config):
        self.config = config
        self._train = self._val = self._test = None

    def setup(self, stage=None, rank=0, world=1, device="cuda"):
        d = self.config.data
        n = int(d.num_jets or 1024)
        ratios = d.split_ratios or [0.8, 0.2]
        n_train = int(ratios[0] * n)
        self._args = dict(N=d.max_num_particles, metadata_dir=d.target_path, Dc=max(d.dim_continuous, 1), V=d.vocab_size,
                          modality=d.modality, device=device)
        self._rank, self._device = rank, device
        nv = n - n_train
        base, rem = divmod(nv, world)
        self._n_test = base + (1 if rank < rem else 0)
        if stage == "predict":
            # generation only needs the SHAPE of the target batch (the masks come from the source sampler, K4): the
            # rank's shard is drawn chunk by chunk in predict_dataloader, never as one 1M-jet host tensor
            self._train = self._val = self._test = None
            return
        full = synthetic_target(n, seed=1234, **self._args)
        base, rem = divmod(n_train, world)
        lo = rank * base + min(rank, rem)
        hi = lo + base + (1 if rank < rem else 0)
        self._train = full[lo:hi]
        base, rem = divmod(nv, world)
        lo = n_train + rank * base + min(rank, rem)
        hi = lo + base + (1 if rank < rem else 0)
        self._val = full[lo:hi]
        self._test = self._val

    def _batches(self, data, perm=None):
        bs = int(self.config.data.batch_size)
        out = []
        for i in range(0, len(data), bs):
            part = data[i:i + bs] if perm is None else data[perm[i:i + bs]]
            out.append(DataCoupling(TensorMultiModal(), part, None))
        return out

    def train_dataloader(self, epoch=None):
        """A list of batches; reshuffled per epoch with a seeded device permutation (shuffle=True upstream,
        jetmodule.py:102)."""
        perm = None
        if epoch is not None:
            g = torch.Generator(device="cpu").manual_seed(0x5EED + 7919 * int(epoch) + self._rank)
            perm = torch.randperm(len(self._train), generator=g).to(self._train.mask.device)
        return self._batches(self._train, perm)

    def val_dataloader(self):
        return self._batches(self._val)

    def predict_dataloader(self):
        """This rank's shard of the test jets in chunks of batch_size (S5: 125 000 jets per rank in chunks of 8192)."""
        if self._test is not None:
            yield from self._batches(self._test)
            return
        bs = int(self.config.data.batch_size)
        for i, lo in enumerate(range(0, self._n_test, bs)):
            nb = min(bs, self._n_test - lo)
            yield DataCoupling(TensorMultiModal(), synthetic_target(nb, seed=99991 * (self._rank + 1) + i, **self._args),
                               None)
