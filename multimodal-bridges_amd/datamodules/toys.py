"""ToyDataModule: the toy experiment's data (BASELINE configs[0]; reference: notebooks/data_toys.ipynb cell 4, which
draws a fresh source/target batch every iteration, and config_toys.yaml's num_jets/batch_size/split_ratios).

config.data.num_jets points are drawn ONCE on device from `target_name` (TwoMoons) and `source_name` (NGaussians)
through the registry; batches couple them index by index (an independent coupling: both sets are shuffled), each
rank iterating its own contiguous shard.  Same setup/train_dataloader/val_dataloader/predict_dataloader surface as
SyntheticJetDataModule, so ExperimentPipeline drives it unchanged."""
import torch

from tensorclass import TensorMultiModal
from datamodules.datasets import DataCoupling


class ToyDataModule:
    def __init__(self, config):
        self.config = config
        self._train = self._val = None

    def setup(self, stage=None, rank=0, world=1, device="cuda"):
        from pipeline.registry import registered_distributions
        d = self.config.data
        n = int(d.num_jets or 1024)
        tgt = registered_distributions[d.target_name](self.config)
        src = registered_distributions[d.source_name](self.config)
        tgt.manual_seed(1234)
        src.manual_seed(4321)
        target = tgt(n, device=device)
        source = src(n, device=device)
        n = min(len(target), len(source))                     # NGaussians rounds down to a multiple of V
        ratios = d.split_ratios or [0.9, 0.1]
        n_train = int(ratios[0] * n)

        def shard(lo, hi):
            base, rem = divmod(hi - lo, world)
            a = lo + rank * base + min(rank, rem)
            return a, a + base + (1 if rank < rem else 0)

        a, b = shard(0, n_train)
        self._train = (source[a:b], target[a:b])
        a, b = shard(n_train, n)
        self._val = (source[a:b], target[a:b])
        self._rank = rank

    def _batches(self, pair, perm=None):
        bs = int(self.config.data.batch_size)
        src, tgt = pair
        out = []
        for i in range(0, len(tgt), bs):
            sl = slice(i, i + bs) if perm is None else perm[i:i + bs]
            out.append(DataCoupling(src[sl], tgt[sl], None))
        return out

    def train_dataloader(self, epoch=None):
        perm = None
        if epoch is not None:
            g = torch.Generator(device="cpu").manual_seed(0x70F5 + 7919 * int(epoch) + self._rank)
            perm = torch.randperm(len(self._train[1]), generator=g).to(self._train[1].mask.device)
        return self._batches(self._train, perm)

    def val_dataloader(self):
        return self._batches(self._val)

    def predict_dataloader(self):
        """Generation starts from the held-out SOURCE points; the target only fixes the shape."""
        return self._batches(self._val)
