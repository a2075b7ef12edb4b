"""JetDataModule(config): source/target coupling of particle-cloud datasets looked up in `registered_datasets`
(reference: datamodules/jetmodule.py:13-197, the Lightning datamodule ExperimentPipeline is built around).

Same constructor and the same setup / train_dataloader / val_dataloader / predict_dataloader surface, attributes
`.target`, `.source`, `.metadata`, contiguous train/val split by `split_ratios` (:178-190).  Without Lightning and
torch DataLoader workers: the pre-processed dataset lives in HBM and a "dataloader" is the list of DataCoupling
batches of this rank's shard (shuffle=True upstream -> a seeded permutation per epoch here); `setup` takes the
optional rank / world / device the pipeline passes.  A batch's absent source is an empty TensorMultiModal, so the
model samples it (datasets.py:25-35)."""
import numpy as np
import torch

from tensorclass import TensorMultiModal
from datamodules.datasets import DataCoupling


class JetDataModule:
    def __init__(self, config):
        from pipeline.registry import registered_datasets as JetDataset
        self.config = config
        d = config.data
        self.transform = d.transform
        self.features = {"continuous": d.continuous_features, "discrete": d.discrete_features}
        self.batch_size = d.batch_size
        self.num_workers = d.num_workers
        self.pin_memory = d.pin_memory
        self.split_ratios = d.split_ratios
        self.num_jets = d.num_jets
        self.max_num_particles = d.max_num_particles
        self.metadata = {}
        if d.source_path:
            self.source_dataset_train = JetDataset[d.source_name](data_dir=d.source_path, data_files=d.source_train_files)
            self.source_dataset_pred = JetDataset[d.source_name](data_dir=d.source_path, data_files=d.source_test_files)
        if d.target_path:
            self.target_dataset_train = JetDataset[d.target_name](data_dir=d.target_path, data_files=d.target_train_files)
            self.target_dataset_pred = JetDataset[d.target_name](data_dir=d.target_path, data_files=d.target_test_files)

    # ------------------------------------------------------------------ Lightning-style surface
    def setup(self, stage=None, rank=0, world=1, device="cuda"):
        self._rank, self._world, self._device = rank, world, device
        if stage == "fit" or stage is None:
            self._prepare_fit_datasets()
            n = len(self.target)
            idx0, idx1 = self._idx_data_split(n, self.split_ratios)
            self.train_dataset = self._shard(idx0)
            self.val_dataset = self._shard(idx1)
        if stage == "predict":
            self._prepare_predict_datasets()
            self.predict_dataset = self._shard((0, len(self.target)))

    def train_dataloader(self, epoch=None):
        perm = None
        if self.train_dataset is not None:
            lo, hi = self.train_dataset
            g = torch.Generator(device="cpu").manual_seed(0x5EED + 7919 * int(epoch or 0) + self._rank)
            perm = lo + torch.randperm(hi - lo, generator=g)                 # shuffle=True (:90)
        return self._get_dataloader(self.train_dataset, perm)

    def val_dataloader(self):
        return self._get_dataloader(self.val_dataset)

    def predict_dataloader(self):
        return self._get_dataloader(self.predict_dataset)

    # ------------------------------------------------------------------ data prep
    def _shard(self, span):
        """this rank's contiguous part of [lo, hi)"""
        if span is None:
            return None
        lo, hi = span
        base, rem = divmod(hi - lo, self._world)
        a = lo + self._rank * base + min(self._rank, rem)
        return a, a + base + (1 if self._rank < rem else 0)

    def _get_dataloader(self, span, perm=None):
        if span is None:
            return []
        lo, hi = span
        out = []
        for i in range(lo, hi, self.batch_size):
            if perm is None:
                sl = slice(i, min(i + self.batch_size, hi))
            else:
                sl = perm[i - lo:i - lo + self.batch_size].to(self.target.mask.device)
            src = self.source[sl] if hasattr(self, "source") else TensorMultiModal()
            out.append(DataCoupling(src, self.target[sl], None))
        return out

    def _call(self, dataset, **kw):
        return dataset(self.num_jets, self.max_num_particles, download=True, features=self.features,
                       device=self._device, **kw)

    def _prepare_fit_datasets(self):
        if hasattr(self, "source_dataset_train"):
            self.source, self.metadata["source"] = self._call(self.source_dataset_train, transform=self.transform)
        if hasattr(self, "target_dataset_train"):
            self.target, self.metadata["target"] = self._call(self.target_dataset_train, transform=self.transform)
        self._clear_unavailable_modes()

    def _prepare_predict_datasets(self):
        if hasattr(self, "source_dataset_pred"):
            self.source, _ = self._call(self.source_dataset_pred, transform=self.transform)
        if hasattr(self, "target_dataset_pred"):
            self.target, _ = self._call(self.target_dataset_pred)              # un-transformed, as upstream (:149-154)
        self._clear_unavailable_modes()

    def _clear_unavailable_modes(self):
        m = self.config.data.modality
        if m == "continuous":
            if hasattr(self, "source"):
                self.source.discrete = None
            self.target.discrete = None
        elif m == "discrete":
            if hasattr(self, "source"):
                self.source.continuous = None
            self.target.continuous = None
        elif m != "multi-modal":
            raise ValueError("Invalid data modality. Specify 'continuous', 'discrete', or 'multi-modal'.")

    def _idx_data_split(self, total_size, ratios):
        assert np.abs(1.0 - sum(ratios)) < 1e-3, "Split ratios not sum to 1!"
        train_size = int(total_size * ratios[0])
        valid_size = int(total_size * ratios[1])
        return ((0, train_size) if train_size > 0 else None,
                (train_size, train_size + valid_size) if valid_size > 0 else None)
