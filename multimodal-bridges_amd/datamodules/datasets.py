"""DataCoupling: (source, target, context) triple of TensorMultiModal states -- the batch
type every hot-path entry point takes (reference: datamodules/datasets.py:8-41).
`has_*` are truthiness tests that bottom out in TensorMultiModal.__len__, so an all-None
TensorMultiModal() means "absent" and the model samples the source itself."""
from dataclasses import dataclass
from typing import Optional

import torch

from tensorclass import TensorMultiModal


@dataclass
class DataCoupling:
    source: Optional[TensorMultiModal] = None
    target: Optional[TensorMultiModal] = None
    context: Optional[TensorMultiModal] = None

    def __len__(self):
        return len(self.target)

    @property
    def ndim(self):
        return self.target.ndim

    @property
    def shape(self):
        return self.target.shape

    @property
    def has_source(self):
        return bool(self.source)

    @property
    def has_target(self):
        return bool(self.target)

    @property
    def has_context(self):
        return bool(self.context)

    def to(self, device):
        mv = lambda s: s.to(device) if isinstance(s, TensorMultiModal) else s
        return DataCoupling(mv(self.source), mv(self.target), mv(self.context))


def data_coupling_collate_fn(batch) -> DataCoupling:
    """Stack per-jet namedtuples (fields source_*/target_*/context_*) into a DataCoupling
    (reference: datasets.py:101-148)."""
    def stacked(field):
        return torch.stack([getattr(b, field) for b in batch]) if hasattr(batch[0], field) else None

    source = TensorMultiModal(None, stacked("source_continuous"), stacked("source_discrete"), stacked("source_mask"))
    target = TensorMultiModal(None, stacked("target_continuous"), stacked("target_discrete"), stacked("target_mask"))
    context = TensorMultiModal(None, stacked("context_continuous"), stacked("context_discrete"), None)
    return DataCoupling(source, target, context)
