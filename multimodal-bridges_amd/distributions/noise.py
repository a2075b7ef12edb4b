"""SourceUniform on device (reference: distributions/noise.py:204-249).

`SourceUniform(config)(shape=(B,N), device, sample_masks=False) -> TensorMultiModal`:
x0 ~ N(0,1), k0 ~ U{0..V-1}, mask all ones or -- for generation -- a multiplicity drawn from
metadata["categorical_probs"] with the ones in uniformly random slots.  The reference draws on
the CPU with a Python loop of B randperm calls and copies to the device; here one Philox
kernel (mmb_source_uniform) writes everything in place on the GPU.  Draws are
distribution-equivalent, not bit-equal, to torch's generator.
"""
import json
import os

import torch

from tensorclass import TensorMultiModal
from mmb_hip import ops


class SourceUniform:
    def __init__(self, config):
        self.config = config
        self.vocab_size = config.data.vocab_size
        self.dim_continuous = config.data.dim_continuous
        self.metadata = self._load_metadata(config.data.target_path)
        self.seed = int(getattr(config, "seed", 0xA0A0))
        self._calls = 0
        self._pmf = {}

    def _load_metadata(self, path):
        with open(os.path.join(path, "metadata.json"), "r") as fh:
            return json.load(fh)

    def manual_seed(self, seed, offset=0):
        self.seed, self._calls = int(seed), int(offset)

    def _device_pmf(self, device):
        key = str(device)
        if key not in self._pmf:
            self._pmf[key] = torch.tensor(self.metadata["categorical_probs"], dtype=torch.float32, device=device)
        return self._pmf[key]

    def __call__(self, shape, device, sample_masks=False) -> TensorMultiModal:
        num_jets, max_num_particles = shape
        modality = self.config.data.modality
        self._calls += 1
        pmf = self._device_pmf(device) if sample_masks else None
        x0, k0, mask, maskf, counts = ops.source_uniform(
            num_jets, max_num_particles, max(self.dim_continuous, 1), max(self.vocab_size, 1), device, pmf=pmf,
            sample_masks=sample_masks, seed=self.seed, offset=(1 << 40) + self._calls,
            want_x=modality != "discrete", want_k=modality != "continuous")
        out = TensorMultiModal(None, x0, k0, mask)
        out._mask_info = ops.MaskInfo(maskf, counts, None)      # reused by predict_step (no second pass)
        return out
