"""CPU restatement of GeneratorCallback._postprocess (pipeline/callbacks_generator.py:99-128) followed by
TensorMultiModal.apply_mask (tensorclass.py:96-106).  TEST INFRASTRUCTURE ONLY.
Parity unpinned: the reference module imports Lightning/Comet and cannot be imported here, and its tests hold no
vector for it; this follows the source text line by line."""
import torch


def postprocess(continuous, discrete, mask, transform=None, target_stats=None):
    x = None if continuous is None else continuous.clone()
    if x is not None and transform == "standardize":                      # :102-105
        x = x * torch.tensor(target_stats["std"]) + torch.tensor(target_stats["mean"])
    elif x is not None and transform == "normalize":                      # :107-110
        lo, hi = torch.tensor(target_stats["min"]), torch.tensor(target_stats["max"])
        x = x * (hi - lo) + lo
    if x is not None and transform == "log_pt":                           # :112-116
        x = x * torch.tensor(target_stats["std"]) + torch.tensor(target_stats["mean"])
        x[:, :, 0] = torch.exp(x[:, :, 0]) - 1e-6
    if x is not None:                                                     # apply_mask
        x = x * mask
    k = None if discrete is None else (discrete * mask).long()
    return x, k
