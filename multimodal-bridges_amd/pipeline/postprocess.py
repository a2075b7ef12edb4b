"""Generation post-processing on the device (reference: GeneratorCallback._postprocess,
pipeline/callbacks_generator.py:99-128): inverse of the data transform chosen by `config.data.transform`
("standardize" | "normalize" | "log_pt" | None) with the statistics of `metadata["target"]`, then
`TensorMultiModal.apply_mask`.  The "onehot" discrete representation of the reference (continuous
channels carrying the tokens) is outside the hot path and not supported."""
import torch

from mmb_hip import ops
from tensorclass import TensorMultiModal


def postprocess(paths: TensorMultiModal, transform=None, metadata=None, discrete_features="tokens") -> TensorMultiModal:
    if discrete_features == "onehot":
        raise NotImplementedError("onehot discrete features are not part of the HIP path (tokens only)")
    if transform not in (None, "standardize", "normalize", "log_pt"):
        raise ValueError(f"unknown transform {transform!r}")
    x, k, mask = paths.continuous, paths.discrete, paths.mask
    dev = mask.device
    scale = shift = None
    if transform is not None and x is not None:
        tgt = metadata["target"]
        if transform == "normalize":
            lo = torch.tensor(tgt["min"], dtype=torch.float32)
            hi = torch.tensor(tgt["max"], dtype=torch.float32)
            scale, shift = (hi - lo).to(dev), lo.to(dev)
        else:
            scale = torch.tensor(tgt["std"], dtype=torch.float32, device=dev)
            shift = torch.tensor(tgt["mean"], dtype=torch.float32, device=dev)
    maskf = mask.reshape(-1).float().contiguous()
    if x is not None:
        paths.continuous = x = x.contiguous()
    if k is not None:
        paths.discrete = k = k.contiguous()
    ops.postprocess_(x, k, maskf, scale, shift, log_pt=(transform == "log_pt"))
    if paths.time is not None:
        paths.time = paths.time * mask
    return paths
