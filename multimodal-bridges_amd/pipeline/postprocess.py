"""Generation post-processing on the device (reference: GeneratorCallback._postprocess,
pipeline/callbacks_generator.py:99-128): inverse of the data transform chosen by `config.data.transform`
("standardize" | "normalize" | "log_pt" | None) with the statistics of `metadata["target"]`, the "onehot" branch
(:118-125: the last `vocab_size` continuous channels become the tokens by argmax), then `TensorMultiModal.apply_mask`.

Upstream's `log_pt` writes `paths.continuous[:, :, 0] = exp(.) - 1e-6` (:116).  On a (J,N,Dc) sample (the test set,
:82) that is feature 0; on the stacked (T,J,N,Dc) paths the callback passes for the generated sample (:69) the same
expression selects slot 0 of every jet, all features.  Both are reproduced as written, by the rank of the input."""
import torch

from mmb_hip import ops
from tensorclass import TensorMultiModal


def postprocess(paths: TensorMultiModal, transform=None, metadata=None, discrete_features="tokens",
                vocab_size=None) -> TensorMultiModal:
    if transform not in (None, "standardize", "normalize", "log_pt"):
        raise ValueError(f"unknown transform {transform!r}")
    x, k, mask = paths.continuous, paths.discrete, paths.mask
    dev = mask.device
    scale = shift = None
    log_pt = 0
    if transform is not None and x is not None:
        tgt = metadata["target"]
        if transform == "normalize":
            lo = torch.tensor(tgt["min"], dtype=torch.float32)
            hi = torch.tensor(tgt["max"], dtype=torch.float32)
            scale, shift = (hi - lo).to(dev), lo.to(dev)
        else:
            scale = torch.tensor(tgt["std"], dtype=torch.float32, device=dev)
            shift = torch.tensor(tgt["mean"], dtype=torch.float32, device=dev)
        if scale.numel() != x.shape[-1]:
            raise ValueError(f"metadata['target'] statistics have {scale.numel()} entries, the sample has "
                             f"{x.shape[-1]} continuous channels")
        if transform == "log_pt":
            if x.ndim not in (3, 4):
                raise ValueError("log_pt post-processing needs a (J,N,D) sample or stacked (T,J,N,D) paths")
            log_pt = 1 if x.ndim == 3 else 2
    maskf = mask.reshape(-1).float().contiguous()
    if discrete_features == "onehot":
        if x is None or vocab_size is None or x.shape[-1] < vocab_size:
            raise ValueError("onehot post-processing needs continuous channels holding vocab_size token scores")
        paths.continuous, paths.discrete = ops.postprocess_onehot(x.contiguous(), int(vocab_size), maskf, scale, shift,
                                                                  log_pt)
    else:
        if x is not None:
            paths.continuous = x = x.contiguous()
        if k is not None:
            paths.discrete = k = k.contiguous()
        ops.postprocess_(x, k, maskf, scale, shift, log_pt=log_pt, N=mask.shape[-2])
    if paths.time is not None:
        paths.time = paths.time * mask
    return paths
