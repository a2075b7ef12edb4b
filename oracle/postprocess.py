"""CPU restatement of GeneratorCallback._postprocess (pipeline/callbacks_generator.py:99-128) followed by
TensorMultiModal.apply_mask (tensorclass.py:96-106).  TEST INFRASTRUCTURE ONLY.
Pinned by tests/golden/postprocess.npz: outputs of the reference's own method (tools/make_golden.py:
golden_postprocess) for the four transforms, (J,N,D) samples and stacked (T,J,N,D) paths, tokens and onehot."""
import torch


def postprocess(continuous, discrete, mask, transform=None, target_stats=None, discrete_features="tokens",
                vocab_size=None):
    x = None if continuous is None else continuous.clone()
    if x is not None and transform == "standardize":                      # :102-105
        x = x * torch.tensor(target_stats["std"]) + torch.tensor(target_stats["mean"])
    elif x is not None and transform == "normalize":                      # :107-110
        lo, hi = torch.tensor(target_stats["min"]), torch.tensor(target_stats["max"])
        x = x * (hi - lo) + lo
    if x is not None and transform == "log_pt":                           # :112-116
        x = x * torch.tensor(target_stats["std"]) + torch.tensor(target_stats["mean"])
        x[:, :, 0] = torch.exp(x[:, :, 0]) - 1e-6                         # feature 0 on 3-D input, slot 0 on 4-D paths
    k = discrete
    if discrete_features == "onehot":                                     # :118-125
        k = torch.argmax(x[..., -vocab_size:], dim=-1).unsqueeze(-1)
        x = x[..., :-vocab_size]
        if x.shape[-1] == 0:
            x = None
    if x is not None:                                                     # apply_mask
        x = x * mask
    k = None if k is None else (k * mask).long()
    return x, k
