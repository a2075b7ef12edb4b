"""Oracle: SourceUniform (test infrastructure; see oracle/__init__.py).

Follows distributions/noise.py:204-249 (the second SourceUniform definition wins).
Draws come from torch's CPU generator exactly as in the reference, so the result is
distribution-equivalent to -- not bit-equal with -- the on-device Philox sampler.
"""
import torch


def source_uniform(shape, modality, vocab_size, dim_continuous, categorical_probs=None,
                   sample_masks=False, generator=None):
    B, N = shape
    if sample_masks:
        probs = torch.tensor(categorical_probs)
        mult = torch.multinomial(probs / probs.sum(), B, True, generator=generator)
        mask = torch.zeros((B, N))
        for i, n in enumerate(mult):                      # noise.py:221-224 (python loop!)
            mask[i, :n] = 1
            idx = torch.randperm(N, generator=generator)
            mask[i] = mask[i][idx]
        mask = mask.long().unsqueeze(-1)
    else:
        mask = torch.ones((B, N, 1)).long()
    x = k = None
    if modality in ("continuous", "multi-modal"):
        x = torch.randn((B, N, dim_continuous), generator=generator)
    if modality in ("discrete", "multi-modal"):
        k = torch.randint(0, vocab_size, (B, N, 1), generator=generator)
    return {"continuous": x, "discrete": k, "mask": mask}
