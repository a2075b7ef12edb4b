"""CPU restatement of the reference's jet-level observables (datamodules/utils.py:21-42 ParticleClouds kinematics,
:437-475 JetFeatures, :548-556 jet charge, :558-574 flavour counts).  TEST INFRASTRUCTURE ONLY.
Parity unpinned: the reference module imports fastjet/matplotlib (absent here) and its tests hold no vector for
these quantities; this follows the source text."""
import torch

CHARGE = torch.tensor([0., 0., -1., 1., -1., 1., -1., 1.])


def jet_observables(x, k, mask):
    """x (B,N,3) | None, k (B,N,1) | None, mask (B,N,1) -> dict of (B,) tensors"""
    out = {"numParticles": mask.sum(1).squeeze(-1).float()}
    if x is not None:
        pt, eta, phi = x[..., 0], x[..., 1], x[..., 2]                      # :36-42
        px, py, pz, e = (pt * torch.cos(phi)).sum(-1), (pt * torch.sin(phi)).sum(-1), (pt * torch.sinh(eta)).sum(-1), \
            (pt * torch.cosh(eta)).sum(-1)
        jpt = torch.sqrt(px ** 2 + py ** 2)                                 # :447-454
        out.update(px=px, py=py, pz=pz, e=e, pt=jpt, m=torch.sqrt(e ** 2 - jpt ** 2 - pz ** 2),
                   eta=0.5 * torch.log((jpt + pz) / (jpt - pz)), phi=torch.atan2(py, px))
    if k is not None:
        tok = k.squeeze(-1)
        mb = mask.squeeze(-1) > 0
        B = tok.shape[0]
        flat = (torch.arange(B).unsqueeze(1) * 9 + tok * mb)[mb]            # :558-574
        cnt = torch.bincount(flat, minlength=B * 9).view(B, 9).float()
        names = ("numPhotons", "numNeutralHadrons", "numNegativeHadrons", "numPositiveHadrons", "numElectrons",
                 "numPositrons", "numMuons", "numAntiMuons")
        for i, n in enumerate(names):
            out[n] = cnt[:, i]
        q = CHARGE[tok]                                                     # map_tokens_to_basis(...)[..., -1]
        out["charge"] = q.sum(1)                                            # kappa = 0
        if x is not None:
            out["jet_charge"] = (q * x[..., 0]).sum(1) / out["pt"]          # kappa = 1
    return out
