"""Oracle: AspenOpenJets pre-processing (test infrastructure; see oracle/__init__.py).

numpy restatement of datamodules/aoj.py:115-145 (_preprocess), :179-245 (_filter_particles, _map_pid_to_tokens,
_compute_continuous_coordinates) and :254-275 (_extract_metadata), pinned by tests/golden/aoj.npz -- the outputs of
the reference's own AspenOpenJets methods on a synthetic PFCands block (tools/make_golden.py: golden_aoj).  What is
NOT pinned: the HDF5 read itself (aoj.py:150-154; h5py is absent here, and so no reader for the reference's
tests/resources/2016H_job0_mini.h5).
"""
import numpy as np
import torch

PID_MAP = {22: 0, 130: 1, -211: 2, 211: 3, -11: 4, 11: 5, -13: 6, 13: 7}


def filter_particles(pf):
    """aoj.py:179-184"""
    pf = pf.copy()
    bad = np.abs(pf[:, :, -2]) < 11
    pf[bad] = np.zeros_like(pf[bad])
    return pf


def map_pid_to_tokens(pid):
    """aoj.py:186-200"""
    return np.vectorize(lambda x: PID_MAP.get(x, 0))(pid)


def continuous_coordinates(pf):
    """aoj.py:202-245 -> ({pt, eta_rel, phi_rel} each (J,M,1), mask (J,M) bool)"""
    px, py, pz, e = pf[:, :, 0], pf[:, :, 1], pf[:, :, 2], pf[:, :, 3]
    pt = np.sqrt(px ** 2 + py ** 2)
    eta = np.arcsinh(np.divide(pz, pt, out=np.zeros_like(pz), where=pt != 0))
    phi = np.arctan2(py, px)
    jet = pf[:, :, :4].sum(axis=1)
    jet_eta = np.arcsinh(jet[:, 2] / np.sqrt(jet[:, 0] ** 2 + jet[:, 1] ** 2))
    jet_phi = np.arctan2(jet[:, 1], jet[:, 0])
    eta_rel = eta - jet_eta[:, None]
    phi_rel = (phi - jet_phi[:, None] + np.pi) % (2 * np.pi) - np.pi
    mask = pf[:, :, 3] > 0
    feats = {"pt": (pt * mask)[:, :, None], "eta_rel": (eta_rel * mask)[:, :, None],
             "phi_rel": (phi_rel * mask)[:, :, None]}
    return feats, mask


def read_block(pf):
    """aoj.py:156-165 on an in-memory PFCands block -> (continuous (J,M,3) f32, tokens (J,M,1) f32, mask (J,M,1) i64)"""
    pf = filter_particles(pf)
    feats, mask = continuous_coordinates(pf)
    tokens = map_pid_to_tokens(pf[:, :, -2])[:, :, None]
    cont = torch.cat([torch.tensor(feats[k], dtype=torch.float32) for k in ("pt", "eta_rel", "phi_rel")], dim=-1)
    return cont, torch.tensor(tokens, dtype=torch.float32), torch.tensor(mask[:, :, None], dtype=torch.long)


def extract_metadata(continuous, mask):
    """aoj.py:254-275"""
    mb = mask.squeeze(-1) > 0
    nums = mask.sum(dim=1).squeeze()
    hist, _ = np.histogram(nums, bins=np.arange(0, mask.shape[1] + 2, 1), density=True)
    md = {"num_jets_sample": mask.shape[0], "num_particles_sample": nums.sum().item(),
          "max_num_particles_per_jet": mask.shape[1]}
    if continuous is not None:
        md["mean"] = continuous[mb].mean(0).tolist()
        md["std"] = continuous[mb].std(0).tolist()
        md["min"] = continuous[mb].min(0).values.tolist()
        md["max"] = continuous[mb].max(0).values.tolist()
    md["categorical_probs"] = hist.tolist()
    return md


def preprocess(continuous, discrete, mask, transform, perm):
    """aoj.py:115-145 with the particle permutation (torch.randperm, :137) as an input."""
    md = extract_metadata(continuous, mask)
    if transform == "standardize":
        continuous = (continuous - torch.tensor(md["mean"])) / torch.tensor(md["std"])
    if transform == "normalize":
        lo, hi = torch.tensor(md["min"]), torch.tensor(md["max"])
        continuous = (continuous - lo) / (hi - lo)
    if transform == "log_pt":
        continuous = continuous.clone()
        continuous[:, :, 0] = torch.log(continuous[:, :, 0] + 1e-6)
        md = extract_metadata(continuous, mask)
        continuous = (continuous - torch.tensor(md["mean"])) / torch.tensor(md["std"])
    continuous = continuous[:, perm, :]
    discrete = discrete[:, perm, :] if discrete is not None else None
    return continuous, discrete, mask[:, perm, :], md
