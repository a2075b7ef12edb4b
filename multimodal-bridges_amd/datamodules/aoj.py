"""AspenOpenJets reader on device (reference: datamodules/aoj.py:15-275).

`AspenOpenJets(data_dir, data_files)(num_jets, max_num_particles, download, transform, features)
 -> (TensorMultiModal(None, continuous, discrete, mask), metadata)`: the reference's constructor and call.

The raw PFCands block goes to the GPU as it is read and ONE kernel (mmb_aoj_features) does what the reference does in
numpy on the host: the pid filter, (pt, eta_rel, phi_rel) relative to the jet axis, the 8-token flavour x charge map
and the mask.  The statistics of `_extract_metadata`, the standardize / normalize / log_pt transforms and the particle
shuffle of `_preprocess` run once per dataset as device tensor ops.

Files: `.h5` through h5py when it is importable (dataset "PFCands", as upstream); `.npz` / `.npy` holding the same
array otherwise (h5py is not installed in this image).  There is no network here: a missing file raises
FileNotFoundError naming the URL the reference would download it from; `download=True` does not fetch.
Supported features: continuous any of pt / eta_rel / phi_rel, discrete "tokens" (the hot path's configs)."""
import json
import os

import numpy as np
import torch

from tensorclass import TensorMultiModal
from mmb_hip import ops

_COLS = {"pt": 0, "eta_rel": 1, "phi_rel": 2}


class AspenOpenJets:
    """data constructor for the Aspen OpenJets dataset."""

    def __init__(self, data_dir, data_files=None, url="https://www.fdr.uni-hamburg.de/record/16505/files"):
        self.url = url
        self.data_dir = data_dir
        self.data_files = data_files

    def __call__(self, num_jets=None, max_num_particles=150, download=False, transform=None,
                 features={"continuous": ["pt", "eta_rel", "phi_rel"], "discrete": "tokens"}, device="cuda"):
        if isinstance(self.data_files, str):
            self.data_files = [self.data_files]
        if not self.data_files:
            raise FileNotFoundError(f"AspenOpenJets: no data files given for {self.data_dir}")
        cont_names = features.get("continuous")
        if features.get("discrete") not in (None, "tokens"):
            raise NotImplementedError("AspenOpenJets on device serves discrete_features 'tokens' (or none)")
        for n in cont_names or []:
            if n not in _COLS:
                raise NotImplementedError(f"AspenOpenJets on device serves pt / eta_rel / phi_rel, not '{n}'")
        conts, discs, masks = [], [], []
        jet_count = 0
        for datafile in self.data_files:
            path = os.path.join(self.data_dir, datafile)
            if not os.path.isfile(path):
                raise FileNotFoundError(f"File {datafile} not found in {self.data_dir}."
                                        + (f" (no network here: fetch {self.url}/{os.path.basename(path)} by hand)"
                                           if download else ""))
            feats, tokens, mask = self._read_aoj_file(path, num_jets, device)
            if cont_names:
                conts.append(feats[:, :, [_COLS[n] for n in cont_names]])
            if features.get("discrete") == "tokens":
                discs.append(tokens.float())                      # float32, as upstream (aoj.py:162-163)
            masks.append(mask)
            if num_jets:                                          # halt if enough jets (aoj.py:86-90)
                jet_count += len(masks[-1])
                if jet_count > num_jets:
                    break
        cut = lambda parts: torch.cat(parts, dim=0)[:num_jets, :max_num_particles, :].contiguous() if parts else None
        continuous, discrete, mask = cut(conts), cut(discs), cut(masks)
        continuous, discrete, mask, metadata = self._preprocess(continuous, discrete, mask, transform)
        output = TensorMultiModal(None, continuous, discrete, mask)
        output.apply_mask()
        return output, metadata

    # ------------------------------------------------------------------ file -> device
    @staticmethod
    def _load_pfcands(filepath, num_jets=None):
        ext = os.path.splitext(filepath)[1].lower()
        if ext in (".h5", ".hdf5"):
            try:
                import h5py
            except ImportError as e:
                raise ValueError(f"Error reading file {filepath}: h5py is not installed (convert the PFCands dataset to "
                                 f".npz, key 'PFCands')") from e
            try:
                with h5py.File(filepath, "r") as f:
                    return f["PFCands"][:num_jets] if num_jets else f["PFCands"][:]
            except (OSError, KeyError) as e:
                raise ValueError(f"Error reading file {filepath}: {e}")
        try:
            if ext == ".npy":
                arr = np.load(filepath, mmap_mode="r", allow_pickle=False)
            else:
                arr = np.load(filepath, allow_pickle=False)["PFCands"]
        except (OSError, KeyError, ValueError) as e:
            raise ValueError(f"Error reading file {filepath}: {e}")
        return np.asarray(arr[:num_jets] if num_jets else arr)

    def _read_aoj_file(self, filepath, num_jets=None, device="cuda"):
        """-> feats (J,M,3) = (pt, eta_rel, phi_rel), tokens (J,M,1) i64, mask (J,M,1) i64, all on `device`."""
        pf = torch.from_numpy(np.ascontiguousarray(self._load_pfcands(filepath, num_jets), dtype=np.float32))
        return ops.aoj_features(pf.to(device))

    # ------------------------------------------------------------------ aoj.py:115-145, :254-275
    def _preprocess(self, continuous, discrete, mask, transform, perm=None):
        metadata = self._extract_metadata(continuous, mask)
        dev = mask.device
        if continuous is not None:
            if transform == "standardize":
                continuous = (continuous - torch.tensor(metadata["mean"], device=dev)) / torch.tensor(metadata["std"], device=dev)
            if transform == "normalize":
                lo, hi = torch.tensor(metadata["min"], device=dev), torch.tensor(metadata["max"], device=dev)
                continuous = (continuous - lo) / (hi - lo)
            if transform == "log_pt":
                continuous = continuous.clone()
                continuous[:, :, 0] = torch.log(continuous[:, :, 0] + 1e-6)
                metadata = self._extract_metadata(continuous, mask)
                continuous = (continuous - torch.tensor(metadata["mean"], device=dev)) / torch.tensor(metadata["std"], device=dev)
        if perm is None:                                          # shuffle particles within jets (:137)
            perm = torch.randperm(mask.shape[1])
        idx = perm.to(dev)
        if continuous is not None:
            continuous = continuous[:, idx, :]
        if discrete is not None:
            discrete = discrete[:, idx, :]
        return continuous, discrete, mask[:, idx, :], metadata

    def _extract_metadata(self, continuous, mask):
        mask_bool = mask.squeeze(-1) > 0
        nums = mask.sum(dim=1).reshape(-1)
        M = mask.shape[1]
        hist = torch.bincount(nums, minlength=M + 1)[:M + 1].double()
        hist = hist / hist.sum()                                  # np.histogram(..., bins of width 1, density=True)
        metadata = {"num_jets_sample": mask.shape[0], "num_particles_sample": int(nums.sum().item()),
                    "max_num_particles_per_jet": M}
        if continuous is not None:
            sel = continuous[mask_bool]
            metadata["mean"] = sel.mean(0).tolist()
            metadata["std"] = sel.std(0).tolist()
            metadata["min"] = sel.min(0).values.tolist()
            metadata["max"] = sel.max(0).values.tolist()
        metadata["categorical_probs"] = hist.tolist()
        return metadata

    def _load_metadata(self, path):
        with open(os.path.join(path, "metadata.json"), "r") as f:
            return json.load(f)
