"""AspenOpenJets reader + JetDataModule facade.

tests/golden/aoj.npz holds the outputs of the reference's own AspenOpenJets methods (_filter_particles,
_compute_continuous_coordinates, _map_pid_to_tokens, _preprocess, _extract_metadata; datamodules/aoj.py) on a
synthetic PFCands block (tools/make_golden.py: golden_aoj).  CPU: the numpy restatement (oracle/aoj.py) against it,
and the token <-> (flavour one-hot, charge) table against the known-answer vector of the reference's own test
(tests/test_datamodules_multimodal.py:93-123).  GPU: mmb_aoj_features and the device pre-processing against it
(tokens and masks bit-exact, kinematics rtol 1e-5), then JetDataModule end to end from an .npz file.
Parity unpinned: the HDF5 read (h5py is absent; the reference's 2016H_job0_mini.h5 cannot be opened here).
"""
import copy
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_npz, GOLDEN
from oracle import aoj as OA
from oracle import observables as OO


def close(a, b, rtol=1e-5, atol=1e-6):
    torch.testing.assert_close(torch.as_tensor(a).detach().cpu().float(), torch.as_tensor(b).detach().cpu().float(),
                               rtol=rtol, atol=atol)


def test_oracle_aoj_features():
    z = load_npz("aoj")
    cont, tok, mask = OA.read_block(z["pfcands"])
    for i, k in enumerate(("pt", "eta_rel", "phi_rel")):
        assert np.array_equal(cont[:, :, i:i + 1].numpy(), z["feat:" + k])          # same numpy ops: bit-equal
    assert np.array_equal(tok.numpy().astype(np.int64), z["tokens"])
    assert np.array_equal(mask.squeeze(-1).numpy(), z["mask"])
    # the block exercises every branch: filtered pids, unknown pids, pt == 0, padding
    raw_pid = z["pfcands"][:, :, -2]
    assert (np.abs(raw_pid[raw_pid != 0]) < 11).any() and (raw_pid == 310).any() and (z["mask"] == 0).any()


@pytest.mark.parametrize("tr", ["none", "standardize", "normalize", "log_pt"])
def test_oracle_aoj_preprocess(tr):
    z = load_npz("aoj")
    cont, tok, mask = OA.read_block(z["pfcands"])
    c, d, m, md = OA.preprocess(cont, tok, mask, None if tr == "none" else tr, torch.from_numpy(z[f"pre:{tr}:perm"]))
    ref = json.loads(str(z[f"pre:{tr}:metadata_json"]))
    assert md["num_particles_sample"] == ref["num_particles_sample"]
    close(torch.tensor(md["categorical_probs"]), torch.tensor(ref["categorical_probs"]), rtol=1e-6, atol=1e-9)
    for k in ("mean", "std", "min", "max"):
        close(torch.tensor(md[k]), torch.tensor(ref[k]), rtol=1e-6, atol=1e-7)
    close(c, z[f"pre:{tr}:continuous"], rtol=1e-6, atol=1e-7)
    assert np.array_equal(d.numpy(), z[f"pre:{tr}:discrete"]) and np.array_equal(m.numpy(), z[f"pre:{tr}:mask"])


def test_token_basis_known_answer():
    """The reference's own known-answer vector (tests/test_datamodules_multimodal.py:93-123): tokens 0..7 <->
    (photon, neutral hadron, charged hadron, electron, muon | charge).  Pins the charge column the jet-charge
    observables use (oracle/observables.py CHARGE, and the device kernel's table through it)."""
    basis = torch.tensor([[[1, 0, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0]], [[0, 0, 1, 0, 0, -1], [0, 0, 1, 0, 0, 1]],
                          [[0, 0, 0, 1, 0, -1], [0, 0, 0, 1, 0, 1]], [[0, 0, 0, 0, 1, -1], [0, 0, 0, 0, 1, 1]]]).long()
    tokens = torch.tensor([[[0], [1]], [[2], [3]], [[4], [5]], [[6], [7]]]).long()
    assert torch.equal(OO.CHARGE[tokens.squeeze(-1)].long(), basis[..., -1])
    # ... and the pid map of the reader lands on the same tokens: (pid -> token) o (token -> charge) = sign of the pid's charge
    pid_charge = {22: 0, 130: 0, -211: -1, 211: 1, -11: 1, 11: -1, -13: 1, 13: -1}       # PDG: 11 = e-, 13 = mu-
    for pid, tok in OA.PID_MAP.items():
        q = int(OO.CHARGE[tok])
        if abs(pid) == 211:
            assert q == pid_charge[pid]
        assert (q == 0) == (pid in (22, 130))


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_aoj_features_and_preprocess():
    from mmb_hip import ops
    from datamodules.aoj import AspenOpenJets
    z = load_npz("aoj")
    feats, tokens, mask = ops.aoj_features(torch.from_numpy(z["pfcands"]).cuda())
    assert np.array_equal(tokens.squeeze(-1).cpu().numpy(), z["tokens"].squeeze(-1))
    assert np.array_equal(mask.squeeze(-1).cpu().numpy(), z["mask"])
    for i, k in enumerate(("pt", "eta_rel", "phi_rel")):
        close(feats[:, :, i:i + 1], z["feat:" + k], rtol=1e-5, atol=2e-6)
    aoj = AspenOpenJets(data_dir="/nonexistent")
    for tr in ("none", "standardize", "normalize", "log_pt"):
        c, d, m, md = aoj._preprocess(feats.clone(), tokens.float(), mask, None if tr == "none" else tr,
                                      perm=torch.from_numpy(z[f"pre:{tr}:perm"]))
        ref = json.loads(str(z[f"pre:{tr}:metadata_json"]))
        assert md["num_particles_sample"] == ref["num_particles_sample"] and md["num_jets_sample"] == ref["num_jets_sample"]
        close(torch.tensor(md["categorical_probs"]), torch.tensor(ref["categorical_probs"]), rtol=1e-6, atol=1e-9)
        for k in ("mean", "std", "min", "max"):
            close(torch.tensor(md[k]), torch.tensor(ref[k]), rtol=2e-5, atol=2e-6)
        close(c, z[f"pre:{tr}:continuous"], rtol=1e-4, atol=2e-5)
        assert np.array_equal(d.cpu().numpy(), z[f"pre:{tr}:discrete"]) and np.array_equal(m.cpu().numpy(), z[f"pre:{tr}:mask"])


@pytest.mark.gpu
def test_jet_datamodule_from_npz(tmp_path, golden):
    """JetDataModule(config) as the reference's tests drive it (test_datamodules_multimodal.py:30-90), on an .npz
    copy of a PFCands block; then one training step of the model on its first batch."""
    from pipeline.configs import ExperimentConfigs
    from datamodules.jetmodule import JetDataModule
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    from tensorclass import TensorMultiModal
    z = load_npz("aoj")
    pf = np.concatenate([z["pfcands"]] * 3, 0)                          # 72 jets, 30 candidates
    np.savez(tmp_path / "block.npz", PFCands=pf)
    with open(os.path.join(GOLDEN, "metadata.json")) as fh:
        (tmp_path / "metadata.json").write_text(fh.read())
    cfg = copy.deepcopy(golden("mm_mid").cfg)
    cfg["data"].update(target_name="AspenOpenJets", target_path=str(tmp_path), target_train_files=["block.npz"],
                       target_test_files=["block.npz"], num_jets=64, max_num_particles=16, batch_size=16,
                       split_ratios=[0.75, 0.25], transform="standardize")
    config = ExperimentConfigs(cfg)
    jets = JetDataModule(config)
    jets.setup()
    N, M = 64, 16
    assert jets.target.continuous.shape == (N, M, 3) and jets.target.discrete.shape == (N, M, 1)
    assert jets.target.mask.shape == (N, M, 1) and jets.target.discrete.dtype == torch.int64
    assert set(jets.metadata["target"]) >= {"mean", "std", "min", "max", "categorical_probs"}
    loader = jets.train_dataloader()
    assert len(loader) == N * 0.75 // 16 and len(jets.val_dataloader()) == 1
    for batch in loader:
        assert isinstance(batch.source, TensorMultiModal) and not batch.has_source and batch.source.continuous is None
        assert batch.target.ndim == 3 and batch.target.continuous.shape == (16, M, 3) and batch.target.mask.shape[2] == 1
    jets.setup(stage="predict")
    assert len(jets.predict_dataloader()) == N // 16
    # standardised: masked mean 0 / std 1 per feature
    sel = jets.target.continuous  # predict target is NOT transformed upstream: compare the fit one instead
    jets.setup(stage="fit")
    x, m = jets.target.continuous, jets.target.mask.squeeze(-1) > 0
    close(x[m].mean(0), torch.zeros(3), rtol=0, atol=1e-4)
    close(x[m].std(0), torch.ones(3), rtol=1e-4, atol=0)
    model = MultiModalBridgeMatching(config).cuda().train()
    out = model.training_step(jets.train_dataloader()[0], 0)
    out["loss"].backward()
    assert torch.isfinite(out["loss"]).item()
    # a missing file is an error that names the file (aoj.py:69-72)
    cfg["data"]["target_train_files"] = ["absent.h5"]
    with pytest.raises(FileNotFoundError):
        JetDataModule(ExperimentConfigs(cfg)).setup()
