"""GPU: end-to-end ExperimentPipeline train -> checkpoint -> resume -> generate on synthetic
AOJ-shaped jets (the reference's tests/test_multimode_experiment.py flow), loss must go down."""
import copy
import os

import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_train_resume_generate(golden, tmp_path):
    from pipeline.experiment import ExperimentPipeline
    from datamodules.synthetic import SyntheticJetDataModule
    cfg = copy.deepcopy(golden("mm_mid").cfg)
    cfg["data"].update(target_path=GOLDEN, num_jets=512, batch_size=64, split_ratios=[0.75, 0.25], max_num_particles=16)
    cfg["trainer"].update(max_epochs=6, optimizer_params={"lr": 2e-3, "weight_decay": 5e-5}, gradient_clip_val=1.0,
                          scheduler_name="CosineAnnealingLR", scheduler_params={"T_max": 6, "eta_min": 1e-5})
    cfg["encoder"]["dropout"] = 0.1
    cfg["checkpoints"] = {"dirpath": str(tmp_path / "run")}
    torch.manual_seed(0)
    pipe = ExperimentPipeline(SyntheticJetDataModule, config=cfg)
    hist = pipe.train()
    assert len(hist) == 6 and all(torch.isfinite(torch.tensor(h["train_loss"])) for h in hist)
    assert hist[-1]["train_loss"] < hist[0]["train_loss"]
    assert os.path.exists(tmp_path / "run" / "checkpoints" / "last.ckpt")
    assert os.path.exists(tmp_path / "run" / "checkpoints" / "best.ckpt")
    # resume for two more epochs
    pipe2 = ExperimentPipeline(SyntheticJetDataModule, experiment_path=str(tmp_path / "run"), load_ckpt="last",
                               trainer={"max_epochs": 8})
    hist2 = pipe2.train()
    assert [h["epoch"] for h in hist2] == [6, 7]
    gen = pipe2.generate()
    n_test = 128
    assert gen.continuous.shape == (n_test, 16, 3) and gen.discrete.shape == (n_test, 16, 1)
    assert int(gen.discrete.min()) >= 0 and int(gen.discrete.max()) < 8
    assert torch.isfinite(gen.continuous).all()


def test_evaluate_report_on_device():
    """pipeline.evaluate: device observables + W1 of two synthetic post-processed samples; identical samples -> 0."""
    import torch
    from tensorclass import TensorMultiModal
    from pipeline.evaluate import wasserstein_report
    g = torch.Generator().manual_seed(1)
    def sample(shift):
        mask = (torch.rand(300, 64, 1, generator=g) < 0.5).long()
        x = (torch.randn(300, 64, 3, generator=g) * 0.3 + shift).abs() * mask
        k = torch.randint(0, 8, (300, 64, 1), generator=g) * mask
        return TensorMultiModal(None, x.cuda(), k.cuda(), mask.cuda())
    a, b = sample(0.0), sample(0.2)
    rep = wasserstein_report(a, b)
    assert set(rep) >= {"pt", "m", "numParticles", "numPhotons", "charge", "jet_charge"}
    assert all(v >= 0.0 for v in rep.values()) and rep["pt"] > 0.0
    same = wasserstein_report(a, a, features=["pt", "numParticles"])
    assert same == {"pt": 0.0, "numParticles": 0.0}
