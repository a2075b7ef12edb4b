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


@pytest.mark.parametrize("name", ["mm_mid", "uni_mid"])
def test_flat_adamw_matches_torch_adamw_with_clipping(golden, name):
    """FlatAdamW (mmb_adamw_clip_step: clip + AdamW in two launches on flat buffers) against
    torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW over several steps with a changing learning rate; the
    parameters whose gradient is None upstream (Q1 embedding, K2 dead layer) must not move -- torch skips them."""
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    from pipeline.optim import FlatAdamW, unused_parameter_names
    g = golden(name)
    cfg = copy.deepcopy(g.cfg)
    cfg["data"]["target_path"] = GOLDEN
    a = MultiModalBridgeMatching(ExperimentConfigs(copy.deepcopy(cfg)))
    a.load_state_dict(g.params)
    b = MultiModalBridgeMatching(ExperimentConfigs(copy.deepcopy(cfg)))
    b.load_state_dict(g.params)
    a, b = a.cuda(), b.cuda()
    unused = unused_parameter_names(a)
    assert any(n.startswith("embedder.discrete_embedding") for n in unused)
    if name == "mm_mid":
        assert any("fused_encoder.epic_layers.2." in n for n in unused)
    hp = dict(lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    fa = FlatAdamW(a.named_parameters(), unused=unused, **hp)
    ta = torch.optim.AdamW([p for p in b.parameters() if p.requires_grad], **hp)
    before = {n: p.detach().clone() for n, p in a.named_parameters()}
    gen = torch.Generator().manual_seed(5)
    for it in range(6):
        lr = 3e-3 * (1.0 - 0.1 * it)
        fa.param_groups[0]["lr"] = ta.param_groups[0]["lr"] = lr
        fa.zero()
        for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
            if not pa.requires_grad:
                continue
            if n in unused:
                pb.grad = None                                 # as upstream: never touched by backward
                continue
            gr = (torch.randn(pa.shape, generator=gen) * (3.0 if it % 2 else 0.01)).cuda()     # clipped / not clipped
            pa.grad.copy_(gr)
            pb.grad = gr.clone()
        norm = torch.nn.utils.clip_grad_norm_([p for p in b.parameters() if p.grad is not None], 1.0)
        ta.step()
        fa.step(max_norm=1.0)
        torch.testing.assert_close(fa.last_norm[0].cpu(), norm.cpu(), rtol=1e-5, atol=1e-7)
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        torch.testing.assert_close(pa.detach().cpu(), pb.detach().cpu(), rtol=2e-5, atol=2e-7, msg=lambda m: f"{n}: {m}")
        if n in unused:
            assert torch.equal(pa.detach().cpu(), before[n].cpu())
    # the weight caches see the raw-pointer update
    assert all(p._version > 0 for p in a.parameters() if p.requires_grad)
    sd = fa.state_dict()
    fa2 = FlatAdamW(a.named_parameters(), unused=unused, **hp)
    fa2.load_state_dict(sd)
    assert fa2.t == 6 and torch.equal(fa2.exp_avg, fa.exp_avg)
    # a torch.optim.AdamW state_dict (reference checkpoints, earlier checkpoints of this package) loads into the flat
    # buffers: same moments, same step count, and the next step of both optimizers agrees
    fa3 = FlatAdamW(a.named_parameters(), unused=unused, foreach=None, fused=None, **hp)
    fa3.load_state_dict(ta.state_dict())
    assert fa3.t == 6
    torch.testing.assert_close(fa3.exp_avg.cpu(), fa.exp_avg.cpu(), rtol=2e-5, atol=1e-9)
    torch.testing.assert_close(fa3.exp_avg_sq.cpu(), fa.exp_avg_sq.cpu(), rtol=2e-5, atol=1e-12)
    with pytest.raises(ValueError):
        FlatAdamW(a.named_parameters(), unused=unused, maximize=True, **hp)
    with pytest.raises(ValueError):
        fa3.load_state_dict({"something": "else"})
