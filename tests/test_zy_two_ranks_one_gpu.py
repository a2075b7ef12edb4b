"""Two rank processes on the ONE GPU of the box (gloo; RCCL refuses two ranks on one device): what an N-GPU run does that
a single process never exercises -- weight broadcast, per-rank data shards and Philox rank keys, the flat gradient
bucket averaged over ranks, FlatAdamW on every rank, the sharded generation and its single gather.  The children are
fresh interpreters started as child processes (tests/two_rank_worker.py under torch.distributed.run); nothing is
re-executed in this process.  Reference: pipeline/experiment.py:208-224 (Trainer(strategy="ddp")),
callbacks_generator.py:39-70 (per-rank results, rank-0 concatenation).

Collected late on purpose: the round-end run is `pytest -x`."""
import copy
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

GEN_CALLS = 1000


def _cfg(golden):
    cfg = copy.deepcopy(golden("mm_mid").cfg)
    cfg["data"].update(target_path=GOLDEN, num_jets=256, batch_size=32, split_ratios=[0.75, 0.25], max_num_particles=16)
    cfg["trainer"].update(max_epochs=1, optimizer_params={"lr": 2e-3, "weight_decay": 5e-5}, gradient_clip_val=1.0,
                          scheduler_name=None, scheduler_params=None)
    cfg["encoder"]["dropout"] = 0.1
    cfg["checkpoints"] = {"dirpath": None}
    return cfg


@pytest.fixture(scope="module")
def two_rank_run(golden, tmp_path_factory):
    out = tmp_path_factory.mktemp("two_ranks")
    cfg = _cfg(golden)
    cfg_path = os.path.join(out, "cfg.json")
    with open(cfg_path, "w") as fh:
        json.dump(cfg, fh)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "two_rank_worker.py"), str(out), cfg_path]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return cfg, out


def _single_process_model(cfg):
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    torch.manual_seed(1000)                                   # rank 0's initial weights: what the broadcast hands out
    config = ExperimentConfigs(copy.deepcopy(cfg))
    model = MultiModalBridgeMatching(config).cuda()
    model.manual_seed(0xA0A0, 0)
    return model, config


def test_world2_training_equals_one_process_on_both_shards(two_rank_run):
    """Three steps of ExperimentPipeline.train() at world 2 == one process that runs BOTH ranks' batches with their
    Philox rank keys, averages the two gradients and takes the same FlatAdamW steps (tolerance: the fp32 mean and the
    atomics of the weight-gradient kernels, then AdamW's normalisation)."""
    from pipeline.experiment import make_optimizer, optimizer_step
    from datamodules.synthetic import SyntheticJetDataModule
    cfg, out = two_rank_run
    got = [torch.load(os.path.join(out, f"train_rank{r}.pt"), weights_only=True) for r in (0, 1)]
    for n, w in got[0]["state_dict"].items():                 # ranks end bit-identical: same averaged gradient, same update
        assert torch.equal(w, got[1]["state_dict"][n]), n
    assert len(got[0]["history"]) == 1

    model, config = _single_process_model(cfg)
    model.train()
    opt, _, flat = make_optimizer(model, config)
    loaders = []
    for r in (0, 1):
        dm = SyntheticJetDataModule(config)
        dm.setup("fit", rank=r, world=2, device="cuda")
        loaders.append(dm.train_dataloader(epoch=0))
    n_steps = min(len(l) for l in loaders)
    assert n_steps >= 2
    acc = torch.zeros_like(flat.flat)
    for n in range(n_steps):
        acc.zero_()
        c0 = model._calls
        for r in (0, 1):
            flat.zero()
            model.rank, model._calls = r, c0
            model.training_step(loaders[r][n], n)["loss"].backward()
            acc += flat.flat
        flat.flat.copy_(acc / 2)
        optimizer_step(opt, flat, config.trainer.gradient_clip_val)
    lr, worst = 2e-3, 0.0
    moved = 0.0
    init, _ = _single_process_model(cfg)
    init_sd = init.state_dict()
    for name, w in model.state_dict().items():
        d = (w.detach().cpu() - got[0]["state_dict"][name]).abs().max().item()
        worst = max(worst, d)
        moved = max(moved, (w.detach().cpu() - init_sd[name].cpu()).abs().max().item())
    print(f"two ranks vs one process: max |dw| {worst:.3e}; the weights moved by up to {moved:.3e} (lr {lr})")
    assert moved > 2 * lr                                     # three AdamW steps did happen
    assert worst < 1e-5, worst                                # measured 2.1e-6 (atomics + fp32 mean, through AdamW)


def test_world2_generation_gather_equals_shards_generated_alone(two_rank_run):
    """generate() + gather_generated on rank 0 == the concatenation of the two shards, and each shard == what a single
    process generates with that rank's key, shard size and weights (bit-exact: no atomics in the generation path)."""
    from pipeline.experiment import ExperimentPipeline
    from datamodules.synthetic import SyntheticJetDataModule
    cfg, out = two_rank_run
    gens = [torch.load(os.path.join(out, f"gen_rank{r}.pt"), weights_only=True) for r in (0, 1)]
    for f in "xkm":
        assert torch.equal(gens[0]["full_" + f], torch.cat([gens[0]["shard_" + f], gens[1]["shard_" + f]])), f
    assert gens[0]["full_k"].dtype == torch.int64 and gens[0]["full_x"].shape[0] == 64
    assert not torch.equal(gens[0]["shard_x"], gens[1]["shard_x"])          # rank keys differ
    weights = torch.load(os.path.join(out, "train_rank0.pt"), weights_only=True)["state_dict"]
    pipe = ExperimentPipeline(SyntheticJetDataModule, config=cfg)
    pipe.model.load_state_dict(weights)
    for r in (0, 1):
        pipe.rank, pipe.world = r, 2                          # shard bookkeeping only: no process group here
        pipe.model.manual_seed(0xA0A0, r)
        pipe.model._calls = GEN_CALLS
        alone = pipe.generate(gather=False, postprocess=False)
        assert torch.equal(alone.continuous.cpu(), gens[r]["shard_x"]), r
        assert torch.equal(alone.discrete.cpu(), gens[r]["shard_k"]), r
        assert torch.equal(alone.mask.cpu(), gens[r]["shard_m"]), r
