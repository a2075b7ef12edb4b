"""One rank of the two-ranks-on-ONE-GPU rehearsal (tests/test_zy_two_ranks_one_gpu.py starts two of these under
torch.distributed.run).  RCCL refuses two ranks on one device, so the process group is gloo and the collectives stage
through the host (pipeline/experiment.py: all_reduce_mean, all_reduce_, broadcast_, gather_generated) -- every kernel,
the flat gradient bucket, FlatAdamW, the Philox rank keys and the sharding are the ones an 8-GPU run uses.

    two_rank_worker.py <outdir> <cfg.json>
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-bridges_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

GEN_CALLS = 1000          # Philox call counter both generation passes start from


def main():
    outdir, cfg_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ["LOCAL_RANK"] = "0"                      # both ranks on the one card
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pipeline.experiment import ExperimentPipeline
    from datamodules.synthetic import SyntheticJetDataModule
    with open(cfg_path) as fh:
        cfg = json.load(fh)
    torch.manual_seed(1000 + rank)                      # DIFFERENT initial weights per rank: the broadcast must fix that
    pipe = ExperimentPipeline(SyntheticJetDataModule, config=cfg)
    hist = pipe.train()
    sd = {k: v.detach().cpu() for k, v in pipe.model.state_dict().items()}
    torch.save({"state_dict": sd, "history": hist, "calls": int(pipe.model._calls)}, os.path.join(outdir, f"train_rank{rank}.pt"))
    pipe.model.manual_seed(0xA0A0, rank)               # rewinds the source sampler's stream and the call counter
    pipe.model._calls = GEN_CALLS
    shard = pipe.generate(gather=False, postprocess=False)
    pipe.model.manual_seed(0xA0A0, rank)
    pipe.model._calls = GEN_CALLS
    full = pipe.generate(gather=True, postprocess=False)
    assert (full is None) == (rank != 0)
    out = {"shard_x": shard.continuous.cpu(), "shard_k": shard.discrete.cpu(), "shard_m": shard.mask.cpu()}
    if rank == 0:
        out.update(full_x=full.continuous.cpu(), full_k=full.discrete.cpu(), full_m=full.mask.cpu())
    torch.save(out, os.path.join(outdir, f"gen_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
