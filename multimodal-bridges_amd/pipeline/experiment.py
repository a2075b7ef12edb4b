"""ExperimentPipeline: train() / generate() without Lightning (reference:
pipeline/experiment.py:18-224).  One process per GPU (launch with torchrun); data-parallel
training exchanges ONE flat fp32 gradient bucket per step through RCCL
(torch.distributed backend "nccl" on ROCm); generation shards jets across ranks and gathers
once at the end.

    pipeline = ExperimentPipeline(datamodule_cls, config="configs/S3_multimodal.yaml")
    pipeline.train(); pipeline.generate()

The gradient payload is tiny (2.4 MB for S3) so the all-reduce is latency-bound: every
parameter's .grad is a VIEW into one pre-allocated flat buffer (parameters without a gradient
-- the truncated embedding Q1, the dead fused layer K2 -- stay zero there, which is what the
reference's find_unused_parameters papers over), so a step costs exactly one collective, and
the global-norm clip runs on the already averaged buffer.
"""
import json
import os
import inspect
import time
from typing import Optional

import torch

from pipeline.configs import ExperimentConfigs
from tensorclass import TensorMultiModal
from datamodules.datasets import DataCoupling


class FlatGrads:
    """All gradients as views of one flat fp32 buffer (single-bucket all-reduce)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()
        off = 0
        for p in self.params:                      # re-attach views if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def all_reduce_mean(self, dist, world):
        if dist is not None and world > 1:
            if dist.get_backend() == "gloo" and self.flat.is_cuda:     # rehearsal only: gloo on device
                host = self.flat.cpu()                                 # tensors is unreliable on ROCm
                dist.all_reduce(host)
                self.flat.copy_(host)
            else:
                dist.all_reduce(self.flat)
            self.flat.div_(world)

    def clip_(self, max_norm):
        """Global L2-norm clip (Lightning gradient_clip_val semantics), no host sync."""
        if not max_norm:
            return None
        norm = torch.linalg.vector_norm(self.flat)
        self.flat.mul_(torch.clamp(max_norm / (norm + 1e-6), max=1.0))
        return norm


def make_optimizer(model, config):
    """-> (optimizer, scheduler | None, flat): FlatAdamW (clip + AdamW in two launches on flat buffers, its own
    gradient bucket) when the config asks for AdamW, else the registry's torch optimizer + a FlatGrads bucket."""
    from pipeline.registry import registered_optimizers, registered_schedulers
    tr = config.trainer
    params = dict(tr.optimizer_params or {})
    if tr.optimizer_name == "AdamW" and not params.get("amsgrad"):
        from pipeline.optim import FlatAdamW, unused_parameter_names
        params.pop("amsgrad", None)
        opt = FlatAdamW(model.named_parameters(), unused=unused_parameter_names(model), **params)
        flat = opt
    else:
        opt = registered_optimizers[tr.optimizer_name](model.parameters(), **params)
        flat = FlatGrads(model.parameters())
    sched = None
    if tr.scheduler_name:
        sched = registered_schedulers[tr.scheduler_name](opt, **(tr.scheduler_params or {}))
    return opt, sched, flat


def optimizer_step(opt, flat, max_norm):
    if flat is opt:
        opt.step(max_norm=max_norm)
    else:
        flat.clip_(max_norm)
        opt.step()


def _staged(dist, t):
    """gloo moves host memory only (the two-ranks-on-one-GPU rehearsal; RCCL refuses two ranks on one device): device
    tensors of the small control-plane collectives go through the host then."""
    return dist.get_backend() == "gloo" and t.is_cuda


def all_reduce_(dist, t, op=None):
    op = dist.ReduceOp.SUM if op is None else op
    if _staged(dist, t):
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def broadcast_(dist, t, src=0):
    if _staged(dist, t):
        h = t.cpu()
        dist.broadcast(h, src=src)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src)
    return t


def gather_generated(gen, dist, rank, world):
    """ONE tensor gather of the generated shards to rank 0 (the reference concatenates per-rank files,
    callbacks_generator.py:53-70).  x (J,N,Dc) f32, k (J,N,1), mask (J,N,1) are packed into one (Jmax, N, Dc+2) f32
    tensor per rank (tokens and masks are small integers: exact in f32); shards may differ by one jet, so they are
    padded to the largest and cut back on rank 0.  Returns the concatenation on rank 0, None elsewhere."""
    dev = gen.mask.device
    J, N = gen.mask.shape[0], gen.mask.shape[1]
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    counts[rank] = J
    all_reduce_(dist, counts)
    jmax = int(counts.max())
    has_c, has_d = gen.has_continuous, gen.has_discrete
    Dc = gen.continuous.shape[-1] if has_c else 0
    width = Dc + (1 if has_d else 0) + 1
    pack = torch.zeros(jmax, N, width, dtype=torch.float32, device=dev)
    if has_c:
        pack[:J, :, :Dc] = gen.continuous
    if has_d:
        pack[:J, :, Dc] = gen.discrete.squeeze(-1).float()
    pack[:J, :, -1] = gen.mask.squeeze(-1).float()
    host = dist.get_backend() == "gloo" and pack.is_cuda                     # rehearsal only
    buf = pack.cpu() if host else pack
    parts = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, parts, dst=0)
    if rank != 0:
        return None
    outs = []
    for r, part in enumerate(parts):
        part = part[:int(counts[r])].to(dev)
        outs.append(TensorMultiModal(None, part[..., :Dc].contiguous() if has_c else None,
                                     part[..., Dc:Dc + 1].long() if has_d else None, part[..., -1:].long()))
    return TensorMultiModal.cat(outs, dim=0)


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of rank; shards differ by at most one item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _dist_env():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return world, rank, local


class ExperimentPipeline:
    def __init__(self, datamodule_cls=None, config=None, experiment_path: Optional[str] = None,
                 load_ckpt: str = "last", accelerator: str = "gpu", strategy: str = "ddp", devices="auto",
                 num_nodes: int = 1, sync_batchnorm: bool = False, val_check_interval=None, **config_updates):
        from model.multimodal_bridge_matching import MultiModalBridgeMatching
        self.world, self.rank, self.local = _dist_env()
        if not torch.cuda.is_available():
            raise RuntimeError("ExperimentPipeline needs an MI355X: the HIP path has no CPU fallback")
        torch.cuda.set_device(self.local)
        self.device = torch.device("cuda", self.local)
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group("nccl", device_id=self.device)
            self.dist = dist
        self.ckpt_path = None
        if experiment_path is not None:
            self.experiment_path = experiment_path
            self.ckpt_path = os.path.join(experiment_path, "checkpoints", f"{load_ckpt}.ckpt")
            blob = torch.load(self.ckpt_path, map_location="cpu", weights_only=True)
            self.config = ExperimentConfigs(blob["config"])
            self.config.update(config_updates or {})
            self.model = MultiModalBridgeMatching(self.config)
            self.model.load_state_dict(blob["state_dict"])
            self._resume = blob
        else:
            self.config = config if isinstance(config, ExperimentConfigs) else ExperimentConfigs(config)
            self.config.update(config_updates or {})
            self.experiment_path = self.config.path or self.config.checkpoints.dirpath
            if self.experiment_path is None and self.rank == 0:
                print("[ExperimentPipeline] checkpoints.dirpath is null: nothing will be saved (no checkpoints, no "
                      "metrics.jsonl, no generated sample)", flush=True)
            self.model = MultiModalBridgeMatching(self.config)
            self._resume = None
        self.model.to(self.device)
        self.model.manual_seed(0xA0A0, self.rank)
        if self._resume is not None and "philox_calls" in self._resume:
            self.model._calls = int(self._resume["philox_calls"])          # resumed epochs draw NEW noise streams
        if self.dist is not None:                                  # identical initial weights on every rank
            for p in self.model.parameters():
                broadcast_(self.dist, p.data, src=0)
        self.datamodule = datamodule_cls(self.config) if datamodule_cls is not None else None
        self.history = []

    # ------------------------------------------------------------------ training
    def _log(self, rec):
        self.history.append(rec)
        if self.rank == 0 and self.experiment_path:
            os.makedirs(self.experiment_path, exist_ok=True)
            with open(os.path.join(self.experiment_path, "metrics.jsonl"), "a") as fh:
                fh.write(json.dumps(rec) + "\n")

    def _save(self, name, epoch, optimizer, scheduler):
        if self.rank != 0 or not self.experiment_path:
            return
        d = os.path.join(self.experiment_path, "checkpoints")
        os.makedirs(d, exist_ok=True)
        torch.save({"state_dict": {k: v.cpu() for k, v in self.model.state_dict().items()},
                    "optimizer": optimizer.state_dict(),
                    "scheduler": scheduler.state_dict() if scheduler else None, "epoch": epoch,
                    "philox_calls": int(self.model._calls),
                    "config": self.config.to_dict()}, os.path.join(d, f"{name}.ckpt"))

    def _write_metadata(self):
        """Target statistics next to the checkpoints (the reference's datamodule writes metadata.json into the
        experiment directory; generate() reads it back to invert the data transform)."""
        meta = getattr(self.datamodule, "metadata", None)
        if meta and self.rank == 0 and self.experiment_path:
            os.makedirs(self.experiment_path, exist_ok=True)
            with open(os.path.join(self.experiment_path, "metadata.json"), "w") as fh:
                json.dump(meta, fh)

    def _min_over_ranks(self, n):
        if self.dist is None:
            return n
        t = torch.tensor([n], dtype=torch.int64, device=self.device)
        all_reduce_(self.dist, t, op=self.dist.ReduceOp.MIN)
        return int(t.item())

    def train(self):
        cfg = self.config
        model = self.model.train()
        opt, sched, flat = make_optimizer(model, cfg)
        start = 0
        if self._resume is not None:
            opt.load_state_dict(self._resume["optimizer"])
            if sched is not None and self._resume.get("scheduler"):
                sched.load_state_dict(self._resume["scheduler"])
            start = int(self._resume["epoch"]) + 1
        self.datamodule.setup("fit", rank=self.rank, world=self.world, device=self.device)
        self._write_metadata()
        best = float("inf")
        for epoch in range(start, cfg.trainer.max_epochs):
            model.train()
            acc = torch.zeros(3, device=self.device)
            n = 0
            if "epoch" in inspect.signature(self.datamodule.train_dataloader).parameters:
                loader = self.datamodule.train_dataloader(epoch=epoch)      # reshuffled every epoch (shuffle=True upstream)
            else:
                loader = self.datamodule.train_dataloader()
            batches = list(loader) if not hasattr(loader, "__len__") else loader
            # every step ends in a collective: all ranks must run the SAME number of steps (shards differ by a jet)
            n_steps = self._min_over_ranks(len(batches))
            for batch in batches:
                if n >= n_steps:
                    break
                flat.zero()
                out = model.training_step(batch, n)
                out["loss"].backward()
                flat.all_reduce_mean(self.dist, self.world)
                optimizer_step(opt, flat, cfg.trainer.gradient_clip_val)
                acc += torch.stack([out["loss"].detach(), out["train_loss_continuous"], out["train_loss_discrete"]])
                n += 1
            if sched is not None:
                sched.step()
            val = torch.zeros(1, device=self.device)
            nv = 0
            model.eval()
            for batch in self.datamodule.val_dataloader():
                val += model.validation_step(batch, nv)["val_loss"]
                nv += 1
            stats = torch.cat([acc / max(n, 1), val / max(nv, 1)])
            if self.dist is not None:
                all_reduce_(self.dist, stats)
                stats /= self.world
            tl, tc, td, vl = [float(v) for v in stats.tolist()]       # one host sync per EPOCH
            self._log({"epoch": epoch, "train_loss": tl, "train_loss_continuous": tc, "train_loss_discrete": td,
                       "val_loss": vl, "lr": opt.param_groups[0]["lr"]})
            if nv and vl < best:
                best = vl
                self._save("best", epoch, opt, sched)
            self._save("last", epoch, opt, sched)
        return self.history

    # ------------------------------------------------------------------ generation
    @torch.no_grad()
    def generate(self, gather=True, postprocess=True):
        """Each rank generates its shard of the test jets, inverts the data transform and applies the mask on
        the device (pipeline/postprocess.py; reference GeneratorCallback._postprocess); rank 0 receives the
        concatenation."""
        model = self.model.eval()
        transform = getattr(self.config.data, "transform", None)
        metadata = None
        if postprocess and transform is not None:            # fail BEFORE generating, not after
            if not self.experiment_path:
                raise RuntimeError("generate(): data.transform is set but there is no experiment_path to read the "
                                   "target statistics (metadata.json) from; set checkpoints.dirpath")
            with open(os.path.join(self.experiment_path, "metadata.json")) as fh:
                metadata = json.load(fh)
        self.datamodule.setup("predict", rank=self.rank, world=self.world, device=self.device)
        outs = []
        for i, batch in enumerate(self.datamodule.predict_dataloader()):
            paths, _ = model.predict_step(batch, i, to_cpu=False)
            outs.append(paths[1])
        gen = TensorMultiModal.cat(outs, dim=0)
        if postprocess:
            from pipeline.postprocess import postprocess as _post
            # the final state (J,N,.) goes in: for "log_pt" that exponentiates feature 0 of every particle; upstream
            # hands the stacked 4-D paths over, where the same index expression hits slot 0 of every jet instead
            # (pipeline/postprocess.py reproduces either, by the rank of its input)
            gen = _post(gen, transform, metadata, getattr(self.config.data, "discrete_features", "tokens"),
                        getattr(self.config.data, "vocab_size", None))
        if gather and self.dist is not None:
            gen = gather_generated(gen, self.dist, self.rank, self.world)
        if self.rank == 0 and gen is not None and self.experiment_path:
            os.makedirs(self.experiment_path, exist_ok=True)
            gen.cpu().save_to(os.path.join(self.experiment_path, "generated_sample.h5"))
        return gen


def bench_training(model, config, B, device, dist, steps=10, warmup=3):
    """Training jets/s: K-A bridge sampling -> taped encoder forward -> K-F loss -> HIP backward ->
    flat-bucket all-reduce -> clip -> AdamW, on a resident synthetic AOJ-shaped batch."""
    from datamodules.synthetic import synthetic_target
    world, rank, _ = _dist_env()
    was_training = model.training
    model.train()
    tgt = synthetic_target(B, config.data.max_num_particles, config.data.target_path, seed=1234 + rank,
                           Dc=max(config.data.dim_continuous, 1), V=config.data.vocab_size,
                           modality=config.data.modality, device=device)
    opt, _, flat = make_optimizer(model, config)

    def step(i):
        flat.zero()
        batch = DataCoupling(TensorMultiModal(), TensorMultiModal(None, tgt.continuous, tgt.discrete, tgt.mask), None)
        out = model.training_step(batch, i)
        out["loss"].backward()
        flat.all_reduce_mean(dist, world)
        optimizer_step(opt, flat, config.trainer.gradient_clip_val)
        return out["loss"]

    for i in range(warmup):
        step(i)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(i)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        all_reduce_(dist, tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # untimed extra steps with device events between the phases (one stream: the events split the step exactly), and
    # with every C-ABI call bracketed (mmb_hip.PROFILE) -- what a weak scaling curve is diagnosed with
    import mmb_hip
    n_ph = 5
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n_ph)]
    for j in range(n_ph):
        flat.zero()
        ev[j][0].record()
        batch = DataCoupling(TensorMultiModal(), TensorMultiModal(None, tgt.continuous, tgt.discrete, tgt.mask), None)
        model.training_step(batch, steps + j)["loss"].backward()
        ev[j][1].record()
        flat.all_reduce_mean(dist, world)
        ev[j][2].record()
        optimizer_step(opt, flat, config.trainer.gradient_clip_val)
        ev[j][3].record()
    torch.cuda.synchronize()
    ph = [sum(e[i].elapsed_time(e[i + 1]) for e in ev) / n_ph for i in range(3)]
    mmb_hip.PROFILE = []
    for j in range(3):
        step(steps + n_ph + j)
    torch.cuda.synchronize()
    rec, mmb_hip.PROFILE = mmb_hip.PROFILE, None
    kern = {}
    for name, _, e0, e1 in rec:
        kern[name] = kern.get(name, 0.0) + e0.elapsed_time(e1) / 3
    model.train(was_training)
    for p in model.parameters():
        p.grad = None
    # what the fused kernels skip on this batch: tiles without a real particle (forward, reverse), rows of padded slots
    # (weight gradients) -- the bench's roofline counts executed work with these
    n = tgt.mask.sum(dim=(1, 2))
    N = tgt.mask.shape[1]
    tiles = ((n + 31) // 32).clamp(max=(N + 31) // 32)
    work = {"working_tiles_frac": float(tiles.sum()) / (B * ((N + 31) // 32)), "real_rows_frac": float(n.sum()) / (B * N)}
    return {"value": B * world * steps / dt, "batch_work": work, "unit": "jets/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "global_batch": B * world, "dropout": config.encoder.dropout, "final_loss": float(loss.detach()),
            "optimizer": type(opt).__name__,
            "includes": "bridge sampling, fwd, loss, bwd, grad all-reduce, clip, AdamW",
            "fwd_loss_bwd_ms": ph[0], "allreduce_ms": ph[1], "optimizer_ms": ph[2],
            "entry_point_ms_per_step": {k: round(v, 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1])}}
