#!/usr/bin/env python3
"""Headline benchmark: generated jets/s @128 particles, 100 solver steps (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload S3|S2] [--batch B]

One "step" = one full pass of the generation hot path over one batch of synthetic
AOJ-shaped clouds per rank: on-device SourceUniform draw (multiplicity masks from the AOJ
pmf) followed by the 100-point Euler / tau-leap loop (2 encoder forwards + K-B + K-C per grid
point for S3).  Everything is resident in HBM; outputs stay on the device.  Ranks shard the
jets (weak scaling), no collective on the data path.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     -- the dominant kernel (the fp32-MFMA per-particle linear layer), timed live with
                  HIP events on the launch stream during one extra profiled pass
  cpu_baseline -- the CPU oracle (op-for-op eager PyTorch restatement of the reference) timed
                  on the host cores over a bounded sample of the same workload (rank 0, N=1)
  training     -- training jets/s of the same model (fwd+bwd+AdamW), when available
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "multimodal-bridges_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense (the unfused per-layer kernels)
PEAK_BF16_MFMA_TFLOPS = 2516.8     # same guide: v_mfma_f32_32x32x16_bf16, dense (16 x the f32 rate; no sparsity)
# The fused kernels compute every fp32 product as six bf16 x bf16 partial products of a three-plane exact operand
# split (csrc/fused_common.h: mma_x), so an fp32 flop costs six bf16 flops: the peak they are priced against.
PEAK_FP32_ON_BF16_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6
PEAK_HBM_GBS = 8000.0

WORKLOADS = {"S3": "configs/S3_multimodal.yaml", "S2": "configs/S2_discrete.yaml",
             "S5": "configs/S5_large_generation.yaml"}


def build_model(workload, steps_per_gen, device):
    import yaml
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    with open(os.path.join(ROOT, WORKLOADS[workload])) as fh:
        cfg = yaml.safe_load(fh)
    cfg["data"]["target_path"] = os.path.join(ROOT, "tests", "golden")
    if steps_per_gen:
        cfg["model"]["num_timesteps"] = steps_per_gen
    config = ExperimentConfigs(cfg)
    torch.manual_seed(0)
    model = MultiModalBridgeMatching(config)
    return model.to(device), config


def gen_batch(config, B, device):
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    N = config.data.max_num_particles
    shape_only = TensorMultiModal(None, None, None, torch.ones(B, N, 1, dtype=torch.int64, device=device))
    if config.data.modality != "discrete":
        shape_only.continuous = torch.zeros(B, N, config.data.dim_continuous, device=device)
    shape_only.discrete = torch.zeros(B, N, 1, dtype=torch.int64, device=device)
    return DataCoupling(TensorMultiModal(), shape_only, None)


def one_generation(model, config, B, device):
    batch = gen_batch(config, B, device)
    paths, _ = model.predict_step(batch, 0, to_cpu=False)
    return paths


def profile_dominant_kernel(model, config, B, device):
    """One extra generation pass with every C-ABI call bracketed by HIP events."""
    import mmb_hip
    mmb_hip.PROFILE = []
    one_generation(model, config, B, device)
    torch.cuda.synchronize()
    rec, mmb_hip.PROFILE = mmb_hip.PROFILE, None
    groups = {}
    for name, meta, e0, e1 in rec:
        ms = e0.elapsed_time(e1)
        key = name
        if name == "mmb_linear_cat":
            key = "mmb_linear_cat[per-particle]" if meta["M"] > 4096 else "mmb_linear_cat[per-jet]"
        g = groups.setdefault(key, {"ms": 0.0, "n": 0, "flops": 0.0})
        g["ms"] += ms
        g["n"] += 1
        if meta:
            g["flops"] += meta["flops"]
    return groups


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r1/pmc_traffic.json; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes)."""
    path = os.path.join(ROOT, "profiles", "r1", "pmc_traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh).get(workload)
    except (OSError, ValueError):
        return None


def cpu_baseline(model, config, workload, n_jets, n_steps):
    """CPU oracle on a bounded sample: n_jets jets, the first n_steps of the time grid."""
    from oracle import model as OM, source as OS
    from datamodules.synthetic import load_pmf
    p = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    m = config.model
    flat = {"time_eps": m.time_eps, "num_timesteps": m.num_timesteps, "gamma": m.gamma, "sigma": m.sigma,
            "thermostat_fn": m.thermostat_fn, "solver_continuous": m.solver_continuous,
            "bridge_continuous": getattr(m, "bridge_continuous", None),
            "solver_discrete": m.solver_discrete, "vocab_size": config.data.vocab_size,
            "encoder_name": config.encoder.name, "skip_connection": config.encoder.skip_connection}
    N = config.data.max_num_particles
    pmf = load_pmf(os.path.join(ROOT, "tests", "golden"))
    threads = torch.get_num_threads()
    t0 = time.perf_counter()
    src = OS.source_uniform((n_jets, N), config.data.modality, config.data.vocab_size, config.data.dim_continuous,
                            pmf, sample_masks=True)
    with torch.no_grad():
        OM.simulate_dynamics(p, flat, src, noise=None, max_steps=n_steps)
    dt = time.perf_counter() - t0
    full = dt / n_steps * m.num_timesteps
    return {"value": n_jets / full, "unit": "jets/s", "cores": threads, "kind": "port",
            "sample": f"{n_jets} jets x first {n_steps} of {m.num_timesteps} grid points of {workload} "
                      f"(source sampling included), {dt:.1f} s measured, scaled to the full grid"}


def trace(msg):
    if os.environ.get("MMB_BENCH_TRACE"):
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def main():
    if os.environ.get("MMB_BENCH_TRACE"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MMB_BENCH_TRACE"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="S3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=1024, help="jets per rank per step")
    ap.add_argument("--solver-steps", type=int, default=0, help="grid points (0: the workload's own: 100, S5: 200)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-jets", type=int, default=128)
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--no-train", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)            # (rehearsals put several ranks on one card; the driver never does)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("MMB_DIST_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm; gloo only for rehearsal
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    trace("process group up")
    model, config = build_model(args.workload, args.solver_steps, device)
    trace("model built")
    args.solver_steps = config.model.num_timesteps
    model.eval()
    model.manual_seed(0xA0A0, rank)
    B = args.batch

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_generation(model, config, B, device)
    trace("warm-up done")
    barrier()
    trace("barrier passed")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_generation(model, config, B, device)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    jets = B * world * args.steps
    value = jets / elapsed
    trace("generation timed")

    training = None
    if not args.no_train:
        try:
            from pipeline.experiment import bench_training
            training = bench_training(model, config, B, device, dist, steps=max(args.steps * 5, 10), warmup=3)
        except Exception as exc:                                   # reported, never hidden
            training = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        groups = profile_dominant_kernel(model, config, B, device)
        dom = groups.get("mmb_linear_cat[per-particle]")
        total_ms = sum(g["ms"] for g in groups.values())
        roof = None
        fz = groups.get("mmb_fused_epic_generate") or groups.get("mmb_fused_uni_generate")
        if fz:
            from mmb_hip.fused import algorithmic_flops_per_particle
            fpp = algorithmic_flops_per_particle(model)
            hybrid = config.data.modality == "multi-modal"
            n_fwd = (2 if hybrid else 1) * args.solver_steps     # hybrid: one encoder forward per solver per grid point
            alg = float(fpp) * config.data.max_num_particles * B * n_fwd * fz["n"]
            achieved = alg / (fz["ms"] * 1e-3) / 1e12
            tr = measured_traffic(args.workload)
            # what the kernel's MFMAs really execute: the time/global-vector columns of every layer are hoisted
            # into per-jet matvecs, so this is below the reference's algorithmic count (frac may exceed
            # executed_frac, and 1.0, for that reason)
            ex_fpp = model._fused_pack()["mfma_flops_per_forward"]
            executed = float(ex_fpp) * config.data.max_num_particles * B * n_fwd * fz["n"] / (fz["ms"] * 1e-3) / 1e12
            kname = "fused_generate_kernel" if hybrid else "fused_uni_generate_kernel"
            roof = {"kernel": kname + " (whole generation loop; fp32 products as 6 x v_mfma_f32_32x32x16_bf16 over "
                                      "a 3-plane exact operand split)", "bound": "mfma",
                    "achieved": achieved, "peak": PEAK_FP32_ON_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP32_ON_BF16_TFLOPS, "traffic": tr,
                    "peak_basis": "bf16 dense MFMA peak 2516.8 TFLOP/s / 6 partial products per fp32 product; "
                                  "achieved counts fp32 flops of the reference's layers",
                    "frac_of_native_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
                    "executed_mfma_flops_per_particle_per_forward": ex_fpp,
                    "executed_bf16_tflops": 6 * executed,
                    "executed_frac": 6 * executed / PEAK_BF16_MFMA_TFLOPS,
                    "launches": fz["n"], "avg_launch_us": fz["ms"] * 1e3 / fz["n"],
                    "algorithmic_flops_per_particle_per_forward": fpp, "forwards_per_launch": n_fwd,
                    "share_of_gpu_time": fz["ms"] / total_ms}
        elif dom:
            achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            roof = {"kernel": "linear_cat_kernel (fp32 MFMA 32x32x2, per-particle layers)", "bound": "mfma",
                    "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                    "launches": dom["n"], "avg_launch_us": dom["ms"] * 1e3 / dom["n"],
                    "share_of_gpu_time": dom["ms"] / total_ms}
        line = {
            "metric": f"generated jets/sec @128 particles, {args.solver_steps} steps", "value": value, "unit": "jets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload]} -- {config.encoder.name}, "
                                   f"N={config.data.max_num_particles}, {args.solver_steps} grid points, "
                                   f"{config.model.solver_continuous}+{config.model.solver_discrete}",
                       "jets_per_rank_per_step": B, "parallelism": f"jets sharded over {world} rank(s), no collective"},
            "roofline": roof,
            "kernel_time_ms": {k: round(v["ms"], 3) for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])},
        }
        if training is not None:
            tr_step = (measured_traffic("training_step") or {}).get(args.workload)
            if tr_step and B == 1024:                       # PMC passes were taken at 1024 jets per rank and step
                per_rank = tr_step["bytes_per_step"]
                training["hbm_traffic_bytes_per_rank_step"] = per_rank
                training["hbm_frac"] = per_rank / (training["ms_per_step"] * 1e-3) / 8.0e12
            line["training"] = training
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model, config, args.workload, args.cpu_jets, args.cpu_steps)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
