#!/usr/bin/env python3
"""Headline benchmark: generated jets/s @128 particles, 100 solver steps (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload S3|S2|S5|S4|S1] [--batch B]

One "step" = one full pass of the generation hot path over one batch of synthetic
AOJ-shaped clouds per rank: on-device SourceUniform draw (multiplicity masks from the AOJ
pmf) followed by the 100-point Euler / tau-leap loop (2 encoder forwards + K-B + K-C per grid
point for S3).  Everything is resident in HBM; outputs stay on the device (the reference's
predict_step ends in .cpu(): 9 MB per 1024 jets, 0.15 ms over PCIe against ~70 ms of compute --
not in the timed region, see DESIGN.md "Host-buffer note").  Ranks shard the jets (weak scaling), no
collective on the data path.  Rank 0 prints ONE JSON line.

`--gpus N` with N > 1 and no torchrun environment: this process starts N ranks itself (a fresh
`python -m torch.distributed.run` child, BEFORE anything here touches the GPU) and relays their output;
under torchrun (WORLD_SIZE set) it is one of the ranks and checks that WORLD_SIZE == --gpus.

Extra objects on the line:
  roofline     -- the dominant kernel (the single-launch fused generation kernel: fp32 products formed from
                  six bf16 MFMA partial products), timed live with HIP events on the launch stream over extra
                  profiled passes; peak = the guide's dense bf16 MFMA peak, frac = EXECUTED MFMA flops / peak --
                  executed on the 32-particle tiles that hold a real particle only (roofline.working_tiles: half of
                  them on AOJ multiplicities); algorithmic_* count every slot, as the reference computes them
  cpu_baseline -- the CPU oracle (op-for-op eager PyTorch restatement of the reference) timed
                  on the host cores over a bounded sample of the same workload (rank 0, N=1)
  training     -- training jets/s of the same model (fwd+bwd+all-reduce+clip+AdamW)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "multimodal-bridges_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense (the unfused per-layer kernels)
PEAK_BF16_MFMA_TFLOPS = 2516.8     # same guide: v_mfma_f32_32x32x16_bf16, dense (16 x the f32 rate; no sparsity)
# The fused kernels compute every fp32 product as six bf16 x bf16 partial products of a three-plane exact operand
# split (csrc/fused_common.h: mma_x): the roofline counts those executed bf16 flops against the bf16 peak.
PEAK_HBM_GBS = 8000.0

WORKLOADS = {"S3": "configs/S3_multimodal.yaml", "S2": "configs/S2_discrete.yaml",
             "S5": "configs/S5_large_generation.yaml", "S4": "configs/S4_transformer.yaml",
             "S1": "configs/S1_toy.yaml"}


PARITY_NOTE = {"S5": "parity unpinned (reference crashes): Euler-Maruyama with SchrodingerBridge raises upstream (SURVEY Q5); "
                     "the repaired solver is pinned against this repository's CPU oracle only",
               "S1": "goldens from the notebook's MultiModalMLP inside the reference's model; TwoMoons sampler parity unpinned"}


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


def profile_dominant_kernel(model, config, B, device, passes=3):
    """Extra generation passes with every C-ABI call bracketed by HIP events (recorded on the launch stream)."""
    import mmb_hip
    from mmb_hip.fused import working_tiles
    mmb_hip.PROFILE = []
    tiles = [0, 0]
    for _ in range(passes):
        paths = one_generation(model, config, B, device)
        if paths.mask is not None and paths.mask.dim() == 4:
            w = working_tiles(paths.mask[-1].sum(dim=(1, 2)), paths.mask.shape[2])
            tiles[0] += int(w.sum())
            tiles[1] += B * ((paths.mask.shape[2] + 31) // 32)
    torch.cuda.synchronize()
    rec, mmb_hip.PROFILE = mmb_hip.PROFILE, None
    groups = {"_tiles": tiles}
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


TRAFFIC_FILES = ("profiles/r3/pmc_traffic.json", "profiles/r2/pmc_traffic.json", "profiles/r1/pmc_traffic.json")


def measured_traffic(workload):
    """(HBM bytes per launch of the dominant kernel, file) from the newest committed rocprofv3 PMC passes of this
    build's kernels (separate --pmc runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes) -- counters cannot
    be read from inside this process, so this is the profile of the same command, not of this very run."""
    for rel in TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, rel)) as fh:
                v = json.load(fh).get(workload)
            if v is not None:
                return v, rel
        except (OSError, ValueError):
            continue
    return None, None


def training_roofline(model, config, workload, B, training):
    """Both bounds of the training step (forward + reverse + weight gradients of the set encoder): algorithmic GEMM
    flops (3 x the forward's: fwd, dX, dW) / step time against the native fp32 and -- for the flops the kernels
    execute as six bf16 products -- the bf16 MFMA peak; HBM bytes of the step (rocprofv3 PMC passes of this
    command, B = 1024) / step time against 8 TB/s.  The step is bounded by neither: see DESIGN.md."""
    sec = training["ms_per_step"] * 1e-3
    P = B * config.data.max_num_particles
    out = {"unit": "per rank and step", "ms_per_step": training["ms_per_step"]}
    try:
        from mmb_hip.fused import algorithmic_flops_per_particle
        fpp = float(algorithmic_flops_per_particle(model))
        alg = 3.0 * fpp * P / sec / 1e12
        out["mfma"] = {"algorithmic_fp32_tflops": alg, "frac_of_native_fp32_mfma_peak": alg / PEAK_FP32_MFMA_TFLOPS,
                       "peak_fp32": PEAK_FP32_MFMA_TFLOPS, "flops_per_particle_fwd": fpp, "passes": 3}
        pack = model._fused_pack() if hasattr(model, "_fused_pack") else None
        if pack and pack.get("mfma_flops_per_forward"):
            # executed: forward and reverse pass on the tiles that hold a real particle, weight gradients on the rows
            # of real particles (training["batch_work"]; 1.0 each when the switches are off or the masks are full)
            w = training.get("batch_work") or {}
            tiles_on = int(os.environ.get("MMB_TRAIN_TILES", "3")) & 1
            rows_on = os.environ.get("MMB_GW_ROWS", "1") != "0"
            share = 2.0 * (w.get("working_tiles_frac", 1.0) if tiles_on else 1.0) + \
                (w.get("real_rows_frac", 1.0) if rows_on else 1.0)
            ex = 6.0 * share * float(pack["mfma_flops_per_forward"]) * P / sec / 1e12
            out["mfma"].update({"executed_bf16_tflops": ex, "frac_of_bf16_mfma_peak": ex / PEAK_BF16_MFMA_TFLOPS,
                                "peak_bf16": PEAK_BF16_MFMA_TFLOPS, "executed_passes": share,
                                "note": "algorithmic_* count every slot of every jet, as the reference computes them; "
                                        "executed_* what the kernels run after idle tiles and padded rows are skipped"})
    except Exception:                                          # not an EPiC encoder: count the Linear layers generically
        fpp = 0.0
        for m in model.encoder.modules():
            if isinstance(m, torch.nn.MultiheadAttention):
                fpp += 2.0 * m.in_proj_weight.shape[0] * m.in_proj_weight.shape[1]
            elif isinstance(m, torch.nn.Linear):
                fpp += 2.0 * m.in_features * m.out_features
            elif hasattr(m, "dim_in") and hasattr(m, "dim_out") and hasattr(m, "weight_v"):
                fpp += 2.0 * m.dim_in * m.dim_out
        alg = 3.0 * fpp * P / sec / 1e12
        out["mfma"] = {"algorithmic_fp32_tflops": alg, "frac_of_native_fp32_mfma_peak": alg / PEAK_FP32_MFMA_TFLOPS,
                       "peak_fp32": PEAK_FP32_MFMA_TFLOPS, "flops_per_particle_fwd": fpp, "passes": 3,
                       "note": "Linear layers only (the attention products depend on the key counts and are left out)"}
    tr_step, src = measured_traffic("training_step")
    tr_step = (tr_step or {}).get(workload)
    if tr_step and B == 1024:                                  # PMC passes were taken at 1024 jets per rank and step
        by = tr_step["bytes_per_step"]
        out["hbm"] = {"traffic_bytes": by, "achieved_gbs": by / sec / 1e9, "peak_gbs": PEAK_HBM_GBS,
                      "frac": by / sec / 1e9 / PEAK_HBM_GBS, "traffic_source": src}
    else:
        out["hbm"] = None
    out["bound"] = "hbm" if out.get("hbm") else "mfma"
    return out


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
            "encoder_name": config.encoder.name, "skip_connection": getattr(config.encoder, "skip_connection", True),
            "num_heads": getattr(config.encoder, "num_heads", None)}
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
    scaled = "" if n_steps >= m.num_timesteps else ", scaled to the full grid"
    return {"value": n_jets / full, "unit": "jets/s", "cores": threads, "kind": "port",
            "sample": f"{n_jets} jets x first {min(n_steps, m.num_timesteps)} of {m.num_timesteps} grid points of "
                      f"{workload} (source sampling included), {dt:.1f} s measured{scaled}"}


def trace(msg):
    if os.environ.get("MMB_BENCH_TRACE"):
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def launch_ranks(n):
    """Start n ranks of this script under torch.distributed.run as a CHILD process and relay its exit code.  Called
    before this process has made any GPU call (importing torch and counting devices do not initialise HIP), so no
    GPU-initialised process is ever replaced or forked."""
    ndev = torch.cuda.device_count()
    if ndev < n and not os.environ.get("MMB_BENCH_REHEARSAL"):        # (rehearsal: CPU test of the launcher, gloo)
        raise SystemExit(f"bench.py --gpus {n}: only {ndev} GPU(s) visible; refusing to report an n_gpus it did not use")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    if os.environ.get("MMB_BENCH_TRACE"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MMB_BENCH_TRACE"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="S3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0,
                    help="jets per rank per step (0: 1024, the reference's batch_size; S5, whose driver shards a million jets "
                         "in chunks of its own choosing: the chunk of configs/S5_large_generation.yaml, 8192)")
    ap.add_argument("--solver-steps", type=int, default=0, help="grid points (0: the workload's own: 100, S5: 200)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-jets", type=int, default=256)
    ap.add_argument("--cpu-steps", type=int, default=100)
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--train-only", action="store_true", help="time only the training step (S4 has no fused generation)")
    ap.add_argument("--rendezvous-check", action="store_true",
                    help="form the process group, all-reduce one number, print the world size and exit (launcher test)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher formed WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rendezvous_check:
        import torch.distributed as dist
        dist.init_process_group(os.environ.get("MMB_DIST_BACKEND", "nccl"))
        one = torch.ones(1)
        dist.all_reduce(one)
        if dist.get_world_size() != args.gpus or int(one.item()) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
        if rank == 0:
            print(json.dumps({"n_gpus": dist.get_world_size(), "rendezvous": "ok"}), flush=True)
        dist.destroy_process_group()
        return
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

        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
        world = dist.get_world_size()

    trace("process group up")
    model, config = build_model(args.workload, args.solver_steps, device)
    trace("model built")
    args.solver_steps = config.model.num_timesteps
    model.eval()
    model.manual_seed(0xA0A0, rank)
    B = args.batch or (int(config.data.batch_size) if args.workload == "S5" else 1024)
    B_train = min(B, 1024)                                      # the training leg stays at the reference's batch size

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.train_only:
        # S4 (BASELINE configs[3]) is a TRAINING config: data-parallel step of the transformer set encoder
        from pipeline.experiment import bench_training
        try:
            tr = bench_training(model, config, B, device, dist, steps=args.steps, warmup=args.warmup)
        except Exception:
            if dist is None:
                raise
            import traceback
            traceback.print_exc()
            sys.stderr.flush()
            os._exit(13)
        if rank == 0:
            tr["roofline"] = training_roofline(model, config, args.workload, B, tr)
            print(json.dumps({
                "metric": f"training jets/sec @{config.data.max_num_particles} particles", "value": tr["value"],
                "unit": "jets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": tr["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload]} -- {config.encoder.name}, "
                                       f"N={config.data.max_num_particles}, training step",
                           "jets_per_rank_per_step": B,
                           "parallelism": f"data-parallel over {world} rank(s), one flat fp32 gradient all-reduce per step"},
                "training": tr}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

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
            training = bench_training(model, config, B_train, device, dist, steps=max(args.steps * 5, 10), warmup=10)
        except Exception as exc:                                   # reported, never hidden
            if dist is not None:
                # the other ranks are (or will be) inside a collective this rank never joins: leave at once with a
                # failure code so that the launcher takes every rank down instead of letting them hang
                import traceback
                traceback.print_exc()
                sys.stderr.flush()
                os._exit(13)
            training = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        groups = profile_dominant_kernel(model, config, B, device)
        tiles = groups.pop("_tiles")
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
            tr, tr_file = measured_traffic(args.workload)
            # what the kernel's MFMAs really execute: the time/global-vector columns of every layer are hoisted
            # into per-jet matvecs, so this is below the reference's algorithmic count; every fp32 product is six
            # bf16 MFMA products, all of which count against the guide's dense bf16 peak
            # ... and only the 32-particle tiles that hold a real particle (or the one representative padded slot)
            # run them: csrc/fused_epic.hip TilePlan; `tiles` = [working, all] over the profiled launches
            ex_fpp = model._fused_pack()["mfma_flops_per_forward"]
            executed = float(ex_fpp) * 32 * tiles[0] * n_fwd / (fz["ms"] * 1e-3) / 1e12
            kname = "fused_generate_kernel" if hybrid else "fused_uni_generate_kernel"
            roof = {"kernel": kname + " (whole generation loop; fp32 products as 6 x v_mfma_f32_32x32x16_bf16 over "
                                      "a 3-plane exact operand split)", "bound": "mfma",
                    "achieved": 6 * executed, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": 6 * executed / PEAK_BF16_MFMA_TFLOPS, "traffic": tr, "traffic_source": tr_file,
                    "peak_basis": "MI355X_MICROARCH.md dense bf16 MFMA peak; achieved = bf16 MFMA flops the kernel "
                                  "executes (6 per fp32 multiply-add pair of the hoisted per-particle GEMMs)",
                    "executed_mfma_flops_per_particle_per_forward": ex_fpp,
                    "working_tiles": {"working": tiles[0], "all": tiles[1], "frac": tiles[0] / max(tiles[1], 1),
                                      "note": "tiles of 32 slots without a real particle do no per-particle work; "
                                              "algorithmic_* still count every slot, as the reference computes them"},
                    "algorithmic_fp32_tflops": achieved,
                    "algorithmic_frac_of_native_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
                    # round 1 quoted this one: algorithmic fp32 flops against what six bf16 products per fp32 product
                    # can deliver at most (bf16 peak / 6)
                    "algorithmic_frac_of_six_product_bound": achieved / (PEAK_BF16_MFMA_TFLOPS / 6.0),
                    "frac_basis": "frac = executed bf16 MFMA flops / dense bf16 peak (rounds 2, 3); "
                                  "algorithmic_frac_of_six_product_bound = SURVEY 8(d) flops / (bf16 peak / 6) (round 1); "
                                  "algorithmic_frac_of_native_fp32_mfma_peak = SURVEY 8(d) flops / fp32 MFMA peak",
                    "algorithmic_flops_per_particle_per_forward": fpp, "forwards_per_launch": n_fwd,
                    "launches": fz["n"], "avg_launch_us": fz["ms"] * 1e3 / fz["n"],
                    "share_of_gpu_time": fz["ms"] / total_ms}
        elif groups.get("mmb_toy_generate"):
            tz = groups["mmb_toy_generate"]
            enc = model.encoder
            H, Dc, V = enc.dim_hidden, enc.dim, enc.vocab_size
            fpp = 2.0 * ((1 + Dc + H // 4) * H + 3 * H * H + H * (Dc + V))        # per point and forward
            alg = fpp * config.data.max_num_particles * B * 2 * args.solver_steps * tz["n"]
            achieved = alg / (tz["ms"] * 1e-3) / 1e12
            roof = {"kernel": "toy_generate_kernel (whole generation loop of the toy MLP; v_mfma_f32_32x32x2_f32, "
                              "32 points per wave)", "bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                    "note": f"{(B * config.data.max_num_particles + 31) // 32} waves of work on a 1024-SIMD chip: the "
                            "configured batch is latency-bound; --batch 32768 fills the chip",
                    "launches": tz["n"], "avg_launch_us": tz["ms"] * 1e3 / tz["n"],
                    "share_of_gpu_time": tz["ms"] / total_ms}
        elif groups.get("mmb_transformer_tail") and groups["mmb_transformer_tail"]["ms"] >= max(g["ms"] for g in groups.values()):
            tl = groups["mmb_transformer_tail"]
            achieved = tl["flops"] / (tl["ms"] * 1e-3) / 1e12
            roof = {"kernel": "transformer_tail_kernel (out_proj + LayerNorm1 + FFN + LayerNorm2 + next in_proj per launch; "
                              "fp32 products as 6 x v_mfma_f32_32x32x16_bf16 over a 3-plane exact operand split)",
                    "bound": "mfma", "achieved": 6 * achieved, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": 6 * achieved / PEAK_BF16_MFMA_TFLOPS, "traffic": None,
                    "algorithmic_fp32_tflops": achieved,
                    "algorithmic_frac_of_native_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
                    "launches": tl["n"], "avg_launch_us": tl["ms"] * 1e3 / tl["n"],
                    "share_of_gpu_time": tl["ms"] / total_ms,
                    "note": "the S4 generation pass is three kernels of similar weight: this one, mmb_attention_fwd "
                            "and the remaining mmb_linear_cat launches (kernel_time_ms)"}
        elif dom:
            achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            # the group holds every per-particle mmb_linear_cat launch: most run linear_stream_x6_kernel (fp32 products
            # as six bf16 MFMA products), unaligned first layers the tiled fp32-MFMA kernel, and MMB_LINEAR_PATH can
            # force either -- so `achieved` is the ALGORITHMIC fp32 rate against the native fp32 MFMA peak, and the
            # bf16 figure (valid only for the x6 launches) is a separate, labelled estimate
            roof = {"kernel": "mmb_linear_cat, per-particle layers (linear_stream_x6_kernel for aligned widths, tiled "
                              "fp32-MFMA kernel otherwise)", "bound": "mfma",
                    "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                    "linear_path": os.environ.get("MMB_LINEAR_PATH", "auto"),
                    "x6_estimate_bf16_tflops_if_all_launches_ran_x6": 6 * achieved,
                    "x6_estimate_frac_of_bf16_peak": 6 * achieved / PEAK_BF16_MFMA_TFLOPS,
                    "launches": dom["n"], "avg_launch_us": dom["ms"] * 1e3 / dom["n"],
                    "share_of_gpu_time": dom["ms"] / total_ms}
        line = {
            "metric": f"generated jets/sec @{config.data.max_num_particles} particles, {args.solver_steps} steps",
            "value": value, "unit": "jets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload]} -- {config.encoder.name}, "
                                   f"N={config.data.max_num_particles}, {args.solver_steps} grid points, "
                                   f"{config.model.solver_continuous}+{config.model.solver_discrete}",
                       "jets_per_rank_per_step": B, "parallelism": f"jets sharded over {world} rank(s), no collective",
                       "parity": PARITY_NOTE.get(args.workload, "reference goldens (tests/golden/model_*.npz)")},
            "roofline": roof,
            "kernel_time_ms": {k: round(v["ms"], 3) for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])},
        }
        if training is not None:
            if "ms_per_step" in training:
                training["roofline"] = training_roofline(model, config, args.workload, B_train, training)
            line["training"] = training
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model, config, args.workload, args.cpu_jets, args.cpu_steps)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
