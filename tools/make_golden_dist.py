#!/usr/bin/env python3
"""Distribution pins (SURVEY.md 8c golden 9): run the REFERENCE generator (imported from
/root/reference, same harness as tools/make_golden.py) R times with different seeds on the
S3 model with the name-synthesised weight set, and store per-observable quantile functions
of every run plus the run-to-run Wasserstein-1 spread.  Observables are those of the
reference's generation callback minus the fastjet ones: particle pt/eta_rel/phi_rel on
mask==1 slots, multiplicity, per-flavour counts per jet.

Runs ONLY in the authoring container.  Output: tests/golden/dist_S3.npz (data only).
"""
import copy
import json
import os
import sys
import time

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # noqa: E402

NQ = 257
OBS = ["pt", "eta_rel", "phi_rel", "multiplicity"] + [f"count_flavour_{v}" for v in range(8)]


def observables(x, k, mask):
    """x (B,N,3) f32, k (B,N,1) i64, mask (B,N,1) -> dict name -> 1-D float64 sample"""
    m = mask[..., 0] > 0
    out = {"pt": x[..., 0][m], "eta_rel": x[..., 1][m], "phi_rel": x[..., 2][m],
           "multiplicity": m.sum(1).double()}
    for v in range(8):
        out[f"count_flavour_{v}"] = ((k[..., 0] == v) & m).sum(1).double()
    return {n: np.sort(np.asarray(t.double().numpy())) for n, t in out.items()}


def quantile_fn(sample):
    return np.quantile(sample, np.linspace(0.0, 1.0, NQ))


def w1(qa, qb):
    """W1 between two distributions given their quantile functions on a common grid."""
    return float(np.trapz(np.abs(qa - qb), np.linspace(0.0, 1.0, NQ)))


def main():
    R = int(os.environ.get("DIST_RUNS", 8))
    B = int(os.environ.get("DIST_B", 1024))
    steps = 100
    MG._install_stubs()
    torch.set_num_threads(8)
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    cfg = copy.deepcopy(MG.variants()["S3"])
    cfg["model"]["num_timesteps"] = steps
    seed = 9
    model = MG.build_model(cfg, seed)
    N = cfg["data"]["max_num_particles"]
    Q = np.zeros((R, len(OBS), NQ))
    for r in range(R):
        t0 = time.time()
        shape_only = TensorMultiModal(None, torch.zeros(B, N, 3), torch.zeros(B, N, 1, dtype=torch.int64),
                                      torch.ones(B, N, 1, dtype=torch.int64))
        batch = DataCoupling(TensorMultiModal(), shape_only, None)
        torch.manual_seed(9000 + r)
        with torch.no_grad():
            paths, _ = model.predict_step(batch, 0)
        obs = observables(paths.continuous[1], paths.discrete[1], paths.mask[0])
        for i, n in enumerate(OBS):
            Q[r, i] = quantile_fn(obs[n])
        print(f"run {r}: {time.time() - t0:.1f}s  mult mean {obs['multiplicity'].mean():.2f}", flush=True)
    spread = np.zeros((len(OBS), R, R))
    for i in range(len(OBS)):
        for a in range(R):
            for b in range(R):
                spread[i, a, b] = w1(Q[a, i], Q[b, i])
    iu = np.triu_indices(R, 1)
    w1_mean = np.array([spread[i][iu].mean() for i in range(len(OBS))])
    w1_max = np.array([spread[i][iu].max() for i in range(len(OBS))])
    out = os.path.join(MG.OUT, "dist_S3.npz")
    np.savez_compressed(out, quantiles=Q.astype(np.float32), observables=json.dumps(OBS), w1_mean=w1_mean,
                        w1_max=w1_max, runs=R, jets_per_run=B, steps=steps, weight_seed=seed,
                        cfg_json=json.dumps(cfg))
    for n, a, b in zip(OBS, w1_mean, w1_max):
        print(f"{n:18s} run-to-run W1 mean {a:.5f} max {b:.5f}")
    print("wrote", out, os.path.getsize(out) / 1024, "KiB")


if __name__ == "__main__":
    main()
