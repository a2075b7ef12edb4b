#!/usr/bin/env python3
"""Distribution pins (SURVEY.md 8c golden 9): run the REFERENCE generator (imported from
/root/reference, same harness as tools/make_golden.py) R times with different seeds on the
S3 model with the name-synthesised weight set, and store per-observable quantile functions
of every run plus the run-to-run Wasserstein-1 spread.  Observables are those of the
reference's generation callback minus the fastjet ones: particle pt/eta_rel/phi_rel on
mask==1 slots, multiplicity, per-flavour counts per jet.

`python tools/make_golden_dist.py trained`: the same with a BRIEFLY TRAINED weight set -- 300 steps of the reference's
own training_step on CPU on a synthetic AOJ-shaped target whose kinematics depend on the flavour; the trained
state_dict is stored in the fixture (tests/golden/dist_S3_trained.npz) so that the HIP generator runs the same weights.

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


def structured_target(B, N, seed):
    """Synthetic AOJ-shaped target with STRUCTURE for the brief training: the flavour mix of AOJ,
    kinematics that depend on the flavour (so that the learnt jump rates and drifts are far from their random-init
    values): x0 skewed and shifted by token, x1/x2 narrow."""
    from oracle.synth import aoj_like_batch
    b = aoj_like_batch(B, N, seed=seed)
    g = torch.Generator().manual_seed(seed + 77)
    k, m = b["discrete"], b["mask"]
    z = torch.randn(B, N, 3, generator=g)
    shift = torch.tensor([0.6, -0.2, -0.5, 0.3, 1.0, 1.0, -1.0, -1.0])[k[..., 0]]
    x = torch.stack([0.55 * z[..., 0] + 0.25 * z[..., 0] ** 2 - 0.25 + shift, 0.45 * z[..., 1], 0.45 * z[..., 2] + 0.2 * shift], -1)
    return {"continuous": x * m, "discrete": k, "mask": m}


def train_reference(model, cfg, steps, B, N):
    """A few hundred steps of the reference's own training_step (AdamW, lr 1e-3, clip 1.0: configs.py:71) on CPU."""
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=5e-5)
    t0 = time.time()
    for it in range(steps):
        tb = structured_target(B, N, 5000 + it)
        batch = DataCoupling(TensorMultiModal(), TensorMultiModal(None, tb["continuous"], tb["discrete"], tb["mask"]), None)
        torch.manual_seed(7000 + it)
        out = model.training_step(batch, it)
        opt.zero_grad(set_to_none=True)
        out["loss"].backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        if it % 25 == 0 or it == steps - 1:
            print(f"  train {it}: loss {float(out['loss']):.4f}  ({time.time() - t0:.0f}s)", flush=True)
    return model.eval()


def main():
    R = int(os.environ.get("DIST_RUNS", 8))
    B = int(os.environ.get("DIST_B", 1024))
    trained = len(sys.argv) > 1 and sys.argv[1] == "trained"
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
    extra = {}
    if trained:
        tsteps, tB = int(os.environ.get("DIST_TRAIN_STEPS", 300)), int(os.environ.get("DIST_TRAIN_B", 128))
        model = train_reference(model, cfg, tsteps, tB, N)
        extra = {"param:" + n: p.detach().numpy().copy() for n, p in model.state_dict().items()}
        extra["train_steps"], extra["train_batch"] = tsteps, tB
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
    out = os.path.join(MG.OUT, "dist_S3_trained.npz" if trained else "dist_S3.npz")
    np.savez_compressed(out, quantiles=Q.astype(np.float32), observables=json.dumps(OBS), w1_mean=w1_mean,
                        w1_max=w1_max, runs=R, jets_per_run=B, steps=steps, weight_seed=seed,
                        cfg_json=json.dumps(cfg), **extra)
    for n, a, b in zip(OBS, w1_mean, w1_max):
        print(f"{n:18s} run-to-run W1 mean {a:.5f} max {b:.5f}")
    print("wrote", out, os.path.getsize(out) / 1024, "KiB")


if __name__ == "__main__":
    main()
