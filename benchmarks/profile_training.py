"""Where does a training step go?  Per-C-ABI-call GPU time (HIP events) vs wall time."""
import os, sys, time, json, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-bridges_amd")); sys.path.insert(0, ROOT)
import bench, mmb_hip
from pipeline.experiment import FlatGrads
from datamodules.synthetic import synthetic_target
from tensorclass import TensorMultiModal
from datamodules.datasets import DataCoupling

wl = os.environ.get("WL", "S3"); B = int(os.environ.get("B", 1024))
dev = torch.device("cuda")
model, config = bench.build_model(wl, 100, dev)
model.train()
tgt = synthetic_target(B, 128, config.data.target_path, seed=1, Dc=max(config.data.dim_continuous, 1), V=8,
                       modality=config.data.modality, device=dev)
opt = torch.optim.AdamW(model.parameters(), lr=2e-4)
flat = FlatGrads(model.parameters())

def step():
    flat.zero()
    out = model.training_step(DataCoupling(TensorMultiModal(), TensorMultiModal(None, tgt.continuous, tgt.discrete, tgt.mask), None), 0)
    out["loss"].backward()
    flat.clip_(1.0)
    opt.step()

for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10
mmb_hip.PROFILE = []
step(); torch.cuda.synchronize()
rec, mmb_hip.PROFILE = mmb_hip.PROFILE, None
agg = collections.OrderedDict()
for name, meta, e0, e1 in rec:
    key = name
    if name == "mmb_linear_cat": key += "[M=%s]" % ("BN" if meta["M"] > 4096 else "B")
    a = agg.setdefault(key, [0.0, 0]); a[0] += e0.elapsed_time(e1); a[1] += 1
tot = sum(a[0] for a in agg.values())
print(f"wall {wall*1e3:.2f} ms/step  | sum of mmb kernel time {tot:.2f} ms over {len(rec)} launches")
for k, (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:34s} {ms:8.3f} ms  {n:4d} launches")
