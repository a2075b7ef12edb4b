"""Host-side cost of a training step: run at a tiny batch (GPU time negligible) under cProfile."""
import os, sys, time, cProfile, pstats, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-bridges_amd")); sys.path.insert(0, ROOT)
import bench
from pipeline.experiment import FlatGrads
from datamodules.synthetic import synthetic_target
from tensorclass import TensorMultiModal
from datamodules.datasets import DataCoupling

wl = os.environ.get("WL", "S3"); B = int(os.environ.get("B", 16))
dev = torch.device("cuda")
model, config = bench.build_model(wl, 100, dev)
model.train()
model.use_fused_training = os.environ.get("FUSED_TRAIN", "1") == "1"
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

for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print(f"B={B}: wall {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
