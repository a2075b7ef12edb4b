"""Per-shape time of mmb_linear_cat in one S4 generation (HIP events around every C-ABI call)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.join(os.getcwd(), "multimodal-bridges_amd"))
sys.argv = ["bench.py"]
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import mmb_hip
dev = torch.device("cuda:0")
model, config = bench.build_model("S4", 0, dev)
model.eval()
B = 1024
bench.one_generation(model, config, B, dev)
torch.cuda.synchronize()
mmb_hip.PROFILE = []
bench.one_generation(model, config, B, dev)
torch.cuda.synchronize()
rec, mmb_hip.PROFILE = mmb_hip.PROFILE, None
agg = collections.defaultdict(lambda: [0.0, 0])
for name, meta, e0, e1 in rec:
    key = (name, meta["M"], meta["K"], meta["Nout"]) if meta else (name,)
    a = agg[key]; a[0] += e0.elapsed_time(e1); a[1] += 1
tot = sum(v[0] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{v[0]:9.2f} ms {v[1]:6d} calls {v[0] / v[1] * 1e3:8.1f} us  {k}")
print("total", tot)
