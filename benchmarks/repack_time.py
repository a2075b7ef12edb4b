import os, sys, time, torch
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "multimodal-bridges_amd")); sys.path.insert(0, ROOT)
import bench
from mmb_hip import fused
for wl in ("S3", "S2"):
    model, config = bench.build_model(wl, 100, torch.device("cuda"))
    pack = model._fused_pack()
    fused.packed_bwd(pack, 1)
    cnt = [0]
    def ver():
        cnt[0] += 1; return 1000 + cnt[0]
    def go():
        fused.repack(pack, pack.get("version", 0) + 1) if False else None
    # time the pieces the training step runs after an optimizer step
    torch.cuda.synchronize()
    for name, fn in (("repack", lambda: fused.repack(pack, ver())), ("packed_bwd", lambda: fused.packed_bwd(pack, ver()))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); print(wl, name, "%.1f us" % ((time.perf_counter() - t0) / 20 * 1e6))
