"""Micro-benchmark of mmb_linear_cat on the S3/S2 layer shapes (B=1024, N=128) vs rocBLAS via torch."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-bridges_amd"))
from mmb_hip import ops

B, N = int(os.environ.get("B", 1024)), 128
M = B * N
dev = "cuda"
shapes = [("branch fc_loc1", [(32, 2), (64, 0), (16, 1)], 64, "lrelu"),
          ("branch fc_loc2", [(64, 0)], 64, None),
          ("fused fc_loc1", [(32, 2), (128, 0), (32, 1)], 128, "lrelu"),
          ("fused fc_loc2", [(128, 0)], 128, None),
          ("proj local0", [(32, 2), (16, 0)], 64, "gelu"),
          ("head0", [(32, 2), (64, 0), (64, 0)], 64, "gelu"),
          ("head2", [(64, 0)], 8, None),
          ("S2 fc_loc1", [(128, 2), (128, 0), (16, 1)], 128, "lrelu")]
maskf = (torch.rand(M, device=dev) > 0.5).float()
for name, segs, Nout, act in shapes:
    ts = []
    for w, kind in segs:
        ts.append((torch.randn(M if kind == 0 else B, w, device=dev), kind))
    K = sum(w for w, _ in segs)
    W = torch.randn(Nout, K, device=dev) / K ** 0.5
    b = torch.randn(Nout, device=dev)
    out = torch.empty(M, Nout, device=dev)
    for _ in range(3):
        ops.linear_cat(ts, W, b, N, maskf=maskf, act=act, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ops.linear_cat(ts, W, b, N, maskf=maskf, act=act, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    flops = 2.0 * M * K * Nout
    X = torch.randn(M, K, device=dev)
    for _ in range(3):
        torch.matmul(X, W.t())
    e0.record()
    for _ in range(reps):
        torch.matmul(X, W.t())
    e1.record(); torch.cuda.synchronize()
    us2 = e0.elapsed_time(e1) / reps * 1e3
    byts = 4.0 * (M * sum(w for w, k in segs if k == 0) + M * Nout)
    print(f"{name:16s} K={K:4d} N={Nout:4d}  mmb {us:8.1f} us {flops/us/1e6:7.1f} TF {byts/us/1e3:7.1f} GB/s | rocBLAS {us2:8.1f} us {flops/us2/1e6:7.1f} TF")
