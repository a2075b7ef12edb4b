"""Micro-benchmark of the weight-gradient kernels on the S3 layer shapes (M = 1024*128 rows)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-bridges_amd"))
from mmb_hip import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    M = 1024 * 128
    for K, N in ((64, 64), (128, 128), (64, 128), (128, 64), (16, 64), (64, 3), (192, 64)):
        X = torch.randn(M, K, device="cuda")
        dZ = torch.randn(M, N, device="cuda")
        R = int(os.environ.get("GW_REP", "8"))
        dWr = torch.zeros(R, N * K, device="cuda")
        dbr = torch.zeros(R, N, device="cuda")
        dW, db = dWr[0].view(N, K), dbr[0]
        us = timeit(lambda: ops.grad_weight_rows(X, dZ, dW, db, replicas=R, rep_stride=dWr.stride(0),
                                                 db_stride=dbr.stride(0)))
        us_old = timeit(lambda: ops.grad_weight([(X, ops.SEG_ROW)], 128, None, dZ, dW, db))
        us_t = timeit(lambda: torch.matmul(dZ.t(), X))
        gb = M * (K + N) * 4 / 1e9
        print(f"K={K:4d} N={N:4d}: rows {us:8.1f} us ({gb / us * 1e6:7.0f} GB/s, {2 * M * K * N / us / 1e6:6.1f} TF)"
              f" | old {us_old:8.1f} us | torch(rocBLAS) {us_t:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
