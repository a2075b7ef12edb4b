"""Developer tool: mmb_transformer_tail at the S4 shapes (M = 1024 x 150 rows), HIP-event time per call, the executed
bf16 MFMA rate and the algorithmic HBM bytes (2 d in, 4 d out per row)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-bridges_amd"))
from mmb_hip import fused  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 153600
    for d, ff in ((64, 256), (128, 256), (128, 512)):
        mk = lambda: torch.nn.TransformerEncoderLayer(d_model=d, nhead=4, dim_feedforward=ff, dropout=0.0, batch_first=True).cuda()
        l0, l1 = mk(), mk()
        pack = fused.pack_transformer_tail(l0, l1)
        a, h = torch.randn(M, d, device="cuda"), torch.randn(M, d, device="cuda")
        for _ in range(3):
            fused.transformer_tail(pack, a, h)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            fused.transformer_tail(pack, a, h)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        macs = d * d + 2 * d * ff + 3 * d * d
        print(f"d={d} ff={ff}: {us:7.1f} us/call  executed bf16 {12.0 * macs * M / us / 1e6:7.1f} TFLOP/s  "
              f"HBM {6 * d * 4 * M / us / 1e3:6.1f} GB/s")


if __name__ == "__main__":
    main()
