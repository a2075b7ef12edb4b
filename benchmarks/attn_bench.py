"""Micro-benchmark of mmb_attention_fwd / mmb_attention_bwd at the S4 shapes (B jets x 150 particles, 4 heads of 16
or 32): matrix-core kernels against the one-query-per-thread kernels (MMB_ATTN_PATH=valu).  Prints us per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multimodal-bridges_amd"))
from mmb_hip import ops  # noqa: E402


def timed(fn, n=20):
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
    B, N = 1024, 150
    g = torch.Generator().manual_seed(0)
    nreal = torch.randint(20, 151, (B,), generator=g)
    mask = (torch.arange(N)[None, :] < nreal[:, None]).float().cuda()
    mi = ops.mask_prepare(mask)
    for H, nh in ((64, 4), (128, 4)):
        qkv = torch.randn(B * N, 3 * H, generator=g).cuda()
        dout = torch.randn(B * N, H, generator=g).cuda()
        res = {}
        for path in ("valu", "mfma"):
            os.environ["MMB_ATTN_PATH"] = path
            out, lse = ops.attention_fwd(qkv, mi, nh, want_lse=True)
            dq = ops.attention_bwd(qkv, mi, nh, out, dout, lse)
            res[path] = (out, dq)
            tf = timed(lambda: ops.attention_fwd(qkv, mi, nh, want_lse=True))
            tb = timed(lambda: ops.attention_bwd(qkv, mi, nh, out, dout, lse))
            print(f"H={H} nh={nh} {path}: fwd {tf:8.1f} us  bwd {tb:8.1f} us", flush=True)
        do = float((res["valu"][0] - res["mfma"][0]).abs().max())
        dg = float((res["valu"][1] - res["mfma"][1]).abs().max())
        print(f"   max |valu - mfma|: out {do:.2e}  dqkv {dg:.2e} (scale {float(res['valu'][1].abs().max()):.2e})")


def layernorm():
    M = 1024 * 150
    for D in (64, 128):
        x = torch.randn(M, D, device="cuda")
        gm, bt = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
        dy, dx = torch.randn(M, D, device="cuda"), torch.empty(M, D, device="cuda")
        dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
        y, stats = ops.layernorm_fwd(x, gm, bt, 1e-5, want_stats=True)
        tf = timed(lambda: ops.layernorm_fwd(x, gm, bt, 1e-5, want_stats=True))
        tb = timed(lambda: ops.layernorm_bwd(x, gm, stats, dy, dx, False, dg, db))
        print(f"layernorm D={D}: fwd {tf:7.1f} us ({8.0 * M * D / tf * 1e-6:5.2f} TB/s)  "
              f"bwd {tb:7.1f} us ({12.0 * M * D / tb * 1e-6:5.2f} TB/s)", flush=True)


if __name__ == "__main__":
    layernorm()
    main()
