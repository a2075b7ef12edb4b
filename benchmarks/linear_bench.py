"""Micro-benchmark of mmb_linear_cat at the shapes of the S4 transformer encoder (M = 1024 jets x 150 particles):
us per call, achieved fp32 TFLOP/s and the HBM rate of the algorithmic bytes (input + output (+ residual))."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multimodal-bridges_amd"))
from mmb_hip import ops  # noqa: E402

SHAPES = [  # (K, Nout, residual, act)
    (64, 192, False, None), (64, 64, True, None), (64, 256, False, "relu"), (256, 64, True, None),
    (128, 384, False, None), (128, 128, True, None), (128, 256, False, "relu"), (256, 128, True, None),
    (192, 128, False, None), (256, 256, False, "gelu"),
]


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
    M, N = 1024 * 150, 150
    g = torch.Generator().manual_seed(0)
    tot = 0.0
    only = os.environ.get("MMB_BENCH_SHAPE")                     # "K,N": one shape (for counter passes)
    for K, Nout, res, act in SHAPES:
        if only and only != f"{K},{Nout}":
            continue
        x = torch.randn(M, K, generator=g).cuda()
        W = (torch.randn(Nout, K, generator=g) / K ** 0.5).cuda()
        b = torch.randn(Nout, generator=g).cuda()
        r = torch.randn(M, Nout, generator=g).cuda() if res else None
        out = torch.empty(M, Nout, device="cuda")
        fn = lambda: ops.linear_cat([(x, ops.SEG_ROW)], W, b, N, act=act, residual=r, out=out)
        y = fn()
        ref = torch.nn.functional.linear(x.double(), W.double(), b.double())
        if act == "relu":
            ref = ref.relu()
        if act == "gelu":
            ref = torch.nn.functional.gelu(ref)
        if res:
            ref = ref + r.double()
        err = float((y.double() - ref).abs().max())
        us = timed(fn)
        tot += us
        by = 4.0 * M * (K + Nout * (2 if res else 1))
        print(f"K={K:4d} N={Nout:4d} res={int(res)} act={act}: {us:8.1f} us  {2.0 * M * K * Nout / us * 1e-6:6.1f} TFLOP/s  "
              f"{by / us * 1e-6:6.2f} TB/s  max err {err:.1e}", flush=True)
    print(f"sum {tot:.0f} us")
    # weight gradients dW = dZ^T X of the same layers (mmb_grad_weight_rows, 8 accumulator replicas as the reverse pass)
    R = 8
    for gpath in ("f32", "x6"):
        os.environ["MMB_GW_PATH"] = gpath
        tot = 0.0
        for K, Nout, res, act in SHAPES:
            if only and only != f"{K},{Nout}":
                continue
            x = torch.randn(M, K, generator=g).cuda()
            dz = torch.randn(M, Nout, generator=g).cuda()
            dwr = torch.zeros(R, Nout * K, device="cuda")
            dbr = torch.zeros(R, Nout, device="cuda")
            fn = lambda: ops.grad_weight_rows(x, dz, dwr[0].view(Nout, K), dbr[0], replicas=R, rep_stride=dwr.stride(0),
                                              db_stride=dbr.stride(0))
            fn()
            ref = dz[:20000].double().t() @ x[:20000].double()
            dwr.zero_(); dbr.zero_()
            ops.grad_weight_rows(x[:20000], dz[:20000], dwr[0].view(Nout, K), dbr[0], replicas=R, rep_stride=dwr.stride(0),
                                 db_stride=dbr.stride(0))
            got = dwr.sum(0).view(Nout, K).double()
            err = float((got - ref).abs().max() / ref.abs().max())
            us = timed(fn)
            tot += us
            print(f"gradw[{gpath}] K={K:4d} N={Nout:4d}: {us:8.1f} us  {2.0 * M * K * Nout / us * 1e-6:6.1f} TFLOP/s  "
                  f"{4.0 * M * (K + Nout) / us * 1e-6:5.2f} TB/s  rel err {err:.1e}", flush=True)
        print(f"gradw[{gpath}] sum {tot:.0f} us")


if __name__ == "__main__":
    main()
