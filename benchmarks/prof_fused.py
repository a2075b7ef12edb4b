"""Developer tool: phase breakdown of fused_generate_kernel (block 0, thread 0 core-clock cycles).
Needs a library built with  MMB_EXTRA_FLAGS=-DMMB_FE_PROF python -m mmb_hip.build --force  (touch the
source afterwards and rebuild without the flag for the product library)."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-bridges_amd"))
sys.path.insert(0, ROOT)
import mmb_hip  # noqa: E402
import bench  # noqa: E402

NAMES = ["embed", "pool", "glob1", "glob2", "dvec", "loc1_mma", "loc2_mma", "epilogue", "output_mma", "proj_mma",
         "heads", "step_solver"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    model, config = bench.build_model("S3", 100, "cuda")
    lib = mmb_hip.lib()
    fn = lib.mmb_debug_fused_prof
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    bench.one_generation(model, config, B, "cuda")                # warm-up (packs weights)
    torch.cuda.synchronize()
    fn(None, 1)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    bench.one_generation(model, config, B, "cuda")
    t1.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    fn(buf, 0)
    tot = sum(buf[:len(NAMES)])
    res = {n: buf[i] for i, n in enumerate(NAMES)}
    print(f"B={B} wall {t0.elapsed_time(t1):.1f} ms; block-0 cycles {tot} ({tot / 2.4e6:.1f} ms @2.4GHz)")
    for n, v in sorted(res.items(), key=lambda kv: -kv[1]):
        print(f"  {n:12s} {v:12d}  {100.0 * v / max(tot, 1):5.1f} %")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
