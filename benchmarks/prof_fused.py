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
         "heads", "step_solver", "x1_issue", "x2_poolpart", "x3_combine"]


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
    if hasattr(lib, "mmb_debug_fused_map"):
        nb = min(B, 512)
        mp = (ctypes.c_uint * (nb * 4))()
        lib.mmb_debug_fused_map.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.mmb_debug_fused_map(mp, nb * 4)
        # HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13]; XCC_ID[3:0] in the high half
        where = {}
        for blk in range(nb):
            ws = [mp[blk * 4 + w] for w in range(4)]
            simd = [(v >> 4) & 3 for v in ws]
            cu = ((ws[0] >> 8) & 15, (ws[0] >> 12) & 1, (ws[0] >> 13) & 7, (ws[0] >> 16) & 15)
            where.setdefault(cu, []).append((blk, simd))
        shown = 0
        for cu, lst in sorted(where.items()):
            if shown < 12:
                print("CU(cu,sh,se,xcc)=", cu, lst)
                shown += 1
        import collections
        print("blocks per CU histogram:", collections.Counter(len(v) for v in where.values()))
        print("wave->simd patterns:", collections.Counter(tuple(sd) for v in where.values() for _, sd in v).most_common(6))
        d = collections.Counter()
        for v in where.values():
            if len(v) == 2:
                d[abs(v[0][0] - v[1][0])] += 1
        print("block-index distance of co-resident pairs:", d.most_common(8))


if __name__ == "__main__":
    main()
