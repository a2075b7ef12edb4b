"""Summarise the SQ / instruction-cache --pmc passes and the kernel stats of tools/profile_round.sh <tag>
(gpurun_out/prof_<tag>/) for fused_generate_kernel into profiles/<round>/<tag>_sq_counters.json, and copy the kernel
stats and the bench line taken under rocprofv3 beside it.
usage: python tools/sq_summary.py <tag> [round dir, default r3]"""
import collections, csv, glob, json, shutil, sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r3"
base = f"gpurun_out/prof_{tag}"
out = {}
for sub in ("sq", "sq2", "sq3", "ic"):
    acc, ids = collections.defaultdict(float), set()
    for f in glob.glob(f"{base}/{sub}_S3/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fused_generate_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                ids.add(r["Dispatch_Id"])
    n = len(ids)
    out[sub] = {"launches": n, "per_launch": {k: v / n for k, v in sorted(acc.items())}}
ks = next(r for r in csv.DictReader(open(f"{base}/stats/s_kernel_stats.csv")) if "fused_generate_kernel" in r["Name"])
ms = float(ks["AverageNs"]) / 1e6
g = out["sq2"]["per_launch"]["GRBM_GUI_ACTIVE"] / 8
sq, sq2, sq3 = (out[k]["per_launch"] for k in ("sq", "sq2", "sq3"))
out["derived"] = {
    "kernel_ms_rocprofv3_stats": ms, "calls": int(ks["Calls"]),
    "shader_cycles_per_launch (GRBM_GUI_ACTIVE / 8 XCDs)": g,
    "effective_clock_GHz": g / (ms * 1e6),
    "mfma_busy_frac_of_simd_cycles (SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs))": sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 1024),
    "valu_insts_per_mfma": sq["SQ_INSTS_VALU"] / sq2["SQ_INSTS_MFMA"],
    "mfma_valu_coexec_frac_of_mfma_busy": sq3["SQ_VALU_MFMA_COEXEC_CYCLES"] / sq3["SQ_VALU_MFMA_BUSY_CYCLES"],
}
json.dump(out, open(f"profiles/{rnd}/{tag}_sq_counters.json", "w"), indent=1)
shutil.copy(f"{base}/stats/s_kernel_stats.csv", f"profiles/{rnd}/{tag}_bench_S3_kernel_stats.csv")
shutil.copy(f"{base}/bench_S3_under_rocprof.json", f"profiles/{rnd}/{tag}_bench_S3_under_rocprof.json")
print(json.dumps(out["derived"], indent=1))
