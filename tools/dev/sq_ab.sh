#!/bin/bash
# developer A/B of SQ counters of the generation-only bench: tools/dev/sq_ab.sh <tag> <workload> lib1.so lib2.so ...
tag=$1; wl=$2; shift; shift
root=$PWD
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  out=$root/gpurun_out/sq_${tag}_${lib%.so}_$wl
  mkdir -p $out
  export MMB_LIB_NAME=$lib
  timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/p1 -o p -- python3 $root/bench.py --workload $wl --no-cpu-baseline --no-train --steps 1 --warmup 1 > $out/p1.json 2> $out/p1.err || { echo "pass1 failed $lib"; tail -3 $out/p1.err; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $out/p2 -o p -- python3 $root/bench.py --workload $wl --no-cpu-baseline --no-train --steps 1 --warmup 1 > $out/p2.json 2> $out/p2.err || { echo "pass2 failed $lib"; tail -3 $out/p2.err; exit 1; }
  python3 - <<P
import csv,glob,collections
tot=collections.defaultdict(float); n=0
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "generate" in r["Kernel_Name"]:
            tot[r["Counter_Name"]]+=float(r["Counter_Value"])
disp=len([1 for f in glob.glob("$out/p1/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "generate" in r["Kernel_Name"] and r["Counter_Name"]=="SQ_WAVE_CYCLES"])
print("$lib $wl dispatches",disp)
for k in sorted(tot): print("  %-28s %.4g per dispatch"%(k, tot[k]/max(disp,1)))
w=tot["SQ_WAVE_CYCLES"]
if w: print("  issue %.3f wait_inst %.3f wait_any %.3f | valu/mfma %.2f | mfma_busy/busy %.3f"%(tot["SQ_ACTIVE_INST_ANY"]/w, tot["SQ_WAIT_INST_ANY"]/w, tot["SQ_WAIT_ANY"]/w, tot["SQ_INSTS_VALU"]/max(tot["SQ_INSTS_MFMA"],1), tot["SQ_VALU_MFMA_BUSY_CYCLES"]/max(tot["SQ_BUSY_CYCLES"],1)))
P
done
