#!/bin/bash
# developer: MFMA / VALU co-execution counters of the generation bench: tools/dev/coexec.sh <tag> <workload> [batch]
tag=$1; wl=$2; b=${3:-1024}
root=$PWD
out=$root/gpurun_out/coexec_${tag}_${wl}_$b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $out/p1 -o p -- python3 $root/bench.py --workload $wl --batch $b --no-cpu-baseline --no-train --steps 1 --warmup 1 > $out/p1.json 2> $out/p1.err || { echo "pass failed"; tail -5 $out/p1.err; exit 1; }
python3 - <<P
import csv,glob,collections
tot=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob("$out/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "generate" in r["Kernel_Name"]:
            tot[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
for k in sorted(tot): print("  %-30s %.4g per dispatch"%(k, tot[k]/max(n[k],1)))
P
