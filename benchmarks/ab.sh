#!/bin/bash
# developer A/B of library builds on one box: benchmarks/ab.sh <tag> lib1.so lib2.so ...  (S3 and S2 generation only)
tag=$1; shift
for lib in "$@"; do
  for wl in S3 S2; do
    MMB_LIB_NAME=$lib timeout -k 10 120 python bench.py --workload $wl --steps 5 --warmup 2 --no-train --no-cpu-baseline \
      > gpurun_out/${tag}_${lib%.so}_$wl.json 2> gpurun_out/${tag}_${lib%.so}_$wl.err || { echo "FAILED $lib $wl"; exit 1; }
    python - <<P
import json
d=json.load(open("gpurun_out/${tag}_${lib%.so}_$wl.json"))
print("$lib $wl", round(d["value"],1), "jets/s", round(d["ms_per_step"],2), "ms  frac", round(d["roofline"]["frac"],4))
P
  done
done
