#!/bin/bash
# developer A/B of MMB_STAGGER (start offset of the second jet of a CU, core clocks): benchmarks/ab_stagger.sh <cycles>...
for t in "$@"; do
  for wl in S3; do
    MMB_STAGGER=$t timeout -k 10 200 python bench.py --workload $wl --steps 5 --warmup 2 --no-train --no-cpu-baseline \
      > gpurun_out/abst_${t}_$wl.json 2> gpurun_out/abst_${t}_$wl.err || { echo "FAILED $t $wl"; exit 1; }
    python - <<P
import json
d=json.load(open("gpurun_out/abst_${t}_$wl.json"))
print("stagger=$t $wl", round(d["value"],1), "jets/s", round(d["ms_per_step"],2), "ms")
P
  done
done
