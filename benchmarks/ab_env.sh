#!/bin/bash
# developer A/B of MMB_GEN_TILES settings at a batch size: benchmarks/ab_env.sh <batch> <tiles>...
B=$1; shift
for t in "$@"; do
  for wl in S3; do
    MMB_GEN_TILES=$t timeout -k 10 200 python bench.py --workload $wl --batch $B --steps 3 --warmup 1 --no-train --no-cpu-baseline \
      > gpurun_out/abenv_${B}_${t}_$wl.json 2> gpurun_out/abenv_${B}_${t}_$wl.err || { echo "FAILED $t $wl"; exit 1; }
    python - <<P
import json
d=json.load(open("gpurun_out/abenv_${B}_${t}_$wl.json"))
print("B=$B tiles=$t $wl", round(d["value"],1), "jets/s", round(d["ms_per_step"],2), "ms")
P
  done
done
