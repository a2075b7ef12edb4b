#!/bin/bash
# developer A/B of MMB_GEN_PAIR settings (round size of the jet schedule): benchmarks/ab_pair.sh <chunk>...
for t in "$@"; do
  for wl in S3 S2; do
    MMB_GEN_PAIR=$t timeout -k 10 200 python bench.py --workload $wl --steps 5 --warmup 2 --no-train --no-cpu-baseline \
      > gpurun_out/abpair_${t}_$wl.json 2> gpurun_out/abpair_${t}_$wl.err || { echo "FAILED $t $wl"; exit 1; }
    python - <<P
import json
d=json.load(open("gpurun_out/abpair_${t}_$wl.json"))
print("pair=$t $wl", round(d["value"],1), "jets/s", round(d["ms_per_step"],2), "ms")
P
  done
done
