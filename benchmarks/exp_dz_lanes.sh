#!/bin/bash
# A/B on one box: padded lanes of WORKING tiles drop their dZ stores too (MMB_BWD_DEAD_LANES=1) against idle tiles only.
for wl in S3 S2; do
for t in 0 1 0 1; do
  MMB_BWD_DEAD_LANES=$t timeout -k 10 120 python bench.py --workload $wl --no-cpu-baseline --steps 1 --warmup 1 > gpurun_out/exp_lanes_${wl}_$t.json 2> gpurun_out/exp_lanes_${wl}_$t.err || exit 1
  python - <<P
import json
d=json.load(open("gpurun_out/exp_lanes_${wl}_$t.json"))
t=d["training"]
print("$wl dead_lanes=$t", round(t["ms_per_step"],4), round(t["value"]), {k:v for k,v in list(t["entry_point_ms_per_step"].items())[:3]})
P
done
done
