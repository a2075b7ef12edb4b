#!/bin/bash
# A/B on one box: padded slots of working tiles leave their tape rows unwritten in the training forward
# (default) against storing them (MMB_FWD_DEAD_LANES=0).
for wl in S3 S2; do
for t in 0 1 0 1; do
  MMB_FWD_DEAD_LANES=$t timeout -k 10 120 python bench.py --workload $wl --no-cpu-baseline --steps 1 --warmup 1 > gpurun_out/exp_fwd_${wl}_$t.json 2> gpurun_out/exp_fwd_${wl}_$t.err || exit 1
  python - <<P
import json
d=json.load(open("gpurun_out/exp_fwd_${wl}_$t.json"))
t=d["training"]
print("$wl fwd_dead_lanes=$t", round(t["ms_per_step"],4), round(t["value"]), {k:v for k,v in list(t["entry_point_ms_per_step"].items())[:3]})
P
done
done
