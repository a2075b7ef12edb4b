#!/bin/bash
# A/B on one box: idle tiles of the reverse kernel leave their dZ rows unwritten (default) against writing every row
# (MMB_BWD_DEAD_ROWS=0).  usage (through gpurun): bash benchmarks/exp_dz.sh  ->  gpurun_out/exp_dz_<workload>_<0|1>.json
for wl in S3 S2; do
for t in 0 1 0 1; do
  MMB_BWD_DEAD_ROWS=$t timeout -k 10 120 python bench.py --workload $wl --no-cpu-baseline --steps 1 --warmup 1 > gpurun_out/exp_dz_${wl}_$t.json 2> gpurun_out/exp_dz_${wl}_$t.err || exit 1
  python - <<P
import json
d=json.load(open("gpurun_out/exp_dz_${wl}_$t.json"))
t=d["training"]
print("$wl dead_rows=$t", round(t["ms_per_step"],4), round(t["value"]), {k:v for k,v in list(t["entry_point_ms_per_step"].items())[:4]})
P
done
done
