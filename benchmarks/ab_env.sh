#!/bin/bash
# developer sweep of an environment switch: benchmarks/ab_env.sh <tag> <workload> <VAR> "<values>" [batch]
tag=$1; wl=$2; var=$3; vals=$4; b=${5:-1024}
for v in $vals; do
  env $var=$v timeout -k 10 120 python bench.py --workload $wl --batch $b --steps 5 --warmup 2 --no-train --no-cpu-baseline \
    > gpurun_out/${tag}_${wl}_$v.json 2> gpurun_out/${tag}_${wl}_$v.err || { echo "FAILED $var=$v"; exit 1; }
  python - <<P
import json
d=json.load(open("gpurun_out/${tag}_${wl}_$v.json"))
print("$wl $var=$v B=$b", round(d["value"],1), "jets/s", round(d["ms_per_step"],2), "ms")
P
done
