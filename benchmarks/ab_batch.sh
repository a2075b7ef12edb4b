#!/bin/bash
# developer A/B over batch sizes (jets per launch): benchmarks/ab_batch.sh <tag> <workload> "<batches>" lib1.so lib2.so ...
tag=$1; wl=$2; batches=$3; shift; shift; shift
for lib in "$@"; do
  for b in $batches; do
    MMB_LIB_NAME=$lib timeout -k 10 120 python bench.py --workload $wl --batch $b --steps 5 --warmup 2 --no-train --no-cpu-baseline \
      > gpurun_out/${tag}_${lib%.so}_${wl}_$b.json 2> gpurun_out/${tag}_${lib%.so}_${wl}_$b.err || { echo "FAILED $lib $wl $b"; exit 1; }
    python - <<P
import json
d=json.load(open("gpurun_out/${tag}_${lib%.so}_${wl}_$b.json"))
print("$lib $wl B=$b", round(d["value"],1), "jets/s", round(d["ms_per_step"],2), "ms")
P
  done
done
