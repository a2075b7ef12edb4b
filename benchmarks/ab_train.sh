#!/bin/bash
# developer A/B of MMB_TRAIN_TILES settings on the training step: benchmarks/ab_train.sh <workload> <tiles>...
wl=$1; shift
for t in "$@"; do
  MMB_TRAIN_TILES=$t timeout -k 10 200 python bench.py --workload $wl --train-only > gpurun_out/abtrain_${t}_$wl.json 2> gpurun_out/abtrain_${t}_$wl.err || { echo "FAILED $t $wl"; exit 1; }
  python - <<P
import json
d=json.load(open("gpurun_out/abtrain_${t}_$wl.json"))
print("train tiles=$t $wl", round(d["value"],1), "jets/s", round(d["ms_per_step"],3), "ms", "fwd_loss_bwd", round(d.get("fwd_loss_bwd_ms", 0), 3))
P
done
