#!/bin/bash
# Sweep of the jet schedule's round size (MMB_GEN_PAIR; default = number of CUs) on the S3 generation line, one box.
for c in 256 0 64 128 192 320 384 512 256; do
  MMB_GEN_PAIR=$c timeout -k 10 120 python bench.py --no-cpu-baseline --no-train --steps 5 --warmup 2 > gpurun_out/pair_$c.json 2> gpurun_out/pair_$c.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/pair_$c.json')); print('chunk $c', round(d['ms_per_step'],3), round(d['value']))"
done
