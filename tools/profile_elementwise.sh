#!/bin/bash
# HBM-bound kernels (K-A/K-B/K-C/K-F, mask_prepare) at P = 8.4 M particles under rocprofv3: kernel durations and, in
# separate passes, FETCH_SIZE / WRITE_SIZE.  usage (GPU box): tools/profile_elementwise.sh <tag>
tag=${1:-r2}
root=$PWD
out=$root/gpurun_out/elem_$tag
mkdir -p $out
timeout -k 10 200 python3 $root/benchmarks/bench_elementwise.py > $out/elementwise.txt 2> $out/elementwise.err || { tail -5 $out/elementwise.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $root/benchmarks/bench_elementwise.py > /dev/null 2> $out/stats.err || { echo stats failed; tail -5 $out/stats.err; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $out/$c -o p -- python3 $root/benchmarks/bench_elementwise.py > /dev/null 2> $out/$c.err || { echo "pmc $c failed"; tail -5 $out/$c.err; exit 1; }
done
cat $out/elementwise.txt | grep -v "^{" | grep -v amdgpu.ids
