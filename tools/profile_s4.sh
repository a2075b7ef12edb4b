#!/bin/bash
# S4 (transformer) profile set, run on the GPU box through gpurun: micro-benchmarks of the attention / LayerNorm /
# linear kernels, rocprofv3 kernel stats of the S4 training step and of S4 generation, and separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE) over the micro-benchmarks.  usage: tools/profile_s4.sh <tag>
tag=${1:-r2d}
root=$PWD
out=$root/gpurun_out/s4_$tag
mkdir -p $out
timeout -k 10 200 python3 $root/benchmarks/attn_bench.py > $out/attn_bench.txt 2> $out/attn_bench.err || { tail -5 $out/attn_bench.err; exit 1; }
timeout -k 10 200 python3 $root/benchmarks/linear_bench.py > $out/linear_bench.txt 2> $out/linear_bench.err || { tail -5 $out/linear_bench.err; exit 1; }
for path in stream tiled; do
  MMB_LINEAR_PATH=$path timeout -k 10 200 python3 $root/benchmarks/linear_bench.py > $out/linear_bench_$path.txt 2>> $out/linear_bench.err || { tail -5 $out/linear_bench.err; exit 1; }
done
timeout -k 10 300 python3 $root/bench.py --workload S4 --no-cpu-baseline > $out/bench_S4.json 2> $out/bench_S4.err || { tail -5 $out/bench_S4.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/train -o s -- python3 $root/bench.py --workload S4 --train-only --no-cpu-baseline > $out/bench_S4_train_under_rocprof.json 2> $out/train.err || { echo "train stats failed"; tail -5 $out/train.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/gen -o s -- python3 $root/bench.py --workload S4 --no-train --no-cpu-baseline --steps 1 --warmup 1 > $out/bench_S4_gen_under_rocprof.json 2> $out/gen.err || { echo "gen stats failed"; tail -5 $out/gen.err; exit 1; }
for b in attn_bench linear_bench; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $out/${b}_$c -o p -- python3 $root/benchmarks/$b.py > /dev/null 2> $out/${b}_$c.err || { echo "pmc $c $b failed"; tail -5 $out/${b}_$c.err; exit 1; }
  done
done
cd $root
cat $out/attn_bench.txt $out/linear_bench.txt
