#!/bin/bash
# Round profile set (run on the GPU box through gpurun): rocprofv3 kernel stats of the default bench line, separate
# --pmc passes (FETCH_SIZE / WRITE_SIZE; SQ; instruction cache) of the generation-only bench.  usage: tools/profile_round.sh <tag>
# Every rocprofv3 run has the program itself after `--` and a time limit of its own.
tag=${1:-r2}
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_S3_under_rocprof.json 2> $out/stats.err || { echo "stats pass failed"; tail -5 $out/stats.err; exit 1; }
for wl in S3 S2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    kind=$(echo $c | cut -d_ -f1 | tr A-Z a-z)
    timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $out/${kind}_$wl -o p -- python3 $root/bench.py --workload $wl --no-cpu-baseline --steps 1 --warmup 1 > $out/pmc_${kind}_$wl.json 2> $out/pmc_${kind}_$wl.err || { echo "pmc $c $wl failed"; tail -5 $out/pmc_${kind}_$wl.err; exit 1; }
  done
done
timeout -s KILL 240 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/sq_S3 -o p -- python3 $root/bench.py --no-cpu-baseline --no-train --steps 1 --warmup 1 > $out/pmc_sq_S3.json 2> $out/pmc_sq_S3.err || { echo "sq pass failed"; tail -5 $out/pmc_sq_S3.err; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $out/sq2_S3 -o p -- python3 $root/bench.py --no-cpu-baseline --no-train --steps 1 --warmup 1 > $out/pmc_sq2_S3.json 2> $out/pmc_sq2_S3.err || { echo "sq2 pass failed"; tail -5 $out/pmc_sq2_S3.err; }
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $out/sq3_S3 -o p -- python3 $root/bench.py --no-cpu-baseline --no-train --steps 1 --warmup 1 > $out/pmc_sq3_S3.json 2> $out/pmc_sq3_S3.err || { echo "sq3 pass failed"; tail -5 $out/pmc_sq3_S3.err; }
timeout -s KILL 240 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $out/ic_S3 -o p -- python3 $root/bench.py --no-cpu-baseline --no-train --steps 1 --warmup 1 > $out/pmc_ic_S3.json 2> $out/pmc_ic_S3.err || { echo "icache pass failed"; tail -5 $out/pmc_ic_S3.err; }
find $out -name "*.csv" | head -40
