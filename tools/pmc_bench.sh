#!/bin/bash
# HBM-traffic PMC passes over the bench workload (one pass per run; see MI355X_MICROARCH.md
# "rocprofv3 PMC slots"):  tools/pmc_bench.sh TAG [bench args]
# Writes gpurun_out/pmc_bench_TAG/summary.txt with per-kernel sums.
set -e
TAG=$1; shift
R=$PWD
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_bench_$TAG
mkdir -p $OUT
run() { name=$1; shift; timeout -s KILL 300 rocprofv3 --kernel-include-regex "lz4hip" --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $BENCH_ARGS > $OUT/$name.log 2>&1; }
BENCH_ARGS="$*"
cd /tmp
run ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R
for k in k_fast k_decode k_xxh32_blocks; do
  for p in ea fetch write; do python3 tools/pmc_summary.py $OUT/$p $k; done
done > $OUT/summary.txt
cat $OUT/summary.txt
