#!/bin/bash
# The round's measurement set on one GPU box: tools/round_profile.sh TAG
#   bench line (default flags), rocprofv3 kernel stats of the same workload, HBM-traffic PMC
#   passes, LDS / occupancy counters.  Results under gpurun_out/TAG/.
TAG=$1
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
( time timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time || exit 1
cat $O/bench.json
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err || exit 1
cd $R
DB=$(ls $O/prof/*/x_results.db $O/prof/x_results.db 2>/dev/null | head -1)
python3 tools/rocprof_summary.py $DB > $O/kernel_stats.txt
head -12 $O/kernel_stats.txt
bash tools/pmc_bench.sh $TAG --size-gib 2 > /dev/null || exit 1
python3 tools/traffic_json.py gpurun_out/pmc_bench_$TAG 2 > $O/traffic.json
cat $O/traffic.json
bash tools/pmc_lds.sh lds_$TAG 2 > $O/lds.txt || exit 1
cat $O/lds.txt
