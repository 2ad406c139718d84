#!/bin/bash
# LDS bank-conflict and wave-occupancy counters for the default Fast and decode kernels:
#   tools/pmc_lds.sh TAG GIB
set -e
TAG=$1; GIB=$2
R=$PWD
export TMPDIR=/tmp NOPARITY=1
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
CTRS="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES"
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc $CTRS --kernel-include-regex k_fast --output-format csv -d $OUT/fast -o p -- python3 $R/tools/exp_fast4.py text $GIB 16 > $OUT/fast.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc $CTRS --kernel-include-regex k_decode --output-format csv -d $OUT/decode -o p -- python3 $R/tools/exp_decode.py $GIB text > $OUT/decode.log 2>&1
cd $R
N=$(python3 -c "print(int($GIB*2**30)*3)")
{ python3 tools/pmc_summary.py $OUT/fast k_fast $N; python3 tools/pmc_summary.py $OUT/decode k_decode $N; } > $OUT/summary.txt
cat $OUT/summary.txt
