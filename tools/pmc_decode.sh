#!/bin/bash
# PMC passes over the decode kernel: tools/pmc_decode.sh TAG GIB   (env LZ4HIP_DECODE selects the kernel)
set -e
TAG=$1; GIB=$2
R=$PWD
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --kernel-include-regex "lz4hip" --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 $R/tools/exp_decode.py $GIB text > $OUT/$name.log 2>&1; }
cd /tmp
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
cd $R
for p in tcp sq sq2; do python3 tools/pmc_summary.py $OUT/$p k_decode $(python3 -c "print(int($GIB*2**30)*3)"); done > $OUT/summary.txt
cat $OUT/summary.txt
