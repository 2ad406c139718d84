#!/bin/bash
# PMC passes over the Fast kernel: tools/pmc_fast.sh TAG KIND GIB BSLOG   (env LZ4HIP_FAST selects the kernel)
set -e
TAG=$1; KIND=$2; GIB=$3; BSL=$4
R=$PWD
export TMPDIR=/tmp NOPARITY=1
run() { name=$1; shift; mkdir -p $R/gpurun_out/pmc_$TAG; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$TAG/$name -o p -- python3 $R/tools/exp_fast4.py $KIND $GIB $BSL > $R/gpurun_out/pmc_$TAG/$name.log 2>&1; }
cd /tmp
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run ea TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
cd $R
for p in tcp ea sq; do python3 tools/pmc_summary.py gpurun_out/pmc_$TAG/$p k_fast $(python3 -c "print(int($GIB*2**30)*3)"); done > gpurun_out/pmc_$TAG/summary.txt
cat gpurun_out/pmc_$TAG/summary.txt
