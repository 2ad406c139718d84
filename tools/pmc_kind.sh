KIND=$1
R=$PWD
export TMPDIR=/tmp NOPARITY=1
mkdir -p $R/gpurun_out/pmc_kind_$KIND
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-include-regex "k_fast3" --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $R/gpurun_out/pmc_kind_$KIND/tcp -o p -- python3 $R/tools/exp_fast4.py $KIND 2 18 > $R/gpurun_out/pmc_kind_$KIND/tcp.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-include-regex "k_fast3" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_kind_$KIND/sq -o p -- python3 $R/tools/exp_fast4.py $KIND 2 18 > $R/gpurun_out/pmc_kind_$KIND/sq.log 2>&1
cd $R
for p in tcp sq; do python3 tools/pmc_summary.py gpurun_out/pmc_kind_$KIND/$p k_fast $(python3 -c "print(int(2*2**30)*3)"); done
grep GB/s gpurun_out/pmc_kind_$KIND/tcp.log | tail -1
