#!/bin/bash
# The round's kernel measurements on one GPU box, torch-free (tools/prof/prof_bench over the
# C ABI): timings, rocprofv3 kernel stats and PMC passes at the bench size, per dispatch.
#   tools/prof/profile.sh TAG [fast|decode|hc|lds|all]      results under gpurun_out/TAG/
TAG=$1; WHAT=${2:-all}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
B=$R/tools/prof/prof_bench
TD=$R/tests/golden/testdata
cd /tmp
pmc() {  # pmc NAME "COUNTERS" args...
  name=$1; ctr=$2; shift 2
  timeout -s KILL ${PMC_KILL:-240} rocprofv3 --kernel-include-regex "lz4hip" --pmc $ctr --output-format csv -d $O/pmc_$name -o p -- $B $TD "$@" > $O/pmc_$name.log 2>&1 || return 1
}
stats() {  # stats NAME args...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace_$name -o x -- $B $TD "$@" > $O/trace_$name.log 2>&1 || return 1
  DB=$(ls $O/trace_$name/*/x_results.db $O/trace_$name/x_results.db 2>/dev/null | head -1)
  python3 $R/tools/rocprof_summary.py $DB > $O/kernel_stats_$name.txt
}
if [ $WHAT = all ] || [ $WHAT = fast ]; then
  timeout -k 10 200 $B $TD mixed 8 18 fast 5 > $O/fast_mixed.json || exit 1
  timeout -k 10 200 $B $TD text 8 16 fast 3 > $O/fast_text.json || exit 1
  timeout -k 10 200 $B $TD binary 8 18 fast 3 > $O/fast_binary.json || exit 1
  cat $O/fast_*.json
  stats fast mixed 8 18 fast 3 || exit 1
  N=$((8<<30))
  pmc fast_mem "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" mixed 8 18 fast 1 || exit 1
  pmc fast_ea "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" mixed 8 18 fast 1 || exit 1
  pmc fast_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" mixed 8 18 fast 1 || exit 1
  for p in fast_mem fast_ea fast_sq; do echo "# pass $p: prof_bench mixed 8 GiB, 256 KiB blocks, 1 rep"; python3 $R/tools/prof/pmc_rows.py $O/pmc_$p $N; done > $O/pmc_fast_rows.txt
fi
if [ $WHAT = all ] || [ $WHAT = decode ]; then
  timeout -k 10 200 $B $TD text 8 16 decode 5 > $O/decode_text.json || exit 1
  timeout -k 10 200 $B $TD mixed 8 18 decode 5 > $O/decode_mixed.json || exit 1
  cat $O/decode_*.json
  stats decode text 8 16 decode 3 || exit 1
  N=$((8<<30))
  pmc dec_mem "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" text 8 16 decode 1 || exit 1
  pmc dec_ea "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" text 8 16 decode 1 || exit 1
  pmc dec_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" text 8 16 decode 1 || exit 1
  for p in dec_mem dec_ea dec_sq; do echo "# pass $p: prof_bench text 8 GiB, 64 KiB blocks, decode, 1 rep"; python3 $R/tools/prof/pmc_rows.py $O/pmc_$p $N; done > $O/pmc_decode_rows.txt
fi
if [ $WHAT = all ] || [ $WHAT = hc ]; then
  timeout -k 10 200 $B $TD binary 4 20 hc 1 131072 > $O/hc_binary.json || exit 1
  timeout -k 10 200 $B $TD text 4 20 hc 1 131072 > $O/hc_text.json || exit 1
  cat $O/hc_*.json
  # at cfg4's own size (4 GiB = 4096 blocks of 1 MiB, one wave each: 4 waves per SIMD), not at 1 GiB
  PMC_KILL=600
  stats hc binary 4 20 hc 1 131072 || exit 1
  stats hc_text text 4 20 hc 1 131072 || exit 1
  N=$((4<<30))
  pmc hc_mem "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" binary 4 20 hc 1 131072 || exit 1
  pmc hc_ea "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" binary 4 20 hc 1 131072 || exit 1
  pmc hc_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" binary 4 20 hc 1 131072 || exit 1
  pmc hct_mem "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" text 4 20 hc 1 131072 || exit 1
  pmc hct_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" text 4 20 hc 1 131072 || exit 1
  { for p in hc_mem hc_ea hc_sq; do echo "# pass $p: prof_bench binary 4 GiB, 1 MiB blocks (cfg4), HC depth 131072, 1 rep"; python3 $R/tools/prof/pmc_rows.py $O/pmc_$p $N; done
    for p in hct_mem hct_sq; do echo "# pass $p: prof_bench text 4 GiB, 1 MiB blocks, HC depth 131072, 1 rep"; python3 $R/tools/prof/pmc_rows.py $O/pmc_$p $N; done; } > $O/pmc_hc_rows.txt
  PMC_KILL=240
fi
if [ $WHAT = all ] || [ $WHAT = lds ]; then
  # LDS bank conflicts and wave occupancy of the three kernels (north star: "rocprof counters for ... LDS bank
  # conflicts and wave occupancy"); resident waves per CU = SQ_WAVE_CYCLES / SQ_BUSY_CU_CYCLES
  L="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES"
  pmc lds_fast "$L" mixed 8 18 fast 1 || exit 1
  pmc lds_decode "$L" text 8 16 decode 1 || exit 1
  pmc lds_hc "$L" binary 1 20 hc 1 131072 || exit 1
  { echo "# pass lds_fast: prof_bench mixed 8 GiB, 256 KiB blocks, Fast, 1 rep"; python3 $R/tools/prof/pmc_rows.py $O/pmc_lds_fast $((8<<30))
    echo "# pass lds_decode: prof_bench text 8 GiB, 64 KiB blocks, decode, 1 rep"; python3 $R/tools/prof/pmc_rows.py $O/pmc_lds_decode $((8<<30))
    echo "# pass lds_hc: prof_bench binary 1 GiB, 1 MiB blocks, HC depth 131072, 1 rep"; python3 $R/tools/prof/pmc_rows.py $O/pmc_lds_hc $((1<<30)); } > $O/pmc_lds_rows.txt
fi
cd $R
ls $O
