# tools/prof/pmc_one.sh NAME "COUNTERS" NBYTES prof_bench-args...: one rocprofv3 --pmc pass over prof_bench, rows per input byte
R=$PWD; O=$R/gpurun_out/pmc1; mkdir -p $O; export TMPDIR=/tmp
name=$1; ctr=$2; N=$3; shift 3
cd /tmp
timeout -s KILL 400 rocprofv3 --kernel-include-regex "lz4hip" --pmc $ctr --output-format csv -d $O/pmc_$name -o p -- $R/tools/prof/prof_bench $R/tests/golden/testdata "$@" > $O/pmc_$name.log 2>&1 || exit 1
python3 $R/tools/prof/pmc_rows.py $O/pmc_$name $N
