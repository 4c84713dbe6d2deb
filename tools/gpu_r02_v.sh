#!/bin/bash
# int8 sample pairs on the int8 matrix kernel (64 taps / 10, FM, 2^28 samples): kernel trace, HBM traffic
# (FETCH_SIZE and WRITE_SIZE in separate passes), SQ counters
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02v
mkdir -p $O
A="28 64 10 1 0 20"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/run_one_int8.py $A > $O/trace.log 2>&1 || { echo trace failed; tail -3 $O/trace.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 tools/run_one_int8.py $A > $O/fetch.log 2>&1 || { echo fetch failed; tail -3 $O/fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 tools/run_one_int8.py $A > $O/write.log 2>&1 || { echo write failed; tail -3 $O/write.log; exit 1; }
RUNNER=tools/run_one_int8.py bash tools/pmc_fir.sh r02v_int8 $A > $O/pmc.txt 2>&1 || { echo pmc failed; tail -5 $O/pmc.txt; exit 1; }
grep -h "^ms" $O/trace.log $O/fetch.log
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs head -5
