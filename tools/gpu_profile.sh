#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench + HBM traffic PMC passes (separate runs).
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py > $OUT/bench_stats.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/bench_stats.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 5 --warmup 30 --cpu-seconds 0 > $OUT/bench_fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $OUT/bench_fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 5 --warmup 30 --cpu-seconds 0 > $OUT/bench_write.log 2>&1 || { echo "write pass failed"; tail -5 $OUT/bench_write.log; exit 1; }
find $OUT -name "*.csv" | head -20
