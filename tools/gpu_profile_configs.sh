#!/bin/bash
# rocprofv3 kernel-trace stats of the secondary config benches + HBM traffic PMC passes of the FFT.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_cfg
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/bench_configs.py cfg3 cfg4 cfg5 > $OUT/configs_stats.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/configs_stats.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/run_fft.py 28 4096 2048 3 > $OUT/fft_fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $OUT/fft_fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/run_fft.py 28 4096 2048 3 > $OUT/fft_write.log 2>&1 || { echo "write pass failed"; tail -5 $OUT/fft_write.log; exit 1; }
find $OUT -name "*.csv" | head -20
