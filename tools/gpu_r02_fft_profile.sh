#!/bin/bash
# cfg4 profile of round 2: kernel-trace stats of the secondary configs, A/B of the four 4096-point
# kernels, and SQ counter passes of the old (variant 2) and new (variant 4) kernels
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r02e
mkdir -p $OUT
timeout -k 10 200 python -u tools/fft_variants.py 28 3 > $OUT/fft_variants.jsonl 2>&1 || { echo "variants failed"; tail -5 $OUT/fft_variants.jsonl; exit 1; }
cat $OUT/fft_variants.jsonl | cut -c1-150
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/bench_configs.py cfg3 cfg4 cfg5 > $OUT/configs.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/configs.log; exit 1; }
grep "^{" $OUT/configs.log > $OUT/configs.jsonl
cut -c1-200 $OUT/configs.jsonl
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/run_fft.py 28 4096 2048 3 > $OUT/fft_fetch.log 2>&1 || { echo "fetch pass failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/run_fft.py 28 4096 2048 3 > $OUT/fft_write.log 2>&1 || { echo "write pass failed"; exit 1; }
VARIANTS="4 2" bash tools/gpu_r02_fft_pmc.sh > $OUT/pmc_sq.log 2>&1 || { echo "sq passes failed"; tail -5 $OUT/pmc_sq.log; exit 1; }
echo done
