#!/bin/bash
# end-of-round refresh: kernel-trace stats of the secondary configs (profiles/r02_configs*), SQ counter
# passes of the rewritten int8 kernel at the cfg5 shape and at the headline shape
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r02o
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/bench_configs.py cfg3 cfg4 cfg5 > $OUT/configs.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/configs.log; exit 1; }
grep "^{" $OUT/configs.log > $OUT/configs.jsonl
cut -c1-220 $OUT/configs.jsonl
timeout -k 10 300 python3 tools/bench_configs.py handoff > $OUT/handoff.log 2>&1 || { echo "handoff failed"; tail -5 $OUT/handoff.log; exit 1; }
grep "^{" $OUT/handoff.log | cut -c1-300
bash tools/pmc_fir.sh r02i8b 28 511 10 1 0 0 5 3 > $OUT/pmc_i8_cfg5.txt 2>&1 || { echo i8 cfg5 failed; tail -5 $OUT/pmc_i8_cfg5.txt; exit 1; }
bash tools/pmc_fir.sh r02i8h 30 64 10 1 0 0 5 3 > $OUT/pmc_i8_headline.txt 2>&1 || { echo i8 headline failed; tail -5 $OUT/pmc_i8_headline.txt; exit 1; }
grep -h "^ms" gpurun_out/pmc_r02i8b/a.log gpurun_out/pmc_r02i8h/a.log
echo done
