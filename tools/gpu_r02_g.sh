#!/bin/bash
# headline A/B raw tile vs int8 matrix kernel at 2^30, then SQ counters of the int8 kernel after the round-2 changes
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02g
for r in 1 2; do for p in 0 5; do
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --pipeline $p > gpurun_out/r02g/bench_p${p}_$r.log 2>&1 || { echo "bench p=$p failed"; tail -3 gpurun_out/r02g/bench_p${p}_$r.log; exit 1; }
  tail -1 gpurun_out/r02g/bench_p${p}_$r.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['config']['kernel']['kernel'], j['roofline']['kernel_ms'], j['roofline']['kernel_ms_min'], j['roofline']['frac'], j['parity_max_abs_err'])"
done; done
bash tools/pmc_fir.sh r02i8 28 511 10 1 0 0 5 3 > gpurun_out/pmc_r02i8.txt 2>&1 || { echo i8 pmc failed; tail -5 gpurun_out/pmc_r02i8.txt; exit 1; }
grep -h "^ms" gpurun_out/pmc_r02i8/a.log
