#!/bin/bash
# end-of-round refresh of the headline profiles with the final build (kernel signature gained the int8-sample flag):
# line plain, under a kernel trace, with the raw tile kernel pinned for comparison, and the HBM counter passes
set -u
export TMPDIR=/tmp
O=gpurun_out/r02y
mkdir -p $O
for r in 1 2; do
  timeout -k 10 300 python3 -u bench.py --steps 20 --warmup 5 > $O/bench_auto_$r.log 2>&1 || { echo "bench failed"; tail -5 $O/bench_auto_$r.log; exit 1; }
  tail -n 1 $O/bench_auto_$r.log | cut -c1-400
  timeout -k 10 300 python3 -u bench.py --steps 20 --warmup 5 --cpu-seconds 0 --pipeline 1 > $O/bench_raw_$r.log 2>&1 || { echo "bench raw failed"; tail -5 $O/bench_raw_$r.log; exit 1; }
  tail -n 1 $O/bench_raw_$r.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('raw tile pinned:', j['roofline']['kernel_ms'], j['roofline']['frac'])"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 5 > $O/bench_under_rocprof.log 2>&1 || { echo "rocprof pass failed"; tail -5 $O/bench_under_rocprof.log; exit 1; }
tail -n 1 $O/bench_under_rocprof.log | cut -c1-300
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 5 --warmup 30 --cpu-seconds 0 > $O/bench_fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $O/bench_fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 5 --warmup 30 --cpu-seconds 0 > $O/bench_write.log 2>&1 || { echo "write pass failed"; tail -5 $O/bench_write.log; exit 1; }
timeout -k 10 300 python3 -u bench.py --steps 20 --warmup 5 --config channels --cpu-seconds 0 > $O/bench_channels.log 2>&1 || { echo "bench channels failed"; tail -5 $O/bench_channels.log; exit 1; }
tail -n 1 $O/bench_channels.log | cut -c1-300
