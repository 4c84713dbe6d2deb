#!/bin/bash
# round 2, first GPU pass: parity tests, the bench with the driver's flags, and a kernel trace of
# exactly that command line (profiles/r02_bench_driver_flags_*)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02a
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread > gpurun_out/r02a/pytest.log 2>&1
rc=$?
tail -n 15 gpurun_out/r02a/pytest.log
echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 > gpurun_out/r02a/bench_driver_flags.log 2>&1 || { echo "bench failed"; tail -5 gpurun_out/r02a/bench_driver_flags.log; exit 1; }
tail -n 1 gpurun_out/r02a/bench_driver_flags.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02a/stats -- python3 bench.py --steps 20 --warmup 5 > gpurun_out/r02a/bench_under_rocprof.log 2>&1 || { echo "rocprof pass failed"; tail -5 gpurun_out/r02a/bench_under_rocprof.log; exit 1; }
tail -n 1 gpurun_out/r02a/bench_under_rocprof.log
timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --prewarm-seconds 0 --cpu-seconds 0 > gpurun_out/r02a/bench_no_prewarm.log 2>&1 || { echo "bench (no prewarm) failed"; exit 1; }
tail -n 1 gpurun_out/r02a/bench_no_prewarm.log
timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --config channels --cpu-seconds 0 > gpurun_out/r02a/bench_channels.log 2>&1 || { echo "bench channels failed"; tail -5 gpurun_out/r02a/bench_channels.log; exit 1; }
tail -n 1 gpurun_out/r02a/bench_channels.log
exit $rc
