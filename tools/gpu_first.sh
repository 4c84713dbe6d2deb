#!/bin/bash
# first GPU pass: parity tests, then (unless something crashed) a short bench
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 120 --timeout-method thread > gpurun_out/pytest.log 2>&1
rc=$?
tail -n 30 gpurun_out/pytest.log
echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 300 python -u bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1
rc2=$?
tail -n 5 gpurun_out/bench.log
echo "bench rc=$rc2"
exit $rc
