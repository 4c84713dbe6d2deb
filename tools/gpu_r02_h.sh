#!/bin/bash
# int8 kernel with the start-up outputs folded in: parity subset, size scaling of stage 1, cfg5 chain
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02h
timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -x -q --timeout 120 --timeout-method thread \
  -k "matrix or pipeline5 or byte_split or apt or stream_blocks or startup or randomised or long_filter or zero" > gpurun_out/r02h/pytest.log 2>&1 || { echo "pytest failed"; tail -30 gpurun_out/r02h/pytest.log; exit 1; }
tail -2 gpurun_out/r02h/pytest.log
for n in 24 26 27 28; do
  timeout -k 10 120 python3 tools/run_one.py $n 511 10 1 0 0 5 20 > gpurun_out/r02h/one_$n.log 2>&1 || { echo "run_one $n failed"; tail -5 gpurun_out/r02h/one_$n.log; exit 1; }
  echo "log2n $n: $(grep '^ms' gpurun_out/r02h/one_$n.log | cut -c1-40)"
done
for r in 1 2; do
  timeout -k 10 200 python3 tools/bench_configs.py cfg5 > gpurun_out/r02h/cfg5_$r.log 2>&1 || { echo "cfg5 failed"; tail -5 gpurun_out/r02h/cfg5_$r.log; exit 1; }
  tail -1 gpurun_out/r02h/cfg5_$r.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg5', j['kernel_ms'], j['stage1_ms'], j['stage2_ms'], j['roofline']['frac'], j['parity_max_abs_err'])"
done
