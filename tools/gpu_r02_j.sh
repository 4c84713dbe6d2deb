#!/bin/bash
# new 512-lane int8 kernel: parity subset, then A/B against the previous 1024-lane build, phase clocks, cfg5
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02j
timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -x -q --timeout 120 --timeout-method thread \
  -k "matrix or pipeline5 or byte_split or apt or stream_blocks or startup or randomised or long_filter or zero" > gpurun_out/r02j/pytest.log 2>&1 || { echo "pytest failed"; tail -40 gpurun_out/r02j/pytest.log; exit 1; }
tail -2 gpurun_out/r02j/pytest.log
for r in 1 2; do
  echo -n "old  "; SDRHIP_LIB=$PWD/tools/exp/libsdrhip_i8old.so timeout -k 10 120 python3 tools/run_one.py 28 511 10 1 0 0 5 20 | cut -c1-30 || exit 1
  echo -n "new  "; timeout -k 10 120 python3 tools/run_one.py 28 511 10 1 0 0 5 20 | cut -c1-30 || exit 1
done
for v in 6 1; do echo -n "ablate=$v "; SDRHIP_LIB=$PWD/tools/exp/libsdrhip_i8abl$v.so timeout -k 10 120 python3 tools/run_one.py 28 511 10 1 0 0 5 20 2>/dev/null | cut -c1-30 || exit 1; done
SDRHIP_LIB=$PWD/tools/exp/libsdrhip_i8abl8.so timeout -k 10 120 python3 tools/i8_stamps.py 28 2>/dev/null | sed -n 2,14p || exit 1
timeout -k 10 200 python3 tools/bench_configs.py cfg5 > gpurun_out/r02j/cfg5.log 2>&1 || { echo "cfg5 failed"; tail -5 gpurun_out/r02j/cfg5.log; exit 1; }
tail -1 gpurun_out/r02j/cfg5.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg5', j['kernel_ms'], j['stage1_ms'], j['stage2_ms'], j['roofline']['frac'], j['parity_max_abs_err'])"
