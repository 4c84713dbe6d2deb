#!/bin/bash
# Measurement builds of the int8 matrix kernel (bit mask SDR_I8_ABLATE, see fir.hip).
#   tools/i8_ablate.sh build      (CPU side; VARS="8 16 32 ...")
#   tools/i8_ablate.sh run        (GPU: times every variant; variant 8 prints the per-role phase clocks)
set -eu
cd "$(dirname "$0")/.."
mkdir -p tools/exp
VARS="${VARS:-8 16 32 48 6 22 54}"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Isdr_amd/csrc -mllvm -amdgpu-atomic-optimizer-strategy=None"
O="sdr_amd/build/lib.o sdr_amd/build/fft.o sdr_amd/build/fft_wave.o sdr_amd/build/stream.o sdr_amd/build/apt.o"
case "${1:-build}" in
build)
  for v in $VARS; do
    hipcc $F -DSDR_MEASUREMENT_BUILD -DSDR_I8_ABLATE=$v -c sdr_amd/csrc/fir.hip -o tools/exp/fir_abl$v.o &
  done
  wait
  hipcc $F -DSDR_MEASUREMENT_BUILD -c sdr_amd/csrc/lib.hip -o tools/exp/lib_meas.o
  for v in $VARS; do
    hipcc --offload-arch=gfx950 -shared -fPIC ${O/sdr_amd\/build\/lib.o/tools\/exp\/lib_meas.o} tools/exp/fir_abl$v.o -o tools/exp/libsdrhip_i8abl$v.so
  done ;;
run)
  for r in 1 2; do
    echo -n "shipped  "; timeout -k 10 120 python3 tools/run_one.py ${LOG2N:-28} ${TD:-511 10} 1 0 0 5 20 | cut -c1-30 || exit 1
    for v in $VARS; do
      echo -n "ablate=$v  "
      if [ $v = 8 ]; then
        SDRHIP_LIB=$PWD/tools/exp/libsdrhip_i8abl$v.so timeout -k 10 120 python3 tools/i8_stamps.py ${LOG2N:-28} ${TD:-511 10} || exit 1
      else
        SDRHIP_LIB=$PWD/tools/exp/libsdrhip_i8abl$v.so timeout -k 10 120 python3 tools/run_one.py ${LOG2N:-28} ${TD:-511 10} 1 0 0 5 20 | cut -c1-30 || exit 1
      fi
    done
  done ;;
esac
