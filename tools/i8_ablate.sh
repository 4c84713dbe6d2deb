#!/bin/bash
# Measurement builds of the int8 matrix kernel (bit mask SDR_I8_ABLATE, see fir_mfma_i8.hip).
#   tools/i8_ablate.sh build      (CPU side, after capi.build(); VARS="8 16 32 ...")  -> sdr_amd/libsdrhip_i8abl<v>.so
#   tools/i8_ablate.sh run        (GPU: times every variant; variant 8 prints the per-role phase clocks)
set -eu
cd "$(dirname "$0")/.."
VARS="${VARS:-8 16 32 48 6 22 54}"
case "${1:-build}" in
build)
  F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Isdr_amd/csrc"
  mkdir -p tools/exp
  hipcc $F -DSDR_MEASUREMENT_BUILD -c sdr_amd/csrc/lib.hip -o tools/exp/lib_meas.o
  for v in $VARS; do
    ( hipcc $F -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize -DSDR_MEASUREMENT_BUILD -DSDR_I8_ABLATE=$v -c sdr_amd/csrc/fir_mfma_i8.hip -o tools/exp/fir_abl$v.o
      objs=""
      for o in sdr_amd/build/*.o; do
        case $(basename $o .o) in
        lib) objs="$objs tools/exp/lib_meas.o" ;;
        fir_mfma_i8) objs="$objs tools/exp/fir_abl$v.o" ;;
        *) objs="$objs $o" ;;
        esac
      done
      hipcc --offload-arch=gfx950 -shared -fPIC $objs -o sdr_amd/libsdrhip_i8abl$v.so ) &
  done
  wait ;;
run)
  for r in 1 2; do
    echo -n "shipped  "; timeout -k 10 120 python3 tools/run_one.py ${LOG2N:-28} ${TD:-511 10} 1 0 0 5 20 | cut -c1-30 || exit 1
    for v in $VARS; do
      echo -n "ablate=$v  "
      if [ $v = 8 ]; then
        SDRHIP_LIB=$PWD/sdr_amd/libsdrhip_i8abl$v.so timeout -k 10 120 python3 tools/i8_stamps.py ${LOG2N:-28} ${TD:-511 10} || exit 1
      else
        SDRHIP_LIB=$PWD/sdr_amd/libsdrhip_i8abl$v.so timeout -k 10 120 python3 tools/run_one.py ${LOG2N:-28} ${TD:-511 10} 1 0 0 5 20 | cut -c1-30 || exit 1
      fi
    done
  done ;;
esac
