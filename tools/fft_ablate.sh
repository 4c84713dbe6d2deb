#!/bin/bash
# Build ablated FFT libraries (CPU side): tools/fft_ablate.sh build ; run on the GPU: tools/fft_ablate.sh run
# Variants (bit mask, see SDR_FFT_ABLATE in fft.hip): 0 shipped, 1 no LDS exchanges, 2 no butterflies,
# 7 memory traffic only, 8 no stores, 16 no HBM reads, 24 neither, 27 LDS-free arithmetic-free shell.
set -eu
cd "$(dirname "$0")/.."
mkdir -p tools/exp
VARS="${VARS:-0 1 2 3 7 8 16 24 9 10 27}"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Isdr_amd/csrc -fno-slp-vectorize"
O="sdr_amd/build/lib.o sdr_amd/build/fir.o sdr_amd/build/stream.o sdr_amd/build/apt.o"
case "${1:-build}" in
build)
  for v in $VARS; do
    hipcc $F -DSDR_MEASUREMENT_BUILD -DSDR_FFT_ABLATE=$v -c sdr_amd/csrc/fft.hip -o tools/exp/fft_abl$v.o &
  done
  wait
  for v in $VARS; do
    hipcc --offload-arch=gfx950 -shared -fPIC $O tools/exp/fft_abl$v.o -o tools/exp/libsdrhip_abl$v.so
  done ;;
run)
  for v in $VARS 0; do
    echo -n "ablate=$v  "
    SDRHIP_LIB=$PWD/tools/exp/libsdrhip_abl$v.so timeout -k 10 120 python3 tools/run_fft.py ${LOG2N:-28} ${NFFT:-4096} ${HOP:-2048} 20 || exit 1
  done ;;
esac
