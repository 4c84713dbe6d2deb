#!/bin/bash
# Build libsdrhip's sources + tools/asan/host_asan_driver.cpp into one executable with host-side
# AddressSanitizer (device code is not instrumented) and run it.  Build here (no GPU needed), run on a GPU box.
set -eu
cd "$(dirname "$0")/../.."
OUT=examples/bin/host_asan_driver     # (git-ignored; examples/bin travels to the GPU box, tools/exp does not)
mkdir -p examples/bin
if [ "${1:-}" = build ]; then
  F="--offload-arch=gfx950 -O1 -g -std=c++17 -Iinclude -Isdr_amd/csrc -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize"   # the last two: required by fir_mfma_i8.hip (sdr_amd/capi.py)
  hipcc $F sdr_amd/csrc/lib.hip sdr_amd/csrc/fir_tiles.hip sdr_amd/csrc/fir_mfma_i8.hip sdr_amd/csrc/fir_plan.hip sdr_amd/csrc/stream.hip sdr_amd/csrc/apt.hip sdr_amd/csrc/fft.hip sdr_amd/csrc/fft_wave.hip tools/asan/host_asan_driver.cpp -o $OUT
  exit 0
fi
ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 $OUT
