#!/bin/bash
# Experiment build of libsdrhip: one source recompiled with extra flags, linked with the product's other objects
# into sdr_amd/libsdrhip_<tag>.so (git-ignored; bound through SDRHIP_LIB by tools/ab_libs.py and friends).
#   tools/build_variant.sh <tag> <source.hip> [extra hipcc flags...]      (run capi.build() first: it makes the objects)
set -eu
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
name=$(basename "$src" .hip)
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Isdr_amd/csrc"
case $name in
fir_mfma_i8) F="$F -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize" ;;
fft) F="$F -fno-slp-vectorize" ;;
fft_wave) F="$F -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp" ;;
esac
mkdir -p tools/exp
hipcc $F "$@" -c "sdr_amd/csrc/$name.hip" -o "tools/exp/${name}_$tag.o"
objs=""
for o in sdr_amd/build/*.o; do
    if [ "$(basename "$o" .o)" = "$name" ]; then objs="$objs tools/exp/${name}_$tag.o"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "sdr_amd/libsdrhip_$tag.so"
echo "built sdr_amd/libsdrhip_$tag.so"
