#!/bin/bash
# int8 kernel built without the SLP vectoriser (no packed-float instructions in the helper roles) vs shipped
set -u
for r in 1 2; do
  echo -n "shipped "; timeout -k 10 120 python3 tools/run_one.py 28 511 10 1 0 0 5 20 | cut -c1-30 || exit 1
  echo -n "no-slp  "; SDRHIP_LIB=$PWD/tools/exp/libsdrhip_noslp.so timeout -k 10 120 python3 tools/run_one.py 28 511 10 1 0 0 5 20 | cut -c1-30 || exit 1
done
SDRHIP_LIB=$PWD/tools/exp/libsdrhip_noslp8.so timeout -k 10 120 python3 tools/i8_stamps.py 28 | sed -n 2,6p || exit 1
echo -n "headline shipped "; timeout -k 10 120 python3 tools/run_one.py 30 64 10 1 0 0 0 20 | cut -c1-30 || exit 1
echo -n "headline no-slp  "; SDRHIP_LIB=$PWD/tools/exp/libsdrhip_noslp.so timeout -k 10 120 python3 tools/run_one.py 30 64 10 1 0 0 0 20 | cut -c1-30 || exit 1
