#!/bin/bash
# SQ counter passes of the FIR kernels VERDICT r1 names: headline raw tile, int8 matrix kernel (cfg5
# stage 1) and the real-tile streaming kernel (cfg5 stage 2)
set -u
export TMPDIR=/tmp
bash tools/pmc_fir.sh r02raw 30 64 10 1 0 0 1 3 > gpurun_out/pmc_r02raw.txt 2>&1 || { echo raw failed; tail -5 gpurun_out/pmc_r02raw.txt; exit 1; }
bash tools/pmc_fir.sh r02i8 28 511 10 1 0 0 5 3 > gpurun_out/pmc_r02i8.txt 2>&1 || { echo i8 failed; tail -5 gpurun_out/pmc_r02i8.txt; exit 1; }
RUNNER=tools/run_stage2.py bash tools/pmc_fir.sh r02st2 23 3 > gpurun_out/pmc_r02st2.txt 2>&1 || { echo stage2 failed; tail -5 gpurun_out/pmc_r02st2.txt; exit 1; }
grep -h "^ms\|WAVE_CYCLES\|MFMA_BUSY" gpurun_out/pmc_r02*.txt | head; grep -h "^ms" gpurun_out/pmc_r02*/a.log
