#!/bin/bash
# SQ counter passes of the 4096-point FFT kernels (variants 2 and 3), summaries under gpurun_out/pmc_fftv*/
set -u
export TMPDIR=/tmp
for v in ${VARIANTS:-3 2}; do
  RUNNER=tools/run_fft.py bash tools/pmc_fir.sh fftv$v 28 4096 2048 3 $v > gpurun_out/pmc_fftv$v.txt 2>&1 || { echo "variant $v failed"; tail -5 gpurun_out/pmc_fftv$v.txt; exit 1; }
  cat gpurun_out/pmc_fftv$v.txt
done
