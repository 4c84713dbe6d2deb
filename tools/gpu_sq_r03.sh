#!/bin/bash
# SQ counter passes of round 3's kernels (gpurun -- 'bash tools/gpu_sq_r03.sh'); summaries -> gpurun_out/pmc_<tag>/
set -u
export RUNNER=tools/run_kind.py
#                    tag            log2n T   D  mode in mix    pipe
bash tools/pmc_fir.sh r03_cfg5_s1    26   511 10 1    0  0      5 || exit 1
bash tools/pmc_fir.sh r03_int8       28   64  10 1    4  0      0 || exit 1
bash tools/pmc_fir.sh r03_i16_mix    28   64  10 1    0  0.0371 0 || exit 1
bash tools/pmc_fir.sh r03_cf32_mix   28   64  10 1    1  0.0371 0 || exit 1
bash tools/pmc_fir.sh r03_cf32_dma   28   64  10 1    1  0      0 || exit 1
bash tools/pmc_fir.sh r03_cf32_511   28   511 10 1    1  0      0 || exit 1
