#!/bin/bash
# host-side AddressSanitizer run of the library after round 2's rewrites, and SQ counters of the K-split kernel
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02q
timeout -k 10 600 bash tools/asan/run_host_asan.sh > gpurun_out/r02q/host_asan.txt 2>&1; rc=$?
tail -6 gpurun_out/r02q/host_asan.txt
[ $rc -eq 0 ] || exit $rc
RUNNER=tools/run_stage2.py bash tools/pmc_fir.sh r02ks 23 3 6710886 > gpurun_out/r02q/pmc_ksplit.txt 2>&1 || { echo ksplit failed; tail -5 gpurun_out/r02q/pmc_ksplit.txt; exit 1; }
grep -h "^ms" gpurun_out/pmc_r02ks/a.log
