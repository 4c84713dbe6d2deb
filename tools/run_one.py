#!/usr/bin/env python3
"""Launch one FIR configuration a few times (for rocprofv3 counter passes).
usage: run_one.py log2n T D mode R NT pipe [iters]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdr_amd  # noqa: E402

log2n, T, D, mode, R, NT, pipe = [int(x) for x in sys.argv[1:8]]
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 3
sdr_amd.init(0)
n = 1 << log2n
d_in = sdr_amd.DevBuf(n * 4 + 64)
sdr_amd.capi.synth_fm_i16_dev(d_in, n, fs=20e6, noise_lsb=8)
d_out = sdr_amd.DevBuf((n // D + 64) * 8)              # room for complex outputs at any D
h = sdr_amd.create_fir_filter(T, 1.0 / D, "hann")
p = sdr_amd.FirPlan(h, D, mode=mode, outputs_per_lane=R, threads=NT, pipeline=pipe)
p.time_dev(d_in, n, d_out, 40)          # clocks ramp over the first ~20 launches after idle
ms = min(p.time_dev(d_in, n, d_out, iters) for _ in range(3))
print("ms", ms, p.info())
