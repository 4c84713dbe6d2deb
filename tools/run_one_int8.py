#!/usr/bin/env python3
"""Launch one FIR configuration on int8 sample pairs (SDR_IN_I8_IQ) a few times (for rocprofv3 passes).
usage: run_one_int8.py log2n T D mode pipe [iters]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdr_amd  # noqa: E402

log2n, T, D, mode, pipe = [int(x) for x in sys.argv[1:6]]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 3
sdr_amd.init(0)
n = 1 << log2n
d_in = sdr_amd.DevBuf(n * 2 + 64)
sdr_amd.capi.synth_fm_i16_dev(d_in, n // 2, fs=20e6, noise_lsb=8)      # any bytes will do
d_out = sdr_amd.DevBuf((n // D + 64) * 8)
h = sdr_amd.create_fir_filter(T, 1.0 / D, "hann")
p = sdr_amd.FirPlan(h, D, mode=mode, input=sdr_amd.IN_I8_IQ, pipeline=pipe)
p.time_dev(d_in, n, d_out, 40)          # clocks ramp over the first ~20 launches after idle
ms = min(p.time_dev(d_in, n, d_out, iters) for _ in range(3))
print("ms", ms, p.info())
