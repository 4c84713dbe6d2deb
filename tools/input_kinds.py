#!/usr/bin/env python3
"""Kernel time per input kind on the headline filter shape (64 taps / 10, FM), 2^28 samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdr_amd
sdr_amd.init(0)
n, T, D = 1 << 28, 64, 10
h = sdr_amd.create_fir_filter(T, 1.0 / D, "hann")
d_in = sdr_amd.DevBuf(n * 8 + 64)
sdr_amd.capi.synth_fm_i16_dev(d_in, 2 * n, fs=20e6, noise_lsb=8)       # any finite bit pattern will do for floats? no: use int16 data as bytes
d_out = sdr_amd.DevBuf(n // D * 8 + 64)
kinds = [("int16 IQ raw tile", sdr_amd.IN_I16_IQ, 0, 4), ("int16 IQ streaming", sdr_amd.IN_I16_IQ, 2, 4),
         ("int16 QI raw tile", sdr_amd.IN_I16_QI, 0, 4), ("int8 IQ streaming", sdr_amd.IN_I8_IQ, 0, 2)]
for name, kind, pipe, eb in kinds:
    for mode, mname in ((sdr_amd.MODE_FM, "FM"), (sdr_amd.MODE_ENV, "ENV")):
        p = sdr_amd.FirPlan(h, D, mode=mode, input=kind, pipeline=pipe)
        p.time_dev(d_in, n, d_out, 30)
        ms = min(p.time_dev(d_in, n, d_out, 10) for _ in range(3))
        bps = eb + 4.0 / D
        print(f"{name:22s} {mname:3s} {ms:7.4f} ms  {n / ms / 1e3:9.0f} MS/s  {bps * n / ms / 1e6:6.0f} GB/s  {p.info()}", flush=True)
