#!/usr/bin/env python3
"""Launch one FIR configuration of any input kind a few times (for rocprofv3 counter passes, tools/pmc_fir.sh with
RUNNER=tools/run_kind.py).  usage: run_kind.py log2n T D mode input mix_fr pipe [iters]
input: 0 int16 IQ, 1 complex float, 2 real float, 3 int16 QI, 4 int8 IQ; mix_fr 0 = no mixer"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdr_amd as S  # noqa: E402

log2n, T, D, mode, inp = [int(x) for x in sys.argv[1:6]]
mix = float(sys.argv[6])
pipe = int(sys.argv[7])
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 3
S.init(0)
n = 1 << log2n
d_i16 = S.DevBuf(n * 4 + 64)
S.capi.synth_fm_i16_dev(d_i16, n, fs=20e6, noise_lsb=8)
d_in = d_i16
if inp in (S.IN_CF32, S.IN_F32_REAL):
    d_in = S.DevBuf(n * 8 + 64)
    S.capi.unpack_i16_dev(d_i16, n, d_in)          # complex float samples (read as real floats: 2n of them)
    S.sync()
d_out = S.DevBuf((n // D + 64) * 8)
h = S.create_fir_filter(T, 1.0 / D, "hann")
p = S.FirPlan(h, D, mode=mode, input=inp, mix_fr=mix, pipeline=pipe)
p.time_dev(d_in, n, d_out, 40)
ms = min(p.time_dev(d_in, n, d_out, iters) for _ in range(3))
print("ms", ms, p.info())
