#!/usr/bin/env python3
"""Long-filter evaluation (BASELINE config 5): the same 511-tap / decimate-by-10 FIR through the
raw-tile (VALU, converts per read), planar (VALU, converts once), float32 MFMA Toeplitz and int8
matrix-pipe Toeplitz kernels, warm, interleaved so that all see the same clocks.  Run under rocprofv3 --kernel-trace --stats for
the per-kernel summary kept in profiles/."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import sdr_amd  # noqa: E402
from oracle import oracle as O  # noqa: E402

sdr_amd.init(0)
n, T, D = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 26), 511, 10
h = sdr_amd.create_fir_filter(T, 20800 / 62400.0 / 5, "hann")
d_in = sdr_amd.DevBuf(n * 4 + 64)
sdr_amd.capi.synth_fm_i16_dev(d_in, n, fs=624e3, ftone=2400.0, dev=17000.0, amp=1800.0, noise_lsb=4, seed=5)
d_out = sdr_amd.DevBuf(n // D * 8 + 64)
plans = {name: sdr_amd.FirPlan(h, D, mode=sdr_amd.MODE_FM, pipeline=p)
         for name, p in (("raw_tile", 1), ("planar", 3), ("mfma_f32", 4), ("mfma_i8", 5))}
m = 1 << 20
iq = d_in.to_numpy(np.int16, 2 * m)
want = O.chain_i16(iq, D, h, O.MODE_FM)
best = {k: 1e9 for k in plans}
for name, p in plans.items():
    p.time_dev(d_in, n, d_out, 30)
    got = d_out.to_numpy(np.float32, want.size).astype(np.float64)
    err = float(np.abs(got - want)[T // D + 2:].max())
    assert err <= 5e-6, (name, err)
for _ in range(4):
    for name, p in plans.items():
        best[name] = min(best[name], p.time_dev(d_in, n, d_out, 10))
flops = 4.0 * T / D * n
for name, ms in best.items():
    print(json.dumps({"kernel": name, "taps": T, "decim": D, "samples": n, "ms": round(ms, 4),
                      "useful_TFLOPs_equiv": round(flops / ms / 1e9, 1),
                      "input_GBps": round(4.0 * n / ms / 1e6, 1), "info": plans[name].info()}), flush=True)
