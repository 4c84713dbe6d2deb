#!/usr/bin/env python3
"""A/B of the 4096-point / hop-2048 kernels on one box, interleaved and warm, with a parity check of
every variant against the oracle on three frames.  usage: fft_variants.py [log2n] [rounds]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdr_amd  # noqa: E402
from oracle import oracle as O  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sdr_amd.init(0)
n, nfft, hop = 1 << log2n, 4096, 2048
w = sdr_amd.window("blackman", nfft)
d_in = sdr_amd.DevBuf(n * 4 + 64)
sdr_amd.capi.synth_fm_i16_dev(d_in, n, fs=61.44e6, ftone=250e3, dev=2.0e6, amp=1200.0, noise_lsb=64, seed=3)
names = {1: "workgroup exchange (fft_db_kernel)", 2: "two frames per lane (fft4096_pair_kernel)",
         3: "one frame per wavefront, 4 waves per CU (fft4096_wave_kernel<false>)",
         4: "one frame per wavefront, 8 waves per CU (fft4096_wave_kernel<true>)"}
plans = {}
for v in names:
    plans[v] = sdr_amd.FftPlan(nfft, hop, w)
    plans[v].set_variant(v)
frames = plans[1].frame_count(n)
d_out = sdr_amd.DevBuf(frames * nfft * 4)
best = {v: 1e9 for v in names}
err = {}
for v, plan in plans.items():
    assert plan.execute_dev(d_in, n, d_out, frames) == frames
    sdr_amd.sync()
    worst = 0.0
    for f0 in (0, frames // 2, frames - 4):
        iq = d_in.to_numpy(np.int16, 2 * (nfft + 3 * hop), f0 * hop * 4)
        _, want_lin = O.fft_power_db_i16(iq, nfft, hop, w, linear=True)
        got = d_out.to_numpy(np.float32, 4 * nfft, f0 * nfft * 4).reshape(4, nfft).astype(np.float64)
        worst = max(worst, float(np.abs(10.0 ** (got / 10.0) - want_lin).max() / want_lin.max()))
    err[v] = worst
    plan.time_dev(d_in, n, d_out, 30)
for _ in range(rounds):
    for v, plan in plans.items():
        best[v] = min(best[v], plan.time_dev(d_in, n, d_out, 20))
for v in names:
    gbs = (4.0 * n + 4.0 * nfft * frames) / best[v] / 1e6
    print(json.dumps({"variant": v, "kernel": names[v], "ms": round(best[v], 4), "GB/s": round(gbs, 1),
                      "frac_of_8TBps": round(gbs / 8000.0, 4), "rel_power_err": err[v], "frames": frames}), flush=True)
