#!/usr/bin/env python3
"""Repeat the multi-role kernels on the same input and require bit-identical outputs every time
(a missed barrier or an LDS buffer reused too early shows up as run-to-run differences)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import sdr_amd  # noqa: E402

sdr_amd.init(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << 24
d_in = sdr_amd.DevBuf(n * 4 + 64)
sdr_amd.capi.synth_fm_i16_dev(d_in, n, fs=624e3, ftone=2400.0, dev=17000.0, amp=1800.0, noise_lsb=8, seed=2)
bad = 0
for (T, D, mode) in ((511, 10, sdr_amd.MODE_FM), (511, 15, sdr_amd.MODE_ENV), (200, 10, sdr_amd.MODE_IQ), (101, 5, sdr_amd.MODE_FM),
                     (64, 10, sdr_amd.MODE_FM), (101, 20, sdr_amd.MODE_FM), (300, 25, sdr_amd.MODE_ENV), (60, 31, sdr_amd.MODE_IQ)):
    h = sdr_amd.create_fir_filter(T, 0.8 / D, "hann")
    p = sdr_amd.FirPlan(h, D, mode=mode, pipeline=5)
    cnt = p.out_count(n)
    d_out = sdr_amd.DevBuf(cnt * 8 + 64)
    ref = None
    for r in range(reps):
        p.execute_dev(d_in, n, d_out, cnt)
        sdr_amd.sync()
        got = d_out.to_numpy(np.uint32, cnt * (2 if mode == sdr_amd.MODE_IQ else 1))
        if ref is None:
            ref = got.copy()
        elif not np.array_equal(ref, got):
            bad += 1
            print("DIFF", T, D, mode, "rep", r, int((ref != got).sum()), flush=True)
    print("fir", T, D, mode, p.info()["kernel"], "ok" if not bad else "FAILED", flush=True)
    d_out.free()
w = sdr_amd.window("blackman", 4096)
plan = sdr_amd.FftPlan(4096, 2048, w)
frames = plan.frame_count(n)
d_out = sdr_amd.DevBuf(frames * 4096 * 4)
ref = None
for r in range(reps):
    plan.execute_dev(d_in, n, d_out, frames)
    sdr_amd.sync()
    got = d_out.to_numpy(np.uint32, frames * 4096)
    if ref is None:
        ref = got.copy()
    elif not np.array_equal(ref, got):
        bad += 1
        print("DIFF fft rep", r, int((ref != got).sum()), flush=True)
print("fft 4096/2048", frames, "frames", "ok" if not bad else "FAILED", flush=True)
sys.exit(1 if bad else 0)
