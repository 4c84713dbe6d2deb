#!/usr/bin/env python3
"""Batched-channel timing sweep: same 2^30 samples split into different channel counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdr_amd
sdr_amd.init(0)
D, T = 10, 64
h = sdr_amd.create_fir_filter(T, 1.0 / D, "hann")
total = 1 << 30
d_in = sdr_amd.DevBuf(total * 4 + (1 << 20))
sdr_amd.capi.synth_fm_i16_dev(d_in, total, fs=20e6, noise_lsb=8)
d_out = sdr_amd.DevBuf(total // D * 4 + (1 << 22))
for mode in (sdr_amd.MODE_FM, sdr_amd.MODE_ENV):
    plan = sdr_amd.FirPlan(h, D, mode=mode)
    for log2ch in (0, 4, 8, 10, 12, 14):
        n_ch = 1 << log2ch
        n = total >> log2ch
        for pad in (0, 64):
            if pad and log2ch == 0:
                continue
            stride = n + pad
            if stride * n_ch > total + (1 << 18):
                continue
            cnt = plan.out_count(n)
            ostride = (cnt + 3) & ~3 if pad else cnt
            f = lambda k: plan.time_dev(d_in, n, d_out, k, in_stride=stride, n_channels=n_ch, out_stride=ostride)
            f(30)
            ms = min(f(10) for _ in range(3))
            print(f"mode={mode} n_ch={n_ch:6d} n={n:10d} in_stride_pad={pad:3d} out_stride={ostride}: {ms:.4f} ms  {4.4 * total / ms / 1e6:.0f} GB/s", flush=True)
