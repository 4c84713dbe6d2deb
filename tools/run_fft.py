#!/usr/bin/env python3
"""Launch the FFT->dB kernel a few times (for rocprofv3 counter passes).
usage: run_fft.py log2n nfft hop [iters [variant]]   (variant: sdr_fft_plan_set_variant)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdr_amd  # noqa: E402

log2n, nfft, hop = [int(x) for x in sys.argv[1:4]]
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
sdr_amd.init(0)
n = 1 << log2n
d_in = sdr_amd.DevBuf(n * 4 + 64)
sdr_amd.capi.synth_fm_i16_dev(d_in, n, fs=61.44e6, ftone=250e3, dev=2.0e6, amp=1200.0, noise_lsb=64, seed=3)
plan = sdr_amd.FftPlan(nfft, hop, sdr_amd.window("blackman", nfft))
if len(sys.argv) > 5:
    plan.set_variant(int(sys.argv[5]))
frames = plan.frame_count(n)
d_out = sdr_amd.DevBuf(frames * nfft * 4)
plan.time_dev(d_in, n, d_out, 20)
ms = min(plan.time_dev(d_in, n, d_out, iters) for _ in range(3))
print("ms", ms, "frames", frames, "GB/s", (4.0 * n + 4.0 * nfft * frames) / ms / 1e6)
