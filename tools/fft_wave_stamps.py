#!/usr/bin/env python3
"""Per-wave start / end times of fft4096_wave_kernel from a measurement build (SDR_FFT_ABLATE bit 5):
how evenly do the persistent waves finish?  usage: SDRHIP_LIB=<stamp build> fft_wave_stamps.py [log2n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdr_amd  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
sdr_amd.init(0)
print(sdr_amd.capi.build_info())
n, nfft, hop = 1 << log2n, 4096, 2048
d_in = sdr_amd.DevBuf(n * 4 + 64)
sdr_amd.capi.synth_fm_i16_dev(d_in, n, fs=61.44e6, ftone=250e3, dev=2.0e6, amp=1200.0, noise_lsb=64, seed=3)
plan = sdr_amd.FftPlan(nfft, hop, sdr_amd.window("blackman", nfft))
plan.set_variant(4)
frames = plan.frame_count(n)
d_out = sdr_amd.DevBuf((frames + 8) * nfft * 4)
plan.time_dev(d_in, n, d_out, 30)
ms = plan.time_dev(d_in, n, d_out, 1)
st = d_out.to_numpy(np.uint32, 2 * 2048, (frames + 1) * nfft * 4).reshape(2048, 2).astype(np.int64)
t0 = st[:, 0].min()
start, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0      # microseconds
dur = end - start
print("kernel ms (events)", ms)
print("wave start us: min %.1f max %.1f" % (start.min(), start.max()))
print("wave end   us: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % (end.min(), np.percentile(end, 10), np.median(end), np.percentile(end, 90), end.max()))
print("wave life  us: min %.1f median %.1f max %.1f" % (dur.min(), np.median(dur), dur.max()))
byxcd = [np.median(end[np.arange(2048) // 8 % 8 == x]) for x in range(8)]
print("median end by blockIdx %% 8:", " ".join("%.0f" % v for v in byxcd))
cyc = d_out.to_numpy(np.uint32, 2048, (frames + 2) * nfft * 4).astype(np.float64)
print("in-kernel clock GHz (shader cycles / wave life): median %.3f min %.3f max %.3f; cycles per frame %.0f"
      % (np.median(cyc / dur) / 1e3, (cyc / dur).min() / 1e3, (cyc / dur).max() / 1e3, np.median(cyc) / (frames / 2048.0)))
w = np.arange(2048) % 8
for k in range(8):
    e = end[w == k]
    print("wave %d of its workgroup: end us min %.0f median %.0f max %.0f" % (k, e.min(), np.median(e), e.max()))
