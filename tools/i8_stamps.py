#!/usr/bin/env python3
"""Per-role phase clocks of fir_mfma_i8_kernel from a measurement build (SDR_I8_ABLATE bit 8): shader
cycles per tile that each role works / is parked at the barrier.
usage: SDRHIP_LIB=<build> i8_stamps.py [log2n [T D [int8|int16 [mix_fr]]]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdr_amd  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
T, D = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (511, 10)
s8 = len(sys.argv) > 4 and sys.argv[4] == "int8"
mix = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
sdr_amd.init(0)
print(sdr_amd.capi.build_info())
n = 1 << log2n
d_in = sdr_amd.DevBuf(n * (2 if s8 else 4) + 64)
sdr_amd.capi.synth_fm_i16_dev(d_in, n // 2 if s8 else n, fs=20e6, noise_lsb=8)
n_out = n // D - 1
d_out = sdr_amd.DevBuf((n_out + 64 + 8 * 8 * 256 + 64) * 4)
h = sdr_amd.create_fir_filter(T, 1.0 / D, "hann")
p = sdr_amd.FirPlan(h, D, mode=1, pipeline=5, input=sdr_amd.IN_I8_IQ if s8 else sdr_amd.IN_I16_IQ, mix_fr=mix)
print("T", T, "D", D, "int8 samples" if s8 else "int16 samples", p.info())
p.time_dev(d_in, n, d_out, 40)
ms = p.time_dev(d_in, n, d_out, 1)
st = d_out.to_numpy(np.uint32, 8 * 8 * 256, (n_out + 64) * 4).reshape(256, 8, 8).astype(np.float64)
assert (st[:, :, 3] == 0x51).all(), "no stamps: not a bit-8 measurement build?"
ph = st[:, :, 2]
print("kernel ms", ms, "phases per workgroup", ph.min(), ph.max())
for name, sl in (("matrix", slice(0, 4)), ("helper", slice(4, 8))):
    busy = (st[:, sl, 0] / ph[:, sl]).ravel()
    wait = (st[:, sl, 1] / ph[:, sl]).ravel()
    print("%-9s cycles per phase: working median %.0f (p10 %.0f p90 %.0f)  at the barrier median %.0f (p10 %.0f p90 %.0f)"
          % (name, np.median(busy), np.percentile(busy, 10), np.percentile(busy, 90),
             np.median(wait), np.percentile(wait, 10), np.percentile(wait, 90)))
sec = st[:, 4:8, 4:7] / ph[:, 4:8, None]
print("helper sections, cycles per phase (median): staging %.0f  request %.0f" % tuple(np.median(sec[:, :, i]) for i in range(2)))
print("matrix waves, cycles per phase (median): finishing %.0f" % np.median(st[:, 0:4, 6] / ph[:, 0:4]))
for w in range(8):
    print("wave %2d working %.0f waiting %.0f" % (w, np.median(st[:, w, 0] / ph[:, w]), np.median(st[:, w, 1] / ph[:, w])))
