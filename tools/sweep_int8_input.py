#!/usr/bin/env python3
"""int8 sample pairs (HackRF): float-tile kernel (pipeline 2) against the int8 matrix kernel (pipeline 5)
across filter shapes, 2^28 samples, FM, ms per launch; the automatic choice's model for this input comes
from this sweep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdr_amd
sdr_amd.init(0)
n = 1 << 28
d_in = sdr_amd.DevBuf(n * 2 + 64)
sdr_amd.capi.synth_fm_i16_dev(d_in, n // 2, fs=20e6, noise_lsb=8)      # any bytes will do
d_out = sdr_amd.DevBuf((n // 2) * 4 + 64)                # room for D = 2: n/2 floats
PIPES = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (2, 5, 0)
SHAPES = ((64, 10), (64, 4), (32, 2), (16, 2), (101, 5), (128, 8), (200, 10), (195, 15), (300, 7), (255, 4), (400, 12), (511, 10), (511, 15), (101, 20), (300, 25), (511, 20), (64, 32))
assert all(D >= 2 for _, D in SHAPES)                    # the output buffer above is sized for that
for (T, D) in SHAPES:
    h = sdr_amd.create_fir_filter(T, 1.0 / D, "hann")
    row = []
    for pipe in PIPES:
        try:
            p = sdr_amd.FirPlan(h, D, mode=sdr_amd.MODE_FM, input=sdr_amd.IN_I8_IQ, pipeline=pipe)
        except sdr_amd.SdrError as e:
            row.append((pipe, None)); continue
        p.time_dev(d_in, n, d_out, 30)
        ms = min(p.time_dev(d_in, n, d_out, 8) for _ in range(3))
        row.append((pipe, round(ms, 4), p.info()["kernel"]))
        p.destroy()
    print(f"T={T} D={D} T/D={T/D:.1f}", row, flush=True)
