#!/usr/bin/env python3
"""Every input kind x output mode on the automatic kernel choice, 2^27 samples, ms per launch and the
algorithmic GB/s (input bytes + output bytes): a table to spot a combination that falls out of line."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdr_amd
sdr_amd.init(0)
n = 1 << 27
d_in = sdr_amd.DevBuf(n * 8 + 64)
sdr_amd.capi.synth_fm_i16_dev(d_in, n, fs=20e6, noise_lsb=8)      # any finite bytes: int16 pairs of |v| < 2100 are small floats too
d_out = sdr_amd.DevBuf(n * 4 + 64)
KINDS = (("i16", sdr_amd.IN_I16_IQ, 4), ("qi16", sdr_amd.IN_I16_QI, 4), ("i8", sdr_amd.IN_I8_IQ, 2),
         ("cf32", sdr_amd.IN_CF32, 8), ("f32", sdr_amd.IN_F32_REAL, 4))
MODES = (("iq", sdr_amd.MODE_IQ, 8), ("fm", sdr_amd.MODE_FM, 4), ("env", sdr_amd.MODE_ENV, 4), ("real", sdr_amd.MODE_REAL, 4))
for (T, D) in ((64, 10), (200, 8), (511, 10), (64, 2)):
    h = sdr_amd.create_fir_filter(T, 1.0 / D, "hann")
    for kname, kind, ib in KINDS:
        row = []
        for mname, mode, ob in MODES:
            for mix in (0.0, 0.0371):
                if mix and (mode != sdr_amd.MODE_ENV):
                    continue
                try:
                    p = sdr_amd.FirPlan(h, D, mode=mode, input=kind, mix_fr=mix)
                except sdr_amd.SdrError as e:
                    row.append(f"{mname}{'+mix' if mix else ''}: refused"); continue
                p.time_dev(d_in, n, d_out, 20)
                ms = min(p.time_dev(d_in, n, d_out, 8) for _ in range(3))
                gbs = n * (ib + ob / D) / ms / 1e6
                row.append(f"{mname}{'+mix' if mix else ''}: {ms:.3f} ms {gbs:5.0f} GB/s {p.info()['kernel']}")
                p.destroy()
        print(f"{T}/{D} {kname:5s} | " + " | ".join(row), flush=True)
