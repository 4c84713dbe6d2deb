#!/usr/bin/env python3
"""Every input kind the boundary takes, on the headline shape (64 taps / 10) and the long-filter shape
(511 taps / 10), 2^28 samples, automatic kernel choice: ms per launch, algorithmic TB/s (input bytes per
sample + output bytes per sample / D) and the fraction of the 8 TB/s HBM peak.  VERDICT r2 #6.
usage: sweep_inputs.py [log2n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import sdr_amd as S  # noqa: E402
from bench_configs import timed  # noqa: E402

S.init(0)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
d_i16 = S.DevBuf(n * 4 + 64)
S.capi.synth_fm_i16_dev(d_i16, n, fs=20e6, ftone=1000.0, dev=5000.0, noise_lsb=8)
d_cf = S.DevBuf(n * 8 + 64)
S.capi.unpack_i16_dev(d_i16, n, d_cf)            # complex float samples = the scaled int16 stream; as real floats: 2n samples
S.sync()
d_out = S.DevBuf((n // 10 + 64) * 8)
fr = 0.0371
KINDS = [  # name, input, bytes per sample, buffer, mode, out bytes, extra plan arguments
    ("int16 IQ", S.IN_I16_IQ, 4, d_i16, S.MODE_FM, 4, {}),
    ("int16 IQ, envelope", S.IN_I16_IQ, 4, d_i16, S.MODE_ENV, 4, {}),
    ("int16 IQ, complex out", S.IN_I16_IQ, 4, d_i16, S.MODE_IQ, 8, {}),
    ("int16 QI (B205)", S.IN_I16_QI, 4, d_i16, S.MODE_FM, 4, {}),
    ("int8 IQ (HackRF)", S.IN_I8_IQ, 2, d_i16, S.MODE_FM, 4, {}),
    ("int8 IQ (HackRF), envelope", S.IN_I8_IQ, 2, d_i16, S.MODE_ENV, 4, {}),
    ("complex float", S.IN_CF32, 8, d_cf, S.MODE_FM, 4, {}),
    ("real float, envelope", S.IN_F32_REAL, 4, d_cf, S.MODE_ENV, 4, {}),
    ("int16 IQ + mixer", S.IN_I16_IQ, 4, d_i16, S.MODE_FM, 4, {"mix_fr": fr}),
    ("int16 IQ + per-channel mixer (1 channel)", S.IN_I16_IQ, 4, d_i16, S.MODE_FM, 4, {"mix_fr_channels": [fr]}),
    ("complex float + mixer", S.IN_CF32, 8, d_cf, S.MODE_FM, 4, {"mix_fr": fr}),
    ("real float + mixer, envelope", S.IN_F32_REAL, 4, d_cf, S.MODE_ENV, 4, {"mix_fr": fr}),
]
print(f"2^{n.bit_length() - 1} samples per launch; ms = best of three event-timed runs of 10 after the clocks have settled")
for (T, D) in ((64, 10), (511, 10)):
    h = S.create_fir_filter(T, 1.0 / D, "hann")
    print(f"--- {T} taps / {D}")
    for name, kind, ib, buf, mode, ob, extra in KINDS:
        p = S.FirPlan(h, D, mode=mode, input=kind, **extra)
        if "mix_fr_channels" in extra:
            import time

            def run(k, p=p, buf=buf):
                S.sync()
                t0 = time.perf_counter()
                for _ in range(k):
                    p.execute_channels_dev(buf, n, 0, 0, 1, d_out, n // D)
                S.sync()
                return (time.perf_counter() - t0) * 1e3 / k
        else:
            def run(k, p=p, buf=buf):
                return p.time_dev(buf, n, d_out, k)
        ms = timed(run, 10)
        bps = ib + ob / D
        tbs = bps * n / (ms * 1e-3) / 1e12
        info = p.info()
        print(f"{name:44s} {info['kernel']:9s} {ms:8.4f} ms {n / ms / 1e3:10.0f} MS/s {tbs:6.2f} TB/s {tbs / 8.0:6.3f} of peak   ({bps:.1f} B/sample)", flush=True)
        p.destroy()
