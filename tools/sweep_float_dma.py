#!/usr/bin/env python3
"""Float input without mixer: the persistent float-tile kernel (pipeline 2) against the DMA-filled one-tile-per-workgroup
kernel (pipeline 7) over geometries, ms per launch.  usage: sweep_float_dma.py [log2n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import sdr_amd as S  # noqa: E402
from bench_configs import timed  # noqa: E402

S.init(0)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 27)
d_i16 = S.DevBuf(n * 4 + 64)
S.capi.synth_fm_i16_dev(d_i16, n, fs=20e6, ftone=1000.0, dev=5000.0, noise_lsb=8)
d_cf = S.DevBuf(n * 8 + 64)
S.capi.unpack_i16_dev(d_i16, n, d_cf)
S.sync()
d_out = S.DevBuf((n // 2 + 64) * 8)
SHAPES = [(64, 10), (101, 5), (64, 4), (32, 2), (128, 8), (128, 16), (200, 10), (195, 15), (101, 20), (300, 7), (511, 10), (511, 15)]
for kind, inp, ib, mode in (("cf32", S.IN_CF32, 8, S.MODE_FM), ("real", S.IN_F32_REAL, 4, S.MODE_ENV)):
    print(f"--- {kind}, 2^{n.bit_length() - 1} samples, mode {mode}")
    for (T, D) in SHAPES:
        h = S.create_fir_filter(T, 1.0 / D, "hann")
        p = S.FirPlan(h, D, mode=mode, input=inp, pipeline=2)
        ms2 = timed(lambda k: p.time_dev(d_cf, n, d_out, k), 10)
        i2 = p.info()
        p.destroy()
        cells = []
        best = None
        for R in (1, 2, 4):
            for NT in (64, 128, 256):
                try:
                    q = S.FirPlan(h, D, mode=mode, input=inp, pipeline=7, outputs_per_lane=R, threads=NT)
                except S.SdrError:
                    cells.append(f"R{R}/{NT}: -")
                    continue
                ms = timed(lambda k: q.time_dev(d_cf, n, d_out, k), 10)
                lds = q.info()["lds_bytes"]
                q.destroy()
                cells.append(f"R{R}/{NT}: {ms:.4f} ({lds // 1024}K)")
                if best is None or ms < best[0]:
                    best = (ms, R, NT)
        q = S.FirPlan(h, D, mode=mode, input=inp, pipeline=7)
        msa = timed(lambda k: q.time_dev(d_cf, n, d_out, k), 10)
        ia = q.info()
        q.destroy()
        tb = lambda ms: (ib + 4.0 / D) * n / (ms * 1e-3) / 1e12
        print(f"{T:4d}/{D:<3d} stream R{i2['outputs_per_lane']}/{i2['threads']} {ms2:.4f} ms ({tb(ms2):.2f} TB/s) | dma auto R{ia['outputs_per_lane']}/{ia['threads']} {msa:.4f} | "
              f"best R{best[1]}/{best[2]} {best[0]:.4f} ({tb(best[0]):.2f} TB/s, x{ms2 / best[0]:.2f}) | " + "  ".join(cells), flush=True)
