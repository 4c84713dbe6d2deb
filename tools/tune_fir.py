#!/usr/bin/env python3
"""Time every FIR kernel geometry (outputs per lane x workgroup size) for a filter shape on the
GPU, with HIP events around back-to-back launches.  Usage: tune_fir.py [log2n] [T D mode]..."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdr_amd  # noqa: E402


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    shapes = []
    rest = sys.argv[2:]
    while len(rest) >= 3:
        shapes.append((int(rest[0]), int(rest[1]), int(rest[2])))
        rest = rest[3:]
    if not shapes:
        shapes = [(64, 10, 1), (511, 10, 1), (200, 10, 1), (101, 5, 1), (101, 20, 2)]
    sdr_amd.init(0)
    n = 1 << log2n
    d_in = sdr_amd.DevBuf(n * 4 + 64)
    sdr_amd.capi.synth_fm_i16_dev(d_in, n, fs=20e6, noise_lsb=8)
    d_out = sdr_amd.DevBuf(n * 8 // 1 + 64)
    sdr_amd.sync()
    for (T, D, mode) in shapes:
        h = sdr_amd.create_fir_filter(T, 1.0 / D, "hann")
        bps = 4.0 + (8.0 if mode == 0 else 4.0) / D
        for R in (1, 2, 4, 8):
            for NT in (64, 128, 256):
                for pipe in [int(x) for x in os.environ.get('PIPES', '1,2,3').split(',')]:
                    try:
                        p = sdr_amd.FirPlan(h, D, mode=mode, outputs_per_lane=R, threads=NT, pipeline=pipe)
                    except sdr_amd.SdrError:
                        continue
                    if pipe >= 2 and p.info()['lds_bytes'] > 160 * 1024:
                        continue
                    p.time_dev(d_in, n, d_out, 15)
                    ms = min(p.time_dev(d_in, n, d_out, 8) for _ in range(2))
                    print(f"T={T} D={D} mode={mode} R={R} NT={NT} pipe={pipe} lds={p.info()['lds_bytes']:6d} "
                          f"ms={ms:8.4f} MS/s={n / ms / 1e3:10.0f} GB/s={bps * n / ms / 1e6:8.1f}", flush=True)
                    p.destroy()


if __name__ == "__main__":
    main()
