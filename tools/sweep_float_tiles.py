"""Float-tile kernel (complex float input): every outputs-per-lane x workgroup-size geometry per shape,
sorted by time (us per 2^24 samples), next to the plan's automatic choice.  The slot-count penalty in
sdr_fir_plan_create's cost model comes from this sweep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdr_amd as sdr
sdr.init(0)
n = 1 << 24
rng = np.random.default_rng(1)
buf = sdr.DevBuf.from_numpy((rng.standard_normal(2 * n) * 0.3).astype(np.float32))
d_out = sdr.DevBuf(n * 8 + 64)
for (T, D) in ((511, 15), (511, 20), (200, 15), (101, 20), (300, 13), (511, 10), (300, 10), (200, 10), (128, 10), (64, 10), (64, 20), (32, 16), (200, 8), (164, 6), (101, 5), (300, 4), (255, 3), (64, 2)):
    h = sdr.create_fir_filter(T, 1.0 / D, "hann")
    row = []
    for R in (1, 2, 4):
        for NT in (64, 128, 256):
            try:
                p = sdr.FirPlan(h, D, mode=sdr.MODE_ENV, input=sdr.IN_CF32, outputs_per_lane=R, threads=NT)
            except sdr.SdrError:
                continue
            p.time_dev(buf, n, d_out, 10)
            ms = min(p.time_dev(buf, n, d_out, 10) for _ in range(3))
            row.append(("%d/%d/%dK" % (R, NT, p.info()["lds_bytes"] // 1024), round(ms * 1000, 1)))
            p.destroy()
    p = sdr.FirPlan(h, D, mode=sdr.MODE_ENV, input=sdr.IN_CF32)
    info = p.info()
    print(f"T={T} D={D} auto={info['outputs_per_lane']}/{info['threads']}", sorted(row, key=lambda r: r[1]), flush=True)
