"""FIR kernels across filter shapes (2^28 samples, FM): raw tile | planar | int8 matrix, ms per launch.
The table in DESIGN.md 3.4 and the cost model in sdr_fir_plan_create come from this sweep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdr_amd
sdr_amd.init(0)
n = 1 << 28
d_in = sdr_amd.DevBuf(n * 4 + 64)
sdr_amd.capi.synth_fm_i16_dev(d_in, n, fs=20e6, noise_lsb=8)
d_out = sdr_amd.DevBuf(n * 4 + 64)
PIPES = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 3, 5)
for (T, D) in ((64, 10), (64, 4), (32, 2), (101, 5), (101, 20), (128, 8), (128, 16), (200, 10), (195, 15), (300, 7), (255, 4), (300, 25), (400, 12), (511, 10), (511, 15), (700, 4)):
    h = sdr_amd.create_fir_filter(T, 1.0 / D, "hann")
    row = []
    for pipe in PIPES:
        try:
            p = sdr_amd.FirPlan(h, D, mode=sdr_amd.MODE_FM, pipeline=pipe)
        except sdr_amd.SdrError as e:
            row.append((pipe, None)); continue
        p.time_dev(d_in, n, d_out, 20)
        ms = min(p.time_dev(d_in, n, d_out, 8) for _ in range(3))
        row.append((pipe, round(ms, 4), p.info()["lds_bytes"] if pipe == 5 else 0, p.info()["outputs_per_lane"] if pipe == 5 else 0))
        p.destroy()
    print(f"T={T} D={D} T/D={T/D:.1f}", row, flush=True)
