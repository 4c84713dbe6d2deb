import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdr_amd as sdr
from oracle import oracle
sdr.init(0)
for T, D in ((64, 10), (101, 5), (128, 8)):
    rng = np.random.default_rng(T * 13 + D)
    n_ch, n = 11, 60_000 + 7 * D + 3
    frs = np.concatenate(([0.0, 0.25, -0.5], rng.uniform(-0.5, 0.5, n_ch - 3)))
    h = oracle.create_fir_filter(T, 0.4 / D, "hann")
    iq = rng.integers(-2047, 2048, 2 * n).astype(np.int16)
    xs = oracle.unpack_scale_i16(iq)
    d_in = sdr.DevBuf.from_numpy(iq)
    for pipe in (0, 2):
        plan = sdr.FirPlan(h, D, mode=sdr.MODE_IQ, mix_fr_channels=frs, pipeline=pipe)
        cnt = plan.out_count(n)
        d_out = sdr.DevBuf(cnt * n_ch * 8)
        plan.execute_channels_dev(d_in, n, 0, 0, n_ch, d_out, cnt)
        sdr.sync()
        out = d_out.to_numpy(np.complex64, cnt * n_ch).reshape(n_ch, cnt)
        errs = []
        for c in range(n_ch):
            y = oracle.fir_decimate(oracle.frequency_shift(xs, frs[c]), D, h)
            errs.append(float(np.abs(out[c] - y).max()))
        print(T, D, plan.info()["kernel"], "max |err| per channel:", " ".join(f"{e:.1e}" for e in errs), "max|y|", float(np.abs(y).max()))
