#!/usr/bin/env python3
"""Launch APT stage 2 (real float input, +2400/62400 mixer folded into complex taps, 511 taps / 15,
envelope: fir_stream_kernel on real tiles) a few times, for rocprofv3 counter passes.
usage: run_stage2.py log2n [iters [hint]]   (log2n = samples of the stage's INPUT, e.g. 23 ~ cfg5's 6.7 M;
hint = 1 passes n as the size hint, a larger value that value: cfg5's 6710886 selects the K-split kernel)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import sdr_amd  # noqa: E402

log2n = int(sys.argv[1])
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
hint = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # 0 none, 1 = n, else the hint itself
sdr_amd.init(0)
n = 1 << log2n
rng = np.random.default_rng(1)
d_in = sdr_amd.DevBuf.from_numpy((rng.standard_normal(n) * 0.2).astype(np.float32))
h = sdr_amd.create_fir_filter(511, 2400 / 62400.0, "hann")
p = sdr_amd.FirPlan(h, 15, mode=sdr_amd.MODE_ENV, input=sdr_amd.IN_F32_REAL, mix_fr=2400 / 62400.0, size_hint=(1 << log2n) if hint == 1 else hint)
d_out = sdr_amd.DevBuf(n // 15 * 4 + 64)
p.time_dev(d_in, n, d_out, 40)
ms = min(p.time_dev(d_in, n, d_out, iters) for _ in range(3))
print("ms", ms, p.info())
