#!/usr/bin/env python3
"""Soak of the int8 matrix kernel on small decimation factors with several tiles per workgroup (the shapes of the
round-3 packed-float32 failure, tests/test_gpu_parity.py::test_matrix_kernel_many_short_tiles): repeated launches against
the oracle, mismatches tabulated by (workgroup, round, column block, quarter of the block, output parity)."""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdr_amd as S
from oracle import oracle as O
S.init(0)
def run(T, D, n, reps=200, pipe=5, mode=S.MODE_FM):
    h = O.create_fir_filter(T, min(0.8 / D, 0.9), "hann")
    rng = np.random.default_rng(T * 100 + D)
    iq = O.synth_fm_i16(n, fs=1.0e6, ftone=3000.0, dev=20000.0, amp=1900.0)
    iq = (iq + rng.integers(-40, 41, iq.size)).astype(np.int16)
    want = O.chain_i16(iq, D, h, mode)
    p5 = S.FirPlan(h, D, mode=mode, pipeline=pipe)
    info = p5.info()
    per_tile = 1020 if info["outputs_per_lane"] == 4 else 508
    hist = collections.Counter()
    bad_runs = 0
    for it in range(reps):
        got = p5.execute_host(iq).astype(np.float64)
        d = np.abs(got - want); d = np.minimum(d, 1.0 - d) if mode == S.MODE_FM else d
        bad = np.flatnonzero(d > 2e-5)
        if bad.size:
            bad_runs += 1
            for b in bad:
                t, o = divmod(int(b), per_tile)
                hist[(t % 256 if True else t, t // 256, o // 128, (o % 128) // 32, o % 2)] += 1
    print(f"T={T} D={D} n={n} {info}: bad runs {bad_runs}/{reps}; (wg, round, block, quarter, parity) -> count, top:", hist.most_common(12), "rounds", collections.Counter(k[1] for k in hist.elements()), "blocks", collections.Counter(k[2] for k in hist.elements()), "quarters", collections.Counter(k[3] for k in hist.elements()), "parity", collections.Counter(k[4] for k in hist.elements()), flush=True)
run(7, 1, 600_000)
run(7, 1, 1_200_000, 100)
run(16, 2, 1_200_000, 100)
run(33, 3, 2_000_000, 60)
run(64, 4, 3_000_000, 60)
run(64, 10, 6_000_000, 40)
run(7, 1, 600_000, 100, mode=S.MODE_ENV)
