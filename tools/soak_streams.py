#!/usr/bin/env python3
"""Stress of the int8 matrix kernel's dynamic tile order: one plan launched over and over on ten HIP streams
at once (the plan has eight counter blocks: two streams fall back to the static order), a second plan
(int8 sample pairs) interleaved on the same streams, launch sizes on both sides of the dynamic threshold.
Every output must equal, bit for bit, the one a lone launch produced."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdr_amd  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sdr_amd.init(0)
hip = C.CDLL("libamdhip64.so")
rng = np.random.default_rng(5)
D, T = 10, 64
n_big = 380_000_123                      # > 128 tiles per workgroup on 256 CUs: dynamic order
sizes = [n_big, n_big - 77_777, 40_000_003, 1 << 20]
h = sdr_amd.create_fir_filter(T, 0.1, "hann")
p16 = sdr_amd.FirPlan(h, D, mode=sdr_amd.MODE_FM)
p8 = sdr_amd.FirPlan(h, D, mode=sdr_amd.MODE_ENV, input=sdr_amd.IN_I8_IQ)
assert p16.info()["kernel"] == "mfma_i8" and p8.info()["kernel"] == "mfma_i8"
d_in = sdr_amd.DevBuf(n_big * 4 + 64)
sdr_amd.capi.synth_fm_i16_dev(d_in, n_big, fs=20e6, ftone=1000.0, dev=5000.0, amp=2000.0, noise_lsb=5, seed=2)
streams = []
for _ in range(10):
    s = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(s)) == 0
    streams.append(s)
# lone launches: the expected bits (int8 plan reads the same bytes as 2n int8 pairs)
want = {}
for plan, tag, per in ((p16, "i16", 1), (p8, "i8", 2)):
    for n in sizes:
        nn = n * per if per == 1 else min(n * 2, n_big * 2)
        cnt = plan.out_count(nn)
        d = sdr_amd.DevBuf(cnt * 4 + 64)
        assert plan.execute_dev(d_in, nn, d, cnt) == cnt
        sdr_amd.sync()
        want[(tag, n)] = (nn, cnt, d.to_numpy(np.uint32, cnt))
outs = [sdr_amd.DevBuf(max(w[1] for w in want.values()) * 4 + 64) for _ in streams]
bad = 0
launches = 0
for r in range(rounds):
    jobs = []
    for si in rng.permutation(len(streams)):
        tag = "i16" if rng.integers(0, 3) else "i8"
        n = sizes[int(rng.integers(0, len(sizes)))]
        nn, cnt, _ = want[(tag, n)]
        plan = p16 if tag == "i16" else p8
        assert plan.execute_dev(d_in, nn, outs[si], cnt, stream=streams[si].value) == cnt
        jobs.append((si, tag, n, cnt))
        launches += 1
    sdr_amd.sync()
    for si, tag, n, cnt in jobs:
        got = outs[si].to_numpy(np.uint32, cnt)
        if not np.array_equal(got, want[(tag, n)][2]):
            bad += 1
            print("MISMATCH round", r, "stream", si, tag, n, int((got != want[(tag, n)][2]).sum()), "words differ", flush=True)
    print("round", r, "ok" if not bad else "BAD", flush=True)
for s in streams:
    hip.hipStreamDestroy(s)
print(f"soak_streams: {launches} launches on {len(streams)} streams, {bad} mismatches")
sys.exit(1 if bad else 0)
