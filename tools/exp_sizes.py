"""launch time against launch size for one FIR shape: t = a + b*n (a = the kernel's fixed cost)
usage: exp_sizes.py T D [pipeline]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import sdr_amd as S
from bench_configs import timed
S.init(0)
T, D = int(sys.argv[1]), int(sys.argv[2])
pipe = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n_max = 1 << 28
d_in = S.DevBuf(n_max * 4 + 64)
S.capi.synth_fm_i16_dev(d_in, n_max, fs=624e3, ftone=2400.0, dev=17000.0, noise_lsb=4)
d_out = S.DevBuf(n_max // D * 4 + 64)
h = S.create_fir_filter(T, 1.0 / D, "hann")
p = S.FirPlan(h, D, mode=S.MODE_FM, pipeline=pipe)
print(p.info())
rows = []
for l2 in (12, 16, 20, 22, 24, 25, 26, 27, 28):
    n = 1 << l2
    ms = timed(lambda k: p.time_dev(d_in, n, d_out, k), 20)
    rows.append((n, ms))
    print(f"2^{l2}: {ms * 1e3:9.2f} us   {n / ms / 1e3:10.0f} MS/s", flush=True)
(n1, t1), (n2, t2) = rows[-3], rows[-1]
b = (t2 - t1) / (n2 - n1)
print(f"slope {b * (1 << 26) * 1e3:.2f} us per 2^26 samples, intercept {(t1 - b * n1) * 1e3:.2f} us")
