"""Re-run fuzz cases by seed and show where FM outputs differ: python tools/fuzz_repro.py seed..."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import sdr_amd as sdr
from oracle import oracle
import fuzz_parity as fz
import test_gpu_parity as tp

def fake_check_fm(got, want, y, scale, what=""):
    fs = abs(scale) * np.pi
    d = np.abs(got.astype(np.float64) - want); d = np.minimum(d, np.abs(2 * fs - d))
    mag = np.abs(y); pm = np.minimum(mag[:-1], mag[1:])
    allowed = abs(scale) * 2 * 1e-5 * max(1.0, mag.max()) / np.maximum(pm, 1e-300) + 1e-5 * fs
    bad = np.nonzero(d > allowed)[0]
    print(what, "n_out", got.size, "bad", bad.size, "max|y|", mag.max(), "first bad", bad[:12], "mod512", bad[:12] % 512, "d", d[bad[:6]], "pm", pm[bad[:6]])
    if bad.size:
        i = bad[0]
        print("   got", got[i-1:i+3], "want", want[i-1:i+3], "y", y[i:i+2])
fz.check_fm = fake_check_fm
sdr.init(0)
for s in sys.argv[1:]:
    rng = np.random.default_rng(int(s))
    try:
        print(fz.one_case(rng, int(s)))
    except Exception as e:
        print("EXC", str(e)[:300])
