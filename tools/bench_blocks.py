#!/usr/bin/env python3
"""One capture block through the drop-in boundary, PCIe included, next to the reference's CPU chain on
the same block: examples/bin/block_latency (C++, the shims and the hand-off) + the reference's own
functions (oracle/_ref, one thread) for the FM receiver's 131072-sample block and the APT receiver's
1 248 000-sample block.  One JSON line per path; the CPU lines carry kind "reference" / "port"."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def cpu(name, fn, n, kind):
    fn()
    ts = []
    t_end = time.perf_counter() + 3.0
    while time.perf_counter() < t_end and len(ts) < 50:
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2] * 1e6
    print(json.dumps({"chain": name, "path": "cpu_" + kind, "block_samples": n, "median_us": round(med, 1),
                      "min_us": round(ts[0] * 1e6, 1), "Msamples_per_s": round(n / med, 2), "cores": 1, "kind": kind}), flush=True)


def main():
    exe = os.path.join(ROOT, "examples", "bin", "block_latency")
    cp = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    sys.stdout.write(cp.stdout)
    if cp.returncode != 0:
        sys.stderr.write(cp.stderr)
        sys.exit(cp.returncode)
    scale = 1.0 / (2.0 * np.pi)
    kind = "reference" if O.have_ref() else "port"
    # FM block: 101 taps (zero-padded to 105 for the reference's T % D rule), decimate by 5, discriminator
    n = 131072
    iq = O.synth_fm_i16(n, fs=1.0e6, ftone=1000.0, dev=60000.0)
    h = O.create_fir_filter(101, 0.1, "hann")
    if O.have_ref():
        cpu("fm 131072 x (101 taps / 5 + discriminator)", lambda: O.ref_chain_i16(iq, 5, h, O.MODE_FM, scale), n, kind)
    else:
        cpu("fm 131072 x (101 taps / 5 + discriminator)", lambda: O.chain_i16(iq, 5, h, O.MODE_FM, scale), n, kind)
    # APT block
    n = 1_248_000
    iq = O.synth_fm_i16(n, fs=624e3, ftone=2400.0, dev=17000.0)
    h1 = O.create_fir_filter(200, 20800 / 62400.0 / 5, "hann")
    h2 = O.create_fir_filter(195, 2400 / 62400.0, "hann")
    fr = 2400 / 62400.0
    name = "apt 1248000 x (200 taps / 10 + discriminator -> shift -> 195 taps / 15 -> abs)"
    if O.have_ref():
        cpu(name, lambda: O.ref_apt_chain_i16(iq, 10, h1, fr, 15, h2), n, kind)
    else:
        cpu(name, lambda: O.apt_chain_i16(iq, 10, h1, fr, 15, h2), n, kind)


if __name__ == "__main__":
    main()
