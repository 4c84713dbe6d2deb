#!/usr/bin/env python3
"""Same-box A/B of builds of libsdrhip: this tree's sdr_amd/libsdrhip.so against other .so files (e.g. the
previous round's build kept as sdr_amd/libsdrhip_r02.so, or experiment builds), interleaved, one child
process per measurement (a process binds one library).  usage: ab_libs.py <a.so>[,<b.so>...] [rounds] [shapes]
Each child times a list of FIR shapes (FM unless stated) and prints JSON; the parent tabulates."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [  # log2n, T, D, mode, input, pipeline
    (30, 64, 10, 1, 0, 0), (28, 64, 10, 1, 0, 0), (28, 64, 10, 2, 0, 0), (28, 64, 10, 0, 0, 0), (26, 511, 10, 1, 0, 0), (28, 511, 10, 1, 0, 0),
    (28, 200, 10, 1, 0, 0), (28, 101, 5, 1, 0, 0), (28, 64, 4, 1, 0, 0), (28, 300, 25, 1, 0, 0), (28, 64, 10, 1, 4, 0), (28, 511, 10, 1, 4, 0),
    (28, 64, 10, 1, 0, 1),
    (28, "fft4096", 2048, 0, 0, 0),          # index 13: the cfg4 waterfall (4096-point Blackman frames, hop 2048) instead of a FIR
]


def child():
    sys.path.insert(0, ROOT)
    import sdr_amd
    sdr_amd.init(0)
    n_max = 1 << 30
    d_in = sdr_amd.DevBuf(n_max * 4 + 64)
    sdr_amd.capi.synth_fm_i16_dev(d_in, n_max, fs=20e6, noise_lsb=8)
    d_out = sdr_amd.DevBuf((n_max // 4 + 64) * 8)
    out = []
    pick = [int(x) for x in os.environ.get("AB_SHAPES", "").split(",") if x] or list(range(len(SHAPES)))
    for (l2, T, D, mode, inp, pipe) in [SHAPES[i] for i in pick]:
        n = 1 << l2
        if T == "fft4096":
            p = sdr_amd.FftPlan(4096, D, sdr_amd.window("blackman", 4096))
            frames = p.frame_count(n)
            d_db = sdr_amd.DevBuf(frames * 4096 * 4)
            for _ in range(6):
                p.time_dev(d_in, n, d_db, 20)
            ms = sorted(p.time_dev(d_in, n, d_db, 10) for _ in range(5))
            out.append({"shape": [l2, T, D, mode, inp, pipe], "ms_min": ms[0], "ms_med": ms[2], "kernel": "fft4096"})
            d_db.free()
            p.destroy()
            continue
        h = sdr_amd.create_fir_filter(T, 1.0 / D, "hann")
        p = sdr_amd.FirPlan(h, D, mode=mode, input=inp, pipeline=pipe)
        # warm the clocks: batches until two agree within 1 %
        prev = None
        for _ in range(12):
            cur = p.time_dev(d_in, n, d_out, 20)
            if prev is not None and abs(cur - prev) <= 0.01 * prev:
                break
            prev = cur
        ms = sorted(p.time_dev(d_in, n, d_out, 10) for _ in range(5))
        out.append({"shape": [l2, T, D, mode, inp, pipe], "ms_min": ms[0], "ms_med": ms[2], "kernel": p.info()["kernel"]})
        p.destroy()
    print("AB " + json.dumps(out), flush=True)


def main():
    """ab_libs.py <other.so>[,<other2.so>...] [rounds] [shape indices, comma separated]"""
    others = [os.path.abspath(x) for x in sys.argv[1].split(",")]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    pick = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(range(len(SHAPES)))
    libs = [("this", None)] + [(os.path.basename(o), o) for o in others]
    res = {name: [] for name, _ in libs}
    for r in range(rounds):
        order = libs if r % 2 == 0 else libs[::-1]
        for name, lib in order + order[::-1]:
            env = dict(os.environ)
            env["AB_SHAPES"] = ",".join(str(i) for i in pick)
            if lib:
                env["SDRHIP_LIB"] = lib
            else:
                env.pop("SDRHIP_LIB", None)
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=280)
            if cp.returncode != 0:
                print(cp.stdout[-2000:], cp.stderr[-2000:])
                sys.exit(cp.returncode or 1)
            line = [l for l in cp.stdout.splitlines() if l.startswith("AB ")][-1]
            res[name].append(json.loads(line[3:]))
    print("ms per launch, min over runs (median of medians); shape = (log2n T D mode input pipe)")
    print(f"{'shape':30s} {'kernel':9s} " + " ".join(f"{name[:22]:>22s}" for name, _ in libs))
    for k, i in enumerate(pick):
        cells = []
        base = None
        for name, _ in libs:
            mn = min(r[k]["ms_min"] for r in res[name])
            md = sorted(r[k]["ms_med"] for r in res[name])[len(res[name]) // 2]
            base = base or mn
            cells.append(f"{mn:8.4f} ({md:7.4f}) {mn / base:5.3f}")
        print(f"{str(SHAPES[i]):30s} {res['this'][0][k]['kernel']:9s} " + " ".join(cells), flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        main()
