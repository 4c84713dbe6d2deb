#!/usr/bin/env python3
"""The channelizer with per-channel NCOs (sdr_fir_execute_channels_dev): 1024 channels, 64 taps / 10, FM,
 (a) from 1024 separate streams of 2^20 samples (BASELINE configs[2] layout, 4 GiB of input), and
 (b) from ONE shared wideband stream of 2^20 .. 2^24 samples (every channel reads the same buffer).
One JSON line each: kernel ms (HIP events, warm), input Msamples/s counted per channel, and the ceiling
that applies -- (a) HBM bytes 4.4 B per sample, or the matrix pipe if that fraction is the larger; (b) the input
is read from HBM once (it then lives in L2 / Infinity Cache), so only the outputs are HBM traffic and the bound
is the arithmetic: int8 matrix operations issued per second against the dense int8 rate (matrix kernel), or
useful float32 operations against the vector peak (float-tile kernel).
usage: bench_channelizer.py [separate] [shared20] [shared22] [shared24] [nomix]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import sdr_amd  # noqa: E402
from oracle import oracle as O  # noqa: E402
import parity_checks as PC  # noqa: E402
from bench_configs import timed, MFMA_I8_PEAK_TOPS  # noqa: E402

N_CH, D, T = 1024, 10, 64
FP32_PEAK_TFLOPS = 157.3


def freqs():
    # channel centres spread over +-0.45 of the sample rate (never exactly 0: every channel is really mixed)
    return (np.arange(N_CH) - N_CH / 2 + 0.37) * (0.9 / N_CH)


def run(name, n, shared, mix=True, emit=True):
    h = sdr_amd.create_fir_filter(T, 1.0 / D, "hann")
    frs = freqs()
    plan = sdr_amd.FirPlan(h, D, mode=sdr_amd.MODE_FM, mix_fr_channels=frs if mix else None)
    cnt = plan.out_count(n)
    n_buf = 1 if shared else N_CH
    d_in = sdr_amd.DevBuf(n_buf * n * 4 + 64)
    for c in range(n_buf):
        sdr_amd.capi.synth_fm_i16_dev(d_in.at(c * n * 4), n, fs=1.0e6, ftone=300.0 + 2700.0 * c / N_CH, dev=5000.0,
                                      amp=1500.0, noise_lsb=200, seed=100 + c)
    d_out = sdr_amd.DevBuf(N_CH * cnt * 4 + 64)
    stride_in = 0 if shared else n

    def launch(k):
        import time
        ev = []
        if mix:
            # no library timing loop for this entry point: HIP events through torch-free ctypes would need more
            # plumbing, so time k launches between two device synchronisations (launches are >= 0.2 ms)
            sdr_amd.sync()
            t0 = time.perf_counter()
            for _ in range(k):
                plan.execute_channels_dev(d_in, n, stride_in, 0, N_CH, d_out, cnt)
            sdr_amd.sync()
            return (time.perf_counter() - t0) * 1e3 / k
        return plan.time_dev(d_in, n, d_out, k, in_stride=n, n_channels=N_CH, out_stride=cnt)

    launch(3)
    # parity: three channels against the oracle
    worst = 0.0
    for c in (0, 511, 1023):
        iq = d_in.to_numpy(np.int16, 2 * n, (0 if shared else c) * n * 4)
        x = O.unpack_scale_i16(iq)
        if mix:
            x = O.frequency_shift(x, frs[c])
        y = O.fir_decimate(x, D, h)
        got = d_out.to_numpy(np.float32, cnt, c * cnt * 4)
        PC.check_fm(got, O.polar_discriminator(y, 1 / (2 * np.pi)), y, 1 / (2 * np.pi), f"{name} channel {c}", good_rel=None)
    ms = timed(launch, 10)
    samples = N_CH * n
    out = {"config": name, "kernel": plan.info(), "kernel_ms": round(ms, 4), "Msamples_per_s": round(samples / ms / 1e3, 1),
           "channels": N_CH, "samples_per_channel": n, "shared_input": bool(shared)}
    gbs = 4.4 * samples / (ms * 1e-3) / 1e9
    kern = plan.info()["kernel"]
    hbm = {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
           "algorithmic_bytes_per_sample": 4.4}
    arith = None
    if kern == "mfma_i8":
        # issued int8 matrix work: per tile of 1020 stored outputs 8 column blocks x (k steps of the window) x
        # 7 byte products, 16*16*64*2 operations each
        tb = (T - 1 + 3) // 4 * 4
        if mix and D % 2 == 0:      # complex taps in row form: groups of eight outputs, steps of 32 window offsets x 2 components
            per_block = (tb + 7 * D + 1 + 31) // 32 * 7
        else:                       # steps of 64 window offsets; complex taps in column form run every product twice
            per_block = (tb + 15 * D + 1 + 63) // 64 * (14 if mix else 7)
        tiles = N_CH * ((cnt + 1019) // 1020)
        tops = tiles * 8 * per_block * 32768.0 / (ms * 1e-3) / 1e12
        arith = {"bound": "mfma", "achieved": round(tops, 1), "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s (int8, issued)",
                 "frac": round(tops / MFMA_I8_PEAK_TOPS, 4)}
    else:
        # float-tile kernel: complex-by-real multiply-adds that reach an output, plus a complex product per sample for the mixer
        flops = samples * (T / D) * 4 + (samples * 6 if mix else 0)
        tf = flops / (ms * 1e-3) / 1e12
        arith = {"bound": "fp32 vector", "achieved": round(tf, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s (useful)", "frac": round(tf / FP32_PEAK_TFLOPS, 4)}
    if shared:
        # the shared input crosses the HBM interface once and then lives in L2 / Infinity Cache: only the outputs are
        # HBM traffic, the arithmetic is the ceiling
        arith["hbm_GBps_actual"] = round((4.0 * n + 0.4 * samples) / (ms * 1e-3) / 1e9, 1)
        out["roofline"] = arith
    else:
        out["roofline"] = hbm if hbm["frac"] >= arith["frac"] else arith
        out["roofline_hbm"], out["roofline_arith"] = hbm, arith
    if emit:
        print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    sdr_amd.init(0)
    which = sys.argv[1:] or ["separate", "shared20", "shared22", "shared24", "nomix"]
    for w in which:
        if w == "separate":
            run("channelizer, 1024 per-channel NCOs, 1024 separate 2^20-sample streams", 1 << 20, False)
        elif w == "nomix":
            run("no mixer (round 2's cfg3): 1024 separate 2^20-sample streams", 1 << 20, False, mix=False)
        elif w.startswith("shared"):
            l2 = int(w[6:])
            run(f"channelizer, 1024 per-channel NCOs, one shared 2^{l2}-sample stream", 1 << l2, True)
