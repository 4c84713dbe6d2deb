#!/usr/bin/env python3
"""Secondary benchmarks for the other BASELINE.json configs on one GPU (the headline metric is
bench.py).  One JSON line per config: kernel time from HIP events on the launch stream,
algorithmic bytes / time against the 8 TB/s HBM peak, and a spot parity check
against the CPU oracle.

  cfg3  1024 independent channels x 2^20 samples, 64 taps /10, FM            (4.4 B/sample)
  cfg4  4096-point Blackman FFT -> dB, hop 2048, 2^28 samples                (12 B/sample)
  cfg5  APT path: 511 taps /10 FM, +2400/62400 mix, 511 taps /15, envelope   (4.03 B/sample)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdr_amd  # noqa: E402
from oracle import oracle as O  # noqa: E402

PEAK = 8000.0


def timed(fn, iters):
    """kernel ms per launch: warm the clocks up first, as bench.py does -- batches of launches until two
    consecutive ones agree within 1 % (at most 2 s; a fresh process ramps for ~60 ms of load, which is
    hundreds of launches of a 70 us kernel) -- then the best of three event-timed runs of `iters`"""
    import time
    t0, prev = time.perf_counter(), None
    while time.perf_counter() - t0 < 2.0:
        cur = fn(max(20, iters))
        if prev is not None and abs(cur - prev) <= 0.01 * prev and time.perf_counter() - t0 > 0.1:
            break
        prev = cur
    return min(fn(iters) for _ in range(3))


def line(name, samples, ms, bytes_per_sample, extra):
    gbs = bytes_per_sample * samples / (ms * 1e-3) / 1e9
    out = {"config": name, "Msamples_per_s": round(samples / ms / 1e3, 1), "kernel_ms": round(ms, 4),
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK, "unit": "GB/s",
                        "frac": round(gbs / PEAK, 4), "algorithmic_bytes_per_sample": bytes_per_sample}}
    out.update(extra)
    print(json.dumps(out), flush=True)


def cfg3(iters):
    n_ch, n, D, T = 1024, 1 << 20, 10, 64
    h = sdr_amd.create_fir_filter(T, 1.0 / D, "hann")
    plan = sdr_amd.FirPlan(h, D, mode=sdr_amd.MODE_FM)
    d_in = sdr_amd.DevBuf(n_ch * n * 4)
    for c in range(n_ch):       # independent streams: per-channel tone 300..3000 Hz, seed = base + channel
        sdr_amd.capi.synth_fm_i16_dev(d_in.at(c * n * 4), n, fs=1.0e6, ftone=300.0 + 2700.0 * c / (n_ch - 1),
                                      dev=5000.0, noise_lsb=2, seed=100 + c)
    cnt = plan.out_count(n)
    d_out = sdr_amd.DevBuf(n_ch * cnt * 4)
    plan.execute_batched_dev(d_in, n, n, n_ch, d_out, cnt)
    sdr_amd.sync()
    worst = 0.0
    for c in (0, 511, 1023):
        iq = d_in.to_numpy(np.int16, 2 * n, c * n * 4)
        want = O.chain_i16(iq, D, h, O.MODE_FM)
        got = d_out.to_numpy(np.float32, cnt, c * cnt * 4).astype(np.float64)
        worst = max(worst, float(np.abs(got - want).max()))
    assert worst <= 5e-6, worst
    ms = timed(lambda k: plan.time_dev(d_in, n, d_out, k, in_stride=n, n_channels=n_ch, out_stride=cnt), iters)
    line("cfg3 channelizer 1024 x 2^20, 64 taps /10 FM", n_ch * n, ms, 4.0 + 4.0 / D,
         {"parity_max_abs_err": worst, "kernel": plan.info()})


def cfg4(iters):
    n, nfft, hop = 1 << 28, 4096, 2048
    w = sdr_amd.window("blackman", nfft)
    plan = sdr_amd.FftPlan(nfft, hop, w)
    d_in = sdr_amd.DevBuf(n * 4)
    sdr_amd.capi.synth_fm_i16_dev(d_in, n, fs=61.44e6, ftone=250e3, dev=2.0e6, amp=1200.0, noise_lsb=64, seed=3)
    frames = plan.frame_count(n)
    d_out = sdr_amd.DevBuf(frames * nfft * 4)
    assert plan.execute_dev(d_in, n, d_out, frames) == frames
    sdr_amd.sync()
    worst = 0.0
    for f0 in (0, frames // 2, frames - 4):
        iq = d_in.to_numpy(np.int16, 2 * (nfft + 3 * hop), f0 * hop * 4)
        want_db, want_lin = O.fft_power_db_i16(iq, nfft, hop, w, linear=True)
        got = d_out.to_numpy(np.float32, 4 * nfft, f0 * nfft * 4).reshape(4, nfft).astype(np.float64)
        lin = 10.0 ** (got / 10.0)
        worst = max(worst, float(np.abs(lin - want_lin).max() / want_lin.max()))
    assert worst <= 1e-5, worst
    ms = timed(lambda k: plan.time_dev(d_in, n, d_out, k), iters)
    line("cfg4 waterfall 4096-pt Blackman FFT -> dB, hop 2048, 2^28 samples", n, ms, 12.0,
         {"parity_max_rel_power_err": worst, "frames": frames})


def cfg5(iters):
    n, D1, D2, T = 1 << 26, 10, 15, 511
    h1 = sdr_amd.create_fir_filter(T, 20800 / 62400.0 / 5, "hann")
    h2 = sdr_amd.create_fir_filter(T, 2400 / 62400.0, "hann")
    fr = 2400 / 62400.0
    p1 = sdr_amd.FirPlan(h1, D1, mode=sdr_amd.MODE_FM)
    p2 = sdr_amd.FirPlan(h2, D2, mode=sdr_amd.MODE_ENV, input=sdr_amd.IN_F32_REAL, mix_fr=fr,
                         size_hint=p1.out_count(n))          # stage 2 sees a tenth of the block: short-stream geometry
    d_in = sdr_amd.DevBuf(n * 4)
    sdr_amd.capi.synth_fm_i16_dev(d_in, n, fs=624e3, ftone=2400.0, dev=17000.0, amp=1800.0, noise_lsb=4, seed=5)
    n6 = p1.out_count(n)
    n9 = p2.out_count(n6)
    d_x6, d_x9 = sdr_amd.DevBuf(n6 * 4), sdr_amd.DevBuf(n9 * 4)
    p1.execute_dev(d_in, n, d_x6, n6)
    p2.execute_dev(d_x6, n6, d_x9, n9)
    sdr_amd.sync()
    m = 1 << 20                                     # parity on the first 2^20 samples (zero pre-history)
    iq = d_in.to_numpy(np.int16, 2 * m)
    w9, w6 = O.apt_chain_i16(iq, D1, h1, fr, D2, h2)
    g6 = d_x6.to_numpy(np.float32, w6.size).astype(np.float64)
    g9 = d_x9.to_numpy(np.float32, w9.size).astype(np.float64)
    s6 = T // D1 + 2
    s9 = s6 // D2 + 2 + T // D2
    worst = max(float(np.abs(g6 - w6)[s6:].max()), float(np.abs(g9 - w9)[s9:-T // D2 - 2].max()))
    assert worst <= 5e-6, worst
    ms1 = timed(lambda k: p1.time_dev(d_in, n, d_x6, k), iters)
    ms2 = timed(lambda k: p2.time_dev(d_x6, n6, d_x9, k), iters)
    line("cfg5 APT 511 taps /10 FM -> mix -> 511 taps /15 envelope, 2^26 samples", n, ms1 + ms2,
         4.0 + 4.0 / (D1 * D2), {"stage1_ms": round(ms1, 4), "stage2_ms": round(ms2, 4),
                                 "fp32_flops_per_sample": round(4.0 * T / D1 + 4.0 * T / (D1 * D2), 1),
                                 "achieved_tflops": round((4.0 * T / D1 + 4.0 * T / (D1 * D2)) * n / ((ms1 + ms2) * 1e-3) / 1e12, 2),
                                 "parity_max_abs_err": worst, "stage1_kernel": p1.info(), "stage2_kernel": p2.info()})


def handoff(_iters):
    """PCIe-inclusive rate of the streaming hand-off (host block in, host outputs back)."""
    import time
    nb, blk, D, T = 16, 1 << 24, 10, 64
    h = sdr_amd.create_fir_filter(T, 1.0 / D, "hann")
    plan = sdr_amd.FirPlan(h, D, mode=sdr_amd.MODE_FM)
    iq = O.synth_fm_i16(blk, fs=20e6, ftone=1000.0, dev=75e3)
    st = sdr_amd.FirStream(plan, blk)
    st.push(iq); st.pop()                           # warm-up (pinned pages, first launch)
    t0 = time.perf_counter()
    st.push(iq)
    got = 0
    for _ in range(nb - 1):
        st.push(iq)
        got += st.pop().size
    got += st.pop().size
    dt = time.perf_counter() - t0
    # zero-copy: the producer writes into the slot's pinned buffer (here: a memcpy stands in for the radio)
    st2 = sdr_amd.FirStream(plan, blk)
    b = st2.next_buffer(np.int16, 2 * blk); b[:] = iq; st2.push_buffer(b, blk); st2.pop()
    bufs = []
    for _ in range(2):
        b = st2.next_buffer(np.int16, 2 * blk); b[:] = iq; st2.push_buffer(b, blk); bufs.append(b)
    st2.pop(); st2.pop()
    t0 = time.perf_counter()
    st2.push_buffer(st2.next_buffer(np.int16, 2 * blk), blk)
    for _ in range(nb - 1):
        st2.push_buffer(st2.next_buffer(np.int16, 2 * blk), blk)
        st2.pop()
    st2.pop()
    dt2 = time.perf_counter() - t0
    print(json.dumps({"config": "hand-off, zero-copy: 16 blocks of 2^24 samples already in the slots' pinned buffers "
                                "(H2D, kernel, D2H)", "Msamples_per_s": round(nb * blk / dt2 / 1e6, 1),
                      "host_GBps_in": round(nb * blk * 4 / dt2 / 1e9, 2)}), flush=True)
    print(json.dumps({"config": "hand-off: 16 host blocks of 2^24 samples through sdr_fir_stream (pinned staging, "
                                "H2D, kernel, D2H), 64 taps /10 FM", "Msamples_per_s": round(nb * blk / dt / 1e6, 1),
                      "host_GBps_in": round(nb * blk * 4 / dt / 1e9, 2), "outputs": got}), flush=True)


if __name__ == "__main__":
    sdr_amd.init(0)
    which = sys.argv[1:] or ["cfg3", "cfg4", "cfg5", "handoff"]
    for w in which:
        {"cfg3": cfg3, "cfg4": cfg4, "cfg5": cfg5, "handoff": handoff}[w](20)
