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
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sdr_amd  # noqa: E402
from oracle import oracle as O  # noqa: E402
import parity_checks as PC  # noqa: E402  (the bars of include/sdr_hip.h as assertions)

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


MFMA_I8_PEAK_TOPS = 5000.0     # dense int8 matrix rate, 2x the ~2.5 PF bf16 figure (MI355X_MICROARCH.md, matrix cores)


def line(name, samples, ms, bytes_per_sample, extra):
    gbs = bytes_per_sample * samples / (ms * 1e-3) / 1e9
    out = {"config": name, "Msamples_per_s": round(samples / ms / 1e3, 1), "kernel_ms": round(ms, 4),
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK, "unit": "GB/s",
                        "frac": round(gbs / PEAK, 4), "algorithmic_bytes_per_sample": bytes_per_sample}}
    out.update(extra)
    return out


def cpu_leg(fn, samples, budget_s, kind, cores, what):
    """time fn() (one pass over `samples` input samples on the host) for about budget_s seconds"""
    import time
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter() - t0
    reps = max(1, int(budget_s / max(t1, 1e-3)) - 1)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t = time.perf_counter() - t0
    return {"value": round(samples * reps / t / 1e6, 2), "unit": "Msamples/s", "cores": cores, "kind": kind,
            "sample": f"{reps} passes over {what}, {t:.1f} s"}


def cfg3(iters, cpu_seconds=0.0):
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
    out = line("cfg3 channelizer 1024 x 2^20, 64 taps /10 FM (BASELINE configs[2])", n_ch * n, ms, 4.0 + 4.0 / D,
               {"parity_max_abs_err": worst, "kernel": plan.info()})
    if cpu_seconds > 0:
        iq4 = d_in.to_numpy(np.int16, 2 * 4 * n)                # four channels
        scale = 1.0 / (2.0 * np.pi)
        if O.have_ref():
            out["cpu_baseline"] = cpu_leg(lambda: [O.ref_time_chain_i16(iq4[2 * c * n: 2 * (c + 1) * n], D, h, O.MODE_FM, scale, 1) for c in range(4)],
                                          4 * n, cpu_seconds, "reference", 1, "channels 0-3 (4 x 2^20 samples), the reference's own functions, float64")
        else:
            out["cpu_baseline"] = cpu_leg(lambda: [O.chain_i16(iq4[2 * c * n: 2 * (c + 1) * n], D, h, O.MODE_FM, scale) for c in range(4)],
                                          4 * n, cpu_seconds, "port", 1, "channels 0-3 (4 x 2^20 samples), oracle port, float64")
    return out


def cfg4(iters, cpu_seconds=0.0):
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
    out = line("cfg4 waterfall 4096-pt Blackman FFT -> dB, hop 2048, 2^28 samples (BASELINE configs[3])", n, ms, 12.0,
               {"parity_max_rel_power_err": worst, "frames": frames})
    if cpu_seconds > 0:
        m = 1 << 21
        iqc = d_in.to_numpy(np.int16, 2 * m)
        out["cpu_baseline"] = cpu_leg(lambda: O.fft_power_db_i16(iqc, nfft, hop, w), m, cpu_seconds, "port", 1,
                                      "the first 2^21 samples (1023 frames), oracle port: float64 radix-2 FFT standing in for "
                                      "af::fftInPlace (ArrayFire is not in the reference tree)")
    return out


def cfg5(iters, cpu_seconds=0.0):
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
    # x6 by the discriminator's bars (start-up transient included), x9 against the reference's second
    # stage on the GPU's own x6; the figure reported is the worst error past the transient
    y4 = O.fir_decimate(O.unpack_scale_i16(iq), D1, h1)
    PC.check_fm(g6.astype(np.float32), w6, y4, 1.0 / (2.0 * np.pi), "cfg5 x6")
    PC.check_linear(g9, PC.apt_stage2(O, g6, fr, D2, h2), "cfg5 x9")
    s6 = T // D1 + 2
    s9 = s6 // D2 + 2 + T // D2
    worst = max(float(np.abs(g6 - w6)[s6:].max()), float(np.abs(g9 - w9)[s9:-T // D2 - 2].max()))
    assert worst <= 5e-6, worst
    ms1 = timed(lambda k: p1.time_dev(d_in, n, d_x6, k), iters)
    ms2 = timed(lambda k: p2.time_dev(d_x6, n6, d_x9, k), iters)
    ms = ms1 + ms2
    out = line("cfg5 APT 511 taps /10 FM -> mix -> 511 taps /15 envelope, 2^26 samples (BASELINE configs[4])", n, ms,
               4.0 + 4.0 / (D1 * D2), {"stage1_ms": round(ms1, 4), "stage2_ms": round(ms2, 4),
                                       "parity_max_abs_err": worst, "stage1_kernel": p1.info(), "stage2_kernel": p2.info()})
    # second ceiling: stage 1 runs on the int8 matrix pipe.  Issued work: per tile of 1020 stored outputs
    # 8 column blocks x (k steps of the window) x 7 byte products, 16*16*64*2 operations each -- Toeplitz
    # padding and the byte split included, so this is pipe occupancy, not useful arithmetic
    if p1.info()["kernel"] == "mfma_i8":
        nks = (((T - 1 + 3) // 4 * 4) + 15 * D1 + 1 + 63) // 64
        tiles = (n6 + 1019) // 1020
        ops = tiles * 8 * nks * 7 * 32768.0
        tops = ops / (ms1 * 1e-3) / 1e12
        hbm = out["roofline"]
        mf = {"bound": "mfma", "achieved": round(tops, 1), "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s (int8, issued)",
              "frac": round(tops / MFMA_I8_PEAK_TOPS, 4), "of": "stage 1 alone", "mfma_per_tile": 8 * nks * 7}
        out["roofline_hbm"] = hbm
        out["roofline_mfma"] = mf
        out["roofline"] = mf if mf["frac"] > hbm["frac"] else hbm       # the ceiling the chain is nearer to
    if cpu_seconds > 0:
        mc = 1 << 21
        iqc = d_in.to_numpy(np.int16, 2 * mc)
        if O.have_ref():
            out["cpu_baseline"] = cpu_leg(lambda: O.ref_apt_chain_i16(iqc, D1, h1, fr, D2, h2), mc, cpu_seconds, "reference", 1,
                                          "the first 2^21 samples, the reference's own functions (taps zero-padded to a multiple of D), float64")
        else:
            out["cpu_baseline"] = cpu_leg(lambda: O.apt_chain_i16(iqc, D1, h1, fr, D2, h2), mc, cpu_seconds, "port", 1,
                                          "the first 2^21 samples, oracle port, float64")
    return out


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
    cpu_s = float(os.environ.get("BENCH_CONFIGS_CPU_SECONDS", "0"))
    for w in which:
        if w == "handoff":
            handoff(20)
        else:
            print(json.dumps({"cfg3": cfg3, "cfg4": cfg4, "cfg5": cfg5}[w](20, cpu_s)), flush=True)
