"""Randomised sweeps of the entry points around the FIR plans: FFT -> dB (random frame length, hop,
stream length, window), the block hand-off (random block sizes, zero-copy or copying push) and
batched channels (random channel count and strides).  Usage: python tools/fuzz_more.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import sdr_amd as sdr                      # noqa: E402
from oracle import oracle                  # noqa: E402
import fuzz_parity as fz                   # noqa: E402


def fft_case(rng, seed):
    nfft = int(rng.choice([64, 128, 256, 512, 1024, 2048, 4096, 4096, 8192, 16384]))
    hop = int(rng.choice([nfft // 2, nfft // 2, nfft, nfft // 4, int(rng.integers(1, 2 * nfft + 1))]))
    max_frames = max(2, 600_000 // nfft)
    frames = int(rng.choice([0, 1, 2, 3, int(rng.integers(1, max_frames))]))
    n = 0 if frames == 0 else nfft + (frames - 1) * hop + int(rng.integers(0, hop))
    if frames == 0:
        n = int(rng.integers(0, nfft))
    wname = rng.choice(["none", "hann", "blackman", "nuttall"])
    w = None if wname == "none" else oracle.window(str(wname), nfft)
    tag = f"fft seed={seed} nfft={nfft} hop={hop} n={n} frames={frames} win={wname}"
    kind = int(rng.integers(0, 3))
    if kind == 0:
        iq = rng.integers(-2048, 2048, 2 * n, dtype=np.int16)
    elif kind == 1:
        iq = oracle.synth_fm_i16(n, fs=1.0e6, ftone=2500.0, dev=60000.0, amp=1800.0) if n else np.zeros(0, np.int16)
        iq = (iq + rng.integers(-3, 4, iq.size)).astype(np.int16)
    else:
        iq = rng.integers(-32768, 32768, 2 * n, dtype=np.int16)
    plan = sdr.FftPlan(nfft, hop, w)
    assert plan.frame_count(n) == frames, tag
    got = plan.execute_host(iq).astype(np.float64)
    if frames == 0:
        assert got.size == 0, tag
        return
    want_db, want_lin = oracle.fft_power_db_i16(iq, nfft, hop, w, linear=True)
    assert got.shape == want_db.shape, tag
    lin = 10.0 ** (got / 10.0)
    peak = want_lin.max(axis=1, keepdims=True)
    assert np.abs(lin - want_lin).max() <= 1e-5 * peak.max(), tag + " linear"
    strong = want_lin > peak * 1e-5
    assert np.abs(got - want_db)[strong].max() <= 1e-3, tag + " dB"
    plan.destroy()


def stream_case(rng, seed):
    T = int(rng.choice([1, 8, 64, 101, 300, 511, int(rng.integers(1, 400))]))
    D = int(rng.choice([1, 2, 5, 10, 15, 20, int(rng.integers(1, 30))]))
    mode = int(rng.choice([sdr.MODE_IQ, sdr.MODE_FM, sdr.MODE_ENV]))
    kind = int(rng.choice([sdr.IN_I16_IQ, sdr.IN_I16_IQ, sdr.IN_I16_QI, sdr.IN_I8_IQ, sdr.IN_CF32, sdr.IN_F32_REAL]))
    mix = float(rng.choice([0.0, 0.0, 0.0371]))
    n = int(rng.integers(1, 120_000))
    max_block = int(rng.choice([64, 1000, 4096, 50_000]))
    h = fz.random_taps(rng, T, D)
    tag = f"stream seed={seed} T={T} D={D} mode={mode} in={kind} mix={mix} n={n} max_block={max_block}"
    if kind in (sdr.IN_I16_IQ, sdr.IN_I16_QI):
        raw = oracle.synth_fm_i16(n, fs=1.0e6, ftone=3000.0, dev=20000.0, amp=1900.0)
        raw = (raw + rng.integers(-40, 41, raw.size)).astype(np.int16)
        per = 2
    elif kind == sdr.IN_I8_IQ:
        raw, per = rng.integers(-128, 128, 2 * n, dtype=np.int8), 2
    elif kind == sdr.IN_CF32:
        raw, per = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.3).astype(np.complex64), 1
    else:
        raw, per = (rng.standard_normal(n) * 0.3).astype(np.float32), 1
    plan = sdr.FirPlan(h, D, mode=mode, input=kind, mix_fr=mix)
    whole = plan.execute_host(raw)
    st = sdr.FirStream(plan, max_block)
    pos, parts, in_flight = 0, [], 0
    while pos < n:
        blk = int(min(n - pos, rng.choice([max_block, max_block, 1, 2, 3, int(rng.integers(1, max_block + 1))])))
        piece = raw[per * pos: per * (pos + blk)]
        if rng.integers(0, 2):
            buf = st.next_buffer(raw.dtype, blk * per)
            buf[:] = piece
            st.push_buffer(buf, blk)
        else:
            st.push(piece)
        pos += blk
        in_flight += 1
        if in_flight == 2:
            parts.append(st.pop())
            in_flight -= 1
    while in_flight:
        parts.append(st.pop())
        in_flight -= 1
    got = np.concatenate(parts) if parts else np.zeros(0, whole.dtype)
    assert got.shape == whole.shape, tag + f" {got.shape} {whole.shape}"
    if got.size == 0:
        return
    # Bit-identical to the one-shot launch for the vector-unit kernels without mixer (one summation
    # order per output whatever the tiling).  An automatic plan on the int8 matrix kernel (exact integer
    # sums) hands launches with fewer resident samples than that kernel takes to a float kernel, and
    # the mixer's phasor is a product that starts at the tile's first sample, so there the blocks'
    # outputs are held to the tolerance (discriminator: to the oracle with the conditioned bound).
    loose = plan.info()["kernel"] == "mfma_i8" or mix != 0.0
    if not loose:
        assert np.array_equal(got, whole), tag + f" {np.abs(got.astype(np.complex128) - whole).max()}"
    elif mode != sdr.MODE_FM:
        assert np.abs(got - whole).max() <= 2e-6 * max(1.0, float(np.abs(whole).max())), tag
    elif kind in (sdr.IN_I16_IQ, sdr.IN_I16_QI):
        x = oracle.unpack_scale_i16_qi(raw) if kind == sdr.IN_I16_QI else oracle.unpack_scale_i16(raw)
        if mix != 0.0:
            x = oracle.frequency_shift(x, mix)
        y = oracle.fir_decimate(x, D, h)
        fz.check_fm(got, oracle.polar_discriminator(y, 1.0 / fz.TWO_PI), y, 1.0 / fz.TWO_PI, tag)
    st.destroy()
    plan.destroy()


def batched_case(rng, seed):
    T = int(rng.choice([16, 64, 101, 300, 511]))
    D = int(rng.choice([2, 5, 8, 10, 15, 20]))
    mode = int(rng.choice([sdr.MODE_IQ, sdr.MODE_FM, sdr.MODE_ENV]))
    n_ch = int(rng.choice([1, 2, 7, 37, 64, 200]))
    n = int(rng.choice([D * 3, 1000, 5120, int(rng.integers(1, 30_000))]))
    pipe = int(rng.choice([0, 0, 1, 2, 5]))
    s8 = bool(rng.integers(0, 3) == 0)                    # HackRF int8 pairs in a third of the cases
    if s8 and pipe == 1:
        pipe = 2
    h = fz.random_taps(rng, T, D)
    tag = f"batched seed={seed} T={T} D={D} mode={mode} n_ch={n_ch} n={n} pipe={pipe} int8={s8}"
    try:
        plan = sdr.FirPlan(h, D, mode=mode, pipeline=pipe, input=sdr.IN_I8_IQ if s8 else sdr.IN_I16_IQ)
    except sdr.SdrError:
        assert pipe != 0, tag
        return
    cnt = plan.out_count(n)
    stride_in = n + int(rng.integers(0, 9))
    stride_out = cnt + int(rng.integers(0, 5))
    buf = (rng.integers(-128, 128, 2 * stride_in * n_ch, dtype=np.int8) if s8
           else rng.integers(-32768, 32768, 2 * stride_in * n_ch, dtype=np.int16))
    eb = 8 if mode == sdr.MODE_IQ else 4
    dt = np.complex64 if mode == sdr.MODE_IQ else np.float32
    d_in = sdr.DevBuf.from_numpy(buf)
    d_out = sdr.DevBuf(max(stride_out * n_ch, 1) * eb)
    try:
        plan.execute_batched_dev(d_in, n, stride_in, n_ch, d_out, stride_out)
    except sdr.SdrError as e:
        assert pipe == 5 and "resident samples" in str(e), tag + str(e)
        return
    sdr.sync()
    out = d_out.to_numpy(dt, stride_out * n_ch)
    if cnt == 0:
        return
    for c in sorted(set([0, n_ch - 1, int(rng.integers(0, n_ch))])):
        one = plan.execute_host(buf[2 * c * stride_in: 2 * (c * stride_in + n)])
        got = out[c * stride_out: c * stride_out + cnt]
        assert np.array_equal(got, one), tag + f" ch {c} kernel {plan.info()['kernel']}"
    plan.destroy()


def elementwise_case(rng, seed):
    """unpack / magnitude / frequency shift / PCM stage at random lengths and pointer offsets, outputs
    inside sentinel buffers"""
    n = int(rng.choice([1, 2, 3, 4, 5, 7, 255, 256, 257, 1023, int(rng.integers(1, 300_000))]))
    so, oo = int(rng.integers(0, 5)), int(rng.integers(0, 5))
    tag = f"elementwise seed={seed} n={n} so={so} oo={oo}"
    SENT = 0x7FC0DEAD

    def guarded(words_per_elem, n_elem):
        g = np.full((oo + n_elem + 6) * words_per_elem, SENT, np.uint32)
        return sdr.DevBuf.from_numpy(g), g.size

    def readback(d, size, w, n_elem, what):
        back = d.to_numpy(np.uint32, size)
        assert np.all(back[: oo * w] == SENT) and np.all(back[(oo + n_elem) * w:] == SENT), tag + " " + what + " wrote outside"
        return back[oo * w: (oo + n_elem) * w]

    iq = rng.integers(-32768, 32768, 2 * (n + so), dtype=np.int16)
    d_iq = sdr.DevBuf.from_numpy(iq)
    x = iq[2 * so:]
    # a2 unpack: bit exact
    d_o, size = guarded(2, n)
    sdr.capi.unpack_i16_dev(d_iq.at(4 * so), n, d_o.at(8 * oo))
    sdr.sync()
    got = readback(d_o, size, 2, n, "unpack").view(np.complex64)
    want = ((x[0::2].astype(np.float32) + 1j * x[1::2].astype(np.float32)) / np.float32(2048)).astype(np.complex64)
    assert got.tobytes() == want.tobytes(), tag + " unpack"
    # a6 magnitude
    d_o, size = guarded(1, n)
    sdr.capi.magnitude_i16_dev(d_iq.at(4 * so), n, d_o.at(4 * oo))
    sdr.sync()
    got = readback(d_o, size, 1, n, "magnitude").view(np.float32)
    fz.check_linear(got, oracle.magnitude_i16(x), tag + " magnitude")
    # a7 frequency shift, real or complex or bare rotation, at a stream offset
    fr = float(rng.choice([2400 / 62400.0, -0.0371, 0.25, 1e-4]))
    n0 = int(rng.choice([0, 1, 12345, 2 ** 31 + 5, 2 ** 40 + 3]))
    which = int(rng.integers(0, 3))
    k = np.arange(n, dtype=np.float64) + float(n0)
    rot = np.exp(2j * np.pi * ((fr * k) % 1.0))
    d_o, size = guarded(2, n)
    if which == 0:
        sdr.capi.frequency_shift_dev(None, True, n, n0, fr, d_o.at(8 * oo))
        want = rot
    elif which == 1:
        r = (rng.standard_normal(n + so) * 0.5).astype(np.float32)
        d_r = sdr.DevBuf.from_numpy(r)
        sdr.capi.frequency_shift_dev(d_r.at(4 * so), False, n, n0, fr, d_o.at(8 * oo))
        want = r[so:].astype(np.float64) * rot
    else:
        c = ((rng.standard_normal(n + so) + 1j * rng.standard_normal(n + so)) * 0.5).astype(np.complex64)
        d_c = sdr.DevBuf.from_numpy(c)
        sdr.capi.frequency_shift_dev(d_c.at(8 * so), True, n, n0, fr, d_o.at(8 * oo))
        want = c[so:].astype(np.complex128) * rot
    sdr.sync()
    got = readback(d_o, size, 2, n, "shift").view(np.complex64)
    # phase fr*(n0+n) is formed in double: its rounding grows with n0 (2^40 * 2^-53 turns)
    tol = 1e-5 + 2 * np.pi * abs(fr) * (n0 + n) * 2.3e-16
    assert np.abs(got - want).max() <= tol * max(1.0, float(np.abs(want).max())), tag + f" shift which={which} n0={n0} {np.abs(got - want).max()}"
    # f1 PCM stage: bit exact against the same float32 arithmetic
    a = (rng.standard_normal(n + so) * 10.0 ** rng.uniform(-3, 2)).astype(np.float32)
    if rng.integers(0, 8) == 0:
        a[:] = 0
    d_a = sdr.DevBuf.from_numpy(a)
    d_max = sdr.DevBuf(16)
    gb = np.full(oo + n + 13, 0xAB, np.uint8)
    d_p = sdr.DevBuf.from_numpy(gb)
    sdr.capi.audio_u8_dev(d_a.at(4 * so), n, d_max, d_p.at(oo))
    sdr.sync()
    back = d_p.to_numpy(np.uint8, gb.size)
    assert np.all(back[:oo] == 0xAB) and np.all(back[oo + n:] == 0xAB), tag + " pcm wrote outside"
    m = np.float32(np.abs(a[so:]).max())
    assert d_max.to_numpy(np.float32, 1)[0] == m, tag + " max"
    r32 = np.float32(1.0 / np.float64(m)) if m > 0 else np.float32(0)
    v = (a[so:] * r32 + np.float32(1.0)) * np.float32(30.0)
    want = v.astype(np.int32).astype(np.uint8)
    diff = back[oo: oo + n].astype(np.int32) - want.astype(np.int32)
    # the device contracts x*r + 1 into one fma: a code may differ by one where v sits on an integer
    assert np.abs(diff).max() <= 1 and np.count_nonzero(diff) <= max(2, n // 1000), tag + f" pcm {np.count_nonzero(diff)}"


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sdr.init(0)
    t0 = time.time()
    last = t0
    counts = {"fft": 0, "stream": 0, "batched": 0, "elementwise": 0}
    n_bad = 0
    i = 0
    cases = (("fft", fft_case), ("stream", stream_case), ("batched", batched_case), ("elementwise", elementwise_case))
    while time.time() - t0 < budget:
        seed = seed0 * 1_000_003 + i
        name, fn = cases[i % len(cases)]
        i += 1
        try:
            fn(np.random.default_rng(seed), seed)
            counts[name] += 1
        except AssertionError as e:
            n_bad += 1
            print("FAIL", str(e)[:400], flush=True)
        except sdr.SdrError as e:
            n_bad += 1
            print("ERROR", name, seed, str(e)[:300], flush=True)
        if time.time() - last > 30:
            last = time.time()
            print(f"... {i} cases, {n_bad} failures", flush=True)
    print(f"fuzz_more: {counts}, {n_bad} failures", flush=True)
    sys.exit(1 if n_bad else 0)


if __name__ == "__main__":
    main()
