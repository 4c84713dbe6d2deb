"""CPU tests: the oracle (oracle/sdr_oracle.c) against the golden vectors that
tests/golden/make_golden.py produced from the reference's own functions (oracle/_ref).
Bit-exact in float64 -- this is what pins the oracle."""
import hashlib

import numpy as np
import pytest


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8).tobytes()


def test_unpack_kat(oracle, golden):
    iq = golden["unpack_in"]
    got = oracle.unpack_scale_i16(iq)
    assert bits(got) == bits(golden["unpack_out"])
    # bit-exact integer unpack: value * 2048 returns the int16s
    assert np.array_equal(np.rint(got.view(np.float64) * 2048).astype(np.int16), iq)


def test_fir_decimate_grid(oracle, golden):
    cases = golden["fir_cases"]
    assert len(cases) == 60
    for ci, (T, D, N) in enumerate(cases):
        h, iq = golden[f"fir{ci}_h"], golden[f"fir{ci}_iq"]
        assert h.size == T and iq.size == 2 * N
        y = oracle.chain_i16(iq, int(D), h, oracle.MODE_IQ)
        assert y.size == N // D                                  # bladerf_apt_demod.cpp:282
        assert bits(y) == bits(golden[f"fir{ci}_y"]), (T, D, N)
        fm = oracle.chain_i16(iq, int(D), h, oracle.MODE_FM)
        assert fm.size == N // D - 1                             # :409
        assert bits(fm) == bits(golden[f"fir{ci}_fm"]), (T, D, N)
        env = oracle.chain_i16(iq, int(D), h, oracle.MODE_ENV)
        assert bits(env) == bits(golden[f"fir{ci}_env"]), (T, D, N)


def test_polyphase_restriction(oracle):
    x = np.ones(100, np.complex128)
    with pytest.raises(ValueError):                              # :257-259
        oracle.polyphase_decimate(x, 10, np.ones(64))
    with pytest.raises(ValueError):                              # :254-255
        oracle.polyphase_decimate(x, 0, np.ones(64))
    # padded taps == unrestricted form
    h = np.linspace(-1, 1, 64)
    a = oracle.polyphase_decimate(x, 10, oracle.pad_taps(h, 10))
    b = oracle.fir_decimate(x, 10, h)
    assert bits(a) == bits(b)


def test_decimation_is_causal_fir_sampled(oracle):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(997) + 1j * rng.standard_normal(997)
    h = rng.standard_normal(33)
    full = np.convolve(x, h)[: x.size]
    for D in (1, 3, 10):
        y = oracle.fir_decimate(x, D, h)
        assert y.size == x.size // D
        assert np.allclose(y, full[::D][: y.size], rtol=0, atol=1e-12)
        p = oracle.fir_pick(x, D, h)                             # af::fir + af::seq twin
        assert p.size == (x.size - 1) // D + 1
        assert np.allclose(p[: y.size], y, rtol=0, atol=1e-12)


def test_history_halo(oracle):
    rng = np.random.default_rng(4)
    x = rng.standard_normal(2000) + 1j * rng.standard_normal(2000)
    h = rng.standard_normal(64)
    whole = oracle.fir_decimate(x, 10, h)
    pre = 70
    part = oracle.fir_decimate(x[1000 - pre:], 10, h, pre=pre)
    assert np.allclose(part, whole[100:], rtol=0, atol=1e-13)


def test_cfg1_fm_chain(oracle, golden):
    n, fs, ftone, dev, amp, D = golden["cfg1_params"]
    iq = oracle.synth_fm_i16(int(n), fs=fs, ftone=ftone, dev=dev, amp=amp)
    assert hashlib.sha256(iq.tobytes()).digest() == golden["cfg1_iq_sha256"].tobytes()
    assert np.array_equal(iq[:512], golden["cfg1_iq_head"])
    fm = oracle.chain_i16(iq, int(D), golden["cfg1_taps"], oracle.MODE_FM)
    assert fm.size == int(golden["cfg1_fm_len"][0]) == int(n) // int(D) - 1
    assert bits(fm[:256]) == bits(golden["cfg1_fm_head"])
    assert bits(fm[-256:]) == bits(golden["cfg1_fm_tail"])
    assert bits(fm[::9973]) == bits(golden["cfg1_fm_stride"])
    sums = np.array([fm.sum(), np.abs(fm).sum(), (fm * fm).sum()])
    assert bits(sums) == bits(golden["cfg1_fm_sum"])
    # physics: 5 kHz deviation at fs/10 = 100 kHz -> peak 0.05 cycles/sample
    assert abs(np.abs(fm[1000:]).max() - 0.05) < 1e-4
    env = oracle.chain_i16(iq[: 2 * 1_000_000], int(D), golden["cfg1_taps"], oracle.MODE_ENV)
    assert bits(env[::997]) == bits(golden["cfg1_env_stride"])


def test_am_chain(oracle, golden):
    env = oracle.chain_i16(golden["am_iq"], 20, golden["am_taps"], oracle.MODE_ENV)
    assert bits(env) == bits(golden["am_env"])
    fm = oracle.chain_i16(golden["am_iq"], 5, golden["am_taps"], oracle.MODE_FM, scale=0.75)
    assert bits(fm) == bits(golden["am_fm"])


@pytest.mark.parametrize("pfx,h1,h2", [("apt", "apt_lpf_fm", "apt_lpf_am"),
                                       ("apt511", "apt511_h1", "apt511_h2")])
def test_apt_two_stage(oracle, golden, pfx, h1, h2):
    x9, x6 = oracle.apt_chain_i16(golden["apt_iq"], 10, golden[h1], 2400 / 62400.0, 15,
                                  golden[h2])
    assert bits(x6) == bits(golden[pfx + "_x6"])
    assert bits(x9) == bits(golden[pfx + "_x9"])
    assert x9.size == (golden["apt_iq"].size // 2 // 10 - 1) // 15


def test_helpers(oracle, golden):
    assert bits(oracle.frequency_shift(golden["fs_real_in"], 2400 / 62400.0)) == \
        bits(golden["fs_real_out"])
    assert bits(oracle.frequency_shift(golden["fs_cplx_in"], -0.0371)) == \
        bits(golden["fs_cplx_out"])
    assert bits(oracle.create_freq_rotation(2048, 0.0123)) == bits(golden["rot_out"])
    for k in range(4):
        rate = float(golden[f"dec{k}_rate"][0])
        res, idx, oob = oracle.decimate_vec(golden["fs_real_in"], rate)
        assert oob == 0
        assert bits(res) == bits(golden[f"dec{k}_out"])
    for T in (7, 64, 101):
        got = oracle.filter_vec(golden["fs_real_in"], golden[f"filt{T}_h"])
        assert bits(got) == bits(golden[f"filt{T}_out"])
    assert bits(oracle.polar_discriminator(golden["disc_in"], 1 / (2 * np.pi))) == \
        bits(golden["disc_out"])


def test_discriminator_signed_zero(oracle, golden):
    got = oracle.polar_discriminator(golden["disc_zero_in"], 1 / (2 * np.pi))
    assert bits(got) == bits(golden["disc_zero_out"])
    # the corner itself: y[m] = 0 and both parts of y[m+1] negative gives +pi, i.e. 0.5
    assert got[0] == 0.5


def test_fft_unpinned(oracle, golden):
    """The reference's FFT is ArrayFire (absent): PARITY UNPINNED.  The oracle's DFT is checked
    against numpy.fft on the same int16 input, and the dB/shift/scale conventions against
    bladerf_main_v1.cpp:139,191-196."""
    iq = golden["fft_iq"]
    for name in ("rect", "blackman"):
        w = None if name == "rect" else oracle.window("blackman", 4096)
        db, lin = oracle.fft_power_db_i16(iq, 4096, 2048, w, linear=True)
        want = golden[f"fft_{name}_pow_unpinned"]
        assert lin.shape == want.shape == (4, 4096)
        assert np.allclose(lin, want, rtol=1e-9, atol=1e-22)
        assert np.allclose(db, 10 * np.log10(want), rtol=0, atol=1e-7)
    # bin-centred tone at +300 lands at shifted index 2048+300 with |X| = A/2048
    db = oracle.fft_power_db_i16(iq, 4096, 2048, None)
    assert np.argmax(db[0]) == 2048 + 300
    assert abs(db[0].max() - 20 * np.log10(1200 / 2048.0)) < 0.05


def test_windows_and_designer_unpinned(oracle):
    """DSP::create_fir_filter lives in davemers0160/Common (absent): PARITY UNPINNED.
    Checks of our documented definition only."""
    for name in oracle.WINDOWS:
        w = oracle.window(name, 65)
        assert np.allclose(w, w[::-1], atol=1e-15) and abs(w[32] - 1.0) < 1e-12
    assert oracle.window("hann", 64)[0] == 0.0
    for N in (64, 101, 200, 195, 511):
        h = oracle.create_fir_filter(N, 0.1, "hann")
        assert h.size == N and np.allclose(h, h[::-1], atol=1e-15)
        assert abs(h.sum() - 1.0) < 0.01                         # ~unity DC gain
        H = np.abs(np.fft.fft(h, 8192))
        assert H[int(0.02 * 8192)] > 0.95 and H[int(0.2 * 8192)] < 1e-2


def test_wire_format_unpackers(oracle):
    """SURVEY 8(f2) layouts: Q-first int16 and int8 pairs.  No reference function exists for them
    in the tree (the readers are in the un-vendored Common repository) -- parity unpinned; the
    known answers below pin the layout and the exact power-of-two scales."""
    qi = np.array([-2048, 2047, 1, -1, 32767, -32768], np.int16)        # (Q, I) pairs
    x = oracle.unpack_scale_i16_qi(qi)
    assert np.array_equal(x, np.array([2047 - 2048j, -1 + 1j, -32768 + 32767j]) / 2048.0)
    # the same samples, I first, through the reference-pinned unpacker
    iq = qi.reshape(-1, 2)[:, ::-1].reshape(-1)
    assert np.array_equal(x, oracle.unpack_scale_i16(iq))
    b = np.array([-128, 127, 0, -1, 64, 5], np.int8)
    y = oracle.unpack_scale_i8(b)
    assert np.array_equal(y, np.array([-128 + 127j, 0 - 1j, 64 + 5j]) / 128.0)


def _apt_envelope(n_lines, width=2080, jitter=0, seed=0):
    """synthetic 4160 Hz APT envelope: every line starts with the 7-cycle sync A pattern (two
    samples high, two low) on a dark porch, followed by a grey ramp and noise"""
    rng = np.random.default_rng(seed)
    x = []
    for ln in range(n_lines):
        row = 0.35 + 0.25 * np.linspace(0, 1, width) + 0.01 * rng.standard_normal(width)
        row[:4] = 0.1
        for c in range(7):
            row[4 + 4 * c: 6 + 4 * c] = 0.9
            row[6 + 4 * c: 8 + 4 * c] = 0.1
        row[32:39] = 0.1
        x.append(row[: width - (jitter if ln % 5 == 4 else 0)])
    return np.concatenate(x)


def _apt_sync_python(x10, width):
    """independent restatement of bladerf_apt_demod.cpp:758-787 + :181-213 (numpy correlation, the
    sequential peak logic in plain Python) -> (peak positions, image rows)"""
    pulse = np.array([-128] * 4 + [127, 127, -128, -128] * 7 + [-128] * 7, np.float64)
    assert pulse.size == 39
    lo, hi = x10.min(), x10.max()
    alpha = 255.0 / (hi - lo)
    v = x10 * alpha + (-lo * alpha)
    x12 = np.clip(np.rint(v - 128.0), -32768, 32767)
    x11 = np.clip(np.rint(v), 0, 255).astype(np.uint8)
    corr = np.correlate(x12, pulse, mode="valid") / 39.0
    peaks = [[0, 0.0]]
    index = 0
    while index <= x10.size - 39:
        if index - peaks[-1][0] > 2000:
            peaks.append([index, corr[index]])
            index += 500
        elif corr[index] > peaks[-1][1]:
            peaks[-1] = [index, corr[index]]
        index += 1
    rows = [x11[p: p + width] for p, _ in peaks[:-1] if p + width < x10.size]
    return [p for p, _ in peaks], rows


def test_apt_line_sync_oracle(oracle):
    """f3: the C oracle against an independent numpy/Python restatement of the same reference
    lines (OpenCV absent -> parity unpinned), over two blocks with carried-over samples."""
    x = _apt_envelope(12)
    s = oracle.AptSync(rows=16)
    added, peaks = s.push(x)
    want_peaks, want_rows = _apt_sync_python(x, 2080)
    assert list(peaks) == want_peaks and peaks[0] == 0
    assert added == len(want_rows) and added >= 4
    for k, row in enumerate(want_rows):
        assert np.array_equal(s.img[16 - added + k], row)
    drop = max(want_peaks[-1] - 500, 0)
    assert s.n10.value == x.size - drop
    # every detected line start sits on a sync pulse of the synthetic signal (multiples of 2080)
    assert all(p % 2080 == 0 for p in peaks[:-1])
    # second block continues with the carried-over samples
    x2 = _apt_envelope(6, seed=1)
    pending = np.concatenate([x[drop:], x2])
    added2, peaks2 = s.push(x2)
    want_peaks2, want_rows2 = _apt_sync_python(pending, 2080)
    assert list(peaks2) == want_peaks2 and added2 == len(want_rows2)
    for k, row in enumerate(want_rows2):
        assert np.array_equal(s.img[16 - added2 + k], row)
    assert s.lines == added + added2
    # fewer than 39 samples / constant input: nothing happens
    t = oracle.AptSync(rows=4)
    assert t.push(np.ones(20))[0] == 0 and t.push(np.ones(5000))[0] == 0
