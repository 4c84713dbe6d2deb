"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the golden vectors from the reference's own functions and against the CPU oracle on the
same seeded inputs.

Tolerances (BASELINE north_star: bit-exact int16 unpack and decimation indexing; <= 1e-5 of
full scale on float32 outputs):
  IQ / envelope : |err| <= 1e-5 * FS, FS = max(1, max|expected|)   (Q11 full scale = 1.0)
  discriminator : |err| <= 1e-5 * FS, FS = scale*pi (0.5 for scale 1/(2 pi)), compared modulo one
                  turn; outputs whose y[m] or y[m+1] is below 1e-3 of the block rms are
                  ill-conditioned angles and get 1e-3*FS; outputs that are exact zeros / +-pi in
                  the reference (y exactly zero) must match exactly.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TWO_PI = 2.0 * np.pi


@pytest.fixture(scope="module")
def sdr():
    import sdr_amd
    sdr_amd.capi.lib()
    sdr_amd.init(0)
    return sdr_amd


from parity_checks import check_fm, check_linear, apt_stage2  # noqa: E402  (tests/ is on sys.path under pytest)


def test_unpack_bit_exact(sdr, golden):
    iq = golden["unpack_in"]
    plan = sdr.FirPlan(np.ones(1), 1, mode=sdr.MODE_IQ)
    got = plan.execute_host(iq)
    want = golden["unpack_out"].astype(np.complex64)
    assert got.tobytes() == want.tobytes()
    # and the stand-alone a2 kernel
    n = iq.size // 2
    d_in = sdr.DevBuf.from_numpy(iq)
    d_out = sdr.DevBuf(n * 8)
    sdr.capi.unpack_i16_dev(d_in, n, d_out)
    sdr.sync()
    assert d_out.to_numpy(np.complex64, n).tobytes() == want.tobytes()


def test_golden_grid(sdr, golden):
    cases = golden["fir_cases"]
    for ci, (T, D, N) in enumerate(cases):
        h, iq = golden[f"fir{ci}_h"], golden[f"fir{ci}_iq"]
        y = golden[f"fir{ci}_y"]
        tag = f"case {ci} T={T} D={D} N={N}"
        for mode, key in ((sdr.MODE_IQ, "y"), (sdr.MODE_ENV, "env"), (sdr.MODE_FM, "fm")):
            plan = sdr.FirPlan(h, int(D), mode=mode)
            got = plan.execute_host(iq)
            want = golden[f"fir{ci}_{key}"]
            assert got.size == plan.out_count(N)
            if mode == sdr.MODE_FM:
                check_fm(got, want, y, 1 / TWO_PI, tag)
            else:
                check_linear(got, want, tag + f" mode {mode}")
            plan.destroy()


def test_decimation_indexing_exact(sdr):
    """output m <-> input D*m, bit-exact: a 1-tap unity filter must return x[D*m]/2048."""
    rng = np.random.default_rng(5)
    for D in (1, 2, 3, 5, 10, 15, 20, 64):
        n = 100_003
        iq = rng.integers(-32768, 32768, 2 * n, dtype=np.int16)
        plan = sdr.FirPlan(np.ones(1), D, mode=sdr.MODE_IQ)
        got = plan.execute_host(iq)
        x = (iq[0::2].astype(np.float32) + 1j * iq[1::2].astype(np.float32)) / np.float32(2048)
        want = x[::D][: n // D].astype(np.complex64)
        assert got.tobytes() == want.tobytes(), D


def test_cfg1_fm_chain(sdr, oracle, golden):
    n, fs, ftone, dev, amp, D = golden["cfg1_params"]
    iq = oracle.synth_fm_i16(int(n), fs=fs, ftone=ftone, dev=dev, amp=amp)
    plan = sdr.FirPlan(golden["cfg1_taps"], int(D), mode=sdr.MODE_FM)
    got = plan.execute_host(iq).astype(np.float64)
    assert got.size == int(golden["cfg1_fm_len"][0])
    tol = 1e-5 * 0.5
    assert np.abs(got[:256] - golden["cfg1_fm_head"]).max() <= tol
    assert np.abs(got[-256:] - golden["cfg1_fm_tail"]).max() <= tol
    assert np.abs(got[::9973] - golden["cfg1_fm_stride"]).max() <= tol
    s = golden["cfg1_fm_sum"]
    assert abs(np.abs(got).sum() - s[1]) <= tol * got.size * 0.1
    # whole output against the oracle
    want = oracle.chain_i16(iq, int(D), golden["cfg1_taps"], oracle.MODE_FM)
    assert np.abs(got - want).max() <= tol
    env = sdr.FirPlan(golden["cfg1_taps"], int(D), mode=sdr.MODE_ENV).execute_host(iq[: 2 * 1_000_000])
    assert np.abs(env[::997].astype(np.float64) - golden["cfg1_env_stride"]).max() <= 1e-5


def test_am_chain(sdr, oracle, golden):
    got = sdr.FirPlan(golden["am_taps"], 20, mode=sdr.MODE_ENV).execute_host(golden["am_iq"])
    check_linear(got, golden["am_env"], "am env")
    got = sdr.FirPlan(golden["am_taps"], 5, mode=sdr.MODE_FM, scale=0.75).execute_host(golden["am_iq"])
    y = oracle.fir_decimate(oracle.unpack_scale_i16(golden["am_iq"]), 5, golden["am_taps"])
    check_fm(got, golden["am_fm"], y, 0.75, "am fm")       # the start-up transient included


@pytest.mark.parametrize("pfx,h1,h2", [("apt", "apt_lpf_fm", "apt_lpf_am"),
                                       ("apt511", "apt511_h1", "apt511_h2")])
def test_apt_two_stage(sdr, oracle, golden, pfx, h1, h2):
    """bladerf_apt_demod.cpp:728-754: FIR/10 + FM, then +2400/62400 shift, FIR/15, envelope.
    Stage 2 runs the fused real-input + NCO-mixer kernel on stage 1's device output."""
    iq = golden["apt_iq"]
    n = iq.size // 2
    p1 = sdr.FirPlan(golden[h1], 10, mode=sdr.MODE_FM)
    p2 = sdr.FirPlan(golden[h2], 15, mode=sdr.MODE_ENV, input=sdr.IN_F32_REAL,
                     mix_fr=2400 / 62400.0)
    d_iq = sdr.DevBuf.from_numpy(iq)
    n6 = p1.out_count(n)
    d_x6 = sdr.DevBuf(n6 * 4)
    assert p1.execute_dev(d_iq, n, d_x6, n6) == n6
    n9 = p2.out_count(n6)
    d_x9 = sdr.DevBuf(n9 * 4)
    assert p2.execute_dev(d_x6, n6, d_x9, n9) == n9
    sdr.sync()
    x6 = d_x6.to_numpy(np.float32, n6).astype(np.float64)
    x9 = d_x9.to_numpy(np.float32, n9).astype(np.float64)
    w6, w9 = golden[pfx + "_x6"], golden[pfx + "_x9"]
    assert x6.shape == w6.shape and x9.shape == w9.shape
    # stage 1 over the whole block, start-up transient included, by the discriminator's bars
    y4 = oracle.fir_decimate(oracle.unpack_scale_i16(iq), 10, golden[h1])
    check_fm(x6, w6, y4, 1 / TWO_PI, pfx + " x6")
    # stage 2 against the reference's arithmetic on the same x6, and against the golden x9 where the
    # transient's (legitimately different) angles have left the second filter
    check_linear(x9, apt_stage2(oracle, x6, 2400 / 62400.0, 15, golden[h2]), pfx + " x9")
    s9 = (golden[h1].size // 10 + 2) // 15 + 2 + golden[h2].size // 15
    assert np.abs(x9 - w9)[s9:].max() <= 1e-5 * 0.5


@pytest.mark.parametrize("T,D", [(64, 10), (101, 5), (101, 20), (200, 10), (195, 15), (511, 10),
                                 (511, 15), (7, 1), (33, 3), (64, 4)])
def test_geometries_against_oracle(sdr, oracle, T, D):
    """every outputs-per-lane / workgroup-size geometry the plan accepts, incl. padded tiles"""
    rng = np.random.default_rng(T * 100 + D)
    n = 300_017
    iq = oracle.synth_fm_i16(n, fs=1.0e6, ftone=3000.0, dev=20000.0, amp=1900.0)
    iq = (iq + rng.integers(-40, 41, iq.size)).astype(np.int16)
    h = oracle.create_fir_filter(T, min(0.8 / D, 0.9), "hann")
    y = oracle.chain_i16(iq, D, h, oracle.MODE_IQ)
    fm = oracle.chain_i16(iq, D, h, oracle.MODE_FM)
    env = oracle.chain_i16(iq, D, h, oracle.MODE_ENV)
    ran = 0
    for R in (1, 2, 4, 8):
        for NT in (64, 128, 256):
            for pipe in (1, 2, 3, 4):   # raw tile / persistent float tiles / planar float tile / MFMA f32
                try:
                    p_fm = sdr.FirPlan(h, D, mode=sdr.MODE_FM, outputs_per_lane=R, threads=NT,
                                       pipeline=pipe)
                except sdr.SdrError as e:
                    assert e.status == sdr.capi.EINVAL   # geometry not available (alignment / LDS)
                    continue
                if pipe >= 2 and p_fm.info()["lds_bytes"] > 160 * 1024:
                    continue
                ran += 1
                tag = f"T={T} D={D} R={R} NT={NT} pipe={pipe}"
                check_fm(p_fm.execute_host(iq), fm, y, 1 / TWO_PI, tag)
                check_linear(sdr.FirPlan(h, D, mode=sdr.MODE_ENV, outputs_per_lane=R, threads=NT,
                                         pipeline=pipe).execute_host(iq), env, tag)
                check_linear(sdr.FirPlan(h, D, mode=sdr.MODE_IQ, outputs_per_lane=R, threads=NT,
                                         pipeline=pipe).execute_host(iq), y, tag)
    assert ran >= 3
    # int8 matrix kernel: one geometry of its own
    tag = f"T={T} D={D} pipe=5"
    try:
        p5 = sdr.FirPlan(h, D, mode=sdr.MODE_FM, pipeline=5)
    except sdr.SdrError as e:
        assert e.status == sdr.capi.EINVAL and D >= 16    # two tile buffers of 16*D*31 samples exceed the LDS
        return
    check_fm(p5.execute_host(iq), fm, y, 1 / TWO_PI, tag)
    check_linear(sdr.FirPlan(h, D, mode=sdr.MODE_ENV, pipeline=5).execute_host(iq), env, tag)
    check_linear(sdr.FirPlan(h, D, mode=sdr.MODE_IQ, pipeline=5).execute_host(iq), y, tag)
    with pytest.raises(sdr.SdrError):
        sdr.FirPlan(h, D, mode=sdr.MODE_IQ, pipeline=5, threads=256)


def test_float_inputs_and_mixer(sdr, oracle):
    rng = np.random.default_rng(11)
    n = 120_000
    h = oracle.create_fir_filter(101, 0.08, "hann")
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.3
    xr = rng.standard_normal(n) * 0.3
    iq = rng.integers(-2048, 2048, 2 * n, dtype=np.int16)
    fr = 0.0371
    # complex float input, no mixer
    want = oracle.fir_decimate(x.astype(np.complex64).astype(np.complex128), 10, h)
    got = sdr.FirPlan(h, 10, mode=sdr.MODE_IQ, input=sdr.IN_CF32).execute_host(x)
    check_linear(got, want, "cf32")
    # complex float + mixer (frequency_shift then polyphase_decimate)
    xs = oracle.frequency_shift(x.astype(np.complex64).astype(np.complex128), fr)
    got = sdr.FirPlan(h, 10, mode=sdr.MODE_IQ, input=sdr.IN_CF32, mix_fr=fr).execute_host(x)
    check_linear(got, oracle.fir_decimate(xs, 10, h), "cf32+mix")
    # real float + mixer, envelope (APT stage 2 shape), negative frequency too
    for f in (fr, -0.123):
        xs = oracle.frequency_shift(xr.astype(np.float32).astype(np.float64), f)
        got = sdr.FirPlan(h, 15, mode=sdr.MODE_ENV, input=sdr.IN_F32_REAL, mix_fr=f).execute_host(xr)
        check_linear(got, oracle.envelope(oracle.fir_decimate(xs, 15, h)), "real+mix")
    # real float, no mixer
    got = sdr.FirPlan(h, 4, mode=sdr.MODE_IQ, input=sdr.IN_F32_REAL).execute_host(xr)
    check_linear(got, oracle.fir_decimate(xr.astype(np.float32).astype(np.complex128), 4, h), "real")
    # int16 + mixer (channeliser front end)
    xs = oracle.frequency_shift(oracle.unpack_scale_i16(iq), fr)
    got = sdr.FirPlan(h, 10, mode=sdr.MODE_IQ, mix_fr=fr).execute_host(iq)
    check_linear(got, oracle.fir_decimate(xs, 10, h), "i16+mix")


def test_time_block_sharding(sdr, oracle):
    """a long stream cut into blocks that carry their FIR history halo gives the same
    outputs as one pass (the reference instead restarts every block from zero state)."""
    n = 1_000_003
    D, T = 10, 64
    iq = oracle.synth_fm_i16(n, fs=2.0e6, ftone=1500.0, dev=30000.0)
    h = oracle.create_fir_filter(T, 0.08, "hann")
    for mode in (sdr.MODE_FM, sdr.MODE_ENV):
        plan = sdr.FirPlan(h, D, mode=mode)
        whole = plan.execute_host(iq)
        cnt = whole.size
        G = 3
        parts = []
        for gidx in range(G):
            ob, oe = cnt * gidx // G, cnt * (gidx + 1) // G
            lo = max(0, ob * D - (T - 1))
            hi = min(n, (oe + 1) * D + 1)
            d_blk = sdr.DevBuf.from_numpy(iq[2 * lo: 2 * hi])      # only the shard + halo is resident
            d_out = sdr.DevBuf((oe - ob) * 4)
            plan.execute_range_dev(d_blk, lo, hi - lo, n, ob, oe, d_out)
            sdr.sync()
            parts.append(d_out.to_numpy(np.float32, oe - ob))
        got = np.concatenate(parts)
        assert got.shape == whole.shape
        assert np.abs(got - whole).max() <= 2e-6
        assert (got == whole).mean() > 0.99
    # missing halo is refused
    with pytest.raises(sdr.SdrError) as e:
        d_blk = sdr.DevBuf.from_numpy(iq[2 * 5000: 2 * 9000])
        plan.execute_range_dev(d_blk, 5000, 4000, n, 500, 800, sdr.DevBuf(4096))
    assert e.value.status == sdr.capi.ERANGE


def test_batched_channels(sdr, oracle):
    """BASELINE config 3 shape: independent channels, channel-major, one launch"""
    n_ch, n = 37, 50_000
    D = 10
    h = oracle.create_fir_filter(64, 0.1, "hann")
    stride_in = n + 24
    buf = np.zeros(2 * stride_in * n_ch, np.int16)
    for c in range(n_ch):
        buf[2 * c * stride_in: 2 * (c * stride_in + n)] = oracle.synth_fm_i16(
            n, fs=1.0e6, ftone=300.0 + 73.0 * c, dev=5000.0)
    plan = sdr.FirPlan(h, D, mode=sdr.MODE_FM)
    cnt = plan.out_count(n)
    stride_out = cnt + 3
    d_in = sdr.DevBuf.from_numpy(buf)
    d_out = sdr.DevBuf(stride_out * n_ch * 4)
    plan.execute_batched_dev(d_in, n, stride_in, n_ch, d_out, stride_out)
    sdr.sync()
    out = d_out.to_numpy(np.float32, stride_out * n_ch)
    for c in (0, 1, 17, 36):
        want = oracle.chain_i16(buf[2 * c * stride_in: 2 * (c * stride_in + n)], D, h, oracle.MODE_FM)
        got = out[c * stride_out: c * stride_out + cnt].astype(np.float64)
        assert np.abs(got - want).max() <= 1e-5 * 0.5, c


def test_empty_and_tiny_inputs(sdr):
    h = np.ones(8) / 8
    for mode in (sdr.MODE_IQ, sdr.MODE_FM, sdr.MODE_ENV):
        plan = sdr.FirPlan(h, 10, mode=mode)
        assert plan.execute_host(np.zeros(0, np.int16)).size == 0
        assert plan.execute_host(np.zeros(2 * 9, np.int16)).size == 0          # n < D
        got = plan.execute_host(np.full(2 * 10, 100, np.int16))               # exactly one y
        assert got.size == (0 if mode == sdr.MODE_FM else 1)
        got = plan.execute_host(np.full(2 * 25, 100, np.int16))
        assert got.size == (1 if mode == sdr.MODE_FM else 2)
    plan = sdr.FirPlan(h, 10, mode=sdr.MODE_ENV)
    d_in, d_out = sdr.DevBuf(4000), sdr.DevBuf(40)
    with pytest.raises(sdr.SdrError) as e:
        plan.execute_dev(d_in, 1000, d_out, 10)                                # needs 100
    assert e.value.status == sdr.capi.ECAPACITY
    with pytest.raises(sdr.SdrError) as e:
        plan.execute_dev(sdr.capi.C.c_void_p(d_in.ptr.value + 2), 100, d_out, 10)
    assert e.value.status == sdr.capi.EALIGN


def test_discriminator_signed_zero_corner(sdr, oracle):
    """h[0] == 0 (Hann end point) makes y[0] exactly zero; the first discriminator output is then
    0 or +-pi*scale depending on the signs of y[1] (bladerf_apt_demod.cpp:419-421 as libstdc++
    evaluates it).  The kernel must reproduce that, not just be 'close'."""
    h = np.array([0.0, 1.0, 0.5, 0.25])
    for s1 in ((-5, -7), (5, -7), (-5, 7), (5, 7)):
        iq = np.zeros(2 * 64, np.int16)
        iq[0::2] = s1[0] * 100
        iq[1::2] = s1[1] * 100
        want = oracle.chain_i16(iq, 1, h, oracle.MODE_FM)
        got = sdr.FirPlan(h, 1, mode=sdr.MODE_FM).execute_host(iq)
        assert got[0] == np.float32(want[0]), (s1, got[0], want[0])
        assert np.abs(got[1:] - want[1:]).max() <= 5e-6


def test_elementwise_kernels(sdr, oracle, golden):
    rng = np.random.default_rng(2)
    n = 100_001
    iq = rng.integers(-32768, 32768, 2 * n, dtype=np.int16)
    d_iq = sdr.DevBuf.from_numpy(iq)
    d_mag = sdr.DevBuf(n * 4)
    sdr.capi.magnitude_i16_dev(d_iq, n, d_mag)
    sdr.sync()
    check_linear(d_mag.to_numpy(np.float32, n), oracle.magnitude_i16(iq), "magnitude")
    # frequency_shift on real and complex float data, and the bare rotation vector
    r = golden["fs_real_in"].astype(np.float32)
    d_r = sdr.DevBuf.from_numpy(r)
    d_o = sdr.DevBuf(r.size * 8)
    sdr.capi.frequency_shift_dev(d_r, False, r.size, 0, 2400 / 62400.0, d_o)
    sdr.sync()
    want = oracle.frequency_shift(r.astype(np.float64), 2400 / 62400.0)
    check_linear(d_o.to_numpy(np.complex64, r.size), want, "fs real")
    c = golden["fs_cplx_in"].astype(np.complex64)
    d_c = sdr.DevBuf.from_numpy(c)
    d_o = sdr.DevBuf(c.size * 8)
    sdr.capi.frequency_shift_dev(d_c, True, c.size, 0, -0.0371, d_o)
    sdr.sync()
    check_linear(d_o.to_numpy(np.complex64, c.size), oracle.frequency_shift(c.astype(np.complex128), -0.0371), "fs cplx")
    d_o = sdr.DevBuf(2048 * 8)
    sdr.capi.frequency_shift_dev(None, True, 2048, 0, 0.0123, d_o)
    sdr.sync()
    check_linear(d_o.to_numpy(np.complex64, 2048), golden["rot_out"], "rotation")
    # phase continuity at a large stream offset: n0 = 2^31+5
    n0 = 2 ** 31 + 5
    sdr.capi.frequency_shift_dev(None, True, 2048, n0, 0.0123, d_o)
    sdr.sync()
    k = np.arange(n0, n0 + 2048, dtype=np.float64)
    ph = (0.0123 * k) % 1.0
    check_linear(d_o.to_numpy(np.complex64, 2048), np.exp(2j * np.pi * ph), "rotation far")


@pytest.mark.parametrize("nfft", [64, 128, 256, 512, 1024, 2048, 4096])
def test_fft_power_db(sdr, oracle, golden, nfft):
    """FFT -> dB.  The reference's FFT is ArrayFire (absent): PARITY UNPINNED; checked against
    the oracle's float64 DFT.  Tolerance: linear power within 1e-5 of the peak power (full
    scale) for every bin, and |err| <= 1e-3 dB for bins within 50 dB of the frame peak (below
    that a float32 FFT's rounding floor, ~1e-7 of the peak amplitude, is visible in dB)."""
    iq = golden["fft_iq"]
    hop = nfft // 2
    for wname in (None, "blackman"):
        w = None if wname is None else oracle.window(wname, nfft)
        want_db, want_lin = oracle.fft_power_db_i16(iq, nfft, hop, w, linear=True)
        plan = sdr.FftPlan(nfft, hop, w)
        got = plan.execute_host(iq).astype(np.float64)
        assert got.shape == want_db.shape
        lin = 10.0 ** (got / 10.0)
        peak = want_lin.max(axis=1, keepdims=True)
        assert np.abs(lin - want_lin).max() <= 1e-5 * peak.max()
        strong = want_lin > peak * 1e-5
        assert np.abs(got - want_db)[strong].max() <= 1e-3
    if nfft == 4096:
        got = sdr.FftPlan(4096, 2048, oracle.window("blackman", 4096)).execute_host(iq)
        want = 10 * np.log10(golden["fft_blackman_pow_unpinned"])
        strong = want > want.max() - 50
        assert np.abs(got - want)[strong].max() <= 1e-3


def test_fft_exact_zero_gives_minus_inf(sdr):
    got = sdr.FftPlan(256, 128).execute_host(np.zeros(2 * 512, np.int16))
    assert got.shape == (3, 256) and np.all(np.isneginf(got))


@pytest.mark.gpu
def test_wire_formats(sdr, oracle):
    """Q-first int16 (B205 recordings) on every kernel, int8 pairs (HackRF) on the
    streaming kernel (test_int8_samples_on_the_matrix_kernel: on the matrix kernel), against the oracle
    chain run on the unpacked samples."""
    rng = np.random.default_rng(77)
    n = 200_003
    T, D = 64, 10
    h = oracle.create_fir_filter(T, 0.08, "hann")
    iq = oracle.synth_fm_i16(n, fs=1.0e6, ftone=2500.0, dev=30000.0, amp=1800.0)
    iq = (iq + rng.integers(-30, 31, iq.size)).astype(np.int16)
    qi = iq.reshape(-1, 2)[:, ::-1].reshape(-1).copy()
    y = oracle.chain_i16(iq, D, h, oracle.MODE_IQ)
    fm = oracle.chain_i16(iq, D, h, oracle.MODE_FM)
    env = oracle.chain_i16(iq, D, h, oracle.MODE_ENV)
    assert np.array_equal(oracle.unpack_scale_i16_qi(qi), oracle.unpack_scale_i16(iq))
    for pipe in (1, 2, 3, 4, 5):
        tag = f"QI pipe={pipe}"
        check_fm(sdr.FirPlan(h, D, mode=sdr.MODE_FM, input=sdr.IN_I16_QI, pipeline=pipe).execute_host(qi),
                 fm, y, 1 / TWO_PI, tag)
        check_linear(sdr.FirPlan(h, D, mode=sdr.MODE_ENV, input=sdr.IN_I16_QI, pipeline=pipe).execute_host(qi), env, tag)
        check_linear(sdr.FirPlan(h, D, mode=sdr.MODE_IQ, input=sdr.IN_I16_QI, pipeline=pipe).execute_host(qi), y, tag)
    # Q-first with the mixer (streaming kernel): the swap must happen before the rotation
    fr = 0.0123
    xs = oracle.frequency_shift(oracle.unpack_scale_i16(iq), fr)
    ys = oracle.fir_decimate(xs, D, h)
    check_linear(sdr.FirPlan(h, D, mode=sdr.MODE_IQ, input=sdr.IN_I16_QI, mix_fr=fr).execute_host(qi), ys, "QI mixer")

    # int8 pairs
    b = rng.integers(-128, 128, 2 * n).astype(np.int8)
    ph = 2 * np.pi * np.cumsum(0.01 + 0.004 * np.sin(2 * np.pi * 1e-3 * np.arange(n)))
    b[0::2] = np.clip(np.rint(100 * np.cos(ph)) + rng.integers(-3, 4, n), -128, 127).astype(np.int8)
    b[1::2] = np.clip(np.rint(100 * np.sin(ph)) + rng.integers(-3, 4, n), -128, 127).astype(np.int8)
    x8 = oracle.unpack_scale_i8(b)
    y8 = oracle.fir_decimate(x8, D, h)
    fm8 = oracle.polar_discriminator(y8, 1 / TWO_PI)
    for R, NT in ((0, 0), (1, 64), (4, 128)):
        tag = f"int8 R={R} NT={NT}"
        p = sdr.FirPlan(h, D, mode=sdr.MODE_FM, input=sdr.IN_I8_IQ, outputs_per_lane=R, threads=NT, pipeline=2)
        assert p.info()["kernel"] == "stream"
        check_fm(p.execute_host(b), fm8, y8, 1 / TWO_PI, tag)
        check_linear(sdr.FirPlan(h, D, mode=sdr.MODE_ENV, input=sdr.IN_I8_IQ, outputs_per_lane=R,
                                 threads=NT, pipeline=2).execute_host(b), np.abs(y8), tag)
        check_linear(sdr.FirPlan(h, D, mode=sdr.MODE_IQ, input=sdr.IN_I8_IQ, outputs_per_lane=R,
                                 threads=NT).execute_host(b), y8, tag)
    # odd byte address (sample-aligned only): the checked path
    d_in = sdr.DevBuf(2 * n + 16)
    import ctypes
    sdr.capi.check(sdr.capi.lib().sdr_dev_upload(d_in.at(2), b.ctypes.data_as(ctypes.c_void_p),
                                                 ctypes.c_size_t(2 * n)))
    p = sdr.FirPlan(h, D, mode=sdr.MODE_IQ, input=sdr.IN_I8_IQ)
    cnt = p.out_count(n)
    d_out = sdr.DevBuf(cnt * 8)
    p.execute_dev(d_in.at(2), n, d_out, cnt)
    sdr.sync()
    check_linear(d_out.to_numpy(np.complex64, cnt), y8, "int8 misaligned")


@pytest.mark.gpu
def test_fm_audio_post_chain(sdr, oracle):
    """f1: bladerf_fm_demod.cpp:301-313 -- de-emphasis + audio low-pass + pick + normalise + u8.
    ArrayFire is not in the tree (parity unpinned); the oracle restates the documented semantics
    in f64.  The GPU folds both filters into one real-input FIR/decimate (SDR_MODE_REAL)."""
    rng = np.random.default_rng(3)
    n = 150_001
    t = np.arange(n)
    x5 = (0.3 * np.sin(2 * np.pi * 0.003 * t) + 0.1 * np.sin(2 * np.pi * 0.041 * t)
          + 0.02 * rng.standard_normal(n)).astype(np.float32)
    h_de = oracle.create_fir_filter(64, 1.0 / (200e3 * 75e-6), "rectangular").astype(np.float32)
    h_a = oracle.create_fir_filter(101, (10e3 / 2.0) / 200e3, "hann").astype(np.float32)
    D = 20
    audio, pcm = oracle.fm_audio_chain(x5, h_de, h_a, D)
    hc = np.convolve(h_de.astype(np.float64), h_a.astype(np.float64))
    plan = sdr.FirPlan(hc, D, mode=sdr.MODE_REAL, input=sdr.IN_F32_REAL)
    got = plan.execute_host(x5)
    assert got.dtype == np.float32 and got.size == audio.size == n // D
    mx = float(np.abs(got).max())
    check_linear(got / mx, audio, "audio")
    m = got.size
    d_x = sdr.DevBuf.from_numpy(got)
    d_max = sdr.DevBuf(4)
    d_pcm = sdr.DevBuf(m)
    sdr.capi.audio_u8_dev(d_x, m, d_max, d_pcm)
    sdr.sync()
    assert d_max.to_numpy(np.float32, 1)[0] == np.float32(mx)
    out = d_pcm.to_numpy(np.uint8, m)
    # bit-exact against the same float32 arithmetic on the host ...
    r = np.float32(1.0 / np.float64(np.float32(mx)))
    ref32 = ((got * r + np.float32(1.0)) * np.float32(30.0)).astype(np.int32).astype(np.uint8)
    assert np.array_equal(out, ref32)
    # ... and within one code of the f64 oracle (values that sit on a truncation boundary)
    d = np.abs(out.astype(np.int32) - pcm.astype(np.int32))
    assert d.max() <= 1 and np.count_nonzero(d) <= max(4, m // 1000), (d.max(), np.count_nonzero(d))
    assert out.min() >= 0 and out.max() <= 60
    # all-zero block: 30 everywhere, max 0
    z = sdr.DevBuf.from_numpy(np.zeros(1000, np.float32))
    sdr.capi.audio_u8_dev(z, 1000, d_max, d_pcm)
    sdr.sync()
    assert np.all(d_pcm.to_numpy(np.uint8, 1000) == 30)
    # other real-output shapes through every geometry that fits
    x = rng.standard_normal(60_007).astype(np.float32)
    for T, Dd in ((7, 1), (64, 4), (165, 20), (511, 15)):
        h = oracle.create_fir_filter(T, min(0.8 / Dd, 0.9), "hann")
        y = oracle.fir_decimate(x.astype(np.complex128), Dd, h).real
        for R in (0, 1, 2, 4):
            try:
                p = sdr.FirPlan(h, Dd, mode=sdr.MODE_REAL, input=sdr.IN_F32_REAL, outputs_per_lane=R)
            except sdr.SdrError:
                continue
            check_linear(p.execute_host(x), y, f"real T={T} D={Dd} R={R}")


@pytest.mark.gpu
@pytest.mark.parametrize("nfft", [64, 512, 4096])
def test_fft_hops_and_long_runs(sdr, oracle, nfft):
    """hop = N/2 keeps half a frame in registers, other hops reload; with more frames than resident
    (workgroup, slot) pairs every lane walks a run of consecutive frames and the last run is
    ragged.  Selected frames against the oracle's float64 DFT of the same slice."""
    rng = np.random.default_rng(nfft)
    fpb = 4096 // nfft
    groups = 256 * 3 * fpb                       # resident pairs on an MI355X (see launch_fft)
    w = oracle.window("blackman", nfft)
    for hop in (nfft // 2, nfft, nfft // 4, 3 * nfft // 4 + 1):
        frames = 2 * groups + groups // 3 + 5    # runs of 3 frames, last workgroups partly idle
        n = (frames - 1) * hop + nfft + 3
        t = np.arange(n)
        iq = np.empty(2 * n, np.int16)
        ph = 2 * np.pi * (0.0371 * t + 2e-9 * t * t)
        iq[0::2] = (1500 * np.cos(ph) + rng.integers(-200, 201, n)).astype(np.int16)
        iq[1::2] = (1500 * np.sin(ph) + rng.integers(-200, 201, n)).astype(np.int16)
        plan = sdr.FftPlan(nfft, hop, w)
        assert plan.frame_count(n) == frames
        got = plan.execute_host(iq).astype(np.float64)
        assert got.shape == (frames, nfft)
        assert np.isfinite(got).all()
        run0 = 3 * (frames // 6)                 # first frame of a run in the middle
        pick = sorted({0, 1, 2, 3, 4, 5, run0 - 1, run0, run0 + 1, run0 + 2, run0 + 3, frames - 7,
                       frames - 2, frames - 1})
        for f in pick:
            seg = iq[2 * f * hop: 2 * (f * hop + nfft)]
            want_db, want_lin = oracle.fft_power_db_i16(seg, nfft, hop, w, linear=True)
            lin = 10.0 ** (got[f] / 10.0)
            peak = want_lin.max()
            assert np.abs(lin - want_lin[0]).max() <= 1e-5 * peak, (nfft, hop, f)
            strong = want_lin[0] > peak * 1e-5
            assert np.abs(got[f] - want_db[0])[strong].max() <= 1e-3, (nfft, hop, f)


@pytest.mark.gpu
def test_full_size_stream_properties(sdr, oracle):
    """BASELINE config 2 size (more than 2^32 bytes of input in one launch): slices at the start,
    across the 2^31- and 2^32-byte marks and at the ragged end against the oracle, and a time-block
    launch over the upper part of the same buffer against the one-piece result."""
    n = (1 << 30) + 12_347
    T, D = 64, 10
    h = oracle.create_fir_filter(T, 1.0 / D, "hann")
    plan = sdr.FirPlan(h, D, mode=sdr.MODE_FM)
    d_in = sdr.DevBuf(4 * n + 64)
    sdr.capi.synth_fm_i16_dev(d_in, n, fs=20.0e6, ftone=1000.0, dev=5000.0, noise_lsb=3, seed=11)
    cnt = plan.out_count(n)
    assert cnt == n // D - 1
    d_out = sdr.DevBuf(4 * cnt + 64)
    assert plan.execute_dev(d_in, n, d_out, cnt) == cnt
    sdr.sync()
    m = 4096
    for j0 in (0, (1 << 29) // D - m // 2, (1 << 30) // D - m // 2, cnt - m):
        s_lo = max(0, j0 * D - (T - 1))
        s_hi = min(n, (j0 + m) * D + 1)
        iq = d_in.to_numpy(np.int16, 2 * (s_hi - s_lo), 4 * s_lo)
        y = oracle.fir_decimate(oracle.unpack_scale_i16(iq), D, h, pre=j0 * D - s_lo)
        want = oracle.polar_discriminator(y, 1 / TWO_PI)[:m]
        got = d_out.to_numpy(np.float32, want.size, 4 * j0)
        check_fm(got, want, y[: want.size + 1], 1 / TWO_PI, f"slice at output {j0}")
    # upper part as a time block with its history halo resident
    ob = (3 << 28) // D + 3
    lo = ob * D - (T - 1)
    d_blk = sdr.DevBuf(4 * (cnt - ob) + 64)
    plan.execute_range_dev(d_in.at(4 * lo), lo, n - lo, n, ob, cnt, d_blk)
    sdr.sync()
    for off in (0, (cnt - ob) // 2, cnt - ob - m):
        a = d_blk.to_numpy(np.float32, m, 4 * off)
        b = d_out.to_numpy(np.float32, m, 4 * (ob + off))
        assert np.array_equal(a, b)          # one summation order per output, whatever the tiling


@pytest.mark.gpu
def test_time_shard_beyond_2_pow_32_samples(sdr, oracle):
    """the last of eight 2^30-sample time shards: stream indices above 2^32 samples (2^34 bytes)
    reach the kernel only through the 64-bit range arguments and the virtual base pointer"""
    from sdr_amd import shard
    T, D = 64, 10
    n_total = 8 << 30
    h = oracle.create_fir_filter(T, 1.0 / D, "hann")
    plan = sdr.FirPlan(h, D, mode=sdr.MODE_FM)
    sh = shard.plan_time_shards(n_total, D, T, shard.MODE_FM, 8)[7]
    assert sh.out_end == plan.out_count(n_total) and sh.lo > (1 << 32)
    # keep only the last 2^27 outputs of the shard resident (the arithmetic is the same)
    ob = sh.out_end - (1 << 27)
    lo = ob * D - (T - 1)
    n_res = sh.hi - lo
    d_in = sdr.DevBuf(4 * n_res + 64)
    sdr.capi.synth_fm_i16_dev(d_in, n_res, n0=lo, fs=20.0e6, ftone=1000.0, dev=5000.0, noise_lsb=2, seed=4)
    d_out = sdr.DevBuf(4 * (sh.out_end - ob) + 64)
    plan.execute_range_dev(d_in, lo, n_res, n_total, ob, sh.out_end, d_out)
    sdr.sync()
    m = 4096
    for j0 in (0, (sh.out_end - ob) // 2 + 77, sh.out_end - ob - m):
        J = ob + j0
        s_lo = J * D - (T - 1)
        s_hi = min(sh.hi, (J + m) * D + 1)
        iq = d_in.to_numpy(np.int16, 2 * (s_hi - s_lo), 4 * (s_lo - lo))
        y = oracle.fir_decimate(oracle.unpack_scale_i16(iq), D, h, pre=T - 1)
        want = oracle.polar_discriminator(y, 1 / TWO_PI)[:m]
        got = d_out.to_numpy(np.float32, want.size, 4 * j0)
        check_fm(got, want, y[: want.size + 1], 1 / TWO_PI, f"shard output {J}")
    # asking for outputs whose samples are not resident is refused
    with pytest.raises(sdr.SdrError) as e:
        plan.execute_range_dev(d_in, lo, n_res, n_total, ob - 10, sh.out_end, d_out)
    assert e.value.status == sdr.capi.ERANGE


@pytest.mark.gpu
@pytest.mark.parametrize("T,D", [(511, 10), (511, 15), (300, 7), (64, 10)])
def test_byte_split_matrix_path_full_range(sdr, oracle, T, D):
    """pipeline 5 feeds the int8 matrix pipe with the samples' two bytes and the taps' four base-256
    digits: the whole int16 range (incl. -32768 and 32767, far beyond Q11) and taps of mixed
    magnitude must come out at float32 accuracy, edges (zero history, ragged end) included"""
    rng = np.random.default_rng(T + D)
    n = 150_001
    iq = rng.integers(-32768, 32768, 2 * n, dtype=np.int64).astype(np.int16)
    iq[:8] = [-32768, 32767, 32767, -32768, -1, 255, 256, -256]
    h = oracle.create_fir_filter(T, 0.7 / D, "hann") * rng.uniform(0.2, 1.0, T)   # no symmetry, mixed exponents
    x = oracle.unpack_scale_i16(iq)
    y = oracle.fir_decimate(x, D, h)
    for mode, want in ((sdr.MODE_IQ, y), (sdr.MODE_ENV, np.abs(y))):
        got = sdr.FirPlan(h, D, mode=mode, pipeline=5).execute_host(iq)
        assert got.shape == want.shape
        # full scale here is 16 (int16 * 2^-11); float32 accumulation of T products
        assert np.abs(got.astype(want.dtype) - want).max() <= 1e-5 * 16 * 0.5, (T, D, mode)
    fm = oracle.polar_discriminator(y, 1 / TWO_PI)
    got = sdr.FirPlan(h, D, mode=sdr.MODE_FM, pipeline=5).execute_host(iq)
    check_fm(got, fm, y, 1 / TWO_PI, f"i8 T={T} D={D}")
    assert sdr.FirPlan(h, D, mode=sdr.MODE_FM, pipeline=5).info()["kernel"] == "mfma_i8"


@pytest.mark.gpu
def test_long_filter_automatic_choice_and_tiny_streams(sdr, oracle):
    """filters with enough work per 512-output tile go to the int8 matrix kernel on their own; streams
    shorter than its 4-sample staging loads are handed to the VALU plan it keeps for that"""
    h = oracle.create_fir_filter(201, 0.3, "hann")
    p = sdr.FirPlan(h, 3, mode=sdr.MODE_IQ)
    assert p.info()["kernel"] == "mfma_i8"
    # (the matrix kernel loses where tiles are many and windows short: small decimation factors, few taps)
    for T, D, want_i8 in ((64, 10, True), (101, 5, True), (101, 20, True), (128, 8, True), (200, 10, True),
                          (195, 15, True), (511, 10, True), (511, 15, True), (64, 4, True), (300, 25, True),
                          (16, 2, False), (32, 4, False), (9, 1, False)):
        k = sdr.FirPlan(oracle.create_fir_filter(T, 0.5 / D, "hann"), D, mode=sdr.MODE_FM).info()["kernel"]
        assert (k == "mfma_i8") == want_i8, (T, D, k)
    # a size hint that says "short launches" keeps the plan off the persistent kernel (its launch latency)
    h64, h511 = oracle.create_fir_filter(64, 0.05, "hann"), oracle.create_fir_filter(511, 0.05, "hann")
    for hh, hint, want_i8 in ((h64, 131072, False), (h64, 1 << 24, False), (h64, 1 << 28, True), (h64, 0, True),
                              (h511, 131072, False), (h511, 1 << 24, True)):
        k = sdr.FirPlan(hh, 10, mode=sdr.MODE_FM, size_hint=hint).info()["kernel"]
        assert (k == "mfma_i8") == want_i8, (hh.size, hint, k)
    rng = np.random.default_rng(5)
    for n in (0, 1, 2, 3, 4, 5, 11, 4097):
        iq = rng.integers(-2048, 2048, 2 * n, dtype=np.int16)
        want = oracle.fir_decimate(oracle.unpack_scale_i16(iq), 3, h)
        got = p.execute_host(iq)
        check_linear(got, want, f"n={n}")


@pytest.mark.gpu
@pytest.mark.parametrize("T,D", [(64, 4), (24, 15), (128, 8), (255, 4), (200, 10), (195, 15), (600, 4), (511, 10),
                                 (511, 15), (101, 20), (128, 16), (300, 25), (60, 31), (300, 20), (100, 25)])
def test_int8_matrix_kernel_instantiations(sdr, oracle, T, D):
    """every instantiation of the int8 matrix kernel (window of 4 / 8 / 12 k steps x tile of 1024 / 512
    outputs x 4 / 6 / 8 octets per helper lane) against the oracle, all modes, on a stream with ragged
    ends and full-range samples"""
    rng = np.random.default_rng(T * 100 + D)
    n = 4 * 1024 * D + 3 * D + 7 + T                       # a few tiles per workgroup slot, ragged end
    iq = oracle.synth_fm_i16(n, fs=1.0e6, ftone=700.0, dev=40000.0, amp=20000.0)
    iq = np.clip(iq.astype(np.int32) + rng.integers(-3000, 3001, iq.size), -32768, 32767).astype(np.int16)
    h = oracle.create_fir_filter(T, 0.8 / D, "hann")
    x = oracle.unpack_scale_i16(iq)
    y = oracle.fir_decimate(x, D, h)
    for mode, want in ((sdr.MODE_IQ, y), (sdr.MODE_ENV, np.abs(y))):
        plan = sdr.FirPlan(h, D, mode=mode, pipeline=5)
        assert plan.info()["kernel"] == "mfma_i8"
        check_linear(plan.execute_host(iq), want, f"i8 {T}/{D} mode {mode}")
    fm = oracle.polar_discriminator(y, 1 / TWO_PI)
    check_fm(sdr.FirPlan(h, D, mode=sdr.MODE_FM, pipeline=5).execute_host(iq), fm, y, 1 / TWO_PI, f"i8 {T}/{D} fm")


@pytest.mark.gpu
@pytest.mark.parametrize("T,D", [(64, 10), (64, 4), (24, 15), (200, 10), (255, 4), (195, 15), (511, 10), (600, 4), (511, 15),
                                 (101, 20), (300, 25), (40, 30), (511, 16)])      # D >= 16: tiles of 512 outputs
def test_int8_samples_on_the_matrix_kernel(sdr, oracle, T, D):
    """HackRF int8 pairs (hackrf_rx_main.cpp:28,54) on the int8 matrix kernel: the sample bytes are the
    operand planes, four byte products per step.  All modes against the oracle on full-range bytes
    with ragged ends, at an odd byte offset inside the device buffer, in two time shards (bit-identical
    to the one-piece run), and the automatic choice."""
    rng = np.random.default_rng(T * 31 + D)
    n = 3 * 1024 * D + 5 * D + 3 + T
    b = rng.integers(-128, 128, 2 * n).astype(np.int8)
    ph = 2 * np.pi * np.cumsum(0.01 + 0.004 * np.sin(2 * np.pi * 1e-3 * np.arange(n)))
    b[0::2] = np.clip(np.rint(90 * np.cos(ph)) + rng.integers(-38, 39, n), -128, 127).astype(np.int8)
    b[1::2] = np.clip(np.rint(90 * np.sin(ph)) + rng.integers(-38, 39, n), -128, 127).astype(np.int8)
    h = oracle.create_fir_filter(T, 0.8 / D, "hann")
    y = oracle.fir_decimate(oracle.unpack_scale_i8(b), D, h)
    fm = oracle.polar_discriminator(y, 1 / TWO_PI)
    for mode, want in ((sdr.MODE_IQ, y), (sdr.MODE_ENV, np.abs(y))):
        plan = sdr.FirPlan(h, D, mode=mode, input=sdr.IN_I8_IQ, pipeline=5)
        assert plan.info()["kernel"] == "mfma_i8"
        check_linear(plan.execute_host(b), want, f"int8 on i8 {T}/{D} mode {mode}")
    plan = sdr.FirPlan(h, D, mode=sdr.MODE_FM, input=sdr.IN_I8_IQ, pipeline=5)
    one = plan.execute_host(b)
    check_fm(one, fm, y, 1 / TWO_PI, f"int8 on i8 {T}/{D} fm")
    # the stream at an odd sample offset inside a device buffer (2-byte aligned 16-byte loads), two shards
    cnt = plan.out_count(n)
    d_all = sdr.DevBuf(2 * (n + 16))
    sdr.capi.check(sdr.capi.lib().sdr_dev_upload(d_all.at(2 * 3), b.ctypes.data_as(C.c_void_p), C.c_size_t(2 * n)))
    parts = []
    for ob, oe in ((0, cnt // 2), (cnt // 2, cnt)):
        lo = max(0, ob * D - (T - 1))
        hi = min(n, (oe + 1) * D + 1)
        d_out = sdr.DevBuf(max(oe - ob, 1) * 4)
        plan.execute_range_dev(d_all.at(2 * (3 + lo)), lo, hi - lo, n, ob, oe, d_out)
        sdr.sync()
        parts.append(d_out.to_numpy(np.float32, oe - ob))
    assert np.array_equal(np.concatenate(parts), one)
    # the automatic choice: the matrix kernel or, for few taps per output, the float tiles
    auto = sdr.FirPlan(h, D, mode=sdr.MODE_FM, input=sdr.IN_I8_IQ).info()["kernel"]
    assert auto in ("mfma_i8", "stream"), auto


@pytest.mark.gpu
@pytest.mark.parametrize("T,D", [(64, 10), (101, 4), (300, 20)])
def test_int8_samples_channels_and_blocks(sdr, oracle, T, D):
    """int8 pairs on the matrix kernel behind the other entry points: channel-major batches with odd
    strides (2-byte aligned channel starts) equal the channels run one by one, and blocks pushed
    through the stream hand-off equal the one-shot launch -- bit for bit, the sums are exact integers."""
    rng = np.random.default_rng(T + D)
    h = oracle.create_fir_filter(T, 0.8 / D, "hann")
    n_ch, n = 5, 2 * 1024 * D + 3 * D + 1
    stride_in = n + 3
    buf = rng.integers(-128, 128, 2 * stride_in * n_ch, dtype=np.int8)
    for mode, dt, eb in ((sdr.MODE_FM, np.float32, 4), (sdr.MODE_IQ, np.complex64, 8)):
        plan = sdr.FirPlan(h, D, mode=mode, input=sdr.IN_I8_IQ, pipeline=5)
        cnt = plan.out_count(n)
        stride_out = cnt + 1
        d_in = sdr.DevBuf.from_numpy(buf)
        d_out = sdr.DevBuf.from_numpy(np.full(stride_out * n_ch * (eb // 4), 0x7FC0DEAD, np.uint32))
        plan.execute_batched_dev(d_in, n, stride_in, n_ch, d_out, stride_out)
        sdr.sync()
        out = d_out.to_numpy(dt, stride_out * n_ch)
        for c in range(n_ch):
            one = plan.execute_host(buf[2 * c * stride_in: 2 * (c * stride_in + n)])
            assert np.array_equal(out[c * stride_out: c * stride_out + cnt], one), (mode, c)
            pad = out[c * stride_out + cnt: (c + 1) * stride_out].view(np.uint32)
            assert np.all(pad == 0x7FC0DEAD), (mode, c)
        if mode == sdr.MODE_IQ:
            y = oracle.fir_decimate(oracle.unpack_scale_i8(buf[2 * 2 * stride_in: 2 * (2 * stride_in + n)]), D, h)
            check_linear(out[2 * stride_out: 2 * stride_out + cnt], y, f"int8 batch {T}/{D}")
    # blocks through the hand-off
    plan = sdr.FirPlan(h, D, mode=sdr.MODE_FM, input=sdr.IN_I8_IQ, pipeline=5)
    b = buf[: 2 * stride_in * 3]
    nb = b.size // 2
    whole = plan.execute_host(b)
    stream = sdr.FirStream(plan, 8192)
    pos, parts, in_flight = 0, [], 0
    while pos < nb:
        blk = int(min(nb - pos, rng.choice([8192, 4096, 1000, 37, 9, 8191])))
        stream.push(b[2 * pos: 2 * (pos + blk)])
        pos += blk
        in_flight += 1
        if in_flight == 2:
            parts.append(stream.pop())
            in_flight -= 1
    while in_flight:
        parts.append(stream.pop())
        in_flight -= 1
    assert np.array_equal(np.concatenate(parts), whole)


@pytest.mark.gpu
def test_int8_matrix_kernel_shards_and_channels(sdr, oracle):
    """the int8 matrix kernel behind the other entry points: time shards with their history halo
    (tiles that start before / end after the resident window), channel-major batches (tiles of
    several channels in one persistent grid) and the streaming hand-off"""
    D, T = 10, 511
    h = oracle.create_fir_filter(T, 0.07, "hann")
    rng = np.random.default_rng(9)
    # ---- time shards -------------------------------------------------------------------------
    n = 400_011
    iq = oracle.synth_fm_i16(n, fs=624e3, ftone=2400.0, dev=17000.0, amp=1800.0)
    iq = (iq + rng.integers(-6, 7, iq.size)).astype(np.int16)
    for mode, omode in ((sdr.MODE_FM, oracle.MODE_FM), (sdr.MODE_ENV, oracle.MODE_ENV)):
        plan = sdr.FirPlan(h, D, mode=mode)
        assert plan.info()["kernel"] == "mfma_i8"
        want = oracle.chain_i16(iq, D, h, omode)
        cnt = want.size
        parts = []
        for gidx in range(3):
            ob, oe = cnt * gidx // 3, cnt * (gidx + 1) // 3
            lo = max(0, ob * D - (T - 1))
            hi = min(n, (oe + 1) * D + 1)
            d_blk = sdr.DevBuf.from_numpy(iq[2 * lo: 2 * hi])
            d_out = sdr.DevBuf((oe - ob) * 4)
            plan.execute_range_dev(d_blk, lo, hi - lo, n, ob, oe, d_out)
            sdr.sync()
            parts.append(d_out.to_numpy(np.float32, oe - ob))
        got = np.concatenate(parts)
        if mode == sdr.MODE_FM:
            check_fm(got, want, oracle.chain_i16(iq, D, h, oracle.MODE_IQ), 1 / TWO_PI, "i8 shards fm")
        else:
            check_linear(got, want, "i8 shards env")
    # ---- channel-major batch -------------------------------------------------------------------
    n_ch, nc = 9, 30_007
    stride_in = nc + 40
    buf = np.zeros(2 * stride_in * n_ch, np.int16)
    for c in range(n_ch):
        buf[2 * c * stride_in: 2 * (c * stride_in + nc)] = oracle.synth_fm_i16(
            nc, fs=624e3, ftone=900.0 + 333.0 * c, dev=9000.0, amp=1500.0)
    plan = sdr.FirPlan(h, D, mode=sdr.MODE_ENV)
    cnt = plan.out_count(nc)
    stride_out = cnt + 5
    d_in = sdr.DevBuf.from_numpy(buf)
    d_out = sdr.DevBuf(stride_out * n_ch * 4)
    plan.execute_batched_dev(d_in, nc, stride_in, n_ch, d_out, stride_out)
    sdr.sync()
    out = d_out.to_numpy(np.float32, stride_out * n_ch)
    for c in range(n_ch):
        want = oracle.chain_i16(buf[2 * c * stride_in: 2 * (c * stride_in + nc)], D, h, oracle.MODE_ENV)
        check_linear(out[c * stride_out: c * stride_out + cnt], want, f"channel {c}")
    # ---- streaming hand-off: history carried across blocks --------------------------------------
    plan = sdr.FirPlan(h, D, mode=sdr.MODE_ENV)
    blk = 20_000
    st = sdr.FirStream(plan, blk)
    outs = []
    for b in range(6):
        st.push(iq[2 * b * blk: 2 * (b + 1) * blk])
        if b:
            outs.append(st.pop())
    outs.append(st.pop())
    got = np.concatenate(outs)
    want = oracle.chain_i16(iq[: 2 * 6 * blk], D, h, oracle.MODE_ENV)
    check_linear(got, want[: got.size], "stream")
    assert got.size == want.size


@pytest.mark.gpu
@pytest.mark.parametrize("nfft,hop", [(8192, 4096), (16384, 16384), (131072, 131072), (131072, 50_000)])
def test_fft_long_frames(sdr, oracle, nfft, hop):
    """a10 at the reference's own size: one 131072-sample capture block per transform
    (bladerf_main_v1.cpp:139, no window, no overlap), plus other long frames with window / overlap;
    the multi-pass path against the oracle's float64 FFT."""
    rng = np.random.default_rng(nfft + hop)
    frames = 3
    n = (frames - 1) * hop + nfft + 17
    t = np.arange(n)
    ph = 2 * np.pi * (0.0913 * t + 3e-8 * t * t)
    iq = np.empty(2 * n, np.int16)
    iq[0::2] = (1400 * np.cos(ph) + 300 * np.cos(0.7 * t) + rng.integers(-150, 151, n)).astype(np.int16)
    iq[1::2] = (1400 * np.sin(ph) + rng.integers(-150, 151, n)).astype(np.int16)
    w = None if hop == nfft == 131072 else oracle.window("blackman", nfft)
    plan = sdr.FftPlan(nfft, hop, w)
    assert plan.frame_count(n) == frames
    got = plan.execute_host(iq).astype(np.float64)
    assert got.shape == (frames, nfft) and np.isfinite(got).all()
    want_db, want_lin = oracle.fft_power_db_i16(iq, nfft, hop, w, linear=True)
    for f in range(frames):
        lin = 10.0 ** (got[f] / 10.0)
        peak = want_lin[f].max()
        assert np.abs(lin - want_lin[f]).max() <= 1e-5 * peak, (nfft, hop, f)
        strong = want_lin[f] > peak * 1e-5
        assert np.abs(got[f] - want_db[f])[strong].max() <= 1e-3, (nfft, hop, f)
    with pytest.raises(sdr.SdrError):
        sdr.FftPlan(1 << 21, 1 << 21)


@pytest.mark.gpu
@pytest.mark.parametrize("T,D,kind,mix", [(65, 39, "real", 0.0), (199, 41, "real", 0.0371),
                                          (2047, 32, "i16", -0.123), (2047, 40, "i16", 0.0371),
                                          (7, 41, "cf32", 0.0), (1000, 25, "i8", 0.0371)])
def test_large_decimation_float_tiles(sdr, oracle, T, D, kind, mix):
    """Float-tile kernel beyond its prefetch registers: decimation factors above ~32 (odd ones need
    two outputs per lane, i.e. D granules per lane) and very long filters at large D take the
    second geometry pass (any tile the LDS holds, rest of the tile fetched without prefetch)."""
    rng = np.random.default_rng(T * 1000 + D)
    n = 150_011
    h = oracle.create_fir_filter(T, min(0.8 / D, 0.9), "hann")
    if kind == "i16":
        raw = rng.integers(-32768, 32768, 2 * n, dtype=np.int16)
        x, inp = oracle.unpack_scale_i16(raw), sdr.IN_I16_IQ
    elif kind == "i8":
        raw = rng.integers(-128, 128, 2 * n, dtype=np.int8)
        x, inp = oracle.unpack_scale_i8(raw), sdr.IN_I8_IQ
    elif kind == "cf32":
        raw = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.3).astype(np.complex64)
        x, inp = raw.astype(np.complex128), sdr.IN_CF32
    else:
        raw = (rng.standard_normal(n) * 0.3).astype(np.float32)
        x, inp = raw.astype(np.complex128), sdr.IN_F32_REAL
    if mix != 0.0:
        x = oracle.frequency_shift(x, mix)
    y = oracle.fir_decimate(x, D, h)
    plan = sdr.FirPlan(h, D, mode=sdr.MODE_IQ, input=inp, mix_fr=mix, pipeline=2)
    assert plan.info()["kernel"] == "stream"
    check_linear(plan.execute_host(raw), y, f"T={T} D={D} {kind}")
    auto = sdr.FirPlan(h, D, mode=sdr.MODE_IQ, input=inp, mix_fr=mix)          # (float input without mixer may take the DMA tile)
    assert auto.info()["kernel"] in ("stream", "float_dma_tile")
    check_linear(auto.execute_host(raw), y, f"T={T} D={D} {kind} auto")
    check_linear(sdr.FirPlan(h, D, mode=sdr.MODE_ENV, input=inp, mix_fr=mix).execute_host(raw),
                 oracle.envelope(y), f"T={T} D={D} {kind} env")
    got = sdr.FirPlan(h, D, mode=sdr.MODE_FM, input=inp, mix_fr=mix).execute_host(raw)
    assert got.shape == (y.size - 1,) and np.isfinite(got).all()
    check_fm(got, oracle.polar_discriminator(y, 1 / TWO_PI), y, 1 / TWO_PI, f"T={T} D={D} {kind} fm")


@pytest.mark.gpu
def test_randomised_shapes(sdr, oracle):
    """A fixed slice of tools/fuzz_parity.py (random T, D, n, mode, input kind, mixer, kernel choice,
    taps of mixed magnitude, a random shard cut per case): 400 seeded cases."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    ran = 0
    for seed in range(7_000_000, 7_000_400):
        try:
            ran += fuzz_parity.one_case(np.random.default_rng(seed), seed) == "ok"
        except sdr.SdrError as e:
            assert "pipeline 5 needs at least" in str(e) and "resident samples" in str(e), (seed, str(e))
    assert ran >= 300


@pytest.mark.gpu
@pytest.mark.parametrize("pipe", [0, 5])
def test_matrix_kernel_startup_zero_vector_angle(sdr, oracle, pipe):
    """Found by tools/fuzz_parity.py: with a Hann design h[0] is exactly zero, so y[0] = 0, and with
    fc*t an integer at t = -(T-1)/2 + 1 the next tap is ~1e-20 -- below the int8 matrix kernel's
    fixed point -- so y[1] consists of a tap that kernel drops.  The reference's first discriminator
    output is then +-pi*scale by the signs of its zeros; fir_startup_fm_kernel recomputes the start-up
    outputs with float32 taps."""
    T, D, n = 193, 1, 50_989
    h = oracle.create_fir_filter(T, 0.8, "hann")
    assert h[0] == 0.0 and 0.0 < abs(h[1]) < abs(h).max() * 2.0 ** -40
    iq = oracle.synth_fm_i16(n, fs=1.0e6, ftone=3000.0, dev=20000.0, amp=1900.0)
    iq[1] = 7                        # Q of sample 0 > 0: the product's real part becomes -0
    plan = sdr.FirPlan(h, D, mode=sdr.MODE_FM, pipeline=pipe)
    assert plan.info()["kernel"] == "mfma_i8"
    y = oracle.fir_decimate(oracle.unpack_scale_i16(iq), D, h)
    want = oracle.polar_discriminator(y, 1 / TWO_PI)
    assert y[0] == 0 and y[1] != 0 and abs(want[0]) == 0.5
    got = plan.execute_host(iq)
    assert got[0] == np.float32(want[0])
    check_fm(got, want, y, 1 / TWO_PI, "whole stream")


@pytest.mark.gpu
def test_af_seq_count_rule(sdr, oracle):
    """SDR_COUNT_AF_SEQ: the ArrayFire loops pick x3(af::seq(0, N, D)) -- every index 0, D, ... <= N-1
    (bladerf_fm_demod.cpp:212,294; bladerf_am_demod.cpp:213,297), one sample more than
    polyphase_decimate's floor(N/D) unless D divides N.  Checked against the oracle's fir_pick."""
    rng = np.random.default_rng(77)
    h = oracle.create_fir_filter(101, 0.09, "hann")
    for n, D in ((131072, 5), (131072, 20), (100_000, 10), (100_001, 10), (99_999, 10), (7, 10), (1, 3), (0, 3)):
        x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.3).astype(np.complex64)
        y = oracle.fir_pick(x.astype(np.complex128), D, h) if n else np.zeros(0, np.complex128)
        want_n = (n - 1) // D + 1 if n else 0
        assert y.size == want_n
        for pipe in (0, 2):
            plan = sdr.FirPlan(h, D, mode=sdr.MODE_IQ, input=sdr.IN_CF32, count_rule=sdr.capi.COUNT_AF_SEQ, pipeline=pipe)
            assert plan.out_count(n) == want_n, (n, D)
            check_linear(plan.execute_host(x), y, f"af seq n={n} D={D}")
        floor_plan = sdr.FirPlan(h, D, mode=sdr.MODE_IQ, input=sdr.IN_CF32)
        assert floor_plan.out_count(n) == n // D
        # the discriminator and the envelope follow the same count (x4(1:end)*conjg(x4(0:end-1)), :297)
        fm = sdr.FirPlan(h, D, mode=sdr.MODE_FM, input=sdr.IN_CF32, scale=0.3, count_rule=sdr.capi.COUNT_AF_SEQ)
        got = fm.execute_host(x)
        assert got.size == max(want_n - 1, 0)
        if got.size:
            check_fm(got, oracle.polar_discriminator(y, 0.3), y, 0.3, f"af seq fm n={n} D={D}")
    # int16 input (raw-tile kernel) with the same rule, and the block hand-off
    n, D = 131072, 5
    iq = rng.integers(-2048, 2048, 2 * n, dtype=np.int16)
    y = oracle.fir_pick(oracle.unpack_scale_i16(iq), D, h)
    plan = sdr.FirPlan(h, D, mode=sdr.MODE_ENV, count_rule=sdr.capi.COUNT_AF_SEQ)
    assert plan.info()["kernel"] in ("raw_tile", "mfma_i8")
    check_linear(plan.execute_host(iq), oracle.envelope(y), "af seq i16 env")
    st = sdr.FirStream(plan, 50_000)
    outs = []
    for a in range(0, n, 50_000):
        st.push(iq[2 * a: 2 * min(n, a + 50_000)])
        outs.append(st.pop())
    got = np.concatenate(outs)
    check_linear(got, oracle.envelope(y), "af seq through the hand-off")


@pytest.mark.gpu
def test_pipeline5_refuses_one_tap(sdr):
    """with one tap there is no history halo and a tile at the stream's end could hold
    fewer than the four valid samples the int8 kernel's 16-byte staging loads need"""
    with pytest.raises(sdr.SdrError) as e:
        sdr.FirPlan(np.ones(1), 1, mode=sdr.MODE_IQ, pipeline=5)
    assert e.value.status == sdr.capi.EINVAL and "at least 2 taps" in str(e.value)
    # two taps at the very end of an exactly sized device buffer: sentinel-free, but every staged load
    # must stay inside [0, n)
    n = 1024 * 3 + 1
    rng = np.random.default_rng(3)
    iq = rng.integers(-2048, 2048, 2 * n, dtype=np.int16)
    plan = sdr.FirPlan(np.array([0.5, 0.25]), 1, mode=sdr.MODE_IQ, pipeline=5)
    d_in = sdr.DevBuf.from_numpy(iq)
    cnt = plan.out_count(n)
    d_out = sdr.DevBuf((cnt + 16) * 8)
    sent = np.full(2 * (cnt + 16), 1234.5, np.float32)
    sdr.capi.check(sdr.capi.lib().sdr_dev_upload(d_out.ptr, sent.ctypes.data_as(sdr.capi.C.c_void_p), sdr.capi.C.c_size_t(sent.nbytes)))
    assert plan.execute_dev(d_in, n, d_out, cnt) == cnt
    sdr.sync()
    got = d_out.to_numpy(np.complex64, cnt + 16)
    x = (iq[0::2] + 1j * iq[1::2]) / 2048.0
    want = 0.5 * x + 0.25 * np.concatenate(([0], x[:-1]))
    check_linear(got[:cnt], want, "two taps, pipeline 5")
    assert (got[cnt:].view(np.float32) == np.float32(1234.5)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("T,D,kind", [(64, 1, "i16"), (64, 10, "i16"), (511, 10, "i16"), (200, 4, "i16"), (300, 20, "i16"),
                                      (64, 10, "i8"), (255, 4, "i8"), (64, 10, "qi")])
def test_matrix_kernel_zero_runs(sdr, oracle, T, D, kind):
    """Round 2's header carried an exemption for the int8 matrix kernel: taps below 2^-31 of the largest
    tap vanish in its fixed point, so after >= T zero samples in mid-stream the output whose window meets
    the first non-zero sample only through such a tap was an exact 0 there and a tiny vector in the
    reference, and two discriminator outputs reported a zero vector's angle.  The kernel now recomputes
    every exactly-zero y that feeds a stored output with float32 taps (i8_zero_fix), so the whole stream
    -- the run's 0 / +-pi outputs included, which check_fm demands exactly -- holds the bars, for the
    pinned and the automatic choice, one piece or sharded."""
    rng = np.random.default_rng(T * 7 + D)
    h = oracle.create_fir_filter(T, min(0.8 / D, 0.45), "hann")
    # sub-resolution taps at both ends and inside (a windowed sinc's numerical zeros), none exactly zero
    h[0], h[1], h[-1] = 1e-13, -3e-14, 2e-15
    h[T // 3] = -1e-14
    assert 0 < abs(h[0]) < np.abs(h).max() * 2.0 ** -31
    n = 40 * 1024 + 1234 + 7 * D
    if kind == "i8":
        raw = rng.integers(-128, 128, 2 * n).astype(np.int8)
        inp, unpack = sdr.IN_I8_IQ, oracle.unpack_scale_i8
    else:
        raw = rng.integers(-2000, 2001, 2 * n).astype(np.int16)
        inp, unpack = (sdr.IN_I16_QI if kind == "qi" else sdr.IN_I16_IQ), \
                      (oracle.unpack_scale_i16_qi if kind == "qi" else oracle.unpack_scale_i16)
    # zero runs in mid-stream: longer than the filter (several, at odd alignments, one crossing a tile
    # border of 1024 outputs), one shorter than the filter, and isolated samples inside a long run
    runs = [(3_000, 3_000 + 3 * T + 5), (1024 * D - T - 3, 1024 * D + 2 * T + 1), (9_001, 9_001 + T - 7),
            (15_000 + D // 2, 15_000 + 4 * T + 2 * D)]
    runs = [(a, b) for a, b in runs if b < n - 2 * T]
    for i, (a, b) in enumerate(runs):
        # ends on an output's sample (that output sees the first sample behind the run through h[0] only)
        # or one before it (h[0] and h[1]); starts so that some output sees the last sample before the
        # run through h[T-1] only
        b = (b + D - 1) // D * D - (i & 1)
        a -= (a + T - 2) % D
        raw[2 * a: 2 * b] = 0
    z0 = 20_000
    z1 = min(n - 3 * T, z0 + 12 * T + 40 * D)
    raw[2 * z0: 2 * z1] = 0
    for k, s in enumerate(range(z0 + T + 3, z1 - T, 2 * T + 3 * D + 1)):   # lone samples: every window sees at most one
        raw[2 * s] = (-1) ** k * 77
        raw[2 * s + 1] = 55 if k % 3 else -91
    y = oracle.fir_decimate(unpack(raw), D, h)
    want = oracle.polar_discriminator(y, 1 / TWO_PI)
    mag = np.abs(y)
    tiny = (mag > 0) & (mag < 1e-11)
    assert tiny.sum() >= 3 and (mag == 0).sum() >= 3 * T // D      # the case is in the stream, several times
    for pipe in (5, 0):
        plan = sdr.FirPlan(h, D, mode=sdr.MODE_FM, input=inp, pipeline=pipe)
        if pipe == 5:
            assert plan.info()["kernel"] == "mfma_i8"
        got = plan.execute_host(raw)
        # (noise input: |y| has no typical magnitude, so every output gets the angle bound -- good_rel=None)
        check_fm(got, want, y, 1 / TWO_PI, f"zero runs {T}/{D} {kind} pipe {pipe}", good_rel=None)
        # outputs that involve a tiny y are bit-equal to what the vector-unit arithmetic gives, and the
        # shards reproduce the one-piece result bit for bit
        cnt = got.size
        esz = 2 if kind == "i8" else 4
        d_all = sdr.DevBuf.from_numpy(raw)
        parts = []
        for ob, oe in ((0, cnt // 3), (cnt // 3, cnt // 2 + 11), (cnt // 2 + 11, cnt)):
            lo = max(0, ob * D - (T - 1))
            hi = min(n, (oe + 1) * D + 1)
            d_out = sdr.DevBuf(max(oe - ob, 1) * 4)
            plan.execute_range_dev(d_all.at(esz * lo), lo, hi - lo, n, ob, oe, d_out)
            sdr.sync()
            parts.append(d_out.to_numpy(np.float32, oe - ob))
        assert np.array_equal(np.concatenate(parts), got)
    # the other modes see the same zeros (no recomputation there: an exact 0 is within the linear bar)
    check_linear(sdr.FirPlan(h, D, mode=sdr.MODE_ENV, input=inp, pipeline=5).execute_host(raw), np.abs(y), "zero runs env")
    # an all-zero stream stays all zero (exact), and a stream that is zero but for its last sample
    z = np.zeros_like(raw)
    p5 = sdr.FirPlan(h, D, mode=sdr.MODE_FM, input=inp, pipeline=5)
    assert not p5.execute_host(z).any()
    z[-2:] = (5, -3)
    yz = oracle.fir_decimate(unpack(z), D, h)
    check_fm(p5.execute_host(z), oracle.polar_discriminator(yz, 1 / TWO_PI), yz, 1 / TWO_PI, "zero but for the last sample", good_rel=None)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,T,D", [("i16", 64, 10), ("i16", 101, 5), ("cf32", 64, 10), ("i8", 64, 10), ("qi", 200, 8),
                                      ("i16", 511, 10), ("i16", 180, 8), ("qi", 32, 4), ("i8", 96, 6), ("i16", 199, 8), ("i8", 33, 14)])
def test_channelizer_per_channel_mixers(sdr, oracle, kind, T, D):
    """The channelizer as SURVEY 8 a7 words BASELINE config 3: one NCO per channel in front of the filter
    (create_freq_rotation / frequency_shift per channel, sdr_functions.h:63-74, bladerf_apt_demod.cpp:476-495;
    negative frequencies = the FM receiver's offset rotator, bladerf_fm_demod.cpp:237-241), then
    polyphase_decimate and the discriminator / envelope, all channels in one launch.  Channels read ONE
    shared wideband buffer (in_stride 0) or a buffer each; every channel against the oracle's
    frequency_shift -> fir_decimate -> demod on the same samples; a block of the channels launched with
    chan_first (a rank's shard) equals the same channels of the whole launch bit for bit."""
    rng = np.random.default_rng(T * 13 + D)
    n_ch, n = 11, 60_000 + 7 * D + 3
    # channel centres 0.09 apart (well outside each other's pass band, so every channel holds ONE constant-
    # envelope FM carrier and its angles are well conditioned), off the grid by a random bit, 0 and -0.45 exact
    frs = (np.arange(n_ch) - 5) * 0.09 + rng.uniform(-0.003, 0.003, n_ch)
    frs[5], frs[0] = 0.0, -0.45
    h = oracle.create_fir_filter(T, min(0.4 / D, 0.05), "hann")
    # wideband test signal: an FM carrier at every channel's frequency (so that each channel, once mixed to
    # zero, holds a constant-envelope FM signal) plus noise
    t = np.arange(n)
    x = np.zeros(n, np.complex128)
    for c in range(n_ch):
        ph = 2 * np.pi * (-frs[c] * t + 0.003 * np.sin(2 * np.pi * (1e-3 + 2e-4 * c) * t) / (1e-3 + 2e-4 * c) / (2 * np.pi) * 2 * np.pi)
        x += np.exp(1j * ph)
    x *= 0.9 / n_ch
    if kind in ("i16", "qi"):
        iq = np.empty(2 * n, np.int16)
        iq[0::2] = np.rint(2047 * x.real) + rng.integers(-3, 4, n)
        iq[1::2] = np.rint(2047 * x.imag) + rng.integers(-3, 4, n)
        xs = oracle.unpack_scale_i16(iq)
        raw, inp, esz = (iq, sdr.IN_I16_IQ, 4) if kind == "i16" else (iq.reshape(-1, 2)[:, ::-1].reshape(-1).copy(), sdr.IN_I16_QI, 4)
    elif kind == "i8":
        raw = np.empty(2 * n, np.int8)
        raw[0::2] = np.clip(np.rint(127 * x.real) + rng.integers(-1, 2, n), -128, 127)
        raw[1::2] = np.clip(np.rint(127 * x.imag) + rng.integers(-1, 2, n), -128, 127)
        xs, inp, esz = oracle.unpack_scale_i8(raw), sdr.IN_I8_IQ, 2
    else:
        raw = x.astype(np.complex64)
        xs, inp, esz = raw.astype(np.complex128), sdr.IN_CF32, 8
    ys = [oracle.fir_decimate(oracle.frequency_shift(xs, frs[c]), D, h) for c in range(n_ch)]
    d_in = sdr.DevBuf.from_numpy(raw)
    for mode, dt, eb in ((sdr.MODE_FM, np.float32, 4), (sdr.MODE_ENV, np.float32, 4), (sdr.MODE_IQ, np.complex64, 8)):
        plan = sdr.FirPlan(h, D, mode=mode, input=inp, mix_fr_channels=frs)
        # integer input with a window of up to 256 samples: the int8 matrix kernel with the mixers folded
        # into complex taps (even D: groups of eight outputs, the row form; odd D: of sixteen, the column form);
        # else the float-tile kernel with per-channel phasor tables
        want_kernel = "mfma_i8" if kind != "cf32" and (T - 1 + 3) // 4 * 4 + (7 if D % 2 == 0 else 15) * D + 1 <= 256 else "stream"
        assert plan.info()["kernel"] == want_kernel, (plan.info(), want_kernel)
        cnt = plan.out_count(n)
        stride = cnt + 3
        SENT = 0x7FC0DEAD
        d_out = sdr.DevBuf.from_numpy(np.full(stride * n_ch * (eb // 4), SENT, np.uint32))
        plan.execute_channels_dev(d_in, n, 0, 0, n_ch, d_out, stride)          # one shared wideband buffer
        sdr.sync()
        out = d_out.to_numpy(dt, stride * n_ch).reshape(n_ch, stride)
        assert (out[:, cnt:].view(np.uint32) == SENT).all()
        for c in range(n_ch):
            tag = f"channelizer {kind} {T}/{D} mode {mode} channel {c} fr {frs[c]:+.4f}"
            if mode == sdr.MODE_FM:
                check_fm(out[c, :cnt], oracle.polar_discriminator(ys[c], 1 / TWO_PI), ys[c], 1 / TWO_PI, tag)
            elif mode == sdr.MODE_ENV:
                check_linear(out[c, :cnt], np.abs(ys[c]), tag)
            else:
                check_linear(out[c, :cnt], ys[c], tag)
        # a rank's block of the channels: channels 4..8 launched on their own with chan_first = 4
        d_blk = sdr.DevBuf.from_numpy(np.full(stride * 5 * (eb // 4), SENT, np.uint32))
        plan.execute_channels_dev(d_in, n, 0, 4, 5, d_blk, stride)
        sdr.sync()
        blk = d_blk.to_numpy(dt, stride * 5).reshape(5, stride)
        assert np.array_equal(blk[:, :cnt].view(np.uint32), out[4:9, :cnt].view(np.uint32))
        # a buffer per channel (in_stride = n + 5): channel c's buffer holds the wideband signal delayed by c samples
        if mode == sdr.MODE_ENV:
            sin = n + 5
            per = 2 if kind != "cf32" else 1
            buf = np.zeros(sin * 3 * per, raw.dtype)
            for k in range(3):
                buf[(k * sin + k) * per: (k * sin + n) * per] = raw[: (n - k) * per]
            d_sep = sdr.DevBuf.from_numpy(buf)
            d_o3 = sdr.DevBuf(stride * 3 * eb)
            plan.execute_channels_dev(d_sep, n, sin, 5, 3, d_o3, stride)
            sdr.sync()
            o3 = d_o3.to_numpy(dt, stride * 3).reshape(3, stride)
            for k in range(3):
                xk = np.concatenate((np.zeros(k, np.complex128), xs[: n - k]))
                yk = oracle.fir_decimate(oracle.frequency_shift(xk, frs[5 + k]), D, h)
                check_linear(o3[k, :cnt], np.abs(yk), f"separate buffers, channel {5 + k}")
        # argument errors: channels beyond the plan's mixers; a plan without mixers
        with pytest.raises(sdr.SdrError):
            plan.execute_channels_dev(d_in, n, 0, 8, 4, d_out, stride)
    with pytest.raises(sdr.SdrError):
        sdr.FirPlan(h, D, mode=sdr.MODE_IQ, input=inp).execute_channels_dev(d_in, n, 0, 0, 1, d_out, stride)
    with pytest.raises(sdr.SdrError):
        sdr.FirPlan(h, D, mode=sdr.MODE_IQ, input=inp, mix_fr=0.1, mix_fr_channels=frs)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,T,D", [("i16", 64, 10), ("i16", 101, 5), ("i8", 64, 10), ("i16", 64, 1), ("i16", 64, 2), ("i8", 150, 4)])
def test_channelizer_matrix_kernel_zero_runs(sdr, oracle, kind, T, D):
    """The complex-tap matrix kernel computes z[m] = y[m] e^{-jw mD} and turns the discriminator's product by
    a constant instead of turning every y (i8_finish_cpx).  Where the reference's y is an exact zero -- zero
    pre-history, zero runs in mid-stream -- its output is 0 or +-pi by the signs of the OTHER y's components,
    and those are the mixed signal's signs: every lane that holds an exact zero goes back to the reference's
    own arithmetic (i8_zero_fix).  Streams with zero runs, lone samples and sub-resolution taps, per channel
    against the oracle with the exact-zero rule of check_fm; all-zero streams stay exactly zero."""
    rng = np.random.default_rng(T * 5 + D)
    h = oracle.create_fir_filter(T, min(0.8 / D, 0.45), "hann")
    h[1], h[-1] = -3e-14, 2e-15                    # sub-resolution taps next to the Hann window's exact zeros
    n = 24 * 1024 + 321 + 3 * D
    if kind == "i8":
        raw = rng.integers(-128, 128, 2 * n).astype(np.int8)
        inp, unpack = sdr.IN_I8_IQ, oracle.unpack_scale_i8
    else:
        raw = rng.integers(-2000, 2001, 2 * n).astype(np.int16)
        inp, unpack = sdr.IN_I16_IQ, oracle.unpack_scale_i16
    for i, (a, b) in enumerate([(2_000, 2_000 + 3 * T + 5), (1024 * D - T - 3, 1024 * D + 2 * T + 1), (7_001, 7_001 + T - 7)]):
        if b >= n - 2 * T:
            continue
        b = (b + D - 1) // D * D - (i & 1)
        raw[2 * a: 2 * b] = 0
    z0 = 12_000
    z1 = min(n - 3 * T, z0 + 10 * T + 30 * D)
    raw[2 * z0: 2 * z1] = 0
    for k, s_ in enumerate(range(z0 + T + 3, z1 - T, 2 * T + 3 * D + 1)):
        raw[2 * s_] = (-1) ** k * 77
        raw[2 * s_ + 1] = 55 if k % 3 else -91
    frs = np.array([0.0371, -0.2, 0.25, 0.0, -0.4999])
    plan = sdr.FirPlan(h, D, mode=sdr.MODE_FM, input=inp, mix_fr_channels=frs)
    assert plan.info()["kernel"] == "mfma_i8"
    cnt = plan.out_count(n)
    d_in = sdr.DevBuf.from_numpy(raw)
    d_out = sdr.DevBuf(cnt * frs.size * 4)
    plan.execute_channels_dev(d_in, n, 0, 0, frs.size, d_out, cnt)
    sdr.sync()
    out = d_out.to_numpy(np.float32, cnt * frs.size).reshape(frs.size, cnt)
    xs = unpack(raw)
    for c, fr in enumerate(frs):
        y = oracle.fir_decimate(oracle.frequency_shift(xs, fr), D, h)
        assert (np.abs(y) == 0).sum() >= 2 * T // D
        check_fm(out[c], oracle.polar_discriminator(y, 1 / TWO_PI), y, 1 / TWO_PI, f"channelizer zero runs {kind} {T}/{D} fr {fr}", good_rel=None)
    d_z = sdr.DevBuf.from_numpy(np.zeros_like(raw))
    plan.execute_channels_dev(d_z, n, 0, 0, frs.size, d_out, cnt)
    sdr.sync()
    assert not d_out.to_numpy(np.float32, cnt * frs.size).any()


@pytest.mark.gpu
@pytest.mark.parametrize("T,D", [(40, 10), (24, 14), (64, 14), (64, 9), (40, 13), (24, 3), (120, 2), (185, 10)])
@pytest.mark.parametrize("kind", ["i16", "i8"])
def test_channelizer_matrix_kernel_instantiations(sdr, oracle, kind, T, D):
    """The complex-tap instantiations of the int8 matrix kernel that test_channelizer_per_channel_mixers does not
    reach: row form (even D) with four and eight steps, column form (odd D), four / six / eight octets per helper
    lane, int16 and int8 pairs -- three channels each, discriminator and envelope against the oracle's
    frequency_shift -> fir_decimate -> demod, over several tiles with a ragged end."""
    rng = np.random.default_rng(1000 * T + D)
    n = 3 * 1024 * D + 5 * D + 3
    if kind == "i8":
        raw = rng.integers(-128, 128, 2 * n).astype(np.int8)
        inp, xs = sdr.IN_I8_IQ, oracle.unpack_scale_i8(raw)
    else:
        raw = rng.integers(-32768, 32768, 2 * n).astype(np.int16)
        inp, xs = sdr.IN_I16_IQ, oracle.unpack_scale_i16(raw)
    h = oracle.create_fir_filter(T, min(0.8 / D, 0.45), "hann")
    frs = np.array([0.0371, -0.31, 0.4999])
    ys = [oracle.fir_decimate(oracle.frequency_shift(xs, fr), D, h) for fr in frs]
    d_in = sdr.DevBuf.from_numpy(raw)
    for mode in (sdr.MODE_FM, sdr.MODE_ENV):
        plan = sdr.FirPlan(h, D, mode=mode, input=inp, mix_fr_channels=frs)
        assert plan.info()["kernel"] == "mfma_i8", plan.info()
        cnt = plan.out_count(n)
        d_out = sdr.DevBuf(cnt * frs.size * 4)
        plan.execute_channels_dev(d_in, n, 0, 0, frs.size, d_out, cnt)
        sdr.sync()
        out = d_out.to_numpy(np.float32, cnt * frs.size).reshape(frs.size, cnt)
        for c, fr in enumerate(frs):
            tag = f"complex-tap instantiation {kind} {T}/{D} mode {mode} fr {fr}"
            if mode == sdr.MODE_FM:
                check_fm(out[c], oracle.polar_discriminator(ys[c], 1 / TWO_PI), ys[c], 1 / TWO_PI, tag, good_rel=None)
            else:
                check_linear(out[c], np.abs(ys[c]), tag)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [1, 2, 3, 4])
def test_fft4096_kernel_variants(sdr, oracle, variant):
    """The 4096-point / hop-2048 waterfall has four kernels (sdr_fft_plan_set_variant): one frame per
    workgroup with barrier-separated exchanges, two frames per lane, and one frame per wavefront with
    four or eight waves per CU (the automatic choice).  Each is held to the oracle
    (bladerf_main_v1.cpp:183-196 semantics + window + hop) on a stream long enough to select it, with
    a ragged tail, and an untouched row behind the last frame."""
    nfft, hop = 4096, 2048
    frames = 4 * 256 + 37                       # >= 4 frames per CU selects the long-stream kernels
    n = (frames - 1) * hop + nfft + 1234        # ragged: 1234 samples past the last frame
    rng = np.random.default_rng(40 + variant)
    iq = oracle.synth_fm_i16(n, fs=61.44e6, ftone=250e3, dev=2.0e6, amp=1200.0)
    iq = (iq + rng.integers(-64, 65, iq.size)).astype(np.int16)
    w = sdr.window("blackman", nfft)
    plan = sdr.FftPlan(nfft, hop, w)
    plan.set_variant(variant)
    assert plan.frame_count(n) == frames
    d_iq = sdr.DevBuf.from_numpy(iq)
    d_out = sdr.DevBuf((frames + 1) * nfft * 4)
    sent = np.full(nfft, 31337.5, np.float32)
    sdr.capi.check(sdr.capi.lib().sdr_dev_upload(d_out.at(frames * nfft * 4), sent.ctypes.data_as(sdr.capi.C.c_void_p),
                                                 sdr.capi.C.c_size_t(sent.nbytes)))
    assert plan.execute_dev(d_iq, n, d_out, frames) == frames
    sdr.sync()
    got = d_out.to_numpy(np.float32, (frames + 1) * nfft).reshape(frames + 1, nfft)
    assert (got[frames] == 31337.5).all()
    want_db, want_lin = oracle.fft_power_db_i16(iq, nfft, hop, w, linear=True)
    assert want_db.shape == (frames, nfft)
    g = got[:frames].astype(np.float64)
    lin = 10.0 ** (g / 10.0)
    peak = want_lin.max(axis=1, keepdims=True)
    assert (np.abs(lin - want_lin) <= 1e-5 * peak).all()
    strong = want_db >= want_db.max(axis=1, keepdims=True) - 50.0
    assert np.abs(g - want_db)[strong].max() <= 1e-3
    with pytest.raises(sdr.SdrError):
        plan.set_variant(5)


@pytest.mark.gpu
@pytest.mark.parametrize("T,D,mix", [(511, 15, 2400 / 62400.0), (195, 15, 2400 / 62400.0), (164, 4, 0.0), (64, 7, -0.123),
                                     (2047, 10, 0.0371)])
def test_ksplit_kernel_short_real_streams(sdr, oracle, T, D, mix):
    """Pipeline 6 (fir_ksplit_kernel): real float input, the filter split over the four waves of a
    workgroup -- the latency kernel for short streams such as APT stage 2
    (bladerf_apt_demod.cpp:744-754: frequency_shift -> polyphase_decimate -> abs) and the FM audio
    filters (bladerf_fm_demod.cpp:301-307).  Against the oracle in every mode it supports, on ragged
    lengths, through a size hint, through time shards (bit-identical: one summation order per output)
    and through the block hand-off."""
    rng = np.random.default_rng(T * 31 + D)
    h = oracle.create_fir_filter(T, min(0.8 / D, 0.9), "hann") * rng.uniform(0.5, 1.0, T)
    for n in (124_799, 4097, D * 256 * 3, 100, D - 1, 0):
        xr = (rng.standard_normal(n) * 0.3).astype(np.float32)
        xs = xr.astype(np.float64).astype(np.complex128)
        if mix != 0.0 and n:
            xs = oracle.frequency_shift(xr.astype(np.float64), mix)
        y = oracle.fir_decimate(xs, D, h) if n >= D else np.zeros(0, np.complex128)
        for mode, want in ((sdr.MODE_IQ, y), (sdr.MODE_ENV, np.abs(y)), (sdr.MODE_REAL, y.real)):
            plan = sdr.FirPlan(h, D, mode=mode, input=sdr.IN_F32_REAL, mix_fr=mix, pipeline=6)
            assert plan.info()["kernel"] == "ksplit"
            check_linear(plan.execute_host(xr), want, f"ksplit T={T} D={D} n={n} mode={mode}")
    # chosen from the size hint, not without it, and never for the discriminator
    small = sdr.FirPlan(h, D, mode=sdr.MODE_ENV, input=sdr.IN_F32_REAL, mix_fr=mix, size_hint=124_799)
    big = sdr.FirPlan(h, D, mode=sdr.MODE_ENV, input=sdr.IN_F32_REAL, mix_fr=mix, size_hint=1 << 28)
    none = sdr.FirPlan(h, D, mode=sdr.MODE_ENV, input=sdr.IN_F32_REAL, mix_fr=mix)
    long_kernels = ("stream", "float_dma_tile")     # long streams: a throughput kernel
    assert small.info()["kernel"] == "ksplit" and big.info()["kernel"] in long_kernels and none.info()["kernel"] in long_kernels
    with pytest.raises(sdr.SdrError):
        sdr.FirPlan(h, D, mode=sdr.MODE_FM, input=sdr.IN_F32_REAL, pipeline=6)
    with pytest.raises(sdr.SdrError):
        sdr.FirPlan(h, D, mode=sdr.MODE_ENV, input=sdr.IN_CF32, pipeline=6)
    # time shards and block hand-off reproduce the one-piece result bit for bit
    n = 300_017
    xr = (rng.standard_normal(n) * 0.3).astype(np.float32)
    whole = small.execute_host(xr)
    cnt = whole.size
    d_all = sdr.DevBuf.from_numpy(xr)
    parts = []
    for gidx in range(3):
        ob, oe = cnt * gidx // 3, cnt * (gidx + 1) // 3
        lo, hi = max(0, ob * D - (T - 1)), min(n, (oe - 1) * D + 1)
        d_out = sdr.DevBuf((oe - ob) * 4)
        small.execute_range_dev(d_all.at(lo * 4), lo, hi - lo, n, ob, oe, d_out)
        sdr.sync()
        parts.append(d_out.to_numpy(np.float32, oe - ob))
    assert np.array_equal(np.concatenate(parts), whole)
    st = sdr.FirStream(small, 50_000)
    outs = []
    for a in range(0, n, 37_123):
        st.push(xr[a: a + 37_123])
        outs.append(st.pop())
    assert np.array_equal(np.concatenate(outs), whole)


@pytest.mark.gpu
@pytest.mark.parametrize("T,D,n", [(7, 1, 1_200_000), (16, 2, 1_200_000), (33, 3, 2_000_003), (12, 1, 700_001)])
def test_matrix_kernel_many_short_tiles(sdr, oracle, T, D, n):
    """Round 3: the int8 matrix kernel on SMALL decimation factors with SEVERAL tiles per workgroup -- short
    phases in which the helper waves finish one tile (discriminator) while the matrix wave of the same SIMD
    multiplies the next.  With packed float32 instructions in the helper waves' arithmetic (hipcc's SLP
    vectoriser) the even outputs of the last 32 of a column block came out wrong in up to 97 % of launches at
    7 taps / 1; earlier tests never put two tiles on one workgroup at such shapes (their streams had fewer than
    256 tiles), so they passed.  Every launch of several must match the oracle and the first launch bit for bit."""
    rng = np.random.default_rng(T * 100 + D)
    iq = oracle.synth_fm_i16(n, fs=1.0e6, ftone=3000.0, dev=20000.0, amp=1900.0)
    iq = (iq + rng.integers(-40, 41, iq.size)).astype(np.int16)
    h = oracle.create_fir_filter(T, min(0.8 / D, 0.9), "hann")
    y = oracle.chain_i16(iq, D, h, oracle.MODE_IQ)
    fm = oracle.polar_discriminator(y, 1 / TWO_PI)
    for mode, inp, raw in ((sdr.MODE_FM, sdr.IN_I16_IQ, iq), (sdr.MODE_FM, sdr.IN_I8_IQ, (iq >> 4).astype(np.int8))):
        plan = sdr.FirPlan(h, D, mode=mode, input=inp, pipeline=5)
        assert plan.info()["kernel"] == "mfma_i8"
        if inp == sdr.IN_I8_IQ:
            y8 = oracle.fir_decimate(oracle.unpack_scale_i8(raw), D, h)
            want, yy = oracle.polar_discriminator(y8, 1 / TWO_PI), y8
        else:
            want, yy = fm, y
        first = None
        for rep in range(6):
            got = plan.execute_host(raw)
            if first is None:
                first = got
                check_fm(got, want, yy, 1 / TWO_PI, f"T={T} D={D} n={n} in={inp}")
            else:
                bad = np.flatnonzero(got.view(np.uint32) != first.view(np.uint32))
                assert bad.size == 0, (T, D, n, inp, rep, bad[:8], got[bad[:4]], first[bad[:4]])
