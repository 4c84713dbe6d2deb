"""GPU tests at the sizes BASELINE.json names (configs[2] channelizer, configs[3] waterfall, configs[4] APT chain) and of
the multi-rank splits with the HIP kernels doing the per-rank compute (tests/test_shard_gloo.py
covers the collectives with the CPU oracle standing in for the kernels).

Full-size runs are checked the way ③ asks: against the oracle on slices it finishes in seconds
(first / middle / last channels and frames) plus size-independent properties -- sentinel-guarded
output strides, untouched tails, and bit-identity between a sharded and a one-piece run.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TWO_PI = 2.0 * np.pi

from parity_checks import check_fm, check_linear, apt_stage2  # noqa: E402


@pytest.fixture(scope="module")
def sdr():
    import sdr_amd
    sdr_amd.capi.lib()
    sdr_amd.init(0)
    return sdr_amd


def fill(sdr, buf, value, count, dtype=np.float32, offset_bytes=0, chunk=1 << 24):
    """fill `count` elements of a DevBuf with a sentinel (chunked uploads)"""
    blk = np.full(min(chunk, count), value, dtype)
    done = 0
    while done < count:
        k = min(chunk, count - done)
        sdr.capi.check(sdr.capi.lib().sdr_dev_upload(buf.at(offset_bytes + done * blk.itemsize),
                                                     blk.ctypes.data_as(C.c_void_p), C.c_size_t(k * blk.itemsize)))
        done += k


def test_cfg3_full_size(sdr, oracle):
    """BASELINE configs[2]: 1024 channels x 2^20 samples, 64 taps / 10, FM -- one batched launch.
    Oracle on channels 0 / 511 / 1023; every channel's output row sits between sentinels that must
    survive (a blockIdx.y or stride slip would land there)."""
    n_ch, n, D, T = 1024, 1 << 20, 10, 64
    h = sdr.create_fir_filter(T, 1.0 / D, "hann")
    plan = sdr.FirPlan(h, D, mode=sdr.MODE_FM)
    d_in = sdr.DevBuf(n_ch * n * 4)
    for c in range(n_ch):
        sdr.capi.synth_fm_i16_dev(d_in.at(c * n * 4), n, fs=1.0e6, ftone=300.0 + 2700.0 * c / (n_ch - 1),
                                  dev=5000.0, noise_lsb=2, seed=100 + c)
    cnt = plan.out_count(n)
    gap = 5
    stride = cnt + gap
    SENT = np.float32(-12345.25)
    d_out = sdr.DevBuf(n_ch * stride * 4)
    fill(sdr, d_out, SENT, n_ch * stride)
    plan.execute_batched_dev(d_in, n, n, n_ch, d_out, stride)
    sdr.sync()
    out = d_out.to_numpy(np.float32, n_ch * stride).reshape(n_ch, stride)
    assert (out[:, cnt:] == SENT).all(), "a channel wrote past its row"
    assert not (out[:, :cnt] == SENT).any(), "outputs missing"
    assert np.isfinite(out[:, :cnt]).all() and np.abs(out[:, :cnt]).max() <= 0.5
    for c in (0, 511, 1023):
        iq = d_in.to_numpy(np.int16, 2 * n, c * n * 4)
        want = oracle.chain_i16(iq, D, h, oracle.MODE_FM)
        assert np.abs(out[c, :cnt].astype(np.float64) - want).max() <= 1e-5 * 0.5, c
    # every channel equals the same channel filtered on its own (same kernel, one channel per launch)
    d_one = sdr.DevBuf(cnt * 4)
    for c in (1, 700):
        plan.execute_dev(d_in.at(c * n * 4), n, d_one, cnt)
        sdr.sync()
        assert np.array_equal(d_one.to_numpy(np.float32, cnt), out[c, :cnt]), c


def test_cfg4_full_size(sdr, oracle):
    """BASELINE configs[3]: 2^28 samples, 4096-point Blackman frames at hop 2048 -> 131071 dB rows.
    Oracle on the first / middle / last frames; the row after the last frame is a sentinel."""
    n, nfft, hop = 1 << 28, 4096, 2048
    w = sdr.window("blackman", nfft)
    plan = sdr.FftPlan(nfft, hop, w)
    d_in = sdr.DevBuf(n * 4)
    sdr.capi.synth_fm_i16_dev(d_in, n, fs=61.44e6, ftone=250e3, dev=2.0e6, amp=1200.0, noise_lsb=64, seed=3)
    frames = plan.frame_count(n)
    assert frames == 131071
    SENT = np.float32(777.125)
    d_out = sdr.DevBuf((frames + 1) * nfft * 4)
    fill(sdr, d_out, SENT, nfft, offset_bytes=frames * nfft * 4)
    assert plan.execute_dev(d_in, n, d_out, frames) == frames
    sdr.sync()
    assert (d_out.to_numpy(np.float32, nfft, frames * nfft * 4) == SENT).all()
    worst = 0.0
    for f0 in (0, 1, frames // 2, 65535, 65536, frames - 4):
        iq = d_in.to_numpy(np.int16, 2 * (nfft + 3 * hop), f0 * hop * 4)
        want_db, want_lin = oracle.fft_power_db_i16(iq, nfft, hop, w, linear=True)
        got = d_out.to_numpy(np.float32, 4 * nfft, f0 * nfft * 4).reshape(4, nfft).astype(np.float64)
        lin = 10.0 ** (got / 10.0)
        worst = max(worst, float(np.abs(lin - want_lin).max() / want_lin.max()))
        strong = want_db >= want_db.max() - 50.0
        assert np.abs(got - want_db)[strong].max() <= 1e-3, f0
    assert worst <= 1e-5, worst
    # property over the whole output: every frame's total power equals the windowed samples' power
    # (Parseval, 1/N^2 scaling of fftInPlace(x, 1/N)) -- checked on a strided subset of frames
    win = (w / 2048.0)
    for f in range(7, frames, 9973):
        iq = d_in.to_numpy(np.int16, 2 * nfft, f * hop * 4).astype(np.float64)
        x = (iq[0::2] + 1j * iq[1::2]) * win
        p_time = float((np.abs(x) ** 2).sum()) / nfft
        row = d_out.to_numpy(np.float32, nfft, f * nfft * 4).astype(np.float64)
        p_freq = float((10.0 ** (row / 10.0)).sum())
        assert abs(p_freq - p_time) <= 1e-4 * p_time, (f, p_freq, p_time)


def test_cfg5_full_size(sdr, oracle):
    """BASELINE configs[4]: the APT chain at 2^26 samples -- 511 taps / 10 + discriminator (int8 matrix
    kernel), +2400/62400 mixer folded into 511 complex taps / 15, envelope (K-split kernel, chosen from
    the size hint).  Oracle on the head of the stream (zero pre-history) and on its tail (a slice with
    its own history: a constant offset of the mixer's phase does not change an envelope); sentinels
    behind both outputs."""
    n, D1, D2, T = 1 << 26, 10, 15, 511
    fr = 2400 / 62400.0
    h1 = sdr.create_fir_filter(T, 20800 / 62400.0 / 5, "hann")
    h2 = sdr.create_fir_filter(T, fr, "hann")
    p1 = sdr.FirPlan(h1, D1, mode=sdr.MODE_FM)
    n6 = p1.out_count(n)
    p2 = sdr.FirPlan(h2, D2, mode=sdr.MODE_ENV, input=sdr.IN_F32_REAL, mix_fr=fr, size_hint=n6)
    n9 = p2.out_count(n6)
    assert p1.info()["kernel"] == "mfma_i8" and p2.info()["kernel"] == "ksplit"
    d_in = sdr.DevBuf(n * 4)
    sdr.capi.synth_fm_i16_dev(d_in, n, fs=624e3, ftone=2400.0, dev=17000.0, amp=1800.0, noise_lsb=4, seed=5)
    SENT = np.float32(-777.5)
    d_x6, d_x9 = sdr.DevBuf((n6 + 8) * 4), sdr.DevBuf((n9 + 8) * 4)
    fill(sdr, d_x6, SENT, 8, offset_bytes=n6 * 4)
    fill(sdr, d_x9, SENT, 8, offset_bytes=n9 * 4)
    assert p1.execute_dev(d_in, n, d_x6, n6) == n6
    assert p2.execute_dev(d_x6, n6, d_x9, n9) == n9
    sdr.sync()
    assert (d_x6.to_numpy(np.float32, 8, n6 * 4) == SENT).all() and (d_x9.to_numpy(np.float32, 8, n9 * 4) == SENT).all()
    # head: the oracle's chain on the first 2^19 samples
    m = 1 << 19
    w9, w6 = oracle.apt_chain_i16(d_in.to_numpy(np.int16, 2 * m), D1, h1, fr, D2, h2)
    g6 = d_x6.to_numpy(np.float32, w6.size).astype(np.float64)
    g9 = d_x9.to_numpy(np.float32, w9.size).astype(np.float64)
    # stage 1: the whole head, start-up transient included, by the discriminator's bars; stage 2: the
    # reference's arithmetic on the GPU's own x6 (all of it), and the oracle chain's x9 past the transient
    y4 = oracle.fir_decimate(oracle.unpack_scale_i16(d_in.to_numpy(np.int16, 2 * m)), D1, h1)
    check_fm(g6.astype(np.float32), w6, y4, 1 / TWO_PI, "cfg5 head x6")
    check_linear(g9, apt_stage2(oracle, g6, fr, D2, h2), "cfg5 head x9")
    s6 = T // D1 + 2
    s9 = s6 // D2 + 2 + T // D2
    assert np.abs(g9 - w9)[s9:-(T // D2) - 2].max() <= 5e-6
    # tail: the slice from sample s0 (a multiple of D1*D2, so both decimators stay in phase) to the end
    s0 = (n - 450_000) // (D1 * D2) * (D1 * D2)
    L = n - s0
    w9t, w6t = oracle.apt_chain_i16(d_in.to_numpy(np.int16, 2 * L, s0 * 4), D1, h1, fr, D2, h2)
    o6, o9 = s0 // D1, s0 // (D1 * D2)
    g6t = d_x6.to_numpy(np.float32, n6 - o6, o6 * 4).astype(np.float64)
    g9t = d_x9.to_numpy(np.float32, n9 - o9, o9 * 4).astype(np.float64)
    k6 = min(g6t.size, w6t.size)
    k9 = min(g9t.size, w9t.size)
    assert k6 >= L // D1 - 2 and k9 >= L // (D1 * D2) - 2
    assert np.abs(g6t[:k6] - w6t[:k6])[s6:].max() <= 5e-6            # past the slice's own start-up transient
    assert np.abs(g9t[:k9] - w9t[:k9])[s9:].max() <= 5e-6
    assert np.isfinite(d_x9.to_numpy(np.float32, n9)).all()


def test_headline_long_launch_dynamic_tile_order(sdr, oracle):
    """A launch long enough for the int8 matrix kernel's dynamic tile hand-out (128 tiles per workgroup
    and more; 64 taps / 10 on 2^29 samples): which workgroup computes a tile must not show in the
    results.  The whole stream is launched on two HIP streams at once (each has its own tile counters),
    both outputs must be identical bit for bit, equal three short time shards (static order) on their
    ranges, match the oracle on the stream's head, and leave the sentinels behind the outputs alone.
    A second launch on the same stream finds the counters zeroed by the first."""
    hip = C.CDLL("libamdhip64.so")
    n, D, T = (1 << 29) + 12345, 10, 64
    h = sdr.create_fir_filter(T, 0.1, "hann")
    plan = sdr.FirPlan(h, D, mode=sdr.MODE_FM)
    assert plan.info()["kernel"] == "mfma_i8"
    cnt = plan.out_count(n)
    d_in = sdr.DevBuf(n * 4)
    sdr.capi.synth_fm_i16_dev(d_in, n, fs=20e6, ftone=1000.0, dev=5000.0, amp=2000.0, noise_lsb=3, seed=11)
    SENT = np.float32(-777.5)
    streams = [C.c_void_p(), C.c_void_p()]
    for s_ in streams:
        assert hip.hipStreamCreate(C.byref(s_)) == 0
    outs = [sdr.DevBuf((cnt + 8) * 4) for _ in range(3)]
    for o in outs:
        fill(sdr, o, SENT, 8, offset_bytes=cnt * 4)
    try:
        assert plan.execute_dev(d_in, n, outs[0], cnt, stream=streams[0].value) == cnt
        assert plan.execute_dev(d_in, n, outs[1], cnt, stream=streams[1].value) == cnt
        assert plan.execute_dev(d_in, n, outs[2], cnt, stream=streams[0].value) == cnt     # same stream again
        sdr.sync()
    finally:
        for s_ in streams:
            hip.hipStreamDestroy(s_)
    a = outs[0].to_numpy(np.float32, cnt + 8)
    assert (a[cnt:] == SENT).all()
    for o in outs[1:]:
        b = o.to_numpy(np.float32, cnt + 8)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        del b
    # the handle hipStreamPerThread ((hipStream_t)2) names a DIFFERENT stream in every host
    # thread: two threads launching the plan on it at once must not share tile counters (such handles run
    # the static order; counter blocks are keyed by hipStreamGetId)
    import threading
    per_thread = C.c_void_p(2)
    t_outs = [sdr.DevBuf((cnt + 8) * 4) for _ in range(2)]
    errs = []

    def worker(o):
        try:
            assert plan.execute_dev(d_in, n, o, cnt, stream=per_thread.value) == cnt
            assert hip.hipStreamSynchronize(per_thread) == 0
        except Exception as e:          # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(o,)) for o in t_outs]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    sdr.sync()
    for o in t_outs:
        b = o.to_numpy(np.float32, cnt)
        assert np.array_equal(a[:cnt].view(np.uint32), b.view(np.uint32))
        del b
    del t_outs
    # three short shards of the same stream (few tiles per workgroup: the static order)
    for ob in (0, cnt // 2 - 7, cnt - 300_000):
        oe = ob + 300_000
        lo = max(0, ob * D - (T - 1))
        hi = min(n, (oe + 1) * D + 1)
        d_out = sdr.DevBuf((oe - ob) * 4)
        plan.execute_range_dev(d_in.at(4 * lo), lo, hi - lo, n, ob, oe, d_out)
        sdr.sync()
        assert np.array_equal(d_out.to_numpy(np.float32, oe - ob), a[ob:oe]), ob
    # the oracle on the head of the stream
    m = 1 << 18
    iq = d_in.to_numpy(np.int16, 2 * m)
    want = oracle.chain_i16(iq, D, h, oracle.MODE_FM, 1.0 / TWO_PI)
    y = oracle.chain_i16(iq, D, h, oracle.MODE_IQ)
    check_fm(a[: want.size], want, y, 1 / TWO_PI, "head of the long launch")     # start-up transient included


@pytest.mark.parametrize("pipe", [1, 5])
def test_two_ranks_time_split_through_hip(sdr, oracle, pipe):
    """shard.plan_time_shards with the HIP kernels as the per-rank compute: two 'ranks' run back to
    back on this GPU, each holding only its own resident window, and their concatenation must equal
    the one-piece launch bit for bit -- for the vector-unit kernel (one summation order per output) and
    for the int8 matrix kernel (an output is one exact integer sum over its whole window, whatever
    tile or shard it falls into)."""
    from sdr_amd import shard
    n_total, D, T = 3_000_017, 10, (64 if pipe == 1 else 511)
    h = sdr.create_fir_filter(T, 0.8 / D, "hann")
    iq = oracle.synth_fm_i16(n_total, fs=2.0e6, ftone=1500.0, dev=30000.0)
    rng = np.random.default_rng(1)
    iq = (iq + rng.integers(-30, 31, iq.size)).astype(np.int16)
    for mode, smode in ((sdr.MODE_FM, shard.MODE_FM), (sdr.MODE_ENV, shard.MODE_ENV)):
        plan = sdr.FirPlan(h, D, mode=mode, pipeline=pipe)
        cnt = plan.out_count(n_total)
        d_all = sdr.DevBuf.from_numpy(iq)
        d_one = sdr.DevBuf(cnt * 4)
        assert plan.execute_dev(d_all, n_total, d_one, cnt) == cnt
        sdr.sync()
        one = d_one.to_numpy(np.float32, cnt)
        parts = []
        for sh in shard.plan_time_shards(n_total, D, T, smode, 2):
            d_res = sdr.DevBuf.from_numpy(iq[2 * sh.lo: 2 * sh.hi])          # this rank's window only
            d_out = sdr.DevBuf(max(sh.n_out, 1) * 4)
            plan.execute_range_dev(d_res, sh.lo, sh.n_resident, n_total, sh.out_begin, sh.out_end, d_out)
            sdr.sync()
            parts.append(d_out.to_numpy(np.float32, sh.n_out))
        got = np.concatenate(parts)
        assert got.shape == one.shape
        assert np.array_equal(got, one)
        # and the one-piece result is the oracle's
        want = oracle.chain_i16(iq, D, h, oracle.MODE_FM if mode == sdr.MODE_FM else oracle.MODE_ENV)
        if mode == sdr.MODE_FM:
            check_fm(one, want, oracle.chain_i16(iq, D, h, oracle.MODE_IQ), 1 / TWO_PI, f"pipe {pipe} fm")
        else:
            check_linear(one, want, f"pipe {pipe} env")


def test_two_ranks_channel_and_frame_split_through_hip(sdr, oracle):
    """shard.plan_channel_shards / plan_frame_shards with the HIP kernels as the per-rank compute"""
    from sdr_amd import shard
    # channels: 37 channels over 2 ranks (19 + 18)
    n_ch, n, D, T = 37, 65_536, 10, 64
    h = sdr.create_fir_filter(T, 1.0 / D, "hann")
    plan = sdr.FirPlan(h, D, mode=sdr.MODE_FM)
    d_in = sdr.DevBuf(n_ch * n * 4)
    for c in range(n_ch):
        sdr.capi.synth_fm_i16_dev(d_in.at(c * n * 4), n, fs=1.0e6, ftone=300.0 + 70.0 * c, dev=5000.0, noise_lsb=2, seed=c)
    cnt = plan.out_count(n)
    d_one = sdr.DevBuf(n_ch * cnt * 4)
    plan.execute_batched_dev(d_in, n, n, n_ch, d_one, cnt)
    sdr.sync()
    one = d_one.to_numpy(np.float32, n_ch * cnt)
    parts = []
    for c0, k in shard.plan_channel_shards(n_ch, 2):
        host = d_in.to_numpy(np.int16, 2 * k * n, c0 * n * 4)                  # what this rank would hold
        d_r = sdr.DevBuf.from_numpy(host)
        d_o = sdr.DevBuf(k * cnt * 4)
        plan.execute_batched_dev(d_r, n, n, k, d_o, cnt)
        sdr.sync()
        parts.append(d_o.to_numpy(np.float32, k * cnt))
    assert np.array_equal(np.concatenate(parts), one)
    want = oracle.chain_i16(d_in.to_numpy(np.int16, 2 * n, 36 * n * 4), D, h, oracle.MODE_FM)
    assert np.abs(one[36 * cnt:].astype(np.float64) - want).max() <= 1e-5 * 0.5
    # frames: 4096-point frames at hop 2048 over 2 ranks
    nfft, hop, ns = 4096, 2048, 1 << 22
    w = sdr.window("blackman", nfft)
    fplan = sdr.FftPlan(nfft, hop, w)
    d_s = sdr.DevBuf(ns * 4)
    sdr.capi.synth_fm_i16_dev(d_s, ns, fs=61.44e6, ftone=250e3, dev=2.0e6, amp=1200.0, noise_lsb=64, seed=3)
    frames = fplan.frame_count(ns)
    d_f = sdr.DevBuf(frames * nfft * 4)
    assert fplan.execute_dev(d_s, ns, d_f, frames) == frames
    sdr.sync()
    one = d_f.to_numpy(np.float32, frames * nfft)
    parts = []
    for f0, f1, lo, hi in shard.plan_frame_shards(ns, nfft, hop, 2):
        host = d_s.to_numpy(np.int16, 2 * (hi - lo), lo * 4)
        d_r = sdr.DevBuf.from_numpy(host)
        d_o = sdr.DevBuf((f1 - f0) * nfft * 4)
        assert fplan.execute_dev(d_r, hi - lo, d_o, f1 - f0) == f1 - f0
        sdr.sync()
        parts.append(d_o.to_numpy(np.float32, (f1 - f0) * nfft))
    got = np.concatenate(parts)
    # different kernels serve the two sizes (one frame per wavefront for the long stream, one frame per
    # workgroup for the halves): different summation orders, so the rows agree like each agrees with the
    # oracle -- 1e-3 dB on bins within 50 dB of the frame's peak, 1e-5 of the peak power everywhere
    assert got.shape == one.shape
    g2, o2 = got.reshape(frames, nfft).astype(np.float64), one.reshape(frames, nfft).astype(np.float64)
    strong = o2 >= o2.max(axis=1, keepdims=True) - 50.0
    assert np.abs(g2 - o2)[strong].max() <= 1e-3
    gl, ol = 10.0 ** (g2 / 10.0), 10.0 ** (o2 / 10.0)
    assert (np.abs(gl - ol) <= 1e-5 * ol.max(axis=1, keepdims=True)).all()
