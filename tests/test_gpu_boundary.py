"""GPU tests of the drop-in boundary: the reference's C++ call sites recompiled against the shim
header (examples/apt_chain_example.cpp, built by __graft_entry__.build()), and the block
hand-off with persistent filter history (sdr_fir_stream_*)."""
import os
import subprocess

import numpy as np
import pytest

from parity_checks import check_fm, check_linear, apt_stage2  # noqa: E402  (tests/ is on sys.path under pytest)

pytestmark = pytest.mark.gpu
TWO_PI = 2.0 * np.pi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLE = os.path.join(ROOT, "examples", "bin", "apt_chain_example")


@pytest.fixture(scope="module")
def sdr():
    import sdr_amd
    sdr_amd.init(0)
    return sdr_amd


def test_reference_loop_through_shims(golden, oracle, tmp_path):
    """bladerf_apt_demod.cpp:728-754 compiled against include/sdr_hip_shims.hpp vs the golden
    outputs of the reference's own functions."""
    assert os.path.exists(EXAMPLE), "run __graft_entry__.build() first"
    src = tmp_path / "apt.sc16"
    golden["apt_iq"].tofile(src)
    pfx = str(tmp_path / "out")
    r = subprocess.run([EXAMPLE, str(src), pfx], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    x6 = np.fromfile(pfx + ".x6.f64", np.float64)
    x9 = np.fromfile(pfx + ".x9.f64", np.float64)
    x6f = np.fromfile(pfx + ".x6f.f32", np.float32).astype(np.float64)
    x9f = np.fromfile(pfx + ".x9f.f32", np.float32).astype(np.float64)
    w6, w9 = golden["apt_x6"], golden["apt_x9"]
    assert x6.shape == w6.shape == x6f.shape and x9.shape == w9.shape == x9f.shape
    h1, h2 = golden["apt_lpf_fm"], golden["apt_lpf_am"]
    y4 = oracle.fir_decimate(oracle.unpack_scale_i16(golden["apt_iq"]), 10, h1)
    s9 = (200 // 10 + 2) // 15 + 2 + 195 // 15
    for got6, got9 in ((x6, x9), (x6f, x9f)):
        check_fm(got6, w6, y4, 1 / TWO_PI, "shim x6")                       # start-up transient included
        check_linear(got9, apt_stage2(oracle, got6, 2400 / 62400.0, 15, h2), "shim x9")
        assert np.abs(got9 - w9)[s9:].max() <= 1e-5 * 0.5
    # decimate_vec (rate 2.5) and filter_vec (7 taps) on the shim path's x6, against the oracle
    dec = np.fromfile(pfx + ".dec.f64", np.float64)
    want_dec, _, oob = oracle.decimate_vec(x6, 2.5)
    assert oob == 0 and np.array_equal(dec, want_dec)             # a pick: bit exact
    filt = np.fromfile(pfx + ".filt.f64", np.float64)
    want_filt = oracle.filter_vec(x6, [0.1, -0.2, 0.3, 0.4, 0.3, -0.2, 0.1])
    assert np.abs(filt - want_filt).max() <= 1e-6
    assert np.array_equal(filt[-3:], want_filt[-3:])              # the reference's truncated tail rows
    # FM audio stage through the shim (bladerf_fm_demod.cpp:301-313) against the oracle chain
    pcm = np.fromfile(pfx + ".pcm.u8", np.uint8)
    h_de = np.fromfile(pfx + ".lpf_de.f32", np.float32)
    h_a = np.fromfile(pfx + ".lpf_a.f32", np.float32)
    assert h_de.size == 64 and h_a.size == 101
    _, want_pcm = oracle.fm_audio_chain(np.fromfile(pfx + ".x6f.f32", np.float32), h_de, h_a, 6)
    assert pcm.shape == want_pcm.shape
    d = np.abs(pcm.astype(np.int32) - want_pcm.astype(np.int32))
    assert d.max() <= 1 and np.count_nonzero(d) <= max(4, pcm.size // 500)


def test_shims_steady_state_allocates_nothing():
    """the Level-1 shims keep the plans they have built (per-thread cache) and their device
    scratch, so a second call with the same filter makes no device / pinned allocation (sdr_alloc_count
    stands still) -- at the reference's own block sizes, 131072 samples (FM) and 1 248 000 (APT).  The
    same calls with a fresh plan each time do allocate (the counter counts), and the hand-off allocates
    only at stream creation."""
    import json
    exe = os.path.join(ROOT, "examples", "bin", "block_latency")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    by = {}
    for row in rows:
        by.setdefault(row["path"], []).append(row)
    assert len(by["shim"]) == 3 and len(by["plan_per_call"]) == 2
    for row in by["shim"] + by["stream"] + by["stream_zc"] + by["stream_pipelined"]:
        assert row["allocations_in_steady_state"] == 0, row
    for row in by["plan_per_call"]:
        assert row["allocations_in_steady_state"] > 0, row


def test_fm_loop_through_source_shim(oracle, tmp_path):
    """bladerf_fm_demod.cpp:279-313 compiled against include/sdr_hip_source.hpp + sdr_hip_shims.hpp
    (examples/fm_chain_example.cpp): SDR_BASE::build() -> start / wait_for_samples / stop over a
    recording, af::fir + x3(dec_seq) + discriminator, audio chain.  Every block restarts from zero
    state like the reference's loop; ArrayFire is absent, so the oracle restates its documented
    semantics (fir_pick: inclusive af::seq end) -- parity unpinned beyond that."""
    exe = os.path.join(ROOT, "examples", "bin", "fm_chain_example")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    blk, nblk = 4096 * 32, 3
    n = blk * nblk + 1000                              # the tail shorter than a block is never delivered
    # one block for start_single, then nblk-1 full blocks through start()/wait_for_samples()
    iq = oracle.synth_fm_i16(n, fs=1.0e6, ftone=1000.0, dev=60000.0, amp=1800.0)
    rng = np.random.default_rng(4)
    iq = (iq + rng.integers(-20, 21, iq.size)).astype(np.int16)
    src = tmp_path / "fm.sc16"
    iq.tofile(src)
    pfx = str(tmp_path / "fm")
    r = subprocess.run([exe, str(src), pfx], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"blocks {nblk - 1} " in r.stdout, r.stdout
    lpf = np.fromfile(pfx + ".lpf.f32", np.float32)
    h_de = np.fromfile(pfx + ".lpf_de.f32", np.float32)
    h_a = np.fromfile(pfx + ".lpf_a.f32", np.float32)
    assert lpf.size == 101 and h_de.size == 64 and h_a.size == 101
    D, Da = 5, 4                                       # fs / channel_bw, fs_d / audio_freq (test_case 1)
    scale = float(np.float32(1.0 / (2.0 * np.pi)))     # float phasor_scale = 1 / ((2*M_PI) / (fs_d / channel_bw)), :224
    # start_single: the a2 conversion, bit exact
    single = np.fromfile(pfx + ".single.f32", np.float32)
    want = oracle.unpack_scale_i16(iq[: 2 * blk]).astype(np.complex64).view(np.float32)
    assert single.tobytes() == want.tobytes()
    x5 = np.fromfile(pfx + ".x5.f32", np.float32).astype(np.float64)
    x5raw = np.fromfile(pfx + ".x5raw.f32", np.float32).astype(np.float64)
    pcm = np.fromfile(pfx + ".pcm.u8", np.uint8)
    n4 = (blk - 1) // D + 1                            # x3(dec_seq) keeps 26215 of 131072 / 5 (floor rule: 26214)
    assert n4 == 26215 and x5.size == (nblk - 1) * (n4 - 1) == x5raw.size
    na = (n4 - 2) // Da + 1
    assert pcm.size == (nblk - 1) * na
    for b in range(nblk - 1):
        cf = oracle.unpack_scale_i16(iq[2 * blk * (b + 1): 2 * blk * (b + 2)]).astype(np.complex64).astype(np.complex128)
        y = oracle.fir_pick(cf, D, lpf)
        want5 = oracle.polar_discriminator(y, scale)
        for got in (x5, x5raw):
            g = got[b * (n4 - 1): (b + 1) * (n4 - 1)]
            check_fm(g, want5, y, scale, f"block {b}")  # every block's start-up transient included
        # audio chain on the kernel's own x5 (float32), AF seq count
        g5 = np.fromfile(pfx + ".x5.f32", np.float32)[b * (n4 - 1): (b + 1) * (n4 - 1)]
        _, want_pcm = oracle.fm_audio_chain(g5, h_de, h_a, Da, af_seq=True)
        got_pcm = pcm[b * na: (b + 1) * na]
        assert got_pcm.shape == want_pcm.shape
        dd = np.abs(got_pcm.astype(np.int32) - want_pcm.astype(np.int32))
        assert dd.max() <= 1 and np.count_nonzero(dd) <= max(4, got_pcm.size // 500)


@pytest.mark.parametrize("pipe", [0, 5])
@pytest.mark.parametrize("mode", ["fm", "env", "iq"])
def test_stream_blocks_equal_one_shot(sdr, oracle, mode, pipe):
    """a12: pushing blocks through sdr_fir_stream gives the one-shot result on the concatenation
    (history carried across blocks), whatever the block sizes -- bit for bit on the int8 matrix
    kernel too (pipe 5), whose outputs are exact integer sums over the whole window."""
    m = {"fm": sdr.MODE_FM, "env": sdr.MODE_ENV, "iq": sdr.MODE_IQ}[mode]
    n = 700_001
    D, T = 10, 101
    iq = oracle.synth_fm_i16(n, fs=2.0e6, ftone=1700.0, dev=25000.0)
    h = oracle.create_fir_filter(T, 0.09, "hann")
    plan = sdr.FirPlan(h, D, mode=m, pipeline=pipe)
    whole = plan.execute_host(iq)
    stream = sdr.FirStream(plan, 131072)
    rng = np.random.default_rng(7)
    pos, parts, in_flight = 0, [], 0
    while pos < n:
        blk = int(min(n - pos, rng.choice([131072, 65536, 1000, 37, 9, 131071])))
        stream.push(iq[2 * pos: 2 * (pos + blk)])
        pos += blk
        in_flight += 1
        if in_flight == 2:
            parts.append(stream.pop())
            in_flight -= 1
    while in_flight:
        parts.append(stream.pop())
        in_flight -= 1
    got = np.concatenate(parts)
    assert got.shape == whole.shape
    if mode == "iq":
        assert np.abs(got - whole).max() <= 1e-6
    else:
        assert np.abs(got - whole).max() <= 2e-6
    # every kernel gives an output one summation order whatever the tiling
    assert np.array_equal(got, whole)
    # the reference's per-block restart differs from this at every block start (the transient)
    with pytest.raises(sdr.SdrError):
        stream.push(iq[: 2 * 10]); stream.push(iq[: 2 * 10]); stream.push(iq[: 2 * 10])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sc16", "qi16", "i8"])
def test_file_replay(sdr, oracle, tmp_path, kind):
    """f2: replaying a recording (temp_get_data, bladerf_apt_demod.cpp:92-148) == the one-shot chain
    on the file cut to a multiple of two blocks (:113-117); a second pass continues the same
    stream (history kept across the wrap-around)."""
    rng = np.random.default_rng(5)
    blk = 40_000
    n = 5 * blk + 1234                      # -> 4 blocks kept
    D, T = 10, 64
    h = oracle.create_fir_filter(T, 0.09, "hann")
    iq = oracle.synth_fm_i16(n, fs=1.0e6, ftone=1200.0, dev=25000.0, amp=1500.0)
    iq = (iq + rng.integers(-20, 21, iq.size)).astype(np.int16)
    if kind == "sc16":
        data, inp, x = iq, sdr.IN_I16_IQ, oracle.unpack_scale_i16(iq)
    elif kind == "qi16":
        data, inp, x = iq.reshape(-1, 2)[:, ::-1].reshape(-1).copy(), sdr.IN_I16_QI, oracle.unpack_scale_i16(iq)
    else:
        data = (iq >> 4).astype(np.int8)
        inp, x = sdr.IN_I8_IQ, oracle.unpack_scale_i8((iq >> 4).astype(np.int8))
    path = tmp_path / f"capture.{kind}"
    data.tofile(path)
    kept = 4 * blk
    plan = sdr.FirPlan(h, D, mode=sdr.MODE_FM, input=inp)
    stream = sdr.FirStream(plan, blk)
    got, pushed = stream.replay_file(path, blk, passes=2)
    assert pushed == 2 * kept
    xx = np.concatenate([x[:kept], x[:kept]])
    y = oracle.fir_decimate(xx, D, h)
    want = oracle.polar_discriminator(y, 1 / TWO_PI)
    check_fm(got, want, y, 1 / TWO_PI, kind)
    # errors: missing file, file shorter than two blocks
    with pytest.raises(sdr.SdrError):
        stream.replay_file(tmp_path / "nope.sc16", blk)
    data[: 2 * blk].tofile(tmp_path / "short.bin")
    with pytest.raises(sdr.SdrError):
        stream.replay_file(tmp_path / "short.bin", blk)


def test_apt_image_from_gpu_chain(sdr, oracle):
    """f3 end to end: synthetic APT transmission (line sync pulses + grey ramp, AM on a 2400 Hz
    subcarrier, FM on the carrier, 624 kS/s int16 IQ) -> GPU two-stage chain -> 4160 Hz envelope
    -> sdr_apt_image (host code in libsdrhip) against the oracle's restatement of
    bladerf_apt_demod.cpp:756-793 on the same envelope blocks: same peaks, identical image."""
    from test_oracle_golden import _apt_envelope
    width, lines = 2080, 14
    a = _apt_envelope(lines, width)
    fs, D1, D2 = 624000.0, 10, 15
    up = np.repeat(a, D1 * D2)
    t = np.arange(up.size) / fs
    m = up * np.cos(2 * np.pi * 2400.0 * t)
    ph = 2 * np.pi * 17000.0 * np.cumsum(m) / fs
    iq = np.empty(2 * up.size, np.int16)
    iq[0::2] = np.rint(1800 * np.cos(ph)).astype(np.int16)
    iq[1::2] = np.rint(1800 * np.sin(ph)).astype(np.int16)
    h1 = oracle.create_fir_filter(200, 20800 / 62400.0 / 5, "hann")
    h2 = oracle.create_fir_filter(195, 2400 / 62400.0, "hann")
    p1 = sdr.FirPlan(h1, D1, mode=sdr.MODE_FM)
    p2 = sdr.FirPlan(h2, D2, mode=sdr.MODE_ENV, input=sdr.IN_F32_REAL, mix_fr=2400 / 62400.0)
    x9 = p2.execute_host(p1.execute_host(iq))
    assert x9.dtype == np.float32 and abs(x9.size - lines * width) < 10
    img = sdr.AptImage(width, 32)
    ref = oracle.AptSync(width, 32)
    total = 0
    for blk in np.array_split(x9, 3):
        added = img.push(blk)
        want_added, _ = ref.push(blk.astype(np.float64))
        assert added == want_added
        total += added
    got, n_lines, pending = img.get()
    assert n_lines == total == ref.lines and total >= 5
    assert pending == ref.n10.value
    assert np.array_equal(got, ref.img)
    # the rows really are image lines: sync pattern at the left edge, ramp brighter to the right
    row = got[-1].astype(int)
    assert row[200:400].mean() < row[1600:1800].mean()
    # argument errors
    with pytest.raises(sdr.SdrError):
        sdr.AptImage(0, 10)


def test_stream_zero_copy_buffers(sdr, oracle):
    """a12: the producer fills the slot's pinned buffer in place (iq_data_queue[write_index] handed
    to bladerf_sync_rx, bladerf_apt_demod.cpp:166) and pushes that pointer: same result as the
    copying push, and a third buffer is refused while both slots are in flight."""
    rng = np.random.default_rng(9)
    blk, nblk, D, T = 50_000, 5, 10, 64
    h = oracle.create_fir_filter(T, 0.09, "hann")
    iq = oracle.synth_fm_i16(blk * nblk, fs=1.0e6, ftone=900.0, dev=20000.0, amp=1700.0)
    iq = (iq + rng.integers(-25, 26, iq.size)).astype(np.int16)
    plan = sdr.FirPlan(h, D, mode=sdr.MODE_ENV)
    want = plan.execute_host(iq)
    stream = sdr.FirStream(plan, blk)
    parts = []
    for b in range(nblk):
        buf = stream.next_buffer(np.int16, 2 * blk)
        buf[:] = iq[2 * b * blk: 2 * (b + 1) * blk]
        stream.push_buffer(buf, blk)
        if b >= 1:
            parts.append(stream.pop())
        if b == 0:
            b2 = stream.next_buffer(np.int16, 2 * blk)        # second slot is free
            assert b2.ctypes.data != buf.ctypes.data
    parts.append(stream.pop())
    got = np.concatenate(parts)
    assert np.array_equal(got, want)
    stream.push(iq[: 2 * blk]); stream.push(iq[: 2 * blk])
    with pytest.raises(sdr.SdrError) as e:
        stream.next_buffer(np.int16, 2 * blk)
    assert e.value.status == sdr.capi.ECAPACITY


def test_gather_picks_and_guards_indices(sdr):
    """a8 support (decimate_vec's pick, bladerf_apt_demod.cpp:374-401): bit-exact picks of 4-, 8- and
    16-byte elements; an index past the end gives a zero element, never a stray read."""
    from sdr_amd import capi
    rng = np.random.default_rng(77)
    for dwords, dt in ((1, np.float32), (2, np.complex64), (4, np.complex128)):
        n_in = 1000
        src = (rng.standard_normal(n_in) + 1j * rng.standard_normal(n_in)).astype(dt) if dwords > 1 \
            else rng.standard_normal(n_in).astype(dt)
        idx = rng.integers(0, n_in, 4097).astype(np.uint64)
        idx[[5, 100, 4096]] = [n_in, n_in + 12345, 2 ** 63]
        d_in, d_idx = capi.DevBuf.from_numpy(src), capi.DevBuf.from_numpy(idx)
        d_out = capi.DevBuf(idx.size * 4 * dwords)
        capi.gather_dev(d_in, dwords, n_in, d_idx, idx.size, d_out)
        capi.sync()
        got = d_out.to_numpy(dt, idx.size)
        ok = idx < n_in
        want = np.zeros(idx.size, dt)
        want[ok] = src[idx[ok]]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_randomised_entry_points(sdr):
    """A fixed slice of tools/fuzz_more.py: FFT -> dB (random frame length, hop, stream length,
    window), the block hand-off (random block sizes, copying and zero-copy pushes; bit-identical to
    the one-shot result for the vector-unit kernels) and batched channels (random counts, strides)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_more
    for seed in range(9_000_000, 9_000_060):
        fuzz_more.fft_case(np.random.default_rng(seed), seed)
        fuzz_more.stream_case(np.random.default_rng(seed), seed)
        fuzz_more.batched_case(np.random.default_rng(seed), seed)
