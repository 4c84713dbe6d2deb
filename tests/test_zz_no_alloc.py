"""GPU test of the allocation-free entry point sdr_fft_execute_i16_noalloc_dev (task 7): its output
must equal sdr_fft_execute_i16_dev's bit for bit, on two calls in a row, the second with a smaller
input, alone in a fresh process and after the other tests."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sdr():
    import sdr_amd
    sdr_amd.capi.lib()
    sdr_amd.init(0)
    return sdr_amd


def both(sdr, plan, d_iq, n):
    frames = plan.frame_count(n)
    d_a, d_b = sdr.DevBuf((frames + 1) * plan.nfft * 4), sdr.DevBuf((frames + 1) * plan.nfft * 4)
    sent = np.full(plan.nfft, 4242.5, np.float32)
    for d in (d_a, d_b):
        sdr.capi.check(sdr.capi.lib().sdr_dev_upload(d.at(frames * plan.nfft * 4), sent.ctypes.data_as(sdr.capi.C.c_void_p),
                                                     sdr.capi.C.c_size_t(sent.nbytes)))
    assert plan.execute_noalloc_dev(d_iq, n, d_a, frames) == frames
    sdr.sync()
    assert plan.execute_dev(d_iq, n, d_b, frames) == frames
    sdr.sync()
    a = d_a.to_numpy(np.float32, (frames + 1) * plan.nfft)
    b = d_b.to_numpy(np.float32, (frames + 1) * plan.nfft)
    assert (a[frames * plan.nfft:] == 4242.5).all() and (b[frames * plan.nfft:] == 4242.5).all()
    return a[: frames * plan.nfft], b[: frames * plan.nfft], frames


@pytest.mark.parametrize("nfft,hop", [(8192, 4096), (131072, 131072), (32768, 10_000), (4096, 2048)])
def test_noalloc_equals_allocating_entry_point(sdr, oracle, nfft, hop):
    rng = np.random.default_rng(nfft + hop)
    n1, n2 = 5 * nfft + 3 * hop + 17, 2 * nfft + 5          # second call: smaller input
    iq = oracle.synth_fm_i16(n1, fs=61.44e6, ftone=250e3, dev=2.0e6, amp=1200.0)
    iq = (iq + rng.integers(-64, 65, iq.size)).astype(np.int16)
    w = sdr.window("blackman", nfft)
    plan = sdr.FftPlan(nfft, hop, w)
    d_iq = sdr.DevBuf.from_numpy(iq)
    a1, b1, f1 = both(sdr, plan, d_iq, n1)
    assert f1 >= 3 and a1.tobytes() == b1.tobytes()
    a2, b2, f2 = both(sdr, plan, d_iq, n2)
    assert 1 <= f2 < f1 and a2.tobytes() == b2.tobytes()
    # the smaller run is a prefix of the larger one (frames are independent): stale scratch of the
    # first call cannot have leaked into the second
    assert a2.tobytes() == a1[: a2.size].tobytes()
    # and the result is the oracle's
    want_db, want_lin = oracle.fft_power_db_i16(iq[: 2 * n2], nfft, hop, w, linear=True)
    lin = 10.0 ** (a2.reshape(f2, nfft).astype(np.float64) / 10.0)
    assert np.abs(lin - want_lin).max() <= 1e-5 * want_lin.max()


def test_noalloc_makes_no_allocation_and_runs_under_capture(sdr, oracle):
    """nothing detected an allocation on the no-alloc path.  Two detectors: the library's own
    allocation counter stands still across the call (and moves for the allocating entry point's first
    long-frame call on a fresh plan), and the call is captured into a HIP graph -- hipMalloc is not
    permitted on a capturing stream -- whose replay gives the same bits."""
    import ctypes as C
    L = sdr.capi.lib()
    L.sdr_alloc_count.restype = C.c_ulonglong
    nfft, hop = 16384, 16384
    n = 3 * nfft + 11
    iq = oracle.synth_fm_i16(n, fs=61.44e6, ftone=250e3, dev=2.0e6, amp=1200.0)
    d_iq = sdr.DevBuf.from_numpy(iq)
    plan = sdr.FftPlan(nfft, hop, sdr.window("blackman", nfft))
    frames = plan.frame_count(n)
    d_a, d_b, d_c = (sdr.DevBuf(frames * nfft * 4) for _ in range(3))
    a0 = L.sdr_alloc_count()
    assert plan.execute_noalloc_dev(d_iq, n, d_a, frames) == frames
    sdr.sync()
    assert L.sdr_alloc_count() == a0
    assert plan.execute_dev(d_iq, n, d_b, frames) == frames        # grows the plan's scratch: counted
    sdr.sync()
    assert L.sdr_alloc_count() > a0
    a = d_a.to_numpy(np.float32, frames * nfft)
    assert a.tobytes() == d_b.to_numpy(np.float32, frames * nfft).tobytes()
    # under stream capture
    hip = C.CDLL("libamdhip64.so")
    st, graph, gexec = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert hip.hipStreamCreate(C.byref(st)) == 0
    try:
        assert hip.hipStreamBeginCapture(st, C.c_int(0)) == 0      # hipStreamCaptureModeGlobal
        try:
            assert plan.execute_noalloc_dev(d_iq, n, d_c, frames, stream=st.value) == frames
        finally:
            rc = hip.hipStreamEndCapture(st, C.byref(graph))
        assert rc == 0 and graph.value
        assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)) == 0
        assert hip.hipGraphLaunch(gexec, st) == 0
        assert hip.hipStreamSynchronize(st) == 0
        assert d_c.to_numpy(np.float32, frames * nfft).tobytes() == a.tobytes()
    finally:
        if gexec.value:
            hip.hipGraphExecDestroy(gexec)
        if graph.value:
            hip.hipGraphDestroy(graph)
        hip.hipStreamDestroy(st)


def test_noalloc_limits_and_errors(sdr):
    big = sdr.FftPlan(262144, 262144)
    d_iq, d_out = sdr.DevBuf(262144 * 4), sdr.DevBuf(262144 * 4)
    with pytest.raises(sdr.SdrError) as e:
        big.execute_noalloc_dev(d_iq, 262144, d_out, 1)
    assert e.value.status == sdr.capi.EINVAL and "static scratch limit" in str(e.value)
    assert big.execute_dev(d_iq, 262144, d_out, 1) == 1        # the allocating entry point still takes it
    sdr.sync()
    p = sdr.FftPlan(8192, 8192)
    with pytest.raises(sdr.SdrError) as e:
        p.execute_noalloc_dev(d_iq, 3 * 8192, d_out, 2)        # 3 frames, capacity 2
    assert e.value.status == sdr.capi.ECAPACITY
    assert p.execute_noalloc_dev(d_iq, 100, d_out, 0) == 0     # shorter than a frame: nothing to do
