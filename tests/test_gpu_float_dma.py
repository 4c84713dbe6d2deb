"""Pipeline 7 (float tiles filled by LDS-DMA, one tile per workgroup; complex or real float input without mixer:
the af::fir call shape of the FM receiver, bladerf_fm_demod.cpp:288-294) against the oracle, every geometry."""
import numpy as np
import pytest

from parity_checks import check_fm, check_linear

TWO_PI = 2.0 * np.pi


@pytest.fixture(scope="module")
def sdr():
    import sdr_amd
    sdr_amd.capi.lib()
    sdr_amd.init(0)
    return sdr_amd


@pytest.mark.gpu
@pytest.mark.parametrize("T,D,n", [(64, 10, 200_003), (101, 5, 131_072), (1, 1, 5000), (7, 3, 10), (200, 10, 99_999), (511, 10, 60_001),
                                   (33, 16, 77_777), (64, 10, 0), (64, 10, 5), (12, 2, 40_000), (300, 7, 50_001)])
@pytest.mark.parametrize("kind", ["cf32", "real"])
def test_float_dma_tiles(sdr, oracle, T, D, n, kind):
    rng = np.random.default_rng(T * 131 + D + n)
    h = oracle.create_fir_filter(T, min(0.8 / D, 0.9), "hann") if T > 1 else np.array([0.7])
    if kind == "cf32":
        raw = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.3).astype(np.complex64)
        inp = sdr.IN_CF32
    else:
        raw = (rng.standard_normal(n) * 0.3).astype(np.float32)
        inp = sdr.IN_F32_REAL
    x = raw.astype(np.complex128)
    y = oracle.fir_decimate(x, D, h)
    for R, NT in ((0, 0), (1, 64), (2, 128), (4, 256), (2, 256), (4, 64)):
        try:
            plan = sdr.FirPlan(h, D, mode=sdr.MODE_IQ, input=inp, pipeline=7, outputs_per_lane=R, threads=NT)
        except sdr.SdrError as e:
            assert "no tile geometry" in str(e), str(e)
            continue
        assert plan.info()["kernel"] == "float_dma_tile"
        tag = f"T={T} D={D} n={n} {kind} R={R} NT={NT}"
        got = plan.execute_host(raw)
        check_linear(got, y, tag)
        # bit-identical to the float-tile kernel?  (same float32 taps, same ascending summation order)
        ref = sdr.FirPlan(h, D, mode=sdr.MODE_IQ, input=inp, pipeline=2).execute_host(raw)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)) or np.abs(got - ref).max() < 1e-6, tag
        check_linear(sdr.FirPlan(h, D, mode=sdr.MODE_ENV, input=inp, pipeline=7, outputs_per_lane=R, threads=NT).execute_host(raw),
                     oracle.envelope(y), tag + " env")
        if kind == "cf32":
            gotf = sdr.FirPlan(h, D, mode=sdr.MODE_FM, input=inp, pipeline=7, outputs_per_lane=R, threads=NT).execute_host(raw)
            assert gotf.shape == (max(y.size - 1, 0),)
            if y.size > 1:
                check_fm(gotf, oracle.polar_discriminator(y, 1 / TWO_PI), y, 1 / TWO_PI, tag + " fm", good_rel=None)
        else:
            gotr = sdr.FirPlan(h, D, mode=sdr.MODE_REAL, input=inp, pipeline=7, outputs_per_lane=R, threads=NT).execute_host(raw)
            check_linear(gotr, y.real, tag + " real")
    # a mixer goes along (complex taps + one phasor per output): frequency_shift in front of the filter
    for fr in (0.0371, -0.123):
        ym = oracle.fir_decimate(oracle.frequency_shift(raw.astype(np.complex128 if kind == "cf32" else np.float64), fr), D, h) if n else y
        modes = [(sdr.MODE_IQ, ym), (sdr.MODE_ENV, np.abs(ym))] + ([(sdr.MODE_REAL, ym.real)] if kind == "real" else [])
        for mode, want in modes:
            try:
                pm = sdr.FirPlan(h, D, mode=mode, input=inp, pipeline=7, mix_fr=fr)
            except sdr.SdrError as e:
                assert "no tile geometry" in str(e), str(e)
                continue
            check_linear(pm.execute_host(raw), want, f"T={T} D={D} n={n} {kind} mixer {fr} mode {mode}")
        if kind == "cf32" and ym.size > 1:
            try:
                pm = sdr.FirPlan(h, D, mode=sdr.MODE_FM, input=inp, pipeline=7, mix_fr=fr)
            except sdr.SdrError:
                continue
            check_fm(pm.execute_host(raw), oracle.polar_discriminator(ym, 1 / TWO_PI), ym, 1 / TWO_PI, f"T={T} D={D} n={n} cf32 mixer {fr} fm", good_rel=None)
    with pytest.raises(sdr.SdrError):
        sdr.FirPlan(h, D, mode=sdr.MODE_IQ, input=inp, pipeline=7, mix_fr_channels=[0.01, 0.02])
    with pytest.raises(sdr.SdrError):
        sdr.FirPlan(h, D, mode=sdr.MODE_IQ, input=sdr.IN_I16_IQ, pipeline=7)
