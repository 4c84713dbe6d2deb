"""CPU models of the two mixer formulations the float-tile kernel uses (sdr_amd/csrc/fir_tiles.hip,
fir_stream_kernel), checked against the reference's order of operations -- frequency_shift then
polyphase_decimate (bladerf_apt_demod.cpp:744-748) -- as the oracle restates it.  No GPU."""
import numpy as np


def _phasor32(fr, idx):
    """what the host builds: phase reduced in double, cos/sin rounded to float32"""
    ph = fr * np.asarray(idx, np.float64)
    f = ph - np.rint(ph)
    return (np.cos(2 * np.pi * f).astype(np.float32) + 1j * np.sin(2 * np.pi * f).astype(np.float32)).astype(np.complex64)


def test_real_input_mixer_folds_into_complex_taps(oracle):
    """y[m] = e^{j w m D} * sum_i (h[i] e^{-j w i}) x[mD - i] for real x: the kernel keeps x real in LDS,
    uses complex taps and rotates each output once."""
    rng = np.random.default_rng(3)
    T, D, n, fr = 101, 15, 4000, 2400 / 62400.0
    h = oracle.create_fir_filter(T, 0.05, "hann")
    x = rng.standard_normal(n)
    want = oracle.fir_decimate(oracle.frequency_shift(x, fr), D, h)
    taps = h * np.exp(-2j * np.pi * fr * np.arange(T))
    xp = np.concatenate([np.zeros(T - 1), x])
    m = np.arange(n // D)
    got = np.array([np.dot(taps, xp[mm * D + T - 1 - np.arange(T)]) for mm in m]) * np.exp(2j * np.pi * fr * m * D)
    assert np.abs(got - want).max() <= 1e-12
    # the envelope needs no rotation at all
    assert np.abs(np.abs(got) - oracle.envelope(want)).max() <= 1e-12


def test_table_phasor_factorisation_in_float32():
    """e^{j w (W0 + i)} = B * HI[i >> 6] * LO[i & 63] with every factor and product in float32: within
    3e-7 of the exact phasor for tile offsets up to a whole tile and stream positions up to 2^40."""
    fr = 0.0371
    i = np.arange(0, 20_000)
    hi, lo = _phasor32(fr, 64 * np.arange(i.max() // 64 + 1)), _phasor32(fr, np.arange(64))
    for w0 in (-255, 0, 123_456_789, 2 ** 40 + 77):
        b = _phasor32(fr, [w0])[0]
        bh = (b * hi[i >> 6]).astype(np.complex64)
        p = (bh * lo[i & 63]).astype(np.complex64)
        ph = fr * (np.float64(w0) + i)
        exact = np.exp(2j * np.pi * (ph - np.rint(ph)))
        assert np.abs(p - exact).max() <= 3e-7 + 2 * np.pi * abs(fr) * abs(w0) * 2.3e-16
