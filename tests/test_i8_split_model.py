"""CPU model of the arithmetic behind FIR plan pipeline 5 (fir_mfma_i8_kernel, sdr_amd/csrc/fir_mfma_i8.hip):
int16 samples as two signed bytes, taps as four balanced base-256 digits of a 31-bit fixed-point
value, seven of the eight byte products (the weight-2^0 one is below 2^-30 of full scale and left out)
in four exact integer accumulators per output over the whole window, one float32 combination.  The kernel itself is
checked on the GPU (tests/test_gpu_parity.py); this file pins the scheme's claims -- digit ranges,
accumulator bounds, exact zeros under zero pre-history, accuracy against the float64 oracle --
with numpy integers, no GPU."""
import numpy as np
import pytest


def tap_digits(h):
    """q = round(h * 2^F), |q| <= 2^30, as four balanced digits in [-128, 127] (host code of
    sdr_fir_plan_create)"""
    hmax = float(np.abs(h).max())
    e2 = int(np.frexp(hmax)[1]) if hmax > 0 else 0
    F = 30 - e2
    q = np.rint(np.ldexp(h, F)).astype(np.int64)
    digits, r = [], q.copy()
    for _ in range(4):
        d = ((r + 128) & 255) - 128
        digits.append(d)
        r = (r - d) // 256
    assert not r.any()                                   # four digits hold every tap
    return q, np.stack(digits), F


def sample_bytes(s):
    """s = 256*hi + (lo + 128): the raw bytes, bit 7 of the low one flipped"""
    s = s.astype(np.int64)
    hi = s >> 8
    lo = (s & 255) - 128
    assert np.array_equal(256 * hi + lo + 128, s)
    return hi, lo


def model_output(x_i, q_digits, F, m, D):
    """output m of the decimated FIR for one component, as the kernel accumulates it"""
    T = q_digits.shape[1]
    idx = D * m - np.arange(T)                           # sample index per tap (zero pre-history below 0)
    xs = np.where(idx >= 0, x_i[np.clip(idx, 0, None)], 0)
    hi, lo = sample_bytes(xs)
    acc = np.zeros(5, np.int64)
    for d in range(4):
        if d > 0:
            acc[d] += int((q_digits[d] * lo).sum())      # weight 2^(8d); the kernel leaves weight 2^0 out
        acc[d + 1] += int((q_digits[d] * hi).sum())      # weight 2^(8(d+1))
    assert np.abs(acc).max() < 2 ** 24                   # exact in int32 and in the float32 conversion
    bias = np.array([0] + [128 * int(q_digits[d].sum()) for d in range(1, 4)] + [0], np.int64)
    dropped = int((q_digits[0] * (lo + 128)).sum())      # what the missing product would have added
    f = np.float32

    def fma(a, b, c):                                    # float32 fused multiply-add: one rounding
        return f(np.float64(a) * np.float64(b) + np.float64(c))

    def combine(a):                                      # the kernel's float32 sequence (weights 2^8 .. 2^32):
        lo_ = fma(f(a[2]), f(256.0), f(a[1]))            # every accumulator is converted on its own (|acc| < 2^24
        hi_ = fma(f(a[4]), f(256.0), f(a[3]))            # over a 768-sample window: exact), two fma by 256, one by 65536
        return fma(hi_, f(65536.0), lo_)
    exact = sum(int(acc[w] + bias[w]) << (8 * w) for w in range(5)) + dropped
    y32 = f(f(combine(acc) + combine(bias)) * f(np.ldexp(1.0, -(F + 11) + 8)))
    assert abs(dropped) <= 128 * 255 * q_digits.shape[1]                 # < 2^-30 of the 2^45 full-scale sum per tap
    return exact, y32


def test_digits_and_bytes_are_exact():
    rng = np.random.default_rng(1)
    h = rng.standard_normal(511) * np.hanning(511) * 0.01
    q, dg, F = tap_digits(h)
    assert dg.min() >= -128 and dg.max() <= 127
    assert np.array_equal(sum(dg[d].astype(np.int64) << (8 * d) for d in range(4)), q)
    assert np.abs(np.ldexp(q.astype(np.float64), -F) - h).max() <= np.ldexp(0.5, -F)      # 31-bit taps
    s = np.array([-32768, -32767, -257, -256, -255, -129, -128, -1, 0, 1, 127, 128, 255, 256, 32767])
    hi, lo = sample_bytes(s)
    assert hi.min() >= -128 and hi.max() <= 127 and lo.min() >= -128 and lo.max() <= 127


@pytest.mark.parametrize("T,D", [(511, 10), (64, 3)])
def test_model_matches_float64_filter(T, D):
    rng = np.random.default_rng(T)
    n = 4000
    x = rng.integers(-32768, 32768, n)
    h = np.sinc((np.arange(T) - (T - 1) / 2) * 0.7 / D) * np.hanning(T) * 0.7 / D
    q, dg, F = tap_digits(h)
    for m in (0, 1, T // D, T // D + 1, n // D - 1):
        idx = D * m - np.arange(T)
        xs = np.where(idx >= 0, x[np.clip(idx, 0, None)], 0).astype(np.float64)
        want = float((h * xs).sum()) / 2048.0
        exact, y32 = model_output(x, dg, F, m, D)
        # the integer sum IS the fixed-point dot product
        assert exact == int((q * xs.astype(np.int64)).sum())
        # and the one float32 rounding keeps it within 1e-5 of full scale (16 for raw int16 / 2^11)
        assert abs(float(y32) - want) <= 1e-5 * 16 * 0.1


def test_zero_prehistory_gives_exact_zero():
    """taps that meet only zero samples contribute -128*digit through the low byte and +128*digit
    through the bias: the same float32 operations on negated integers cancel exactly"""
    T, D = 101, 5
    h = np.hanning(T) * 0.02
    h[0] = 0.0                                           # Hann: the one tap that meets x[0] at m = 0
    q, dg, F = tap_digits(h)
    x = np.array([12345] + [0] * 600)
    exact, y32 = model_output(x, dg, F, 0, D)
    assert exact == 0 and float(y32) == 0.0
    x = np.zeros(600, np.int64)
    for m in (0, 7, 50):
        exact, y32 = model_output(x, dg, F, m, D)
        assert exact == 0 and float(y32) == 0.0
