"""The parity bars of include/sdr_hip.h as assertions, shared by the GPU test modules.

  IQ / envelope / real : |err| <= 1e-5 * FS, FS = max(1, max|expected|)   (Q11 full scale = 1.0); no exemption
  discriminator        : see check_fm
"""
import numpy as np

TWO_PI = 2.0 * np.pi


def check_linear(got, want, what=""):
    want = np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if want.size == 0:
        return
    fs = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got.astype(want.dtype) - want).max())
    assert err <= 1e-5 * fs, (what, err, fs)


def check_fm(got, want, y, scale, what="", good_rel=1e-3):
    """y: the oracle's complex decimated stream (len(want)+1).  The bars of include/sdr_hip.h:
    outputs whose y[m] or y[m+1] is exactly zero in the reference (0 / +-scale*pi by the signs of its
    zeros) must match exactly; outputs with both |y| above 1e-3 of the block rms are within 1e-5 of full
    scale; in between the linear bar e = 1e-5 * max(1, max|y|) on y turns into the angle bound
    scale * (asin(e/|y[m]|) + asin(e/|y[m+1]|)) (a vector moved by e turns by at most asin(e/|y|); a
    vector shorter than e can point anywhere).  good_rel=None (the randomised sweeps, whose random taps
    and noise inputs have no "typical" magnitude) applies the angle bound to every non-exact output."""
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if want.size == 0:
        return
    fs = abs(scale) * np.pi
    turn = abs(scale) * TWO_PI
    d = np.abs(got.astype(np.float64) - want)
    d = np.minimum(d, np.abs(turn - d))
    mag = np.abs(y)
    rms = float(np.sqrt(np.mean(mag * mag))) if mag.size else 0.0
    pair_min = np.minimum(mag[:-1], mag[1:])
    exact = pair_min == 0.0
    if exact.any():
        # 0 or +-scale*pi, decided by the signs of the zeros: one float32 ulp of slack for the product
        # scale*pi itself (float32(scale)*float32(pi) vs float32(scale*pi); exact for scale = 1/(2 pi))
        ge, we = got[exact].astype(np.float64), want[exact].astype(np.float32).astype(np.float64)
        nz = we != 0.0
        assert np.array_equal(np.signbit(ge[nz]), np.signbit(we[nz])) and (ge[~nz] == 0.0).all(), what
        assert np.abs(ge - we).max() <= 2.0 ** -23 * fs, (what, float(np.abs(ge - we).max()))
    good = pair_min > good_rel * rms if good_rel is not None else np.zeros(pair_min.shape, bool)
    if good.any():
        assert float(d[good].max()) <= 1e-5 * fs, (what, float(d[good].max()))
    weak = ~good & ~exact
    if weak.any():
        e = 1e-5 * max(1.0, float(mag.max()))
        with np.errstate(divide="ignore", invalid="ignore"):
            turn_by = lambda m: np.where(m > e, np.arcsin(np.minimum(e / np.maximum(m, 1e-300), 1.0)), np.pi)
            bound = abs(scale) * (turn_by(mag[:-1]) + turn_by(mag[1:])) + 1e-5 * fs
        over = weak & (d > bound)
        assert not over.any(), (what, int(np.flatnonzero(over)[0]), float(d[over].max()), float(bound[over].min()))


def apt_stage2(oracle, x6, fr, D2, h2):
    """The reference's second APT stage (bladerf_apt_demod.cpp:744-754: frequency_shift ->
    polyphase_decimate -> abs) in the oracle's float64, on a GIVEN first-stage output.  The GPU chain's x9
    is held to this on the GPU chain's own x6: x6 is checked on its own with check_fm, whose bars allow
    the start-up transient's ill-conditioned angles to differ, and x9 inherits those legitimately."""
    xs = oracle.frequency_shift(np.asarray(x6, np.float64), fr)
    return oracle.envelope(oracle.fir_decimate(xs, D2, h2))
