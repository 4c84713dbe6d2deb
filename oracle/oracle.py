"""ctypes/numpy front-end of the CPU oracle (oracle/liboracle.so) and, when it has been built,
of the reference's own functions (oracle/_ref/libsdrref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under sdr_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libsdrref.so")

WINDOWS = {"rectangular": 0, "hann": 1, "hamming": 2, "blackman": 3, "nuttall": 4,
           "blackman_harris": 5}
MODE_IQ, MODE_FM, MODE_ENV = 0, 1, 2

_dp = C.POINTER(C.c_double)
_sp = C.POINTER(C.c_int16)


def build(force=False):
    """Compile liboracle.so (and _ref/libsdrref.so when /root/reference exists)."""
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "sdr_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and (force or not os.path.exists(_REF)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _d(a):
    return a.ctypes.data_as(_dp)


def _s(a):
    return a.ctypes.data_as(_sp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _c2d(a):
    """complex128 array -> interleaved float64 view"""
    a = np.ascontiguousarray(a, dtype=np.complex128)
    return a.view(np.float64)


def _i16(iq):
    iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1)
    assert iq.size % 2 == 0
    return iq


class _Lib:
    def __init__(self, path):
        self.lib = C.CDLL(path)


_oracle = None
_ref = None


def lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        L.orc_polyphase_decimate.restype = C.c_int
        L.orc_fir_decimate.restype = C.c_int
        L.orc_fir_pick.restype = C.c_size_t
        L.orc_decimate_vec.restype = C.c_size_t
        L.orc_filter_vec.restype = C.c_int
        L.orc_window.restype = C.c_int
        L.orc_create_fir_filter.restype = C.c_int
        L.orc_fft_power_db_i16.restype = C.c_size_t
        L.orc_chain_i16.restype = C.c_size_t
        L.orc_apt_chain_i16.restype = C.c_size_t
        L.orc_fm_audio_chain.restype = C.c_size_t
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(_REF)


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(_REF)
        L.ref_polyphase_decimate.restype = C.c_int
        L.ref_decimate_vec_real.restype = C.c_size_t
        L.ref_chain_i16.restype = C.c_size_t
        L.ref_apt_chain_i16.restype = C.c_size_t
        L.ref_time_chain_i16.restype = C.c_double
        _ref = L
    return _ref


# ------------------------------------------------------------------ oracle (restatement) -----
def unpack_scale_i16(iq):
    iq = _i16(iq)
    n = iq.size // 2
    out = np.empty(2 * n, np.float64)
    lib().orc_unpack_scale_i16(_s(iq), C.c_size_t(n), _d(out))
    return out.view(np.complex128)


def unpack_scale_i16_qi(qi):
    qi = _i16(qi)
    n = qi.size // 2
    out = np.empty(2 * n, np.float64)
    lib().orc_unpack_scale_i16_qi(_s(qi), C.c_size_t(n), _d(out))
    return out.view(np.complex128)


def unpack_scale_i8(iq):
    iq = np.ascontiguousarray(iq, dtype=np.int8).reshape(-1)
    n = iq.size // 2
    out = np.empty(2 * n, np.float64)
    lib().orc_unpack_scale_i8(iq.ctypes.data_as(C.c_void_p), C.c_size_t(n), _d(out))
    return out.view(np.complex128)


def polyphase_decimate(x, D, h):
    """Reference semantics incl. the T % D restriction (raises ValueError like
    std::invalid_argument)."""
    xv = _c2d(x)
    h = _f64(h)
    n = xv.size // 2
    y = np.zeros(2 * max(n // max(D, 1), 1), np.float64)
    rc = lib().orc_polyphase_decimate(_d(xv), C.c_size_t(n), C.c_int(D), _d(h), C.c_int(h.size),
                                      _d(y))
    if rc != 0:
        raise ValueError("polyphase_decimate: invalid argument")
    return y.view(np.complex128)[: n // D]


def fir_decimate(x, D, h, pre=0):
    """x holds `pre` history samples followed by the block."""
    xv = _c2d(x)
    h = _f64(h)
    n = xv.size // 2 - pre
    y = np.zeros(2 * max(n // D, 1), np.float64)
    base = C.cast(C.c_void_p(xv.ctypes.data + 16 * pre), _dp)
    rc = lib().orc_fir_decimate(base, C.c_size_t(n), C.c_size_t(pre), C.c_int(D), _d(h),
                                C.c_int(h.size), _d(y))
    if rc != 0:
        raise ValueError("fir_decimate: invalid argument")
    return y.view(np.complex128)[: n // D]


def fir_pick(x, D, h):
    xv = _c2d(x)
    h = _f64(h)
    n = xv.size // 2
    y = np.zeros(2 * ((n - 1) // D + 1), np.float64)
    m = lib().orc_fir_pick(_d(xv), C.c_size_t(n), C.c_int(D), _d(h), C.c_int(h.size), _d(y))
    return y.view(np.complex128)[:m]


def polar_discriminator(v, scale):
    vv = _c2d(v)
    n = vv.size // 2
    d = np.zeros(max(n - 1, 0), np.float64)
    lib().orc_polar_discriminator(_d(vv), C.c_size_t(n), C.c_double(scale), _d(d))
    return d


def envelope(v):
    vv = _c2d(v)
    n = vv.size // 2
    e = np.zeros(n, np.float64)
    lib().orc_envelope(_d(vv), C.c_size_t(n), _d(e))
    return e


def magnitude_i16(iq):
    iq = _i16(iq)
    n = iq.size // 2
    m = np.zeros(n, np.float64)
    lib().orc_magnitude_i16(_s(iq), C.c_size_t(n), _d(m))
    return m


def frequency_shift(data, fr):
    data = np.asarray(data)
    n = data.size
    out = np.zeros(2 * n, np.float64)
    if np.iscomplexobj(data):
        lib().orc_frequency_shift_cplx(_d(_c2d(data)), C.c_size_t(n), C.c_double(fr), _d(out))
    else:
        lib().orc_frequency_shift_real(_d(_f64(data)), C.c_size_t(n), C.c_double(fr), _d(out))
    return out.view(np.complex128)


def create_freq_rotation(N, fr):
    out = np.zeros(2 * N, np.float64)
    lib().orc_create_freq_rotation(C.c_uint64(N), C.c_double(fr), _d(out))
    return out.view(np.complex128)


def decimate_vec(v, rate):
    v = _f64(v)
    n = v.size
    num = int(np.ceil(n / rate))
    res = np.zeros(num, np.float64)
    idx = np.zeros(num, np.int64)
    oob = C.c_size_t(0)
    m = lib().orc_decimate_vec(_d(v), C.c_size_t(n), C.c_int(1), C.c_double(rate), _d(res),
                               idx.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(oob))
    assert m == num
    return res, idx, oob.value


def filter_vec(v, h):
    v = _f64(v)
    h = _f64(h)
    res = np.zeros(v.size, np.float64)
    rc = lib().orc_filter_vec(_d(v), C.c_size_t(v.size), _d(h), C.c_int(h.size), _d(res))
    if rc != 0:
        raise ValueError("filter_vec: needs n >= taps")
    return res


def window(name, N):
    w = np.zeros(N, np.float64)
    rc = lib().orc_window(C.c_int(WINDOWS[name]), C.c_size_t(N), _d(w))
    if rc != 0:
        raise ValueError("window")
    return w


def create_fir_filter(N, fc, window_name="hann"):
    h = np.zeros(N, np.float64)
    rc = lib().orc_create_fir_filter(C.c_size_t(N), C.c_double(fc), C.c_int(WINDOWS[window_name]),
                                     _d(h))
    if rc != 0:
        raise ValueError("create_fir_filter")
    return h


def fft_power_db_i16(iq, nfft, hop, window=None, linear=False):
    iq = _i16(iq)
    n = iq.size // 2
    if n < nfft:
        return np.zeros((0, nfft))
    frames = (n - nfft) // hop + 1
    out = np.zeros((frames, nfft), np.float64)
    lin = np.zeros((frames, nfft), np.float64) if linear else None
    w = _f64(window) if window is not None else None
    f = lib().orc_fft_power_db_i16(_s(iq), C.c_size_t(n), C.c_size_t(nfft), C.c_size_t(hop),
                                   _d(w) if w is not None else None, _d(out),
                                   _d(lin) if lin is not None else None)
    assert f == frames
    return (out, lin) if linear else out


def chain_i16(iq, D, h, mode, scale=1.0 / (2.0 * np.pi)):
    iq = _i16(iq)
    h = _f64(h)
    n = iq.size // 2
    ny = n // D
    out = np.zeros(2 * max(ny, 1), np.float64)
    m = lib().orc_chain_i16(_s(iq), C.c_size_t(n), C.c_int(D), _d(h), C.c_int(h.size),
                            C.c_int(mode), C.c_double(scale), _d(out))
    return out.view(np.complex128)[:m].copy() if mode == MODE_IQ else out[:m].copy()


def apt_chain_i16(iq, D1, h1, fr, D2, h2, disc_scale=1.0 / (2.0 * np.pi)):
    iq = _i16(iq)
    h1, h2 = _f64(h1), _f64(h2)
    n = iq.size // 2
    n6 = max(n // D1 - 1, 0)
    x6 = np.zeros(max(n6, 1), np.float64)
    x9 = np.zeros(max(n6 // D2, 1), np.float64)
    m = lib().orc_apt_chain_i16(_s(iq), C.c_size_t(n), C.c_int(D1), _d(h1), C.c_int(h1.size),
                                C.c_double(disc_scale), C.c_double(fr), C.c_int(D2), _d(h2),
                                C.c_int(h2.size), _d(x9), _d(x6))
    return x9[:m].copy(), x6[:n6].copy()


def synth_fm_i16(n, n0=0, fs=1.0e6, ftone=1000.0, dev=5000.0, amp=2000.0):
    iq = np.zeros(2 * n, np.int16)
    lib().orc_synth_fm_i16(_s(iq), C.c_size_t(n), C.c_uint64(n0), C.c_double(fs),
                           C.c_double(ftone), C.c_double(dev), C.c_double(amp))
    return iq


def pad_taps(h, D):
    """zero-pad to a multiple of D (what SURVEY 0.6 prescribes to drive the reference)."""
    h = _f64(h)
    T = ((h.size + D - 1) // D) * D
    return np.concatenate([h, np.zeros(T - h.size)])


# ------------------------------------------------------------------ reference (oracle/_ref) --
def ref_polyphase_decimate(x, D, h):
    xv = _c2d(x)
    h = _f64(h)
    n = xv.size // 2
    y = np.zeros(2 * max(n // max(D, 1), 1), np.float64)
    rc = ref().ref_polyphase_decimate(_d(xv), C.c_size_t(n), C.c_int(D), _d(h), C.c_int(h.size),
                                      _d(y))
    if rc != 0:
        raise ValueError("reference threw std::invalid_argument")
    return y.view(np.complex128)[: n // D]


def ref_polar_discriminator(v, scale):
    vv = _c2d(v)
    n = vv.size // 2
    d = np.zeros(n - 1, np.float64)
    ref().ref_polar_discriminator(_d(vv), C.c_size_t(n), C.c_double(scale), _d(d))
    return d


def ref_frequency_shift(data, fr):
    data = np.asarray(data)
    n = data.size
    out = np.zeros(2 * n, np.float64)
    if np.iscomplexobj(data):
        ref().ref_frequency_shift_cplx(_d(_c2d(data)), C.c_size_t(n), C.c_double(fr), _d(out))
    else:
        ref().ref_frequency_shift_real(_d(_f64(data)), C.c_size_t(n), C.c_double(fr), _d(out))
    return out.view(np.complex128)


def ref_create_freq_rotation(N, fr):
    out = np.zeros(2 * N, np.float64)
    ref().ref_create_freq_rotation(C.c_uint64(N), C.c_double(fr), _d(out))
    return out.view(np.complex128)


def ref_decimate_vec(v, rate):
    v = _f64(v)
    num = int(np.ceil(v.size / rate))
    res = np.zeros(num, np.float64)
    m = ref().ref_decimate_vec_real(_d(v), C.c_size_t(v.size), C.c_double(rate), _d(res),
                                    C.c_size_t(num))
    assert m == num
    return res


def ref_filter_vec(v, h):
    v, h = _f64(v), _f64(h)
    res = np.zeros(v.size, np.float64)
    ref().ref_filter_vec_real(_d(v), C.c_size_t(v.size), _d(h), C.c_int(h.size), _d(res))
    return res


def ref_chain_i16(iq, D, h, mode, scale=1.0 / (2.0 * np.pi)):
    iq = _i16(iq)
    h = pad_taps(h, D)
    n = iq.size // 2
    out = np.zeros(2 * max(n // D, 1), np.float64)
    m = ref().ref_chain_i16(_s(iq), C.c_size_t(n), C.c_int(D), _d(h), C.c_int(h.size),
                            C.c_int(mode), C.c_double(scale), _d(out))
    return out.view(np.complex128)[:m].copy() if mode == MODE_IQ else out[:m].copy()


def ref_apt_chain_i16(iq, D1, h1, fr, D2, h2, disc_scale=1.0 / (2.0 * np.pi)):
    iq = _i16(iq)
    h1, h2 = pad_taps(h1, D1), pad_taps(h2, D2)
    n = iq.size // 2
    n6 = max(n // D1 - 1, 0)
    x6 = np.zeros(max(n6, 1), np.float64)
    x9 = np.zeros(max(n6 // D2, 1), np.float64)
    m = ref().ref_apt_chain_i16(_s(iq), C.c_size_t(n), C.c_int(D1), _d(h1), C.c_int(h1.size),
                                C.c_double(disc_scale), C.c_double(fr), C.c_int(D2), _d(h2),
                                C.c_int(h2.size), _d(x9), _d(x6))
    return x9[:m].copy(), x6[:n6].copy()


def ref_time_chain_i16(iq, D, h, mode, scale, reps):
    """seconds for `reps` single-thread passes of the reference FM chain (cpu_baseline)."""
    iq = _i16(iq)
    h = pad_taps(h, D)
    sink = C.c_double(0)
    return ref().ref_time_chain_i16(_s(iq), C.c_size_t(iq.size // 2), C.c_int(D), _d(h),
                                    C.c_int(h.size), C.c_int(mode), C.c_double(scale),
                                    C.c_int(reps), C.byref(sink))


def fm_audio_chain(x5, h_de, h_a, D, af_seq=False):
    """bladerf_fm_demod.cpp:301-313 -> (normalised audio float64, pcm uint8); af_seq: keep the samples
    x6(seq_audio) picks (every index 0, D, ... <= n-1) instead of floor(n/D)"""
    x = _f64(x5)
    h_de = _f64(h_de)
    h_a = _f64(h_a)
    m = x.size // D + 1
    audio = np.zeros(max(m, 1), np.float64)
    pcm = np.zeros(max(m, 1), np.uint8)
    lib().orc_fm_audio_chain_rule.restype = C.c_size_t
    k = lib().orc_fm_audio_chain_rule(_d(x), C.c_size_t(x.size), _d(h_de), C.c_int(h_de.size), _d(h_a),
                                      C.c_int(h_a.size), C.c_int(D), C.c_int(1 if af_seq else 0), _d(audio),
                                      pcm.ctypes.data_as(C.c_void_p))
    return audio[:k], pcm[:k]


class AptSync:
    """bladerf_apt_demod.cpp:756-793 block by block (oracle state: pending samples + image)."""

    def __init__(self, width=2080, rows=700, capacity=1 << 22):
        self.width, self.rows = width, rows
        self.x10 = np.zeros(capacity, np.float64)
        self.n10 = C.c_size_t(0)
        self.img = np.zeros((rows, width), np.uint8)
        self.lines = 0

    def push(self, x9):
        x = _f64(x9)
        peaks = np.zeros(4096, np.int64)
        npk = C.c_int()
        r = lib().orc_apt_sync_block(_d(self.x10), C.byref(self.n10), C.c_size_t(self.x10.size), _d(x),
                                     C.c_size_t(x.size), self.img.ctypes.data_as(C.c_void_p),
                                     C.c_int(self.rows), C.c_int(self.width),
                                     peaks.ctypes.data_as(C.c_void_p), C.c_int(peaks.size), C.byref(npk))
        if r < 0:
            raise ValueError("AptSync: pending buffer overflow")
        self.lines += r
        return r, peaks[: npk.value].copy()
