"""ctypes binding of libsdrhip.so (include/sdr_hip.h).

This is the Python host side used by the tests and bench.py.  It is plumbing only: every
compute call goes through the C ABI into the HIP kernels.  There is no CPU fallback -- if the
shared library is missing, or no gfx950 device is present, the calls raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsdrhip.so")
# A/B measurements of experiment builds (tools/ab_libs.py) bind another build -- only from inside sdr_amd/
# (a stray environment variable cannot make a foreign library stand in for the product), and loudly
if os.environ.get("SDRHIP_LIB"):
    _alt = os.path.abspath(os.environ["SDRHIP_LIB"])
    if os.path.dirname(_alt).startswith(_HERE) and os.path.basename(_alt).startswith("libsdrhip"):
        import sys as _sys
        print(f"sdr_amd: SDRHIP_LIB -> {_alt} (experiment build, not the product library)", file=_sys.stderr)
        LIB_PATH = _alt
    else:
        raise ImportError(f"SDRHIP_LIB={_alt}: only builds under {_HERE} named libsdrhip*.so can be bound")
SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("fir_mfma_i8.hip", "fir_tiles.hip", "fir_plan.hip", "fft.hip", "fft_wave.hip", "stream.hip", "apt.hip", "lib.hip")]
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

OK, EINVAL, ENOMEM, EHIP, EALIGN, ECAPACITY, ENODEV, ERANGE = range(8)
MODE_IQ, MODE_FM, MODE_ENV, MODE_REAL = 0, 1, 2, 3
COUNT_FLOOR, COUNT_AF_SEQ = 0, 1
IN_I16_IQ, IN_CF32, IN_F32_REAL, IN_I16_QI, IN_I8_IQ = 0, 1, 2, 3, 4
WINDOWS = {"rectangular": 0, "hann": 1, "hamming": 2, "blackman": 3, "nuttall": 4,
           "blackman_harris": 5}


class SdrError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"libsdrhip status {status}: {msg}")
        self.status = status


class FirDesc(C.Structure):
    _fields_ = [("input", C.c_int), ("mode", C.c_int), ("decim", C.c_int), ("n_taps", C.c_int),
                ("taps", C.POINTER(C.c_double)), ("scale", C.c_double), ("mix_fr", C.c_double),
                ("outputs_per_lane", C.c_int), ("threads", C.c_int), ("pipeline", C.c_int),
                ("count_rule", C.c_int), ("size_hint", C.c_longlong),
                ("n_mix_channels", C.c_int), ("mix_fr_channels", C.POINTER(C.c_double))]


# per-file compiler flags: the FFT's complex butterflies lose ~12 % when the SLP vectoriser pairs
# them into v_pk_add_f32 (same issue cost as two scalar adds on gfx950, plus the moves that build
# the register pairs); the FIR walk gains from v_pk_fma_f32 and keeps the default
# fft_wave.hip (one 4096-point frame per wavefront) is bound by vector-instruction issue: the max-ILP
# scheduling strategy leaves fewer dependent-issue stalls there (0.723 vs 0.734 ms per 2^28 samples)
# fir_mfma_i8.hip: the int8 matrix kernel draws its tiles from an atomic counter one phase ahead; LLVM's atomic
# optimizer would rewrite that one-lane atomic into a wave-wide form that reads the result back at once
# (s_waitcnt vmcnt(0) right behind the atomic: the round trip, and the acknowledgements of the stores
# before it, in every phase).  Results do not depend on the flag, only the kernel's speed.
# fir_mfma_i8.hip, -fno-slp-vectorize: REQUIRED FOR CORRECT RESULTS, not a tuning flag.  With the SLP vectoriser on,
# hipcc (ROCm 7.2) turns the helper waves' discriminator arithmetic into packed float32 instructions (v_pk_mul_f32 /
# v_pk_fma_f32), and on MI355X the LOW half of those results came out wrong in lanes 48-63 -- intermittently, and only
# while the matrix wave of the same SIMD was inside its MFMA loop (short phases: D = 1, 2; several tiles per workgroup;
# up to 97 % of launches at 7 taps / 1 on 1.2 M samples; tests/test_gpu_parity.py::test_matrix_kernel_many_short_tiles).
# Without packed float32 in that file the failure is gone (and packed float32 cannot issue in the shadow of another
# wave's MFMAs anyway).  An isolated reproduction (tools/ubench/pk_under_mfma.hip) did not show it, so
# the trigger is narrower than "packed float32 beside MFMA"; DESIGN 3.4 records what is known.
EXTRA_FLAGS = {"fir_mfma_i8.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-fno-slp-vectorize"],
               "fft.hip": ["-fno-slp-vectorize"],
               "fft_wave.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


def build(force=False):
    """hipcc -> sdr_amd/libsdrhip.so (cross-compiles for gfx950 without a GPU)."""
    srcs = [s for s in SOURCES if os.path.exists(s)]
    deps = srcs + [os.path.join(_HERE, "csrc", "sdr_internal.h"), os.path.join(_HERE, "csrc", "fft_common.h"),
                   os.path.join(_HERE, "csrc", "fir_common.h"),
                   os.path.join(INCLUDE, "sdr_hip.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(LIB_PATH) and \
            all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    base = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
            "-I" + INCLUDE, "-I" + os.path.join(_HERE, "csrc")]
    objs, procs = [], []
    for src in srcs:
        name = os.path.basename(src)
        obj = os.path.join(objdir, name.replace(".hip", ".o"))
        objs.append(obj)
        procs.append(subprocess.Popen(base + EXTRA_FLAGS.get(name, []) + ["-c", src, "-o", obj]))
    for pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, pr.args)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SdrError(ENODEV, f"{LIB_PATH} not built (run __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        L.sdr_strerror.restype = C.c_char_p
        L.sdr_last_error.restype = C.c_char_p
        L.sdr_build_info.restype = C.c_char_p
        L.sdr_fir_out_count.restype = C.c_size_t
        L.sdr_fir_out_count.argtypes = [C.c_void_p, C.c_size_t]
        L.sdr_fft_frame_count.restype = C.c_size_t
        L.sdr_fft_frame_count.argtypes = [C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def check(status):
    if status != OK:
        L = lib()
        raise SdrError(status, f"{L.sdr_strerror(status).decode()} -- {L.sdr_last_error().decode()}")


def init(device=0):
    check(lib().sdr_init(C.c_int(device)))


def build_info():
    return lib().sdr_build_info().decode()


# ------------------------------------------------------------------------------ host design --
def window(name, n):
    w = np.zeros(n, np.float64)
    check(lib().sdr_window(C.c_int(WINDOWS[name]), C.c_size_t(n),
                           w.ctypes.data_as(C.POINTER(C.c_double))))
    return w


def create_fir_filter(n_taps, fc, window_name="hann"):
    h = np.zeros(n_taps, np.float64)
    check(lib().sdr_create_fir_filter(C.c_size_t(n_taps), C.c_double(fc),
                                      C.c_int(WINDOWS[window_name]),
                                      h.ctypes.data_as(C.POINTER(C.c_double))))
    return h


# ---------------------------------------------------------------------------- device memory --
class DevBuf:
    """A device allocation owned through the C ABI (sdr_dev_alloc / sdr_dev_free)."""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        check(lib().sdr_dev_alloc(C.byref(self.ptr), C.c_size_t(self.nbytes)))

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        check(lib().sdr_dev_upload(b.ptr, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes)))
        return b

    def to_numpy(self, dtype, count, offset_bytes=0):
        out = np.empty(count, dtype)
        check(lib().sdr_dev_download(out.ctypes.data_as(C.c_void_p),
                                     C.c_void_p(self.ptr.value + offset_bytes),
                                     C.c_size_t(out.nbytes)))
        return out

    def at(self, offset_bytes):
        return C.c_void_p(self.ptr.value + int(offset_bytes))

    def free(self):
        if self.ptr and self.ptr.value:
            lib().sdr_dev_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sync():
    check(lib().sdr_dev_sync())


def _vp(x):
    """device pointer argument: DevBuf, c_void_p, int, or a torch CUDA tensor"""
    if isinstance(x, DevBuf):
        return x.ptr
    if isinstance(x, C.c_void_p):
        return x
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    raise TypeError(type(x))


# ---------------------------------------------------------------------------------- FIR plan --
class FirPlan:
    """sdr_fir_plan: FIR + decimate (+ mixer) + {IQ, FM discriminator, envelope}."""

    def __init__(self, taps, decim, mode=MODE_FM, input=IN_I16_IQ, scale=1.0 / (2.0 * np.pi),
                 mix_fr=0.0, outputs_per_lane=0, threads=0, pipeline=0, count_rule=COUNT_FLOOR, size_hint=0,
                 mix_fr_channels=None):
        self._taps = np.ascontiguousarray(taps, dtype=np.float64)
        self.mode, self.input, self.decim = mode, input, decim
        self._chan = None if mix_fr_channels is None else np.ascontiguousarray(mix_fr_channels, dtype=np.float64)
        d = FirDesc(input, mode, decim, self._taps.size,
                    self._taps.ctypes.data_as(C.POINTER(C.c_double)), scale, mix_fr,
                    outputs_per_lane, threads, pipeline, count_rule, int(size_hint),
                    0 if self._chan is None else self._chan.size,
                    None if self._chan is None else self._chan.ctypes.data_as(C.POINTER(C.c_double)))
        self.h = C.c_void_p()
        check(lib().sdr_fir_plan_create(C.byref(d), C.byref(self.h)))

    def info(self):
        r, nt, lds = C.c_int(), C.c_int(), C.c_int()
        check(lib().sdr_fir_plan_info(self.h, C.byref(r), C.byref(nt), C.byref(lds)))
        return {"outputs_per_lane": r.value, "threads": nt.value, "lds_bytes": lds.value,
                "kernel": {1: "raw_tile", 2: "stream", 3: "planar", 4: "mfma", 5: "mfma_i8", 6: "ksplit", 7: "float_dma_tile"}.get(lib().sdr_fir_plan_kernel(self.h), "?")}

    def out_count(self, n_in):
        return lib().sdr_fir_out_count(self.h, n_in)

    @property
    def out_dtype(self):
        return np.complex64 if self.mode == MODE_IQ else np.float32

    def _in_array(self, x):
        if self.input in (IN_I16_IQ, IN_I16_QI):
            a = np.ascontiguousarray(x, dtype=np.int16).reshape(-1)
            return a, a.size // 2
        if self.input == IN_I8_IQ:
            a = np.ascontiguousarray(x, dtype=np.int8).reshape(-1)
            return a, a.size // 2
        if self.input == IN_CF32:
            a = np.ascontiguousarray(x, dtype=np.complex64).reshape(-1)
            return a, a.size
        a = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
        return a, a.size

    def execute_host(self, x):
        a, n = self._in_array(x)
        cnt = self.out_count(n)
        out = np.empty(max(cnt, 1), self.out_dtype)
        n_out = C.c_size_t()
        check(lib().sdr_fir_execute_host(self.h, a.ctypes.data_as(C.c_void_p), C.c_size_t(n),
                                         out.ctypes.data_as(C.c_void_p), C.c_size_t(cnt),
                                         C.byref(n_out)))
        return out[: n_out.value]

    def _in_bytes(self):
        return {IN_I16_IQ: 4, IN_I16_QI: 4, IN_I8_IQ: 2, IN_CF32: 8}.get(self.input, 4)

    def _fits(self, what, buf, need_bytes):
        """The C ABI trusts the caller's device pointers.  When a buffer is a DevBuf its size is known
        here, so a launch that would read or write past it is refused on the host (an undersized
        output buffer in a measurement script is a GPU memory fault otherwise)."""
        if isinstance(buf, DevBuf) and need_bytes > buf.nbytes:
            raise ValueError(f"{what}: the launch needs {need_bytes} bytes, the DevBuf holds {buf.nbytes}")

    def execute_dev(self, d_in, n_in, d_out, out_capacity, stream=None):
        self._fits("input", d_in, n_in * self._in_bytes())
        self._fits("output", d_out, min(out_capacity, self.out_count(n_in)) * np.dtype(self.out_dtype).itemsize)
        n_out = C.c_size_t()
        check(lib().sdr_fir_execute_dev(self.h, _vp(d_in), C.c_size_t(n_in), _vp(d_out),
                                        C.c_size_t(out_capacity), C.byref(n_out),
                                        C.c_void_p(stream or 0)))
        return n_out.value

    def execute_range_dev(self, d_in, in_first, n_resident, n_total, out_begin, out_end, d_out,
                          stream=None):
        self._fits("input", d_in, n_resident * self._in_bytes())
        self._fits("output", d_out, max(out_end - out_begin, 0) * np.dtype(self.out_dtype).itemsize)
        check(lib().sdr_fir_execute_range_dev(self.h, _vp(d_in), C.c_uint64(in_first),
                                              C.c_uint64(n_resident), C.c_uint64(n_total),
                                              C.c_uint64(out_begin), C.c_uint64(out_end),
                                              _vp(d_out), C.c_void_p(stream or 0)))

    def execute_batched_dev(self, d_in, n_in, in_stride, n_channels, d_out, out_stride,
                            stream=None):
        cnt = self.out_count(n_in)
        if n_channels > 0:
            self._fits("input", d_in, ((n_channels - 1) * (in_stride or n_in) + n_in) * self._in_bytes())
            self._fits("output", d_out, ((n_channels - 1) * (out_stride or cnt) + cnt) * np.dtype(self.out_dtype).itemsize)
        check(lib().sdr_fir_execute_batched_dev(self.h, _vp(d_in), C.c_size_t(n_in),
                                                C.c_size_t(in_stride), C.c_size_t(n_channels),
                                                _vp(d_out), C.c_size_t(out_stride),
                                                C.c_void_p(stream or 0)))

    def execute_channels_dev(self, d_in, n_in, in_stride, chan_first, n_channels, d_out, out_stride, stream=None):
        """sdr_fir_execute_channels_dev: per-channel NCO + FIR + decimate + demod; in_stride 0 = one shared buffer"""
        cnt = self.out_count(n_in)
        if n_channels > 0:
            self._fits("input", d_in, ((n_channels - 1) * in_stride + n_in) * self._in_bytes())
            self._fits("output", d_out, ((n_channels - 1) * (out_stride or cnt) + cnt) * np.dtype(self.out_dtype).itemsize)
        check(lib().sdr_fir_execute_channels_dev(self.h, _vp(d_in), C.c_size_t(n_in), C.c_size_t(in_stride),
                                                 C.c_size_t(chan_first), C.c_size_t(n_channels), _vp(d_out),
                                                 C.c_size_t(out_stride), C.c_void_p(stream or 0)))

    def time_dev(self, d_in, n_in, d_out, iters, in_stride=0, n_channels=1, out_stride=0,
                 stream=None):
        """average ms per kernel launch, measured with HIP events on the launch stream"""
        cnt = self.out_count(n_in)
        if n_channels > 0:
            self._fits("input", d_in, ((n_channels - 1) * (in_stride or n_in) + n_in) * self._in_bytes())
            self._fits("output", d_out, ((n_channels - 1) * (out_stride or cnt) + cnt) * np.dtype(self.out_dtype).itemsize)
        ms = C.c_float()
        check(lib().sdr_fir_time_dev(self.h, _vp(d_in), C.c_size_t(n_in), C.c_size_t(in_stride),
                                     C.c_size_t(n_channels), _vp(d_out), C.c_size_t(out_stride),
                                     C.c_int(iters), C.c_void_p(stream or 0), C.byref(ms)))
        return ms.value

    def destroy(self):
        if self.h and self.h.value:
            lib().sdr_fir_plan_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


# ---------------------------------------------------------------------------------- FFT plan --
class FftPlan:
    def __init__(self, nfft, hop, window=None):
        self.nfft, self.hop = nfft, hop
        w = None if window is None else np.ascontiguousarray(window, dtype=np.float64)
        self.h = C.c_void_p()
        check(lib().sdr_fft_plan_create(C.c_int(nfft), C.c_int(hop),
                                        w.ctypes.data_as(C.POINTER(C.c_double)) if w is not None
                                        else None, C.byref(self.h)))

    def frame_count(self, n_in):
        return lib().sdr_fft_frame_count(self.h, n_in)

    def set_variant(self, variant):
        """0 auto, 1 workgroup exchange, 2 two frames per lane, 3 one frame per wavefront (4 waves per CU),
        4 one frame per wavefront with 8 waves per CU (what the automatic choice runs)"""
        check(lib().sdr_fft_plan_set_variant(self.h, C.c_int(variant)))

    def execute_host(self, iq):
        a = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1)
        n = a.size // 2
        frames = self.frame_count(n)
        out = np.empty((max(frames, 1), self.nfft), np.float32)
        nf = C.c_size_t()
        check(lib().sdr_fft_execute_i16_host(self.h, a.ctypes.data_as(C.c_void_p), C.c_size_t(n),
                                             out.ctypes.data_as(C.c_void_p), C.c_size_t(frames),
                                             C.byref(nf)))
        return out[: nf.value]

    def execute_dev(self, d_iq, n_in, d_out, frame_capacity, stream=None):
        nf = C.c_size_t()
        check(lib().sdr_fft_execute_i16_dev(self.h, _vp(d_iq), C.c_size_t(n_in), _vp(d_out),
                                            C.c_size_t(frame_capacity), C.byref(nf),
                                            C.c_void_p(stream or 0)))
        return nf.value

    def execute_noalloc_dev(self, d_iq, n_in, d_out, frame_capacity, stream=None):
        """sdr_fft_execute_i16_noalloc_dev: no device allocation on any path (nfft <= 131072)"""
        nf = C.c_size_t()
        check(lib().sdr_fft_execute_i16_noalloc_dev(self.h, _vp(d_iq), C.c_size_t(n_in), _vp(d_out),
                                                    C.c_size_t(frame_capacity), C.byref(nf),
                                                    C.c_void_p(stream or 0)))
        return nf.value

    def time_dev(self, d_iq, n_in, d_out, iters, stream=None):
        ms = C.c_float()
        check(lib().sdr_fft_time_dev(self.h, _vp(d_iq), C.c_size_t(n_in), _vp(d_out),
                                     C.c_int(iters), C.c_void_p(stream or 0), C.byref(ms)))
        return ms.value

    def destroy(self):
        if self.h and self.h.value:
            lib().sdr_fft_plan_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


# ------------------------------------------------------------------------ element-wise calls --
def unpack_i16_dev(d_iq, n, d_out, stream=None):
    check(lib().sdr_unpack_i16_dev(_vp(d_iq), C.c_size_t(n), _vp(d_out), C.c_void_p(stream or 0)))


def magnitude_i16_dev(d_iq, n, d_out, stream=None):
    check(lib().sdr_magnitude_i16_dev(_vp(d_iq), C.c_size_t(n), _vp(d_out),
                                      C.c_void_p(stream or 0)))


def audio_u8_dev(d_x, n, d_max, d_pcm, stream=None):
    check(lib().sdr_audio_u8_dev(_vp(d_x), C.c_size_t(n), _vp(d_max), _vp(d_pcm), C.c_void_p(stream or 0)))


def frequency_shift_dev(d_in, is_complex, n, n0, fr, d_out, stream=None):
    check(lib().sdr_frequency_shift_dev(_vp(d_in) if d_in is not None else None,
                                        C.c_int(1 if is_complex else 0), C.c_size_t(n),
                                        C.c_uint64(n0), C.c_double(fr), _vp(d_out),
                                        C.c_void_p(stream or 0)))


def gather_dev(d_in, elem_dwords, n_in, d_idx, n_idx, d_out, stream=None):
    check(lib().sdr_gather_dev(_vp(d_in), C.c_int(elem_dwords), C.c_size_t(n_in), _vp(d_idx),
                               C.c_size_t(n_idx), _vp(d_out), C.c_void_p(stream or 0)))


def synth_fm_i16_dev(d_iq, n, n0=0, fs=1.0e6, ftone=1000.0, dev=5000.0, amp=2000.0, noise_lsb=0,
                     seed=1, stream=None):
    check(lib().sdr_synth_fm_i16_dev(_vp(d_iq), C.c_size_t(n), C.c_uint64(n0), C.c_double(fs),
                                     C.c_double(ftone), C.c_double(dev), C.c_double(amp),
                                     C.c_int(noise_lsb), C.c_uint(seed), C.c_void_p(stream or 0)))


class FirStream:
    """sdr_fir_stream: push blocks, pop outputs; filter history persists across blocks."""

    def __init__(self, plan, max_block_samples):
        self.plan = plan
        self.h = C.c_void_p()
        check(lib().sdr_fir_stream_create(plan.h, C.c_size_t(max_block_samples), C.byref(self.h)))
        self._cap = max_block_samples // plan.decim + 8 + 4096

    def push(self, block):
        a, n = self.plan._in_array(block)
        check(lib().sdr_fir_stream_push(self.h, a.ctypes.data_as(C.c_void_p), C.c_size_t(n)))

    def next_buffer(self, dtype, count):
        """numpy view of the pinned staging buffer the next push will use (zero-copy hand-off)"""
        ptr = C.c_void_p()
        check(lib().sdr_fir_stream_next_buffer(self.h, C.byref(ptr)))
        n = count * np.dtype(dtype).itemsize
        return np.frombuffer((C.c_char * n).from_address(ptr.value), dtype=dtype, count=count)

    def push_buffer(self, buf, n_samples):
        check(lib().sdr_fir_stream_push(self.h, C.c_void_p(buf.ctypes.data), C.c_size_t(n_samples)))

    def pop(self):
        out = np.empty(self._cap, self.plan.out_dtype)
        n = C.c_size_t()
        check(lib().sdr_fir_stream_pop(self.h, out.ctypes.data_as(C.c_void_p), C.c_size_t(self._cap),
                                       C.byref(n)))
        return out[: n.value].copy()

    def replay_file(self, path, block_samples, passes=1):
        """sdr_fir_stream_replay_file: returns (outputs of all blocks concatenated, samples pushed)"""
        chunks = []
        dt = self.plan.out_dtype
        size = np.dtype(dt).itemsize

        @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
        def sink(_user, ptr, n):
            if n:
                chunks.append(np.frombuffer(C.string_at(ptr, n * size), dtype=dt).copy())
            return 0

        done = C.c_uint64()
        check(lib().sdr_fir_stream_replay_file(self.h, str(path).encode(), C.c_size_t(block_samples),
                                               C.c_int(passes), sink, None, C.byref(done)))
        out = np.concatenate(chunks) if chunks else np.empty(0, dt)
        return out, done.value

    def destroy(self):
        if self.h and self.h.value:
            lib().sdr_fir_stream_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class AptImage:
    """sdr_apt_image: line sync + scrolling image (bladerf_apt_demod.cpp:756-793)."""

    def __init__(self, width=2080, rows=700):
        self.width, self.rows = width, rows
        self.h = C.c_void_p()
        check(lib().sdr_apt_image_create(C.c_int(width), C.c_int(rows), C.byref(self.h)))

    def push(self, x9):
        a = np.ascontiguousarray(x9, dtype=np.float32).reshape(-1)
        added = C.c_int()
        check(lib().sdr_apt_image_push(self.h, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size), C.byref(added)))
        return added.value

    def get(self):
        img = np.empty((self.rows, self.width), np.uint8)
        lines, pend = C.c_uint64(), C.c_size_t()
        check(lib().sdr_apt_image_get(self.h, img.ctypes.data_as(C.c_void_p), C.byref(lines), C.byref(pend)))
        return img, lines.value, pend.value

    def destroy(self):
        if self.h and self.h.value:
            lib().sdr_apt_image_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
