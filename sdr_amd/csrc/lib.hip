// lib.hip -- library plumbing (device check, errors, device-memory helpers), the host-side
// filter designer, and the small element-wise kernels of the receive path.

#include <cmath>
#include <cstring>

#include <atomic>

#include "sdr_internal.h"

namespace sdr {

static thread_local char g_err[512] = "";
static thread_local int g_ready_device = -1;

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what)
{
    set_error("%s: %s", what, hipGetErrorString(e));
    return SDR_EHIP;
}

bool device_ready() { return g_ready_device >= 0; }

static int check_device(int dev)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device (%s): libsdrhip has no CPU fallback", hipGetErrorString(e));
        return SDR_ENODEV;
    }
    if (dev < 0 || dev >= count) { set_error("device %d out of range (0..%d)", dev, count - 1); return SDR_EINVAL; }
    hipDeviceProp_t prop;
    SDR_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libsdrhip is built for gfx950 only", dev, prop.gcnArchName);
        return SDR_ENODEV;
    }
    SDR_HIP(hipSetDevice(dev));
    g_ready_device = dev;
    // with the device current: resolve the static scratch of the allocation-free FFT entry point
    if (int rc = fft_static_scratch_init(nullptr)) return rc;
    return SDR_OK;
}

int ensure_device()
{
    int cur = 0;
    if (g_ready_device >= 0) {
        if (hipGetDevice(&cur) == hipSuccess && cur == g_ready_device) return SDR_OK;
    }
    if (hipGetDevice(&cur) != hipSuccess) cur = 0;
    return check_device(cur);
}

// A plan's tables and scratch live on the device it was created on; a launch from a thread whose
// current device is another one would hand that device pointers it may not be able to reach.
int same_device(int plan_device)
{
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != plan_device) {
        set_error("plan belongs to device %d but the calling thread's current device is %d (sdr_init / hipSetDevice first)", plan_device, cur);
        return SDR_EINVAL;
    }
    return SDR_OK;
}

}  // namespace sdr

extern "C" const char *sdr_build_info(void)
{
    static char text[256];
    const int fir = sdr::measurement_flags_fir(), fft = sdr::measurement_flags_fft(), fftw = sdr::measurement_flags_fft_wave();
    if (!fir && !fft && !fftw) return "libsdrhip product build (gfx950)";
    snprintf(text, sizeof(text), "libsdrhip MEASUREMENT BUILD -- results are wrong by construction "
             "(ablation masks: fir_mfma_i8.hip %d, fft.hip %d, fft_wave.hip %d; -1 = built with SDR_MEASUREMENT_BUILD, no mask)", fir, fft, fftw);
    return text;
}

extern "C" int sdr_init(int device)
{
    if (sdr::measurement_flags_fir() || sdr::measurement_flags_fft() || sdr::measurement_flags_fft_wave()) fprintf(stderr, "%s\n", sdr_build_info());
    return sdr::check_device(device);
}

extern "C" const char *sdr_strerror(int s)
{
    switch (s) {
    case SDR_OK: return "ok";
    case SDR_EINVAL: return "invalid argument";
    case SDR_ENOMEM: return "out of memory";
    case SDR_EHIP: return "HIP runtime error";
    case SDR_EALIGN: return "misaligned pointer";
    case SDR_ECAPACITY: return "output buffer too small";
    case SDR_ENODEV: return "no gfx950 device";
    case SDR_ERANGE: return "samples outside the resident window";
    default: return "unknown status";
    }
}

extern "C" const char *sdr_last_error(void) { return sdr::g_err; }

namespace sdr {
static std::atomic<unsigned long long> g_allocs{0};
void note_alloc() { g_allocs.fetch_add(1, std::memory_order_relaxed); }
}  // namespace sdr
extern "C" unsigned long long sdr_alloc_count(void) { return sdr::g_allocs.load(std::memory_order_relaxed); }

extern "C" int sdr_dev_alloc(void **ptr, size_t bytes)
{
    if (!ptr) return SDR_EINVAL;
    int rc = sdr::ensure_device();
    if (rc != SDR_OK) return rc;
    SDR_HIP(sdr::dev_malloc(ptr, bytes ? bytes : 16));
    return SDR_OK;
}
extern "C" int sdr_dev_free(void *ptr) { if (ptr) SDR_HIP(hipFree(ptr)); return SDR_OK; }
extern "C" int sdr_dev_upload(void *dst, const void *src, size_t bytes)
{
    SDR_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return SDR_OK;
}
extern "C" int sdr_dev_download(void *dst, const void *src, size_t bytes)
{
    SDR_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return SDR_OK;
}
extern "C" int sdr_dev_sync(void) { SDR_HIP(hipDeviceSynchronize()); return SDR_OK; }

// ---------------------------------------------------------------------------------------------
// a11: windows + windowed-sinc designer (host).  See sdr_hip.h for the definition; PARITY
// UNPINNED because DSP::create_fir_filter's source (davemers0160/Common) is not in the tree.
// ---------------------------------------------------------------------------------------------
extern "C" int sdr_window(int id, size_t N, double *w)
{
    if (!w || N == 0) { sdr::set_error("window: bad argument"); return SDR_EINVAL; }
    if (id < SDR_WIN_RECTANGULAR || id > SDR_WIN_BLACKMAN_HARRIS) { sdr::set_error("unknown window %d", id); return SDR_EINVAL; }
    if (N == 1) { w[0] = 1.0; return SDR_OK; }
    static const double coef[6][4] = {
        {1.0, 0.0, 0.0, 0.0},
        {0.5, 0.5, 0.0, 0.0},
        {0.54, 0.46, 0.0, 0.0},
        {0.42, 0.5, 0.08, 0.0},
        {0.355768, 0.487396, 0.144232, 0.012604},
        {0.35875, 0.48829, 0.14128, 0.01168}};
    const double *a = coef[id];
    const double den = (double)(N - 1);
    for (size_t n = 0; n < N; ++n) {
        const double t = (double)n / den;
        double v = a[0];
        if (id != SDR_WIN_RECTANGULAR) v -= a[1] * cos(2.0 * M_PI * t);
        if (id >= SDR_WIN_BLACKMAN) v += a[2] * cos(4.0 * M_PI * t);
        if (id >= SDR_WIN_NUTTALL) v -= a[3] * cos(6.0 * M_PI * t);
        w[n] = v;
    }
    return SDR_OK;
}

extern "C" int sdr_create_fir_filter(size_t N, double fc, int window_id, double *h)
{
    if (!h || N == 0) { sdr::set_error("create_fir_filter: bad argument"); return SDR_EINVAL; }
    int rc = sdr_window(window_id, N, h);
    if (rc != SDR_OK) return rc;
    const double c = ((double)N - 1.0) / 2.0;
    for (size_t n = 0; n < N; ++n) {
        const double t = (double)n - c;
        const double s = (t == 0.0) ? fc : sin(M_PI * fc * t) / (M_PI * t);
        h[n] = h[n] * s;
    }
    return SDR_OK;
}

// ---------------------------------------------------------------------------------------------
// element-wise kernels (HBM-bound streaming, 16 B per lane)
// ---------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float lo16(uint32_t w) { return (float)(short)(w & 0xffffu); }
__device__ __forceinline__ float hi16(uint32_t w) { return (float)((int)w >> 16); }

// a2: bladerf_sdr.h:288-291 -- complex<float>(re, im) * (1/2048)
__global__ void __launch_bounds__(256) unpack_kernel(const uint32_t *__restrict__ in, size_t n,
                                                     float2 *__restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n / 4;
    const bool vec = ((((uintptr_t)in) & 15) == 0) && ((((uintptr_t)out) & 15) == 0);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        for (; i < n4; i += stride) {
            const uint4 v = ((const uint4 *)in)[i];
            float4 *o = (float4 *)(out + 4 * i);
            o[0] = make_float4(lo16(v.x) * sdr::kQ11, hi16(v.x) * sdr::kQ11, lo16(v.y) * sdr::kQ11, hi16(v.y) * sdr::kQ11);
            o[1] = make_float4(lo16(v.z) * sdr::kQ11, hi16(v.z) * sdr::kQ11, lo16(v.w) * sdr::kQ11, hi16(v.w) * sdr::kQ11);
        }
        i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    }
    for (; i < n; i += stride) {
        const uint32_t w = in[i];
        out[i] = make_float2(lo16(w) * sdr::kQ11, hi16(w) * sdr::kQ11);
    }
}

// a6 waterfall row: bladerf_waterfall_main.cpp:272 -- abs(complex(re*s, im*s))
__global__ void __launch_bounds__(256) magnitude_kernel(const uint32_t *__restrict__ in, size_t n,
                                                        float *__restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t w = in[i];
        out[i] = hypotf(lo16(w) * sdr::kQ11, hi16(w) * sdr::kQ11);
    }
}

// a7: frequency_shift / create_freq_rotation
__global__ void __launch_bounds__(256) freq_shift_kernel(const float *__restrict__ in, int is_complex,
                                                         size_t n, unsigned long long n0, double fr,
                                                         float2 *__restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double ph = fr * (double)(n0 + i);
        const float f = (float)(ph - rint(ph));
        float sn, cs;
        sincospif(2.0f * f, &sn, &cs);
        float2 x = make_float2(1.f, 0.f);
        if (in) x = is_complex ? ((const float2 *)in)[i] : make_float2(in[i], 0.f);
        out[i] = make_float2(__fmaf_rn(x.x, cs, -(x.y * sn)), __fmaf_rn(x.x, sn, x.y * cs));
    }
}

// Synthetic FM test signal: phi(n) = (dev/ftone)*sin(2*pi*ftone*n/fs),
// I = round(amp*cos(phi)), Q = round(amp*sin(phi)), optional uniform noise of +-noise LSB from a
// counter hash (seeded).  Input generator for bench/tests at sizes that should never cross PCIe.
__global__ void __launch_bounds__(256) synth_fm_kernel(uint32_t *__restrict__ out, size_t n,
                                                       unsigned long long n0, double fs, double ftone,
                                                       double dev, double amp, int noise, unsigned seed)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const double beta = dev / ftone;
    const double w = ftone / fs;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned long long t = n0 + i;
        const double cyc = w * (double)t;
        const double ph = beta * sinpi(2.0 * (cyc - rint(cyc)));
        double sn, cs;
        sincos(ph, &sn, &cs);
        int I = (int)rint(amp * cs), Q = (int)rint(amp * sn);
        if (noise > 0) {
            unsigned long long z = (t + 1) * 0x9E3779B97F4A7C15ULL + seed;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            z ^= z >> 31;
            I += (int)((unsigned)z % (unsigned)(2 * noise + 1)) - noise;
            Q += (int)((unsigned)(z >> 32) % (unsigned)(2 * noise + 1)) - noise;
        }
        I = I < -32768 ? -32768 : (I > 32767 ? 32767 : I);
        Q = Q < -32768 ? -32768 : (Q > 32767 ? 32767 : Q);
        out[i] = ((uint32_t)(uint16_t)(int16_t)I) | ((uint32_t)(uint16_t)(int16_t)Q << 16);
    }
}

// a8 support: out[i] = in[idx[i]], elements of `ew` 32-bit words (decimate_vec's pick); an index
// outside [0, n_in) gives a zero element instead of a stray read
__global__ void __launch_bounds__(256) gather_kernel(const uint32_t *__restrict__ in, int ew, size_t n_in,
                                                     const unsigned long long *__restrict__ idx, size_t n,
                                                     uint32_t *__restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t total = n * (size_t)ew;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t e = i / ew, w = i - e * ew;
        const unsigned long long src = idx[e];
        out[i] = src < n_in ? in[src * ew + w] : 0u;
    }
}

unsigned stream_grid(size_t work_items)
{
    size_t blocks = (work_items + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

// max|x| of a float vector: wave reduction, then one global atomic per wave on the bit pattern
// (non-negative floats order like unsigned integers)
__global__ void __launch_bounds__(256) absmax_kernel(const float *__restrict__ x, size_t n,
                                                     unsigned *__restrict__ mx)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(mx, __float_as_uint(m));
}

// bladerf_fm_demod.cpp:310-313: x * (1/max) -> (x + 1) * 30 -> u8 (truncation)
__global__ void __launch_bounds__(256) audio_u8_kernel(const float *__restrict__ x, size_t n,
                                                       const float *__restrict__ mx,
                                                       uint8_t *__restrict__ out)
{
    const float m = *mx;
    const float r = m > 0.f ? (float)(1.0 / (double)m) : 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = (x[i] * r + 1.0f) * 30.0f;
        out[i] = (uint8_t)(int)v;
    }
}

extern "C" int sdr_audio_u8_dev(const float *d_x, size_t n, float *d_max, uint8_t *d_pcm, void *stream)
{
    if (n == 0) return SDR_OK;
    if (!d_x || !d_max || !d_pcm) { sdr::set_error("null argument"); return SDR_EINVAL; }
    if (((uintptr_t)d_x) & 3 || ((uintptr_t)d_max) & 3) { sdr::set_error("misaligned pointer"); return SDR_EALIGN; }
    int rc = sdr::ensure_device();
    if (rc != SDR_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    SDR_HIP(hipMemsetAsync(d_max, 0, sizeof(float), st));
    absmax_kernel<<<stream_grid(n), 256, 0, st>>>(d_x, n, (unsigned *)d_max);
    SDR_HIP(hipGetLastError());
    audio_u8_kernel<<<stream_grid(n), 256, 0, st>>>(d_x, n, d_max, d_pcm);
    SDR_HIP(hipGetLastError());
    return SDR_OK;
}

extern "C" int sdr_unpack_i16_dev(const int16_t *d_iq, size_t n, float *d_out, void *stream)
{
    if (n == 0) return SDR_OK;
    if (!d_iq || !d_out) { sdr::set_error("null argument"); return SDR_EINVAL; }
    if (((uintptr_t)d_iq) & 3 || ((uintptr_t)d_out) & 7) { sdr::set_error("misaligned pointer"); return SDR_EALIGN; }
    int rc = sdr::ensure_device();
    if (rc != SDR_OK) return rc;
    unpack_kernel<<<stream_grid(n / 4 + 1), 256, 0, (hipStream_t)stream>>>((const uint32_t *)d_iq, n, (float2 *)d_out);
    SDR_HIP(hipGetLastError());
    return SDR_OK;
}

extern "C" int sdr_magnitude_i16_dev(const int16_t *d_iq, size_t n, float *d_mag, void *stream)
{
    if (n == 0) return SDR_OK;
    if (!d_iq || !d_mag) { sdr::set_error("null argument"); return SDR_EINVAL; }
    if (((uintptr_t)d_iq) & 3 || ((uintptr_t)d_mag) & 3) { sdr::set_error("misaligned pointer"); return SDR_EALIGN; }
    int rc = sdr::ensure_device();
    if (rc != SDR_OK) return rc;
    magnitude_kernel<<<stream_grid(n), 256, 0, (hipStream_t)stream>>>((const uint32_t *)d_iq, n, d_mag);
    SDR_HIP(hipGetLastError());
    return SDR_OK;
}

extern "C" int sdr_frequency_shift_dev(const float *d_in, int is_complex, size_t n, uint64_t n0,
                                       double fr, float *d_out, void *stream)
{
    if (n == 0) return SDR_OK;
    if (!d_out) { sdr::set_error("null output"); return SDR_EINVAL; }
    if (((uintptr_t)d_out) & 7 || (d_in && (((uintptr_t)d_in) & (is_complex ? 7 : 3)))) { sdr::set_error("misaligned pointer"); return SDR_EALIGN; }
    int rc = sdr::ensure_device();
    if (rc != SDR_OK) return rc;
    freq_shift_kernel<<<stream_grid(n), 256, 0, (hipStream_t)stream>>>(d_in, is_complex, n, n0, fr, (float2 *)d_out);
    SDR_HIP(hipGetLastError());
    return SDR_OK;
}

extern "C" int sdr_synth_fm_i16_dev(int16_t *d_iq, size_t n, uint64_t n0, double fs, double ftone,
                                    double dev, double amp, int noise_lsb, unsigned seed,
                                    void *stream)
{
    if (n == 0) return SDR_OK;
    if (!d_iq || fs <= 0 || ftone <= 0) { sdr::set_error("bad argument"); return SDR_EINVAL; }
    if (((uintptr_t)d_iq) & 3) { sdr::set_error("misaligned pointer"); return SDR_EALIGN; }
    int rc = sdr::ensure_device();
    if (rc != SDR_OK) return rc;
    synth_fm_kernel<<<stream_grid(n), 256, 0, (hipStream_t)stream>>>((uint32_t *)d_iq, n, n0, fs, ftone, dev, amp, noise_lsb, seed);
    SDR_HIP(hipGetLastError());
    return SDR_OK;
}

extern "C" int sdr_gather_dev(const void *d_in, int elem_dwords, size_t n_in, const uint64_t *d_idx,
                              size_t n_idx, void *d_out, void *stream)
{
    if (n_idx == 0) return SDR_OK;
    if (!d_in || !d_idx || !d_out || elem_dwords <= 0) { sdr::set_error("bad argument"); return SDR_EINVAL; }
    if (((uintptr_t)d_in) & 3 || ((uintptr_t)d_out) & 3 || ((uintptr_t)d_idx) & 7) { sdr::set_error("misaligned pointer"); return SDR_EALIGN; }
    int rc = sdr::ensure_device();
    if (rc != SDR_OK) return rc;
    gather_kernel<<<stream_grid(n_idx * elem_dwords), 256, 0, (hipStream_t)stream>>>(
        (const uint32_t *)d_in, elem_dwords, n_in, (const unsigned long long *)d_idx, n_idx, (uint32_t *)d_out);
    SDR_HIP(hipGetLastError());
    return SDR_OK;
}
