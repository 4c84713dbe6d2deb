// fft.hip -- windowed FFT -> power in dB for the waterfall / spectrum path (gfx950).
//
// Replaces bladerf/rx_example/src/bladerf_main_v1.cpp:183-196
//   c = complex<float>(I, Q) * (1/2048);  af::fftInPlace(c, 1/N);
//   fft_data = 20*log10(shift(abs(c), N/2))
// plus the window multiply and the hop (50 % overlap) of BASELINE config 4.
//
// One frame is transformed by N/16 lanes with a Stockham autosort FFT whose passes are
// radix-16 (then radix 16/8/4 to finish): each lane keeps 16 points in registers, so a 4096-point
// frame is three register passes with only two exchanges through LDS.
//   pass 1 reads the int16 IQ samples straight from HBM (coalesced 4-byte loads at stride N/16),
//          multiplies by the window table (which also carries the exact 2^-11 * 1/N scale),
//   the last pass computes |X|^2 -> 10*log10 and stores the fft-shifted row, coalesced.
// A 256-lane workgroup holds 4096/N frames; the LDS exchange image is padded by one float2 per
// 16 so that the stride-16 scatter of pass 1 and the stride-N/16 gathers hit distinct banks.
// Twiddles come from a table exp(-2*pi*i*k/N) in global memory (L1/L2 resident).

#include <cmath>
#include <vector>

#include "sdr_internal.h"

namespace {

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(__fmaf_rn(a.x, b.x, -(a.y * b.y)), __fmaf_rn(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// a - j*b  and  a + j*b
__device__ __forceinline__ float2 csub_j(float2 a, float2 b) { return make_float2(a.x + b.y, a.y - b.x); }
__device__ __forceinline__ float2 cadd_j(float2 a, float2 b) { return make_float2(a.x - b.y, a.y + b.x); }

// forward 4-point DFT (kernel e^{-j 2 pi nk/4}), natural order in and out
__device__ __forceinline__ void dft4(float2 &x0, float2 &x1, float2 &x2, float2 &x3)
{
    const float2 t0 = cadd(x0, x2), t1 = csub(x0, x2), t2 = cadd(x1, x3), t3 = csub(x1, x3);
    x0 = cadd(t0, t2);
    x2 = csub(t0, t2);
    x1 = csub_j(t1, t3);
    x3 = cadd_j(t1, t3);
}

template <int R>
__device__ __forceinline__ void dft(float2 (&x)[R]);

template <>
__device__ __forceinline__ void dft<2>(float2 (&x)[2])
{
    const float2 a = x[0], b = x[1];
    x[0] = cadd(a, b);
    x[1] = csub(a, b);
}

template <>
__device__ __forceinline__ void dft<4>(float2 (&x)[4]) { dft4(x[0], x[1], x[2], x[3]); }

// 8 = 2 x 4:  n = a + 2m, k = b + 4c:  X[b + 4c] = sum_a W8^{ab} (-1)^{ac} DFT4_m(x[a+2m])[b]
template <>
__device__ __forceinline__ void dft<8>(float2 (&x)[8])
{
    const float s = 0.70710678118654752f;
    float2 u0[4] = {x[0], x[2], x[4], x[6]};
    float2 u1[4] = {x[1], x[3], x[5], x[7]};
    dft4(u0[0], u0[1], u0[2], u0[3]);
    dft4(u1[0], u1[1], u1[2], u1[3]);
    u1[1] = cmul(u1[1], make_float2(s, -s));
    u1[2] = make_float2(u1[2].y, -u1[2].x);                 // * (-j)
    u1[3] = cmul(u1[3], make_float2(-s, -s));
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        x[b] = cadd(u0[b], u1[b]);
        x[b + 4] = csub(u0[b], u1[b]);
    }
}

// 16 = 4 x 4:  n = a + 4m, k = b + 4c:  X[b + 4c] = DFT4_a( W16^{ab} * DFT4_m(x[a+4m])[b] )[c]
template <>
__device__ __forceinline__ void dft<16>(float2 (&x)[16])
{
    const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, c2 = 0.70710678118654752f;
#pragma unroll
    for (int a = 0; a < 4; ++a) dft4(x[a], x[a + 4], x[a + 8], x[a + 12]);
    // after this, x[a + 4b] = u[a][b]; twiddle W16^{ab} = exp(-j*2*pi*a*b/16)
    x[1 + 4] = cmul(x[1 + 4], make_float2(c1, -s1));        // ab = 1
    x[2 + 4] = cmul(x[2 + 4], make_float2(c2, -c2));        // 2
    x[3 + 4] = cmul(x[3 + 4], make_float2(s1, -c1));        // 3
    x[1 + 8] = cmul(x[1 + 8], make_float2(c2, -c2));        // 2
    x[2 + 8] = make_float2(x[2 + 8].y, -x[2 + 8].x);        // 4: -j
    x[3 + 8] = cmul(x[3 + 8], make_float2(-c2, -c2));       // 6
    x[1 + 12] = cmul(x[1 + 12], make_float2(s1, -c1));      // 3
    x[2 + 12] = cmul(x[2 + 12], make_float2(-c2, -c2));     // 6
    x[3 + 12] = cmul(x[3 + 12], make_float2(-c1, s1));      // 9
#pragma unroll
    for (int b = 0; b < 4; ++b) dft4(x[4 * b], x[4 * b + 1], x[4 * b + 2], x[4 * b + 3]);
    // now x[4b + c] = X[b + 4c]: transpose to natural order
    float2 y[16];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) y[b + 4 * c] = x[4 * b + c];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = y[i];
}

__device__ __forceinline__ int pidx(int i) { return i + (i >> 4); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it
// would wait for the next frame's prefetched samples and the previous frame's dB stores at every
// exchange; here only LDS (lgkmcnt) has to be complete.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Exchange point of one frame.  Up to N = 1024 a frame's N/16 lanes sit in one wavefront, whose LDS
// operations execute in order: draining lgkmcnt is all the synchronisation the exchange needs (a
// wavefront-local shuffle through LDS); larger frames span the workgroup's waves and take the barrier.
template <int TPF>
__device__ __forceinline__ void frame_sync()
{
    if constexpr (TPF <= 64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else lds_barrier();
}

struct FftGeom {
    long long n_in;      // samples available
    long long frames;    // frames to compute
    long long chunk;     // consecutive frames one (workgroup, slot) pair transforms
    float *dummy;        // one spare row: slots past the last frame store there
    int hop;
};

// twiddles of one Stockham pass for butterfly index j: W_{Ns*R}^{t*k}, k = j mod Ns, t = 1..R-1.
// They depend on the lane only, so a persistent lane loads them once.
template <int N, int R>
__device__ __forceinline__ void pass_twiddles(float2 (&w)[R > 1 ? R - 1 : 1], int j, int Ns,
                                              const float2 *__restrict__ tw)
{
    const int k = j & (Ns - 1);
    const int step = N / (Ns * R);
#pragma unroll
    for (int t = 1; t < R; ++t) w[t - 1] = tw[t * k * step];
}

// One Stockham pass on registers: inputs gathered from `lds` at j + t*N/R, twiddled, R-point DFT.
// Natural-order bin of output t is (j/Ns)*Ns*R + (j mod Ns) + t*Ns.
template <int N, int R>
__device__ __forceinline__ void pass_load(float2 (&x)[R], const float2 *lds, int j,
                                          const float2 (&w)[R > 1 ? R - 1 : 1])
{
#pragma unroll
    for (int t = 0; t < R; ++t) x[t] = lds[pidx(j + t * (N / R))];
#pragma unroll
    for (int t = 1; t < R; ++t) x[t] = cmul(x[t], w[t - 1]);
    dft<R>(x);
}

// 20*log10|X| = 10*log10(|X|^2) through the hardware log2 (v_log_f32, ~1 ulp): exact zero power
// gives -inf like the reference's log10(0).
template <int N>
__device__ __forceinline__ void emit_db(float *__restrict__ row, int bin, float2 v)
{
    const float p = __fmaf_rn(v.x, v.x, v.y * v.y);
    row[(bin + N / 2) & (N - 1)] = 3.01029995663981195f * __log2f(p);
}

// N = 16 * R2 * R3 (R3 == 1: two passes).  HALF: hop == N/2, so the second half of a frame is the
// first half of the next one and stays in the lane's registers (raw int16 pairs): every input
// sample is fetched from HBM once.
//
// Persistent lanes: a workgroup holds FPB = 4096/N frame slots of N/16 lanes; each slot walks its
// own run of `chunk` consecutive frames, so the per-lane constants (window values, twiddles of
// both later passes) are loaded once, and the samples of frame k+1 are requested before frame k
// is transformed.
template <int N, int R2, int R3, bool HALF>
#ifndef SDR_FFT_MINB
#define SDR_FFT_MINB 3
#endif
__global__ void __launch_bounds__(256, SDR_FFT_MINB)
fft_db_kernel(const uint32_t *__restrict__ iq, const float *__restrict__ win,
              const float2 *__restrict__ tw, float *__restrict__ out, const FftGeom g)
{
    constexpr int TPF = N / 16;            // lanes per frame
    constexpr int FPB = 256 / TPF;         // frame slots per workgroup
    constexpr int FR_LDS = N + N / 16;     // padded float2 per frame
    constexpr int NEW = HALF ? 8 : 16;     // samples a lane fetches per frame
    constexpr int B2 = 16 / R2;            // pass-2 butterflies per lane
    constexpr int Ns3 = 16 * R2;
    constexpr int B3 = R3 > 1 ? 16 / R3 : 1;
    constexpr int NT2 = 16 * (R2 - 1);     // pass-2 twiddle table: W_{16*R2}^{t*k}, k = 0..15, t = 1..R2-1
    __shared__ float2 smem[FPB * FR_LDS];
    __shared__ float2 tw2[NT2];
    __shared__ float lwin[N];              // window table (registers are the scarcer resource here)

    const int tid = threadIdx.x;
    const int fl = tid / TPF;              // frame slot in the workgroup
    const int tl = tid - fl * TPF;         // lane within the frame
    float2 *lds = smem + fl * FR_LDS;

    const long long f0 = ((long long)blockIdx.x * FPB + fl) * g.chunk;     // first frame of this slot
    long long f_end = f0 + g.chunk;
    if (f_end > g.frames) f_end = g.frames;

    for (int i = tid; i < N; i += 256) lwin[i] = win[i];
    // pass-2 twiddles depend on (j mod 16, t) only: one small table per workgroup in LDS; the
    // pass-3 ones differ per lane and stay in registers
    for (int i = tid; i < NT2; i += 256) {
        const int k = i / (R2 - 1), t = i - k * (R2 - 1) + 1;
        tw2[i] = tw[t * k * (N / (16 * R2))];
    }
    float2 w3[B3][R3 > 1 ? R3 - 1 : 1];
    if constexpr (R3 > 1) {
#pragma unroll
        for (int i = 0; i < B3; ++i) pass_twiddles<N, R3>(w3[i], tl + i * TPF, Ns3, tw);
    }

    // raw samples: keep[] = those already in registers for the coming frame (HALF), nxt[] = fetched.
    // Loads and stores are issued unconditionally (a slot past the last frame re-reads the last
    // frame and stores to the spare row): with no branch around them hipcc can wait for the
    // prefetch with a counted vmcnt instead of draining the previous frame's stores.
    const long long last = g.frames - 1;
    uint32_t keep[8], nxt[NEW];
    {
        const uint32_t *src = iq + (f0 < last ? f0 : last) * g.hop;
        if constexpr (HALF) {
#pragma unroll
            for (int t = 0; t < 8; ++t) keep[t] = src[tl + t * TPF];
        }
#pragma unroll
        for (int t = 0; t < NEW; ++t) nxt[t] = src[tl + (t + 16 - NEW) * TPF];
    }

    // Sixteen stores after the first prefetch, as after every later one: the loop-top wait for the
    // prefetched samples can then be the same counted vmcnt(16) on the entry edge and on the back
    // edge (hipcc merges the two and would otherwise drain everything, stores included).
#pragma unroll
    for (int t = 0; t < 16; ++t) g.dummy[(tid + 256 * t) & (N - 1)] = 0.f;

    __syncthreads();                       // tables filled
    for (long long it = 0; it < g.chunk; ++it) {
        const long long f = f0 + it;
        const bool live = f < f_end;
        // ---- pass 1: radix 16 on the raw samples, Ns = 1 ---------------------------------------
        float2 x[16];
        uint32_t raw[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) raw[t] = (t < 16 - NEW) ? keep[t % 8] : nxt[t - (16 - NEW)];
        if constexpr (HALF) {
#pragma unroll
            for (int t = 0; t < 8; ++t) keep[t] = nxt[t];
        }
        {                                  // request the next frame's samples now
            const uint32_t *src = iq + (f + 1 < last ? f + 1 : last) * g.hop;
#pragma unroll
            for (int t = 0; t < NEW; ++t) nxt[t] = src[tl + (t + 16 - NEW) * TPF];
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float wv = lwin[tl + t * TPF];
            x[t] = make_float2((float)(short)(raw[t] & 0xffffu) * wv, (float)((int)raw[t] >> 16) * wv);
        }
        dft<16>(x);
#pragma unroll
        for (int t = 0; t < 16; ++t) lds[pidx(tl * 16 + t)] = x[t];
        frame_sync<TPF>();  

        float *row = live ? out + f * N : g.dummy;
        // ---- pass 2: radix R2, Ns = 16 ------------------------------------------------------
        float2 y[B2][R2];
#pragma unroll
        for (int i = 0; i < B2; ++i) {
            const int j = tl + i * TPF;
            float2 w2[R2 - 1];
#pragma unroll
            for (int t = 0; t < R2 - 1; ++t) w2[t] = tw2[(j & 15) * (R2 - 1) + t];
            pass_load<N, R2>(y[i], lds, j, w2);
        }

        if constexpr (R3 == 1) {
#pragma unroll
            for (int i = 0; i < B2; ++i) {
                const int j = tl + i * TPF;
                const int base = (j >> 4) * (16 * R2) + (j & 15);
#pragma unroll
                for (int t = 0; t < R2; ++t) emit_db<N>(row, base + t * 16, y[i][t]);
            }
            frame_sync<TPF>();                 // the slot's image is free for the next frame
        } else {
            frame_sync<TPF>();                 // every lane has gathered its pass-2 inputs
#pragma unroll
            for (int i = 0; i < B2; ++i) {
                const int j = tl + i * TPF;
                const int base = (j >> 4) * (16 * R2) + (j & 15);
#pragma unroll
                for (int t = 0; t < R2; ++t) lds[pidx(base + t * 16)] = y[i][t];
            }
            frame_sync<TPF>();  
            // ---- pass 3: radix R3, Ns = 16*R2 ---------------------------------------------
#pragma unroll
            for (int i = 0; i < B3; ++i) {
                float2 z[R3];
                const int j = tl + i * TPF;
                pass_load<N, R3>(z, lds, j, w3[i]);
                const int base = (j / Ns3) * (Ns3 * R3) + (j & (Ns3 - 1));
#pragma unroll
                for (int t = 0; t < R3; ++t) emit_db<N>(row, base + t * Ns3, z[t]);
            }
            frame_sync<TPF>();                 // the slot's image is free for the next frame
        }
    }
}

}  // namespace

struct sdr_fft_plan {
    int nfft, hop;
    int n_cu = 256;
    float *d_win = nullptr;
    float2 *d_tw = nullptr;
    float *d_dummy = nullptr;
    void *d_in = nullptr, *d_out = nullptr;
    size_t cap_in = 0, cap_out = 0;
};

namespace {

long long frame_count(const sdr_fft_plan *p, long long n)
{
    if (n < p->nfft) return 0;
    return (n - p->nfft) / p->hop + 1;
}

int launch_fft(sdr_fft_plan *p, const int16_t *d_iq, long long n_in, float *d_out,
               long long frames, hipStream_t st)
{
    if (frames == 0) return SDR_OK;
    if (((uintptr_t)d_iq) & 3 || ((uintptr_t)d_out) & 3) { sdr::set_error("misaligned pointer"); return SDR_EALIGN; }
    FftGeom g;
    g.n_in = n_in; g.frames = frames; g.hop = p->hop; g.dummy = p->d_dummy;
    const int fpb = 256 / (p->nfft / 16);
    // persistent grid: the resident workgroups of every CU; every (workgroup, slot) pair gets one run of
    // `chunk` consecutive frames
    long long groups = (long long)p->n_cu * SDR_FFT_MINB * fpb;
    if (groups > frames) groups = frames;
    g.chunk = (frames + groups - 1) / groups;
    const long long blocks = ((frames + g.chunk - 1) / g.chunk + fpb - 1) / fpb;
    const uint32_t *in = (const uint32_t *)d_iq;
    dim3 grid((unsigned)blocks);
    const bool half = (p->hop * 2 == p->nfft);
#define SDR_FFT(N, R2, R3)                                                                        \
    do {                                                                                          \
        if (half) fft_db_kernel<N, R2, R3, true><<<grid, 256, 0, st>>>(in, p->d_win, p->d_tw, d_out, g); \
        else fft_db_kernel<N, R2, R3, false><<<grid, 256, 0, st>>>(in, p->d_win, p->d_tw, d_out, g);     \
    } while (0)
    switch (p->nfft) {
    case 4096: SDR_FFT(4096, 16, 16); break;
    case 2048: SDR_FFT(2048, 16, 8); break;
    case 1024: SDR_FFT(1024, 16, 4); break;
    case 512: SDR_FFT(512, 8, 4); break;
    case 256: SDR_FFT(256, 16, 1); break;
    case 128: SDR_FFT(128, 8, 1); break;
    case 64: SDR_FFT(64, 4, 1); break;
    default: sdr::set_error("unsupported nfft %d", p->nfft); return SDR_EINVAL;
    }
#undef SDR_FFT
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sdr::hip_fail(e, "fft kernel launch");
    return SDR_OK;
}

}  // namespace

extern "C" int sdr_fft_plan_create(int nfft, int hop, const double *window, sdr_fft_plan **out)
{
    if (!out) { sdr::set_error("null argument"); return SDR_EINVAL; }
    *out = nullptr;
    if (nfft < 64 || nfft > 4096 || (nfft & (nfft - 1))) { sdr::set_error("nfft must be a power of two in 64..4096"); return SDR_EINVAL; }
    if (hop <= 0) { sdr::set_error("hop must be positive"); return SDR_EINVAL; }
    int rc = sdr::ensure_device();
    if (rc != SDR_OK) return rc;
    sdr_fft_plan *p = new sdr_fft_plan();
    p->nfft = nfft; p->hop = hop;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            p->n_cu = prop.multiProcessorCount;
    }
    // window table carries the Q11 scale and the 1/N of fftInPlace(x, 1/N): both powers of two
    std::vector<float> w(nfft);
    const double s = 1.0 / (2048.0 * (double)nfft);
    for (int i = 0; i < nfft; ++i) w[i] = (float)((window ? window[i] : 1.0) * s);
    std::vector<float2> tw(nfft);
    for (int i = 0; i < nfft; ++i) {
        const double a = -2.0 * M_PI * (double)i / (double)nfft;
        tw[i] = make_float2((float)cos(a), (float)sin(a));
    }
    hipError_t e = hipMalloc((void **)&p->d_win, sizeof(float) * nfft);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_tw, sizeof(float2) * nfft);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_dummy, sizeof(float) * nfft);
    if (e == hipSuccess) e = hipMemcpy(p->d_win, w.data(), sizeof(float) * nfft, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_tw, tw.data(), sizeof(float2) * nfft, hipMemcpyHostToDevice);
    if (e != hipSuccess) { sdr_fft_plan_destroy(p); return sdr::hip_fail(e, "fft plan upload"); }
    *out = p;
    return SDR_OK;
}

extern "C" int sdr_fft_plan_destroy(sdr_fft_plan *p)
{
    if (!p) return SDR_OK;
    if (p->d_win) (void)hipFree(p->d_win);
    if (p->d_tw) (void)hipFree(p->d_tw);
    if (p->d_dummy) (void)hipFree(p->d_dummy);
    if (p->d_in) (void)hipFree(p->d_in);
    if (p->d_out) (void)hipFree(p->d_out);
    delete p;
    return SDR_OK;
}

extern "C" size_t sdr_fft_frame_count(const sdr_fft_plan *p, size_t n_in)
{
    return p ? (size_t)frame_count(p, (long long)n_in) : 0;
}

extern "C" int sdr_fft_execute_i16_dev(sdr_fft_plan *p, const int16_t *d_iq, size_t n_in,
                                       float *d_out, size_t frame_capacity, size_t *n_frames,
                                       void *stream)
{
    if (!p) { sdr::set_error("null plan"); return SDR_EINVAL; }
    const long long frames = frame_count(p, (long long)n_in);
    if (n_frames) *n_frames = (size_t)frames;
    if ((size_t)frames > frame_capacity) { sdr::set_error("frame capacity %zu < %lld", frame_capacity, frames); return SDR_ECAPACITY; }
    if (frames == 0) return SDR_OK;
    if (!d_iq || !d_out) { sdr::set_error("null argument"); return SDR_EINVAL; }
    return launch_fft(p, d_iq, (long long)n_in, d_out, frames, (hipStream_t)stream);
}

extern "C" int sdr_fft_execute_i16_host(sdr_fft_plan *p, const int16_t *iq, size_t n_in,
                                        float *out, size_t frame_capacity, size_t *n_frames)
{
    if (!p) { sdr::set_error("null plan"); return SDR_EINVAL; }
    const long long frames = frame_count(p, (long long)n_in);
    if (n_frames) *n_frames = (size_t)frames;
    if ((size_t)frames > frame_capacity) { sdr::set_error("frame capacity %zu < %lld", frame_capacity, frames); return SDR_ECAPACITY; }
    if (frames == 0) return SDR_OK;
    if (!iq || !out) { sdr::set_error("null argument"); return SDR_EINVAL; }
    const size_t bin = n_in * 4, bout = (size_t)frames * p->nfft * sizeof(float);
    if (bin > p->cap_in) {
        if (p->d_in) (void)hipFree(p->d_in);
        p->d_in = nullptr; p->cap_in = 0;
        SDR_HIP(hipMalloc(&p->d_in, bin));
        p->cap_in = bin;
    }
    if (bout > p->cap_out) {
        if (p->d_out) (void)hipFree(p->d_out);
        p->d_out = nullptr; p->cap_out = 0;
        SDR_HIP(hipMalloc(&p->d_out, bout));
        p->cap_out = bout;
    }
    SDR_HIP(hipMemcpyAsync(p->d_in, iq, bin, hipMemcpyHostToDevice, 0));
    int rc = launch_fft(p, (const int16_t *)p->d_in, (long long)n_in, (float *)p->d_out, frames, 0);
    if (rc != SDR_OK) return rc;
    SDR_HIP(hipMemcpyAsync(out, p->d_out, bout, hipMemcpyDeviceToHost, 0));
    SDR_HIP(hipStreamSynchronize(0));
    return SDR_OK;
}

extern "C" int sdr_fft_time_dev(sdr_fft_plan *p, const int16_t *d_iq, size_t n_in, float *d_out,
                                int iters, void *stream, float *ms_per_launch)
{
    if (!p || !d_iq || !d_out || iters <= 0 || !ms_per_launch) { sdr::set_error("bad argument"); return SDR_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const long long frames = frame_count(p, (long long)n_in);
    hipEvent_t e0, e1;
    SDR_HIP(hipEventCreate(&e0));
    SDR_HIP(hipEventCreate(&e1));
    int rc = SDR_OK;
    SDR_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters && rc == SDR_OK; ++i) rc = launch_fft(p, d_iq, (long long)n_in, d_out, frames, st);
    SDR_HIP(hipEventRecord(e1, st));
    SDR_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    SDR_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = ms / (float)iters;
    return rc;
}
