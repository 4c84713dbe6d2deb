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
// A 256-lane workgroup holds 4096/N frames; the LDS exchange image is padded by one complex value per
// 16 so that the stride-16 scatter of pass 1 and the stride-N/16 gathers hit distinct banks.
// Twiddles come from a table exp(-2*pi*i*k/N) in global memory (L1/L2 resident).

#include <cmath>
#include <mutex>
#include <vector>

#include "sdr_internal.h"
#include "fft_common.h"

namespace {

// twiddles of one Stockham pass for butterfly index j: W_{Ns*R}^{t*k}, k = j mod Ns, t = 1..R-1.
// They depend on the lane only, so a persistent lane loads them once.
template <int N, int R>
__device__ __forceinline__ void pass_twiddles(cf (&w)[R > 1 ? R - 1 : 1], int j, int Ns,
                                              const cf *__restrict__ tw)
{
    const int k = j & (Ns - 1);
    const int step = N / (Ns * R);
#pragma unroll
    for (int t = 1; t < R; ++t) w[t - 1] = tw[t * k * step];
}

// One Stockham pass on registers: inputs gathered from `lds` at j + t*N/R, twiddled, R-point DFT.
// Natural-order bin of output t is (j/Ns)*Ns*R + (j mod Ns) + t*Ns.
template <int N, int R>
__device__ __forceinline__ void pass_load(cf (&x)[R], const cf *lds, int j,
                                          const cf (&w)[R > 1 ? R - 1 : 1])
{
    if constexpr (!(SDR_FFT_ABLATE & 1)) {
#pragma unroll
        for (int t = 0; t < R; ++t) x[t] = lds[pidx(j + t * (N / R))];
    }
#pragma unroll
    for (int t = 1; t < R; ++t) x[t] = cmul(x[t], w[t - 1]);
    if constexpr (!(SDR_FFT_ABLATE & 2)) dft<R>(x);
}

// 20*log10|X| = 10*log10(|X|^2) through the hardware log2 (v_log_f32, ~1 ulp): exact zero power
// gives -inf like the reference's log10(0).  Two bins at a time so that the squares and the final
// scale are packed operations too.
template <int N>
__device__ __forceinline__ void emit_db2(float *__restrict__ row, int bin0, int bin1, cf v0, cf v1)
{
    const cf s0 = v0 * v0, s1 = v1 * v1;
    cf p = {s0.x + s0.y, s1.x + s1.y};
    if constexpr (!(SDR_FFT_ABLATE & 4)) p = cf{__log2f(p.x), __log2f(p.y)} * 3.01029995663981195f;
    // rows are written once and never read here: non-temporal stores keep them from displacing the
    // input stream in L2 (N = 256: +22 %, 4096: +1..4 %); below SDR_FFT_NT_MIN a wave-store covers several
    // short rows and the partial lines make them slower instead (N = 64: 3.4x)
    if constexpr (N >= SDR_FFT_NT_MIN) {
        __builtin_nontemporal_store(p.x, &row[(bin0 + N / 2) & (N - 1)]);
        __builtin_nontemporal_store(p.y, &row[(bin1 + N / 2) & (N - 1)]);
    } else {
        row[(bin0 + N / 2) & (N - 1)] = p.x;
        row[(bin1 + N / 2) & (N - 1)] = p.y;
    }
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
              const cf *__restrict__ tw, float *__restrict__ out, const FftGeom g)
{
    constexpr int TPF = N / 16;            // lanes per frame
    constexpr int FPB = 256 / TPF;         // frame slots per workgroup
    constexpr int FR_LDS = N + N / 16;     // padded complex values per frame
    constexpr int NEW = HALF ? 8 : 16;     // samples a lane fetches per frame
    constexpr int B2 = 16 / R2;            // pass-2 butterflies per lane
    constexpr int Ns3 = 16 * R2;
    constexpr int B3 = R3 > 1 ? 16 / R3 : 1;
    constexpr int NT2 = 16 * (R2 - 1);     // pass-2 twiddle table: W_{16*R2}^{t*k}, k = 0..15, t = 1..R2-1
    __shared__ cf smem[FPB * FR_LDS];
    __shared__ cf tw2[NT2];
    __shared__ float lwin[N];              // window table (registers are the scarcer resource here)
#ifdef SDR_FFT_PADLDS                      // measurement builds: fewer resident workgroups per CU
    __shared__ float padlds[SDR_FFT_PADLDS / 4];
    if (g.hop < 0) padlds[threadIdx.x] = 1.f, out[0] = padlds[threadIdx.x ^ 1];
#endif

    const int tid = threadIdx.x;
    const int fl = tid / TPF;              // frame slot in the workgroup
    const int tl = tid - fl * TPF;         // lane within the frame
    cf *lds = smem + fl * FR_LDS;

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
    cf w3[B3][R3 > 1 ? R3 - 1 : 1];
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
        cf x[16];
        uint32_t raw[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) raw[t] = (t < 16 - NEW) ? keep[t % 8] : nxt[t - (16 - NEW)];
        if constexpr (HALF) {
#pragma unroll
            for (int t = 0; t < 8; ++t) keep[t] = nxt[t];
        }
        {                                  // request the next frame's samples now
            const uint32_t *src = iq + ((SDR_FFT_ABLATE & 16) ? (f & 1) : (f + 1 < last ? f + 1 : last)) * g.hop;
#pragma unroll
            for (int t = 0; t < NEW; ++t) nxt[t] = src[tl + (t + 16 - NEW) * TPF];
        }
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
            // two window values per register pair; each scales one sample's (I, Q) pair
            const cf wv = {lwin[tl + t * TPF], lwin[tl + (t + 1) * TPF]};
            const cf a = {(float)(short)(raw[t] & 0xffffu), (float)((int)raw[t] >> 16)};
            const cf b = {(float)(short)(raw[t + 1] & 0xffffu), (float)((int)raw[t + 1] >> 16)};
            asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(x[t]) : "v"(a), "v"(wv));
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(x[t + 1]) : "v"(b), "v"(wv));
        }
        if constexpr (!(SDR_FFT_ABLATE & 2)) dft<16>(x);
        if constexpr (!(SDR_FFT_ABLATE & 1)) {
#pragma unroll
            for (int t = 0; t < 16; ++t) lds[pidx(tl * 16 + t)] = x[t];
            frame_sync<TPF>();
        }

        float *row = (SDR_FFT_ABLATE & 8) ? out + (size_t)blockIdx.x * N : live ? out + f * N : g.dummy;
        // ---- pass 2: radix R2, Ns = 16 ------------------------------------------------------
        cf y[B2][R2];
#pragma unroll
        for (int i = 0; i < B2; ++i) {
            const int j = tl + i * TPF;
            cf w2[R2 - 1];
#pragma unroll
            for (int t = 0; t < R2 - 1; ++t) w2[t] = tw2[(j & 15) * (R2 - 1) + t];
            if constexpr (SDR_FFT_ABLATE & 1) {
#pragma unroll
                for (int t = 0; t < R2; ++t) y[i][t] = x[(i * R2 + t) & 15];
            }
            pass_load<N, R2>(y[i], lds, j, w2);
        }

        if constexpr (R3 == 1) {
#pragma unroll
            for (int i = 0; i < B2; ++i) {
                const int j = tl + i * TPF;
                const int base = (j >> 4) * (16 * R2) + (j & 15);
#pragma unroll
                for (int t = 0; t < R2; t += 2) emit_db2<N>(row, base + t * 16, base + (t + 1) * 16, y[i][t], y[i][t + 1]);
            }
            frame_sync<TPF>();                 // the slot's image is free for the next frame
        } else {
            if constexpr (!(SDR_FFT_ABLATE & 1)) {
                    frame_sync<TPF>();             // every lane has gathered its pass-2 inputs
#pragma unroll
                for (int i = 0; i < B2; ++i) {
                    const int j = tl + i * TPF;
                    const int base = (j >> 4) * (16 * R2) + (j & 15);
#pragma unroll
                    for (int t = 0; t < R2; ++t) lds[pidx(base + t * 16)] = y[i][t];
                }
                frame_sync<TPF>();
                }
            // ---- pass 3: radix R3, Ns = 16*R2 ---------------------------------------------
#pragma unroll
            for (int i = 0; i < B3; ++i) {
                cf z[R3];
                const int j = tl + i * TPF;
                if constexpr (SDR_FFT_ABLATE & 1) {
#pragma unroll
                    for (int t = 0; t < R3; ++t) z[t] = y[((i * R3 + t) / R2) % B2][(i * R3 + t) % R2];
                }
                pass_load<N, R3>(z, lds, j, w3[i]);
                const int base = (j / Ns3) * (Ns3 * R3) + (j & (Ns3 - 1));
#pragma unroll
                for (int t = 0; t < R3; t += 2) emit_db2<N>(row, base + t * Ns3, base + (t + 1) * Ns3, z[t], z[t + 1]);
            }
            if constexpr (!(SDR_FFT_ABLATE & 1)) frame_sync<TPF>();   // the slot's image is free for the next frame
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fft4096_pair_kernel: the 4096-point, hop-2048 case with TWO frames per lane.  fft_db_kernel is
// bound by neither HBM nor a saturated unit but by dependent chains: three waves per SIMD leave
// the vector unit ~60 % busy.  Here a lane transforms frames f and f+1 together --
// two independent instruction streams in one wave, one set of window values and twiddles (all in
// registers, no LDS tables), three half frames of raw samples of which the middle one serves both
// frames -- and a workgroup needs one barrier sequence per PAIR.  232 VGPRs -> two waves per SIMD,
// i.e. four frames in flight per SIMD instead of three, at half the barriers per frame.
// LDS: two frame images (70 KB), two workgroups per CU.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
fft4096_pair_kernel(const uint32_t *__restrict__ iq, const float *__restrict__ win,
                    const cf *__restrict__ tw, float *__restrict__ out, const FftGeom g)
{
    constexpr int N = 4096, TPF = 256;
    constexpr int FR_LDS = N + N / 16;     // padded complex values per frame image
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
    cf *img0 = (cf *)dyn_smem, *img1 = img0 + FR_LDS;
    const int tl = threadIdx.x;

    cf wv[8];                              // window values, two per register pair
#pragma unroll
    for (int t = 0; t < 16; t += 2) wv[t >> 1] = cf{win[tl + t * TPF], win[tl + (t + 1) * TPF]};
    cf w2[15], w3[15];
    pass_twiddles<N, 16>(w2, tl, 16, tw);
    pass_twiddles<N, 16>(w3, tl, 256, tw);
    const int base2 = (tl >> 4) * 256 + (tl & 15);

    const long long f0 = (long long)blockIdx.x * g.chunk;          // chunk is even
    long long f_end = f0 + g.chunk;
    if (f_end > g.frames) f_end = g.frames;
    const long long half_last = g.frames;                           // last half frame of the stream
    auto fetch = [&](long long h, uint32_t (&r)[8]) {               // half frame h of the stream
        h = h < half_last ? h : half_last;
        const uint32_t *src = iq + h * g.hop;
#pragma unroll
        for (int t = 0; t < 8; ++t) r[t] = src[tl + t * TPF];
    };
    auto load_frame = [&](cf (&x)[16], const uint32_t (&lo)[8], const uint32_t (&hi)[8]) {
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
            const uint32_t r0 = t < 8 ? lo[t] : hi[t - 8], r1 = t < 8 ? lo[t + 1] : hi[t - 7];
            const cf a = {(float)(short)(r0 & 0xffffu), (float)((int)r0 >> 16)};
            const cf b = {(float)(short)(r1 & 0xffffu), (float)((int)r1 >> 16)};
            asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(x[t]) : "v"(a), "v"(wv[t >> 1]));
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(x[t + 1]) : "v"(b), "v"(wv[t >> 1]));
        }
    };

    // half frames 2i, 2i+1, 2i+2 of this workgroup's run for pair i; the next pair's two new ones in flight
    uint32_t ha[8], hb[8], hc[8], na[8], nb[8];
    fetch(f0, ha);
    fetch(f0 + 1, hb);
    fetch(f0 + 2, hc);
    for (long long fa = f0; fa < f0 + g.chunk; fa += 2) {
        float *row_a = fa < f_end ? out + fa * N : g.dummy;
        float *row_b = fa + 1 < f_end ? out + (fa + 1) * N : g.dummy;
        fetch(fa + 3, na);
        fetch(fa + 4, nb);
        // ---- pass 1 of both frames --------------------------------------------------------------
        {
            cf x0[16], x1[16];
            load_frame(x0, ha, hb);
            load_frame(x1, hb, hc);
            dft<16>(x0);
            dft<16>(x1);
#pragma unroll
            for (int t = 0; t < 16; ++t) { img0[pidx(tl * 16 + t)] = x0[t]; img1[pidx(tl * 16 + t)] = x1[t]; }
        }
        lds_barrier();
        // ---- pass 2 ------------------------------------------------------------------------------
        {
            cf y0[16], y1[16];
            pass_load<N, 16>(y0, img0, tl, w2);
            pass_load<N, 16>(y1, img1, tl, w2);
            lds_barrier();                 // every lane has gathered its inputs
#pragma unroll
            for (int t = 0; t < 16; ++t) { img0[pidx(base2 + t * 16)] = y0[t]; img1[pidx(base2 + t * 16)] = y1[t]; }
        }
        lds_barrier();
        // ---- pass 3, |X|^2 -> dB, fft-shifted rows ---------------------------------------------------
        {
            cf z0[16], z1[16];
            pass_load<N, 16>(z0, img0, tl, w3);
            pass_load<N, 16>(z1, img1, tl, w3);
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                emit_db2<N>(row_a, tl + t * 256, tl + (t + 1) * 256, z0[t], z0[t + 1]);
                emit_db2<N>(row_b, tl + t * 256, tl + (t + 1) * 256, z1[t], z1[t + 1]);
            }
        }
        lds_barrier();                     // the images are free for the next pair
#pragma unroll
        for (int t = 0; t < 8; ++t) { ha[t] = hc[t]; hb[t] = na[t]; hc[t] = nb[t]; }
    }
}

// ---------------------------------------------------------------------------------------------
// Frames longer than 4096 points (the reference transforms one 131072-sample capture block,
// bladerf_main_v1.cpp:139): Stockham autosort passes of radix 4 (one radix-2 pass at the end when
// log2 N is odd) through two complex scratch arrays in HBM.  The first pass reads the int16 samples
// and applies the window (which carries 2^-11 / N), the last one writes the fft-shifted dB row.
// Twiddles come from sincospif of an exact dyadic fraction.  One lane per butterfly, grid.y = frame.
// This path is for drop-in completeness (a capture block is ~2 ms of signal), not for the
// waterfall benchmark, whose frames fit the register kernels above.
// ---------------------------------------------------------------------------------------------
template <int R, bool FIRST, bool LAST>
__global__ void __launch_bounds__(256)
fft_big_pass_kernel(const uint32_t *__restrict__ iq, const float *__restrict__ win,
                    const cf *__restrict__ src, cf *__restrict__ dst, float *__restrict__ out,
                    int n, int ns, long long hop, long long frame0)
{
    const int j = blockIdx.x * 256 + threadIdx.x;            // butterfly index, 0 .. n/R - 1
    if (j >= n / R) return;
    const long long f = blockIdx.y;                          // frame within this launch
    const int k = j & (ns - 1);
    cf x[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const int idx = j + t * (n / R);
        if constexpr (FIRST) {
            const uint32_t w = iq[(frame0 + f) * hop + idx];
            const float wv = win[idx];
            x[t] = cf{(float)(short)(w & 0xffffu) * wv, (float)((int)w >> 16) * wv};
        } else {
            x[t] = src[f * n + idx];
        }
    }
#pragma unroll
    for (int t = 1; t < R; ++t) {
        // exp(-2*pi*i * t*k / (R*ns)): t*k < n <= 2^20 and R*ns is a power of two, so the argument is exact
        float sn, cs;
        sincospif(-2.0f * (float)(t * k) / (float)(R * ns), &sn, &cs);
        x[t] = cf{x[t].x * cs - x[t].y * sn, x[t].x * sn + x[t].y * cs};
    }
    dft<R>(x);
    const int o = (j - k) * R + k;                           // (j / ns) * ns * R + k
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const int bin = o + t * ns;
        if constexpr (LAST) {
            const float p = __fmaf_rn(x[t].x, x[t].x, x[t].y * x[t].y);
            out[(frame0 + f) * n + ((bin + n / 2) & (n - 1))] = 3.01029995663981195f * __log2f(p);
        } else {
            dst[f * n + bin] = x[t];
        }
    }
}

}  // namespace

namespace sdr {
#ifdef SDR_MEASUREMENT_BUILD
int measurement_flags_fft() { return SDR_FFT_ABLATE ? SDR_FFT_ABLATE : -1; }
#else
int measurement_flags_fft() { return 0; }
#endif
}  // namespace sdr

// ---------------------------------------------------------------------------------------------
// Scratch of the allocation-free entry point sdr_fft_execute_i16_noalloc_dev: the two ping-pong
// arrays of the multi-pass transform for frames of up to kStaticFftPoints points, in a file-scope
// device array instead of hipMalloc'ed plan scratch (2 x 131072 x 8 B = 2 MiB, held by every GPU the
// library runs on until the process ends).  In host code the array's NAME is only a stand-in, not a
// device address: the address comes from hipGetSymbolAddress, once per device with that device
// current (each GPU has its own copy at its own address), and everything host-side -- kernel
// arguments included -- uses that pointer.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr size_t kStaticFftPoints = 131072;      // the reference's capture block, bladerf_main_v1.cpp:139
constexpr int kMaxDevices = 64;
}  // namespace
static __device__ __align__(256) unsigned char g_fft_static_scratch[2 * kStaticFftPoints * sizeof(cf)];
namespace {
cf *g_fft_scratch_ptr[kMaxDevices] = {};
std::mutex g_fft_scratch_mtx;
}  // namespace

namespace sdr {
// Resolve the scratch array's address on the CURRENT device (idempotent).  Called by sdr_init() for the
// device it selects and by the entry point on its first call on a device.
int fft_static_scratch_init(void **ptr_out)
{
    int dev = -1;
    SDR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) { set_error("device ordinal %d beyond %d", dev, kMaxDevices - 1); return SDR_EINVAL; }
    std::lock_guard<std::mutex> lock(g_fft_scratch_mtx);
    if (!g_fft_scratch_ptr[dev]) {
        void *ptr = nullptr;
        SDR_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_fft_static_scratch)));
        if (!ptr) { set_error("hipGetSymbolAddress returned a null scratch address"); return SDR_EHIP; }
        g_fft_scratch_ptr[dev] = (cf *)ptr;
    }
    if (ptr_out) *ptr_out = g_fft_scratch_ptr[dev];
    return SDR_OK;
}
}  // namespace sdr

struct sdr_fft_plan {
    int nfft, hop;
    int n_cu = 256;
    int device = 0;
    bool lds_opt_in = false;        // fft4096_pair_kernel's dynamic LDS limit has been raised
    int variant = 0;                // sdr_fft_plan_set_variant
    float *d_win = nullptr;
    float2 *d_tw = nullptr;
    float *d_dummy = nullptr;
    void *d_in = nullptr, *d_out = nullptr;
    size_t cap_in = 0, cap_out = 0;
    // frames longer than 4096 points: two complex scratch arrays, grown on demand
    cf *d_scr[2] = {nullptr, nullptr};
    size_t cap_scr = 0;             // complex values per array
};

namespace {

long long frame_count(const sdr_fft_plan *p, long long n)
{
    if (n < p->nfft) return 0;
    return (n - p->nfft) / p->hop + 1;
}

// the multi-pass transform of `frames` frames, `per` frames per round, through two scratch arrays of
// per * nfft complex values each
int run_fft_big(const sdr_fft_plan *p, const int16_t *d_iq, float *d_out, long long frames, long long per,
                cf *scr0, cf *scr1, hipStream_t st)
{
    const int n = p->nfft;
    cf *scr[2] = {scr0, scr1};
    const uint32_t *in = (const uint32_t *)d_iq;
    int log2n = 0;
    while ((1 << log2n) < n) ++log2n;
    const int n4 = log2n / 2, n2 = log2n & 1;                // radix-4 passes, then one radix-2 pass
    for (long long f0 = 0; f0 < frames; f0 += per) {
        const unsigned nf = (unsigned)(frames - f0 < per ? frames - f0 : per);
        int ns = 1, cur = 0;
        for (int ps = 0; ps < n4 + n2; ++ps) {
            const bool first = ps == 0, last = ps == n4 + n2 - 1;
            const int R = ps < n4 ? 4 : 2;
            const dim3 grid((unsigned)((n / R + 255) / 256), nf);
            const cf *src = scr[cur];
            cf *dst = scr[cur ^ 1];
#define SDR_BIG(RR, FI, LA) fft_big_pass_kernel<RR, FI, LA><<<grid, 256, 0, st>>>(in, p->d_win, src, dst, d_out, n, ns, (long long)p->hop, f0)
            if (R == 4) { if (first) SDR_BIG(4, true, false); else if (last) SDR_BIG(4, false, true); else SDR_BIG(4, false, false); }
            else { if (last) SDR_BIG(2, false, true); else SDR_BIG(2, false, false); }
#undef SDR_BIG
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return sdr::hip_fail(e, "fft pass launch");
            ns *= R;
            cur ^= 1;
        }
    }
    return SDR_OK;
}

int launch_fft_big(sdr_fft_plan *p, const int16_t *d_iq, float *d_out, long long frames, hipStream_t st)
{
    const int n = p->nfft;
    // frames per launch: scratch of at most 2 x 256 MB
    long long per = (long long)(1 << 25) / n;
    if (per < 1) per = 1;
    if (per > frames) per = frames;
    if (per > 65535) per = 65535;
    const size_t need = (size_t)per * n;
    if (need > p->cap_scr) {
        for (int i = 0; i < 2; ++i) { if (p->d_scr[i]) (void)hipFree(p->d_scr[i]); p->d_scr[i] = nullptr; }
        p->cap_scr = 0;
        SDR_HIP(sdr::dev_malloc((void **)&p->d_scr[0], need * sizeof(cf)));
        SDR_HIP(sdr::dev_malloc((void **)&p->d_scr[1], need * sizeof(cf)));
        p->cap_scr = need;
    }
    return run_fft_big(p, d_iq, d_out, frames, per, p->d_scr[0], p->d_scr[1], st);
}

// static_scr != nullptr: the allocation-free entry point -- long frames take their two scratch arrays
// from it (2 x kStaticFftPoints complex values) instead of the plan's heap scratch
int launch_fft(sdr_fft_plan *p, const int16_t *d_iq, long long n_in, float *d_out,
               long long frames, hipStream_t st, cf *static_scr = nullptr)
{
    if (frames == 0) return SDR_OK;
    if (((uintptr_t)d_iq) & 3 || ((uintptr_t)d_out) & 3) { sdr::set_error("misaligned pointer"); return SDR_EALIGN; }
    if (int rc = sdr::same_device(p->device)) return rc;
    if (p->nfft > 4096 && static_scr) {
        long long per = (long long)kStaticFftPoints / p->nfft;          // >= 1: the entry point checked nfft
        if (per > frames) per = frames;
        return run_fft_big(p, d_iq, d_out, frames, per, static_scr, static_scr + kStaticFftPoints, st);
    }
    if (p->nfft > 4096) return launch_fft_big(p, d_iq, d_out, frames, st);
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
        if (half) fft_db_kernel<N, R2, R3, true><<<grid, 256, 0, st>>>(in, p->d_win, (const cf *)p->d_tw, d_out, g); \
        else fft_db_kernel<N, R2, R3, false><<<grid, 256, 0, st>>>(in, p->d_win, (const cf *)p->d_tw, d_out, g);     \
    } while (0)
    if (p->nfft == 4096 && half && frames >= 4LL * p->n_cu && (p->variant == 0 || p->variant >= 3)) {
        // one frame per wavefront (fft_wave.hip): one persistent workgroup per CU of four (variant 3) or
        // eight waves
        return sdr::launch_fft4096_wave(in, p->d_win, (const cf *)p->d_tw, d_out, g, p->n_cu, p->variant != 3, st);
    }
    if (p->nfft == 4096 && half && frames >= 4LL * p->n_cu && p->variant != 1) {
        // two frames per lane, two 256-lane workgroups per CU
        FftGeom gp = g;
        const long long wgs = 2LL * p->n_cu;
        gp.chunk = (frames + wgs - 1) / wgs;
        gp.chunk += gp.chunk & 1;
        const unsigned nb = (unsigned)((frames + gp.chunk - 1) / gp.chunk);
        const size_t lds = 2 * (size_t)(4096 + 256) * sizeof(float) * 2;
        if (!p->lds_opt_in) {
            hipError_t ea = hipFuncSetAttribute((const void *)fft4096_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (ea != hipSuccess) return sdr::hip_fail(ea, "fft kernel attribute");
            p->lds_opt_in = true;
        }
        fft4096_pair_kernel<<<nb, 256, lds, st>>>(in, p->d_win, (const cf *)p->d_tw, d_out, gp);
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) return sdr::hip_fail(e2, "fft kernel launch");
        return SDR_OK;
    }
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
    if (nfft < 64 || nfft > (1 << 20) || (nfft & (nfft - 1))) { sdr::set_error("nfft must be a power of two in 64..1048576"); return SDR_EINVAL; }
    if (hop <= 0) { sdr::set_error("hop must be positive"); return SDR_EINVAL; }
    int rc = sdr::ensure_device();
    if (rc != SDR_OK) return rc;
    sdr_fft_plan *p = new sdr_fft_plan();
    p->nfft = nfft; p->hop = hop;
    {
        hipDeviceProp_t prop;
        if (hipGetDevice(&p->device) == hipSuccess && hipGetDeviceProperties(&prop, p->device) == hipSuccess)
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
    hipError_t e = sdr::dev_malloc((void **)&p->d_win, sizeof(float) * nfft);
    if (e == hipSuccess) e = sdr::dev_malloc((void **)&p->d_tw, sizeof(float2) * nfft);
    if (e == hipSuccess) e = sdr::dev_malloc((void **)&p->d_dummy, sizeof(float) * nfft);
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
    for (int i = 0; i < 2; ++i) if (p->d_scr[i]) (void)hipFree(p->d_scr[i]);
    if (p->d_in) (void)hipFree(p->d_in);
    if (p->d_out) (void)hipFree(p->d_out);
    delete p;
    return SDR_OK;
}

extern "C" int sdr_fft_plan_set_variant(sdr_fft_plan *p, int variant)
{
    if (!p || variant < 0 || variant > 4) { sdr::set_error("variant must be 0..4"); return SDR_EINVAL; }
    p->variant = variant;
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

// The allocation-free variant (no counterpart in the reference; see sdr_hip.h): same result as
// sdr_fft_execute_i16_dev, no hipMalloc / hipFree on any path.  Frames longer than 4096 points run the
// same multi-pass kernels through the static scratch array, kStaticFftPoints / nfft frames per round.
extern "C" int sdr_fft_execute_i16_noalloc_dev(sdr_fft_plan *p, const int16_t *d_iq, size_t n_in,
                                               float *d_out, size_t frame_capacity, size_t *n_frames,
                                               void *stream)
{
    if (!p) { sdr::set_error("null plan"); return SDR_EINVAL; }
    if ((size_t)p->nfft > kStaticFftPoints) {
        sdr::set_error("sdr_fft_execute_i16_noalloc_dev: nfft %d exceeds the static scratch limit of %zu points "
                       "(use sdr_fft_execute_i16_dev, which allocates plan scratch)", p->nfft, kStaticFftPoints);
        return SDR_EINVAL;
    }
    const long long frames = frame_count(p, (long long)n_in);
    if (n_frames) *n_frames = (size_t)frames;
    if ((size_t)frames > frame_capacity) { sdr::set_error("frame capacity %zu < %lld", frame_capacity, frames); return SDR_ECAPACITY; }
    if (frames == 0) return SDR_OK;
    if (!d_iq || !d_out) { sdr::set_error("null argument"); return SDR_EINVAL; }
    if (int rc = sdr::same_device(p->device)) return rc;
    void *scr = nullptr;
    if (int rc = sdr::fft_static_scratch_init(&scr)) return rc;       // a table look-up after the first call
    return launch_fft(p, d_iq, (long long)n_in, d_out, frames, (hipStream_t)stream, (cf *)scr);
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
    if (int rc = sdr::same_device(p->device)) return rc;
    const size_t bin = n_in * 4, bout = (size_t)frames * p->nfft * sizeof(float);
    if (bin > p->cap_in) {
        if (p->d_in) (void)hipFree(p->d_in);
        p->d_in = nullptr; p->cap_in = 0;
        SDR_HIP(sdr::dev_malloc(&p->d_in, bin));
        p->cap_in = bin;
    }
    if (bout > p->cap_out) {
        if (p->d_out) (void)hipFree(p->d_out);
        p->d_out = nullptr; p->cap_out = 0;
        SDR_HIP(sdr::dev_malloc(&p->d_out, bout));
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
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = SDR_OK;
    float ms = 0.f;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, st);
    for (int i = 0; e == hipSuccess && i < iters && rc == SDR_OK; ++i) rc = launch_fft(p, d_iq, (long long)n_in, d_out, frames, st);
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) return sdr::hip_fail(e, "timing events");
    *ms_per_launch = ms / (float)iters;
    return rc;
}
