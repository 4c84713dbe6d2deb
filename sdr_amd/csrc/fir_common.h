// fir_common.h -- shared by the FIR translation units: the launch geometry the kernels take (FirGeom), the plan
// (sdr_fir_plan), the device helpers every FIR kernel uses (discriminator, output stores, the lane window walk)
// and the launchers each kernel file exports to fir_plan.hip.
//   fir_tiles.hip     the vector-unit kernels (fir_raw / fir_stream / fir_planar / fir_ksplit) and fir_mfma_kernel (float32 matrix)
//   fir_mfma_i8.hip   the int8 matrix kernel (fir_mfma_i8_kernel)
//   fir_plan.hip      plan creation (geometry search, cost models, tables), launch set-up, the C ABI
#ifndef SDR_FIR_COMMON_H_
#define SDR_FIR_COMMON_H_

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "sdr_internal.h"

namespace sdr {

struct FirGeom {
    long long in_stride;    // samples between channels
    long long out_stride;   // output elements between channels
    long long lo_valid;     // samples outside [lo_valid, hi_valid) read as zero
    long long hi_valid;
    long long tile0;        // first output of tile 0 (multiple of 4)
    long long out_begin;    // outputs written: [out_begin, out_end); out[0] holds out_begin
    long long out_end;
    long long store_from;   // store_outputs skips outputs below this (= out_begin, or past the start-up outputs
                            // that fir_mfma_i8_kernel recomputes with float32 taps)
    double mix_fr;          // NCO frequency, cycles per sample (0 = off)
    int D;
    int T;
    int Tback;              // roundup4(T-1)
    int ngran;              // 16-byte granules in one lane's (padded) window
    int q;                  // lane stride in sample granules
    int pad;                // 1: one pad granule after every q sample granules
    unsigned magic;         // ceil(2^32/(q+1)): staging division
    unsigned magic_q;       // ceil(2^32/q): walker pad offset
    int s_gran;             // sample granules staged per tile
    int a_gran;             // LDS granules per tile (pads included, rounded up to NT)
    int n_inst;             // a_gran / NT
    int mode;               // sdr_mode
    int swap_iq;            // 1: the first component of an input pair is Q (SDR_IN_I16_QI)
    float scale;            // discriminator scale
    int per_tile;           // outputs a tile emits: NT*R, or (NT-1)*R with the discriminator (the
                            // last lane's y only serves as its neighbour's successor)
    int tiles_per_ch;
    int tiles_per_xcd;      // ceil(tiles_per_ch / 8)
    int total_tiles;
    int ph_off;             // mixer: offset (floats) of the phasor tables behind the tap table
    int n_ph_hi;            // mixer: entries of the coarse table (tile offsets 0, 64, 128, ...)
    // per-channel mixers (channelizer): channel c of the launch is plan channel chan_first + c; its NCO
    // frequency is chan_fr[..] and its phasor tables start at chan_tab + (..) * chan_tab_stride float2
    const double *chan_fr;  // nullptr: one mixer (mix_fr) for every channel
    const float2 *chan_tab; // float-tile kernel: phasor tables; complex-tap matrix kernel: chan_e, one float2
                            // per channel, exp(j*2*pi*fr*D) (what one output step turns the mixer by)
    int chan_first;
    int chan_bcast;         // 1: every channel of the launch uses table entry chan_first (one mixer for all: sdr_fir_desc.mix_fr)
    int chan_tab_stride;
    int zfix;               // fir_mfma_i8_kernel, discriminator: some tap is below the fixed-point resolution (see i8_zero_fix)
};

}  // namespace sdr
using sdr::FirGeom;

// =============================================================================================
// the plan (host side; built by fir_plan.hip, read by the launchers)
// =============================================================================================
struct sdr_fir_plan {
    int input, mode, D, T, R, NT;
    int count_rule = SDR_COUNT_FLOOR;   // sdr_count_rule
    bool raw;               // raw-tile geometry (4 samples per granule): fir_raw_kernel or fir_planar_kernel
    bool rawf = false;      // fir_raw_kernel on FLOAT tiles (complex / real float input without mixer): raw is set too
    bool planar = false;    // fir_planar_kernel (float planes, convert once)
    bool mfma = false;      // fir_mfma_kernel (float planes, Toeplitz GEMM on the matrix cores)
    float4 *d_atab = nullptr;   // its A-operand table
    bool ksplit = false;    // fir_ksplit_kernel (real input, short streams: taps split over the four waves of a workgroup)
    bool mfma_i8 = false;   // fir_mfma_i8_kernel (byte-split Toeplitz GEMM on the int8 matrix pipe, persistent)
    int cpx = 0;            // ... with complex taps per channel (the channelizer's mixers folded into the taps): 1 = the
                            // output's component in the matrix product's columns, 2 = in its rows (even D; fir_mfma_i8.hip)
    float2 *d_chan_e = nullptr;   // its per-channel output-step phasors exp(j*2*pi*fr*D)
    uint4 *d_atab8 = nullptr;   // its A-operand table (tap digits)
    uint4 *d_btab8 = nullptr;   // 128 * row sums of the digits, per matrix wave
    // Tile counters of its dynamic hand-out (i8_draw): a ring of kQueueBlocks blocks of kI8QueueWords, one
    // block per LAUNCH, taken in turn, so that launches which may run at the same time never share one (the
    // last workgroup of a launch leaves its block zeroed and writes the launch's generation to q_done, a
    // word of pinned host memory per block).  A block is handed out again when its last launch has
    // reported done, or -- still pending -- to a launch on the same ordinary stream handle, which runs
    // behind it in stream order.  Not by handle alone: hipStreamPerThread names a different stream in every
    // host thread, so for that handle (and hipStreamLegacy) only "done" counts.  No block free: static
    // order for that launch.  Allocated at the first launch that qualifies (long launches only).
    mutable unsigned *d_i8q = nullptr;
    mutable unsigned *h_q_done = nullptr;       // pinned, mapped: [kQueueBlocks]
    static constexpr int kQueueBlocks = 64;
    mutable std::mutex q_mu;
    mutable hipStream_t q_handle[kQueueBlocks] = {};
    mutable unsigned q_gen[kQueueBlocks] = {};  // generation of the block's last launch (0: never used)
    mutable unsigned q_next_gen = 0;
    mutable int q_ring = 0;
    mutable bool q_failed = false;              // the allocation failed once: static order from then on
    float *d_ftaps = nullptr;   // float32 taps (Q11 folded) for i8_startup_fm_output / i8_zero_fix (discriminator plans)
    bool zfix = false;          // a non-zero tap rounds to zero in the fixed point: exact-zero outputs are looked at again (i8_zero_fix)
    int plane_bytes = 0;
    float yscale = 1.f;         // its output scale 2^-(F+11)
    sdr_fir_plan *tiny = nullptr;   // automatic choice only: the VALU plan that takes streams shorter than 4 samples
    double scale, mix_fr;
    int spg;                // samples per LDS granule: 4 (raw) / 2 (stream)
    int Tback, Lw4, ngran, q, pad, s_gran, a_gran, n_inst, lds_bytes;
    int stream_blocks = 0;  // persistent grid: CUs x resident workgroups per CU
    int ph_off = 0, n_ph_hi = 0;   // mixer phasor tables behind the tap table (float-tile kernel)
    float *d_gtab = nullptr;
    // channelizer: per-channel NCO frequencies (double) and phasor tables (float2, tab_stride per channel)
    int n_chan_mix = 0;
    bool chan_bcast = false;    // one mixer (sdr_fir_desc.mix_fr) kept as a one-entry channel table that every channel uses
    double *d_chan_fr = nullptr;
    float2 *d_chan_tab = nullptr;
    int chan_tab_stride = 0;
    // staging buffers for the *_host entry point
    void *d_in = nullptr, *d_out = nullptr;
    size_t cap_in = 0, cap_out = 0;
    int device = 0;
    // the kernel's dynamic LDS limit has been raised (set after the first successful launch; execute
    // calls of one plan may run on several threads, and a repeated opt-in is harmless)
    mutable std::atomic<bool> lds_opt_in{false};
};

namespace sdr {
namespace fir {

// the dynamic LDS limit is a property of the kernel, not of the plan: raise it to the whole CU's
// 160 KB so that plans with different tiles can share a kernel in any order
constexpr int kLdsPerCu = 160 * 1024;

constexpr int kSlotsSmall = 12, kSlots = 32;   // fir_stream_kernel: granules one lane stages per tile (prefetch registers)
constexpr int kSlotsReal = 20;                 // real tiles: 16 bytes per slot, D*R/4 slots per lane
constexpr int kMfmaAhead = 4;      // fir_mfma_kernel: operand prefetch distance, in k-chunks of 16
constexpr int kI8MaxSteps = 12;    // fir_mfma_i8_kernel: k steps of 64: windows up to 768 samples (192 registers of tap digits)
constexpr int kI8PartSlots = 66;   // float4 slots of one column block's outputs (2 components x 32, padded)
constexpr int kI8QueueWords = 32 * 9;   // one block of tile counters (see i8_draw)

// Raise the kernel's dynamic LDS limit when the plan's tile needs more than the default 64 KB (once
// per plan; the flag is set only after a successful opt-in and launch).  A failure is reported by
// name instead of surfacing as a generic launch error.
#define SDR_OPT_IN(...)                                                                          \
    do {                                                                                         \
        if (p->lds_bytes > 64 * 1024 && !p->lds_opt_in.load(std::memory_order_relaxed)) {        \
            const hipError_t ea_ = hipFuncSetAttribute((const void *)(__VA_ARGS__),              \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsPerCu); \
            if (ea_ != hipSuccess) { *what = "hipFuncSetAttribute(MaxDynamicSharedMemorySize)"; return ea_; } \
        }                                                                                        \
    } while (0)

// Launchers (one per kernel file).  `in` is the virtual pointer to stream sample 0 of channel 0; *what names the
// step that failed.  fir_plan.hip's launch() fills the geometry and picks the launcher by the plan's kernel.
hipError_t launch_tile_kernels(const sdr_fir_plan *p, hipStream_t st, const void *in, void *out, const FirGeom &g,
                               unsigned tiles, unsigned n_ch, const char **what);
hipError_t launch_mfma_i8(const sdr_fir_plan *p, hipStream_t st, const void *in, void *out, const FirGeom &g,
                          const char **what);

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
// kernel-side input kinds (Q-first int16 is IN_I16 with FirGeom::swap_iq)
enum { IN_I16 = SDR_IN_I16_IQ, IN_CF32 = SDR_IN_CF32, IN_REAL = SDR_IN_F32_REAL, IN_I8 = SDR_IN_I8_IQ };

// table entry of launch channel ch (channelizer plans)
__device__ __forceinline__ long long chan_index(const FirGeom &g, long long ch) { return g.chan_bcast ? g.chan_first : g.chan_first + ch; }

// LDS granule that holds sample granule sg
__device__ __forceinline__ int lds_granule(const FirGeom &g, int sg)
{
    return g.pad ? sg + (int)__umulhi((unsigned)sg, g.magic_q) : sg;
}

// sample granule held by LDS granule a (pad granules duplicate their left neighbour)
__device__ __forceinline__ int sample_granule(const FirGeom &g, int a)
{
    if (!g.pad) return a;
    const int blk = (int)__umulhi((unsigned)a, g.magic);        // a / (q+1)
    const int rem = a - blk * (g.q + 1);
    return blk * g.q + (rem < g.q ? rem : g.q - 1);
}

// Tile of workgroup blockIdx.x.  Consecutive workgroups go to the 8 XCDs in turn; giving XCD x the
// x-th contiguous eighth of the channel's tiles (instead of every eighth tile) lets each L2 and
// its memory channels stream one contiguous region: +2-5 % on a bare read stream
// (tools/ubench/stream_read.hip).  The grid is padded to 8 * tiles_per_xcd; surplus workgroups exit.
__device__ __forceinline__ int xcd_tile(const FirGeom &g)
{
    return (int)(blockIdx.x & 7u) * g.tiles_per_xcd + (int)(blockIdx.x >> 3);
}

__device__ __forceinline__ float2 wave_sum(float2 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v.x += __shfl_xor(v.x, off, 64);
        v.y += __shfl_xor(v.y, off, 64);
    }
    return v;
}

__device__ __forceinline__ float lo16(uint32_t w) { return (float)(short)(w & 0xffffu); }
__device__ __forceinline__ float hi16(uint32_t w) { return (float)((int)w >> 16); }

// polar_discriminator, bladerf_apt_demod.cpp:419-421: tmp = y1 * conj(y0); scale*atan2(im, re).
// Written as libstdc++ evaluates the product, (a*c - b*e, a*e + b*c) with e = -d, so that a
// zero y0 produces the same signed zeros (and therefore the same 0 / +-pi) as the reference.
// atan2 for finite arguments, ~3 ulp (|err| < 7e-7 rad, i.e. 1.1e-7 of a turn), signed zeros as
// IEEE atan2 (atan2(+-0, -0) = +-pi, atan2(+-0, +0) = +-0).  About a third of the instructions of
// the library call, which matters because the walk leaves the kernel VALU-bound.
// Reduction to [0, 1] by swapping, one Newton step on v_rcp_f32, odd minimax polynomial.
__device__ __forceinline__ float atan2_finite(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float sc = (mx < 1e-30f) ? 1.8446744e19f : 1.0f;     // keep v_rcp_f32 out of the denormals
    mx *= sc;
    mn *= sc;
    const float r = __builtin_amdgcn_rcpf(mx);
    float s = mn * r;
    s = __fmaf_rn(__fmaf_rn(-mx, s, mn), r, s);           // refine mn/mx
    s = (mx == 0.0f) ? 0.0f : s;
    const float t = s * s;
    float u = 0.00282363896258175373077393f;
    u = __fmaf_rn(u, t, -0.0159569028764963150024414f);
    u = __fmaf_rn(u, t, 0.0425049886107444763183594f);
    u = __fmaf_rn(u, t, -0.0748900920152664184570312f);
    u = __fmaf_rn(u, t, 0.106347933411598205566406f);
    u = __fmaf_rn(u, t, -0.142027363181114196777344f);
    u = __fmaf_rn(u, t, 0.199926957488059997558594f);
    u = __fmaf_rn(u, t, -0.333331018686294555664062f);
    float a = __fmaf_rn(s * t, u, s);                     // atan(mn/mx) in [0, pi/4]
    a = (ay > ax) ? 1.57079632679489661923f - a : a;
    a = (__float_as_uint(x) >> 31) ? 3.14159265358979323846f - a : a;
    return __uint_as_float(__float_as_uint(a) | (__float_as_uint(y) & 0x80000000u));
}

__device__ __forceinline__ float discriminate(float2 y1, float2 y0, float scale)
{
    const float a = y1.x, b = y1.y, c = y0.x, e = -y0.y;
    const float re = __fmaf_rn(a, c, -(b * e));
    const float im = __fmaf_rn(a, e, b * c);
    return scale * atan2_finite(im, re);
}

// Outputs are written once and never read by this kernel: non-temporal stores (the "nt" policy)
// keep them from displacing the input stream in L2; +3 % on a bare stream of the same bytes
// (tools/ubench/stream_read.hip).
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // MFMA accumulators / operands
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(float *p, float a) { __builtin_nontemporal_store(a, p); }
__device__ __forceinline__ void nt_store(float *p, float a, float b)
{
    v2f v = {a, b};
    __builtin_nontemporal_store(v, (v2f *)p);
}
__device__ __forceinline__ void nt_store(float *p, float a, float b, float c, float d)
{
    v4f v = {a, b, c, d};
    __builtin_nontemporal_store(v, (v4f *)p);
}

template <int R>
__device__ __forceinline__ void store_values(const FirGeom &g, void *__restrict__ outv, long long ch, long long m0,
                                             const float (&v)[R], long long out_end);

template <int R>
__device__ __forceinline__ void store_outputs(const FirGeom &g, void *__restrict__ outv,
                                              long long ch, long long m0, const float2 (&y)[R],
                                              float2 ynext, long long tile_end)
{
    const long long out_end = tile_end < g.out_end ? tile_end : g.out_end;   // this tile's share ends here
    if (g.mode == SDR_MODE_IQ) {
        float2 *out = (float2 *)outv + ch * g.out_stride;
        float2 *dst = out + (m0 - g.out_begin);
        // two outputs per 16-byte store where the lane's run is whole and aligned
        if (R >= 2 && m0 >= g.store_from && m0 + R <= out_end && (((uintptr_t)dst) & 15) == 0) {
#pragma unroll
            for (int r = 0; r + 1 < R; r += 2) nt_store((float *)(dst + r), y[r].x, y[r].y, y[r + 1].x, y[r + 1].y);
            return;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long j = m0 + r;
            if (j >= g.store_from && j < out_end) nt_store((float *)(dst + r), y[r].x, y[r].y);
        }
        return;
    }
    float v[R];
    if (g.mode == SDR_MODE_FM) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = discriminate(r + 1 < R ? y[r + 1] : ynext, y[r], g.scale);
    } else if (g.mode == SDR_MODE_REAL) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = y[r].x;
    } else {
#pragma unroll
        // std::abs(complex): |y| <= 16*sum|h| here, so the plain form cannot overflow; a result
        // below 1e-19 (underflow of the squares) is 0 instead of tiny -- far under the tolerance
        for (int r = 0; r < R; ++r) v[r] = sqrtf(__fmaf_rn(y[r].x, y[r].x, y[r].y * y[r].y));
    }
    store_values<R>(g, outv, ch, m0, v, out_end);
}

// the real-valued outputs v[0..R) of lane position m0 (discriminator, envelope, real part)
template <int R>
__device__ __forceinline__ void store_values(const FirGeom &g, void *__restrict__ outv, long long ch, long long m0,
                                             const float (&v)[R], long long out_end)
{
    float *out = (float *)outv + ch * g.out_stride;
    float *dst = out + (m0 - g.out_begin);
    const bool whole = (m0 >= g.store_from) && (m0 + R <= out_end);
    if (R >= 2 && whole && (((uintptr_t)dst) & (R >= 4 ? 15 : 7)) == 0) {
        if (R == 2) {
            nt_store(dst, v[0], v[R > 1 ? 1 : 0]);
        } else {
#pragma unroll
            for (int r = 0; r < R; r += 4)
                nt_store(dst + r, v[r], v[(r + 1) % R], v[(r + 2) % R], v[(r + 3) % R]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long j = m0 + r;
            if (j >= g.store_from && j < out_end) nt_store(dst + r, v[r]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The lane window walk, software-pipelined in source.  The window is consumed in steps of U
// granules; the LDS data and the taps (scalar loads) of step k+1 are requested before the FMAs of
// step k, from two register sets used alternately.  walk_touch() "uses" a set without doing
// anything, which makes hipcc place its s_waitcnt for that set BEFORE the next set is requested
// (scalar loads return out of order, so any later wait would be lgkmcnt(0) and would also wait
// for the set that has only just been requested).  Host side pads the window to an even number
// of steps plus one look-ahead step (zero taps), so the loop needs no bounds test.
//   SPG = samples per 16-byte granule: 4 (raw int16 pairs) or 2 (float2).
// ---------------------------------------------------------------------------------------------
template <int SPG, int R> struct Walk {
    // taps per step N = SPG*R*U = 16 (32 for R = 8)
    static constexpr int U = (SPG * R >= 16) ? 1 : 16 / (SPG * R);
    static constexpr int N = SPG * R * U;
};

template <bool PAD>
__device__ __forceinline__ int walk_index(int c, const FirGeom &g)
{
    return PAD ? c + (int)__umulhi((unsigned)c, g.magic_q) : c;      // scalar: c is wave-uniform
}

template <int U, bool PAD>
__device__ __forceinline__ void walk_load(uint4 (&v)[U], const uint4 *lane_base, int c,
                                          const FirGeom &g)
{
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = lane_base[walk_index<PAD>(c + k, g)];
}

template <int N>
__device__ __forceinline__ void walk_taps(float (&t)[N], const float *__restrict__ gt)
{
#pragma unroll
    for (int k = 0; k < N; ++k) t[k] = gt[k];
}

template <int U, int N>
__device__ __forceinline__ void walk_touch(const uint4 (&v)[U], const float (&t)[N])
{
#pragma unroll
    for (int k = 0; k < U; ++k) asm volatile("" ::"v"(v[k].x), "v"(v[k].y), "v"(v[k].z), "v"(v[k].w));
#pragma unroll
    for (int k = 0; k < N; k += 8)
        asm volatile("" ::"s"(t[k]), "s"(t[k + 1]), "s"(t[k + 2]), "s"(t[k + 3]), "s"(t[k + 4]),
                     "s"(t[k + 5]), "s"(t[k + 6]), "s"(t[k + 7]));
}

// TAPS: 0 = real taps on complex samples (the plain filter), 1 = complex taps on REAL samples (real input, the mixer
// folded into the taps), 2 = complex taps on complex samples (complex float input with a mixer, fir_raw_kernel)
template <int SPG, int R, int U, int TAPS>
__device__ __forceinline__ void walk_fma(const uint4 (&v)[U], const float (&t)[(TAPS ? 2 : 1) * SPG * R * U],
                                         float2 (&acc)[R])
{
#pragma unroll
    for (int k = 0; k < U; ++k) {
        if constexpr (TAPS == 2) {
            static_assert(TAPS != 2 || SPG == 2, "complex taps on complex samples: float2 tiles");
            // two complex samples, complex taps g = h e^{-jwi}: acc += x * g
            const float xr[2] = {__uint_as_float(v[k].x), __uint_as_float(v[k].z)};
            const float xi[2] = {__uint_as_float(v[k].y), __uint_as_float(v[k].w)};
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int i = 2 * ((k * 2 + u) * R + r);
                    acc[r].x = __fmaf_rn(xr[u], t[i], acc[r].x);
                    acc[r].x = __fmaf_rn(-xi[u], t[i + 1], acc[r].x);
                    acc[r].y = __fmaf_rn(xr[u], t[i + 1], acc[r].y);
                    acc[r].y = __fmaf_rn(xi[u], t[i], acc[r].y);
                }
        } else if constexpr (TAPS == 1) {
            // four real samples, complex taps (the mixer folded into them): acc += x * (tr, ti)
            const float x[4] = {__uint_as_float(v[k].x), __uint_as_float(v[k].y), __uint_as_float(v[k].z), __uint_as_float(v[k].w)};
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int i = 2 * ((k * 4 + u) * R + r);
                    acc[r].x = __fmaf_rn(x[u], t[i], acc[r].x);
                    acc[r].y = __fmaf_rn(x[u], t[i + 1], acc[r].y);
                }
        } else {
            float xi[SPG], xq[SPG];
            if (SPG == 4) {
                const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) { xi[u] = lo16(w[u]); xq[u] = hi16(w[u]); }
            } else {
                xi[0] = __uint_as_float(v[k].x); xq[0] = __uint_as_float(v[k].y);
                xi[1] = __uint_as_float(v[k].z); xq[1] = __uint_as_float(v[k].w);
            }
#pragma unroll
            for (int u = 0; u < SPG; ++u)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float h = t[(k * SPG + u) * R + r];
                    acc[r].x = __fmaf_rn(h, xi[u], acc[r].x);
                    acc[r].y = __fmaf_rn(h, xq[u], acc[r].y);
                }
        }
    }
}

// s_begin / s_count (steps, both even): the part of the window this call walks -- the whole of it
// (0, ngran / U) everywhere but in fir_ksplit_kernel, whose four waves take a quarter each.
template <int SPG, int R, bool PAD, int TAPS>
__device__ __forceinline__ void walk_p(const FirGeom &g, const uint4 *lds4,
                                       const float *__restrict__ gtab, int tid, float2 (&acc)[R],
                                       int s_begin = 0, int s_count = -1)
{
    constexpr int U = Walk<SPG, R>::U;
    constexpr int N = (TAPS ? 2 : 1) * Walk<SPG, R>::N;    // floats of taps per step
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float2(0.f, 0.f);
    const uint4 *lane_base = lds4 + tid * (g.q + (PAD ? 1 : 0));
    const int nstep = s_count < 0 ? g.ngran / U : s_begin + s_count;           // even (host pads)
    uint4 va[U], vb[U];
    float ta[N], tb[N];
    walk_load<U, PAD>(va, lane_base, s_begin * U, g);
    walk_taps<N>(ta, gtab + (size_t)s_begin * N);
    for (int s = s_begin; s < nstep; s += 2) {
        walk_touch<U, N>(va, ta);
        __builtin_amdgcn_sched_barrier(0);
        walk_load<U, PAD>(vb, lane_base, (s + 1) * U, g);
        walk_taps<N>(tb, gtab + (size_t)(s + 1) * N);
        __builtin_amdgcn_sched_barrier(0);
        walk_fma<SPG, R, U, TAPS>(va, ta, acc);
        walk_touch<U, N>(vb, tb);
        __builtin_amdgcn_sched_barrier(0);
        walk_load<U, PAD>(va, lane_base, (s + 2) * U, g);      // look-ahead step (zero taps at the end)
        walk_taps<N>(ta, gtab + (size_t)(s + 2) * N);
        __builtin_amdgcn_sched_barrier(0);
        walk_fma<SPG, R, U, TAPS>(vb, tb, acc);
    }
}

template <int SPG, int R, int TAPS = 0>
__device__ __forceinline__ void walk(const FirGeom &g, const uint4 *lds4,
                                     const float *__restrict__ gtab, int tid, float2 (&acc)[R],
                                     int s_begin = 0, int s_count = -1)
{
    if (g.pad) walk_p<SPG, R, true, TAPS>(g, lds4, gtab, tid, acc, s_begin, s_count);
    else walk_p<SPG, R, false, TAPS>(g, lds4, gtab, tid, acc, s_begin, s_count);
}

__device__ __forceinline__ float2 mix(float2 x, double fr, long long s)
{
    // exp(+j*2*pi*fr*s): the phase in revolutions is reduced in double, so it is exact to
    // ~1e-16*s*fr before the float sincospi
    const double ph = fr * (double)s;
    const float f = (float)(ph - rint(ph));
    float sn, cs;
    sincospif(2.0f * f, &sn, &cs);
    return make_float2(__fmaf_rn(x.x, cs, -(x.y * sn)), __fmaf_rn(x.x, sn, x.y * cs));
}

__device__ __forceinline__ float2 cmulf(float2 a, float2 b)
{
    return make_float2(__fmaf_rn(a.x, b.x, -(a.y * b.y)), __fmaf_rn(a.x, b.y, a.y * b.x));
}

}  // namespace fir
}  // namespace sdr

#endif  // SDR_FIR_COMMON_H_
