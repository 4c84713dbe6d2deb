// fir.hip -- fused FIR + decimate (+ NCO mix, + FM discriminator | envelope) for gfx950.
//
// Replaces, in one pass over HBM, the reference's consumer loop
//   int16 -> float scale          bladerf/apt_demod/src/bladerf_apt_demod.cpp:728-732
//   polyphase_decimate            bladerf_apt_demod.cpp:249-313  (== af::fir + stride pick,
//                                 bladerf/rx_fm_demod/src/bladerf_fm_demod.cpp:291-294)
//   frequency_shift               bladerf_apt_demod.cpp:476-495  (optional, before the FIR)
//   polar_discriminator           bladerf_apt_demod.cpp:405-425
//   std::abs envelope             bladerf_apt_demod.cpp:750-754
//
// Work assignment ("output stride"): a workgroup of NT lanes owns a tile of NT*R consecutive
// outputs; lane l owns outputs m0 = M0 + l*R .. m0+R-1.  The input window of the tile,
//   samples W0 = D*M0 - Tback  ..  D*(M0+NT*R)            (Tback = (T-1) rounded up to 4),
// i.e. the tile's own D*NT*R samples plus the overlap-save halo of T-1 history samples, is
// staged once in LDS and every lane
// walks its own window [D*R*l, D*R*l + Lw) with 16-byte LDS reads.  One loaded sample feeds all
// R outputs of the lane: output r uses tap D*r + Tback - j at window offset j, which the host
// lays out as a zero-padded table gtab[j][r]; the table index is wave-uniform, so taps arrive
// through scalar loads and cost no LDS or VALU work.
//
// Three vector-unit kernels share that scheme (fir_planar_kernel: float planes converted once, see
// its header); long filters run as a Toeplitz GEMM on the int8 matrix cores instead
// (fir_mfma_i8_kernel, chosen by sdr_fir_plan_create from a measured cost model):
//   fir_raw_kernel     (automatic choice for int16 input without mixer) keeps the RAW int16 pairs
//       in LDS, filled by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip), and converts
//       after each LDS read.  Half the LDS footprint: 7 workgroups per CU on the headline shape.
//   fir_stream_kernel  keeps FLOAT samples in LDS: persistent workgroups fetch the raw tile k+1
//       from HBM into registers while they walk tile k, then convert it ONCE per sample
//       (int16 -> f32 is a half-rate VALU op on gfx950, and a sample is read by ~4 lanes), apply
//       the optional NCO mixer, and write float2 to LDS.  Any input type; the choice for float
//       inputs, the mixer, and shapes whose raw geometry does not fit (plan `pipeline` = 2).
// Lane stride in LDS is q = D*R/(samples per 16 B) granules; when q is even the tile image
// carries one pad granule per q (stride q+1, odd) so that ds_read_b128's 16-lane groups hit
// distinct banks.
//
// Numerics: float32 fmaf accumulation in ascending window order (identical for every tile
// placement: with the discriminator a tile advances by (NT-1)*R outputs and its last lane only
// computes the successor y of its neighbour, so every output has one summation order); decimation
// indexing is exact (output m <-> input D*m); zero pre-history (samples < 0 read as 0) as the
// reference.  For int16 input the exact 2^-11 Q11 scale is folded into the
// taps.

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "sdr_internal.h"

namespace {

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

template <int SPG, int R, int U, bool REALT>
__device__ __forceinline__ void walk_fma(const uint4 (&v)[U], const float (&t)[(REALT ? 2 : 1) * SPG * R * U],
                                         float2 (&acc)[R])
{
#pragma unroll
    for (int k = 0; k < U; ++k) {
        if constexpr (REALT) {
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
template <int SPG, int R, bool PAD, bool REALT>
__device__ __forceinline__ void walk_p(const FirGeom &g, const uint4 *lds4,
                                       const float *__restrict__ gtab, int tid, float2 (&acc)[R],
                                       int s_begin = 0, int s_count = -1)
{
    constexpr int U = Walk<SPG, R>::U;
    constexpr int N = (REALT ? 2 : 1) * Walk<SPG, R>::N;    // floats of taps per step
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
        walk_fma<SPG, R, U, REALT>(va, ta, acc);
        walk_touch<U, N>(vb, tb);
        __builtin_amdgcn_sched_barrier(0);
        walk_load<U, PAD>(va, lane_base, (s + 2) * U, g);      // look-ahead step (zero taps at the end)
        walk_taps<N>(ta, gtab + (size_t)(s + 2) * N);
        __builtin_amdgcn_sched_barrier(0);
        walk_fma<SPG, R, U, REALT>(vb, tb, acc);
    }
}

template <int SPG, int R, bool REALT = false>
__device__ __forceinline__ void walk(const FirGeom &g, const uint4 *lds4,
                                     const float *__restrict__ gtab, int tid, float2 (&acc)[R],
                                     int s_begin = 0, int s_count = -1)
{
    if (g.pad) walk_p<SPG, R, true, REALT>(g, lds4, gtab, tid, acc, s_begin, s_count);
    else walk_p<SPG, R, false, REALT>(g, lds4, gtab, tid, acc, s_begin, s_count);
}

// ---------------------------------------------------------------------------------------------
// fir_stream_kernel: float2 tiles, convert once, persistent workgroups with register prefetch.
// LDS: [tile (a_gran float4) | ybuf (NT+2 float2)].
// ---------------------------------------------------------------------------------------------
template <int IN> struct RawPair;                        // two consecutive input samples
template <> struct RawPair<IN_I16> { typedef uint2 type; };
template <> struct RawPair<IN_CF32> { typedef float4 type; };
template <> struct RawPair<IN_REAL> { typedef float4 type; };   // four: real tiles hold 4 samples per granule
template <> struct RawPair<IN_I8> { typedef uint32_t type; };

// bytes of one RawPair
template <int IN> struct PairBytes { static constexpr unsigned value = (IN == IN_CF32 || IN == IN_REAL) ? 16u : (IN == IN_I8 ? 4u : 8u); };
// samples per LDS granule of the float-tile kernel: two complex samples, or four real ones (real
// input stays real in LDS; the mixer is folded into complex taps, see fir_stream_kernel)
template <int IN> struct TileSpg { static constexpr int value = (IN == IN_REAL) ? 4 : 2; };

// Slow path for tiles that touch the ends of the valid window or a misaligned stream: every
// sample predicated on its own.
template <int IN>
__device__ __forceinline__ typename RawPair<IN>::type
fetch_pair_checked(const FirGeom &g, const void *__restrict__ src, long long s)
{
    typedef typename RawPair<IN>::type P;
    const bool v0 = (s >= g.lo_valid && s < g.hi_valid);
    const bool v1 = (s + 1 >= g.lo_valid && s + 1 < g.hi_valid);
    P p;
    if (IN == IN_CF32) {
        const float2 a = v0 ? ((const float2 *)src)[s] : make_float2(0.f, 0.f);
        const float2 b = v1 ? ((const float2 *)src)[s + 1] : make_float2(0.f, 0.f);
        float4 r = make_float4(a.x, a.y, b.x, b.y);
        memcpy(&p, &r, sizeof(p));
    } else if (IN == IN_REAL) {
        float r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long su = s + u;
            r[u] = (su >= g.lo_valid && su < g.hi_valid) ? ((const float *)src)[su] : 0.f;
        }
        memcpy(&p, r, sizeof(p));
    } else if (IN == IN_I8) {
        const uint32_t a = v0 ? ((const uint16_t *)src)[s] : 0u;
        const uint32_t b = v1 ? ((const uint16_t *)src)[s + 1] : 0u;
        const uint32_t r = a | (b << 16);
        memcpy(&p, &r, sizeof(p));
    } else {
        const uint32_t a = v0 ? ((const uint32_t *)src)[s] : 0u;
        const uint32_t b = v1 ? ((const uint32_t *)src)[s + 1] : 0u;
        uint2 r = make_uint2(a, b);
        memcpy(&p, &r, sizeof(p));
    }
    return p;
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

template <int IN>
__device__ __forceinline__ float4 convert_pair(const typename RawPair<IN>::type &p, double fr,
                                               long long s, int swap_iq)
{
    float2 x0, x1;
    if (IN == IN_I16) {
        uint2 w; memcpy(&w, &p, sizeof(w));
        x0 = make_float2(lo16(w.x), hi16(w.x));
        x1 = make_float2(lo16(w.y), hi16(w.y));
        if (swap_iq) { x0 = make_float2(x0.y, x0.x); x1 = make_float2(x1.y, x1.x); }
    } else if (IN == IN_I8) {
        uint32_t w; memcpy(&w, &p, sizeof(w));
        x0 = make_float2((float)(signed char)(w & 0xffu), (float)(signed char)((w >> 8) & 0xffu));
        x1 = make_float2((float)(signed char)((w >> 16) & 0xffu), (float)((int)w >> 24));
    } else if (IN == IN_CF32) {
        float4 w; memcpy(&w, &p, sizeof(w));
        x0 = make_float2(w.x, w.y);
        x1 = make_float2(w.z, w.w);
    } else {
        float4 w; memcpy(&w, &p, sizeof(w));           // real tile: four samples, stored as they are
        return w;
    }
    if (fr != 0.0) { x0 = mix(x0, fr, s); x1 = mix(x1, fr, s + 1); }
    return make_float4(x0.x, x0.y, x1.x, x1.y);
}

__device__ __forceinline__ float2 cmulf(float2 a, float2 b)
{
    return make_float2(__fmaf_rn(a.x, b.x, -(a.y * b.y)), __fmaf_rn(a.x, b.y, a.y * b.x));
}

// the sample pair x (tile offsets i, i+1; i even) times the mixer phasors B * HI[i >> 6] * LO[i & 63 (+1)]
template <int IN>
__device__ __forceinline__ float4 rotate_pair(float4 x, float2 B, const float2 *ph_hi, const float2 *ph_lo, int i)
{
    const float2 bh = cmulf(B, ph_hi[i >> 6]);
    const float4 lo = *(const float4 *)(ph_lo + (i & 63));
    const float2 p0 = cmulf(bh, make_float2(lo.x, lo.y)), p1 = cmulf(bh, make_float2(lo.z, lo.w));
    const float2 y0 = cmulf(make_float2(x.x, x.y), p0), y1 = cmulf(make_float2(x.z, x.w), p1);
    return make_float4(y0.x, y0.y, y1.x, y1.y);
}

struct TileRef { long long ch, M0, W0; const void *src; const unsigned char *wbase; bool fast; };

template <int IN>
__device__ __forceinline__ TileRef tile_ref(const FirGeom &g, const void *__restrict__ in, int t,
                                            int per_tile)
{
    TileRef r;
    const int ch = t / g.tiles_per_ch, tile = t - ch * g.tiles_per_ch;
    r.ch = ch;
    r.M0 = g.tile0 + (long long)tile * per_tile;
    r.W0 = r.M0 * g.D - g.Tback;
    const size_t esz = IN == IN_CF32 ? 8 : (IN == IN_I8 ? 2 : 4);
    r.src = (const unsigned char *)in + (size_t)((long long)ch * g.in_stride) * esz;
    r.wbase = (const unsigned char *)r.src + (size_t)r.W0 * esz;          // sample W0
    // fast path: the whole staged window is valid and pairs are naturally aligned
    r.fast = (r.W0 >= g.lo_valid) && (r.W0 + (long long)TileSpg<IN>::value * g.s_gran <= g.hi_valid) &&
             ((((uintptr_t)r.src) & (IN == IN_CF32 ? 15 : (IN == IN_I8 ? 3 : 7))) == 0);
    return r;
}

constexpr int kSlotsSmall = 12, kSlots = 32;   // granules one lane stages per tile (prefetch registers)
constexpr int kSlotsReal = 20;                 // real tiles: 16 bytes per slot, D*R/4 slots per lane

template <int IN, int R, int SLOTS>
__global__ void __launch_bounds__(256, (SLOTS <= kSlotsSmall ? 3 : 2))
fir_stream_kernel(const void *__restrict__ in, void *__restrict__ outv,
                  const float *__restrict__ gtab, const FirGeom g)
{
    typedef typename RawPair<IN>::type P;
    constexpr int SPG = TileSpg<IN>::value;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int NT = blockDim.x;
    float4 *tile = (float4 *)smem;
    float2 *ybuf = (float2 *)(smem + (size_t)g.a_gran * 16);

    // Per-slot constants (the same for every tile): the sample granule this lane stages in slot k
    // and its byte offset from the tile window's first sample.  Slots past the tile are clamped
    // to its last granule, so the fast path below is branch-free: a branch per load would make
    // hipcc wait for each load at the join, one HBM round trip after the other.
    P pre[SLOTS];
    int sgk[SLOTS];
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const int sg = sample_granule(g, k * NT + tid);
        sgk[k] = sg < g.s_gran ? sg : g.s_gran - 1;
    }
    auto fetch_tile = [&](const TileRef &tr) {
        if (tr.fast) {
#pragma unroll
            for (int k = 0; k < SLOTS; ++k)
                pre[k] = *(const P *)(tr.wbase + (unsigned)sgk[k] * PairBytes<IN>::value);
        } else {
#pragma unroll
            for (int k = 0; k < SLOTS; ++k)
                pre[k] = fetch_pair_checked<IN>(g, tr.src, tr.W0 + (long long)SPG * sgk[k]);
        }
    };

    // Mixer: exp(j*2*pi*fr*s) for sample s = W0 + i of a tile is B * HI[i >> 6] * LO[i & 63], with
    // B = the phasor of the tile's first sample (phase reduced in double, one sincos per tile) and
    // two small tables built on the host in double, kept in LDS: three complex products per sample
    // pair instead of a double-precision phase and a sincos per sample (which cost more than the
    // filter itself at 34 taps per sample).  Rounding: three float32 products, ~2e-7.
    // Real input stays real in LDS (four samples per granule, half the LDS traffic of the walk) and
    // the mixer moves into the taps: y[m] = e^{j w m D} * sum_i (h[i] e^{-j w i}) x[mD - i], complex
    // taps built on the host, one rotation per output afterwards (none for the envelope).
    // Channelizer (g.chan_fr): every channel has its own frequency and its own pair of tables, and a
    // persistent workgroup meets a different channel in nearly every tile.  The tables of the NEXT tile's
    // channel are requested before that tile's samples (so waiting for them does not wait for the
    // prefetch), kept in two registers per lane across the walk, and written to LDS behind it; the
    // barrier at the top of the next round publishes them.
    const bool chan = (IN != IN_REAL) && g.chan_fr != nullptr;
    const bool mixing = (IN != IN_REAL) && (g.mix_fr != 0.0 || chan);
    float2 *ph_hi = (float2 *)(smem + (size_t)g.a_gran * 16 + (size_t)(NT + 2) * 8);
    float2 *ph_lo = ph_hi + g.n_ph_hi;
    const int n_tab = g.n_ph_hi + 64;            // channelizer plans: <= 2 * NT (plan creation)
    auto tab_of = [&](int tile_no) {
        const long long c = chan ? chan_index(g, tile_no / g.tiles_per_ch) : 0;
        return chan ? g.chan_tab + c * g.chan_tab_stride : (const float2 *)(gtab + g.ph_off);
    };
    float2 tab_a = make_float2(0.f, 0.f), tab_b = tab_a;
    auto tab_request = [&](int tile_no) {
        const float2 *src = tab_of(tile_no);
        if (tid < n_tab) tab_a = src[tid];
        if (tid + NT < n_tab) tab_b = src[tid + NT];
    };
    auto tab_publish = [&]() {
        if (tid < n_tab) ph_hi[tid] = tab_a;
        if (tid + NT < n_tab) ph_hi[tid + NT] = tab_b;
    };

    int t = blockIdx.x;
    if (mixing && t < g.total_tiles) {
        const float2 *src = tab_of(t);
        for (int i = tid; i < n_tab; i += NT) ph_hi[i] = src[i];
    }
    if (t < g.total_tiles) fetch_tile(tile_ref<IN>(g, in, t, g.per_tile));
    for (; t < g.total_tiles; t += gridDim.x) {
        const TileRef tr = tile_ref<IN>(g, in, t, g.per_tile);
        __syncthreads();                         // every lane has finished with the previous tile
        float2 B = make_float2(1.f, 0.f);
        if (mixing) {
            const double fr = chan ? g.chan_fr[chan_index(g, tr.ch)] : g.mix_fr;
            const double ph = fr * (double)tr.W0;
            sincospif(2.0f * (float)(ph - rint(ph)), &B.y, &B.x);
        }
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            const int a = k * NT + tid;
            float4 x = convert_pair<IN>(pre[k], 0.0, 0, g.swap_iq);
            if (mixing) x = rotate_pair<IN>(x, B, ph_hi, ph_lo, 2 * sgk[k]);
            if (a < g.a_gran) tile[a] = x;
        }
        if constexpr (SLOTS == kSlots) {
            // tiles with more granules per lane than prefetch registers (decimation factors beyond
            // ~32 with float input or the mixer): the rest is fetched here, without prefetch
            for (int k = SLOTS; k < g.n_inst; ++k) {
                const int a = k * NT + tid;
                const int sg0 = sample_granule(g, a);
                const int sg = sg0 < g.s_gran ? sg0 : g.s_gran - 1;
                P raw;
                if (tr.fast) raw = *(const P *)(tr.wbase + (unsigned)sg * PairBytes<IN>::value);
                else raw = fetch_pair_checked<IN>(g, tr.src, tr.W0 + (long long)SPG * sg);
                float4 x = convert_pair<IN>(raw, 0.0, 0, g.swap_iq);
                if (mixing) x = rotate_pair<IN>(x, B, ph_hi, ph_lo, 2 * sg);
                if (a < g.a_gran) tile[a] = x;
            }
        }
        __syncthreads();
        // request the next tile now: its HBM latency is covered by the walk below
        {
            const int tn = t + gridDim.x;
            if (chan && tn < g.total_tiles) tab_request(tn);
            if (tn < g.total_tiles) fetch_tile(tile_ref<IN>(g, in, tn, g.per_tile));
        }
        __builtin_amdgcn_sched_barrier(0);

        float2 acc[R];
        walk<SPG, R, IN == IN_REAL>(g, (const uint4 *)tile, gtab, tid, acc);
        if (chan && t + (int)gridDim.x < g.total_tiles) tab_publish();    // (nobody reads the tables before the next barrier)

        const long long m0 = tr.M0 + (long long)tid * R;
        if (IN == IN_REAL && g.mix_fr != 0.0 && g.mode != SDR_MODE_ENV) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double ph = g.mix_fr * (double)((m0 + r) * g.D);
                float2 P;
                sincospif(2.0f * (float)(ph - rint(ph)), &P.y, &P.x);
                acc[r] = cmulf(acc[r], P);
            }
        }
        float2 ynext = make_float2(0.f, 0.f);
        if (g.mode == SDR_MODE_FM) {
            ybuf[tid] = acc[0];
            __syncthreads();
            ynext = ybuf[tid + 1];                 // the last lane's is unused: it emits nothing
        }
        store_outputs<R>(g, outv, tr.ch, m0, acc, ynext, tr.M0 + g.per_tile);
    }
}

// ---------------------------------------------------------------------------------------------
// fir_ksplit_kernel: real float input, SHORT streams (plan `pipeline` = 6, or chosen from the plan's
// size hint).  The streaming kernel above is built for throughput: a lane walks its whole window, and a
// long filter makes that one dependent chain of T multiply-adds per output behind scalar tap loads.
// On a stream of a few million samples (the second stage of the APT chain runs at a tenth of the
// capture rate, bladerf_apt_demod.cpp:744-754) there are fewer such chains than the chip has SIMDs
// to hide their latency on, and the launch takes as long as ONE chain.  Here a workgroup of four
// waves owns a tile of only 64 lanes x R outputs, stages its window once, and every wave walks a
// QUARTER of the window for all 64 lanes; the four partial sums meet in LDS and are added in a fixed
// order (quarter 0 first), so an output's value does not depend on the tiling or on the shard.  Four
// times as many, four times shorter chains.  Same byte model as the streaming kernel.
// LDS: [tile (a_gran float4) | partial sums (4 x 64*R float2)].  No discriminator (the successor of a
// lane's last output lives in another lane's partial sums; the modes real input is used with --
// envelope, IQ, real -- do not need it).
// ---------------------------------------------------------------------------------------------
template <int R>
__global__ void __launch_bounds__(256)
fir_ksplit_kernel(const float *__restrict__ in, void *__restrict__ outv,
                  const float *__restrict__ gtab, const FirGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *tile = (float4 *)smem;
    float2 *part = (float2 *)(smem + (size_t)g.a_gran * 16);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = (int)blockIdx.y * g.tiles_per_ch + (int)blockIdx.x;
    const TileRef tr = tile_ref<IN_REAL>(g, in, t, g.per_tile);
    for (int a = tid; a < g.a_gran; a += 256) {
        const int sg0 = sample_granule(g, a);
        const int sg = sg0 < g.s_gran ? sg0 : g.s_gran - 1;
        float4 x;
        if (tr.fast) x = *(const float4 *)(tr.wbase + (unsigned)sg * 16u);
        else x = fetch_pair_checked<IN_REAL>(g, tr.src, tr.W0 + 4LL * sg);
        tile[a] = x;
    }
    __syncthreads();
    constexpr int U = Walk<4, R>::U;
    const int quarter = g.ngran / U / 4;           // even (host pads the window to a multiple of 8 steps)
    float2 acc[R];
    walk<4, R, true>(g, (const uint4 *)tile, gtab, lane, acc, wave * quarter, quarter);
#pragma unroll
    for (int r = 0; r < R; ++r) part[(wave * 64 + lane) * R + r] = acc[r];
    __syncthreads();
    const long long tile_end = tr.M0 + g.per_tile < g.out_end ? tr.M0 + g.per_tile : g.out_end;
    for (int o = tid; o < 64 * R; o += 256) {
        const long long m = tr.M0 + o;
        if (m < g.out_begin || m >= tile_end) continue;
        const float2 p0 = part[o], p1 = part[64 * R + o], p2 = part[2 * 64 * R + o], p3 = part[3 * 64 * R + o];
        float2 y = make_float2(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y);
        if (g.mix_fr != 0.0 && g.mode != SDR_MODE_ENV) {      // the mixer's phasor at the output's own sample
            const double ph = g.mix_fr * (double)(m * g.D);
            float2 P;
            sincospif(2.0f * (float)(ph - rint(ph)), &P.y, &P.x);
            y = cmulf(y, P);
        }
        if (g.mode == SDR_MODE_IQ) {
            nt_store((float *)((float2 *)outv + tr.ch * g.out_stride + (m - g.out_begin)), y.x, y.y);
        } else {
            const float v = g.mode == SDR_MODE_REAL ? y.x : sqrtf(__fmaf_rn(y.x, y.x, y.y * y.y));
            nt_store((float *)outv + tr.ch * g.out_stride + (m - g.out_begin), v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fir_raw_kernel: int16 IQ without mixer, raw samples in LDS (4 per granule) filled by LDS-DMA,
// one tile per workgroup.  LDS: [tile (a_gran uint4) | ybuf (NT+2 float2)].
// ---------------------------------------------------------------------------------------------
// One LDS-DMA piece (global_load_lds_dwordx4: 16 B per lane, 1 KiB per wave), from inline asm so
// that it stays off hipcc's vmcnt book-keeping; the kernel waits vmcnt(0) itself before the
// barrier.  lds_byte must be wave-uniform (it goes to M0); M0 is saved and restored.
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_byte)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte)
                 : "memory");
}

__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char *)p;
}

template <int R>
__global__ void __launch_bounds__(256)
fir_raw_kernel(const uint32_t *__restrict__ in, void *__restrict__ outv,
               const float *__restrict__ gtab, const FirGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *lds4 = (uint4 *)smem;
    const int tid = threadIdx.x;
    const int NT = blockDim.x;
    const long long ch = blockIdx.y;
    const int tile = xcd_tile(g);
    if (tile >= g.tiles_per_ch) return;
    const long long M0 = g.tile0 + (long long)tile * g.per_tile;
    const long long W0 = M0 * g.D - g.Tback;                 // first staged sample (multiple of 4)
    const uint32_t *src = in + ch * g.in_stride;             // sample 0 of this channel

    const bool interior = (W0 >= g.lo_valid) && (W0 + 4LL * g.s_gran <= g.hi_valid);
    if (interior) {
        const uint32_t *wsrc = src + W0;
        for (int k = 0; k < g.n_inst; ++k) {
            const int a = k * NT + tid;
            int sg = sample_granule(g, a);
            sg = sg < g.s_gran ? sg : 0;
            if (a < g.a_gran)              // last round: lanes past the image stay masked
                dma16(wsrc + 4 * sg, __builtin_amdgcn_readfirstlane(lds_addr(lds4 + k * NT + (tid & ~63))));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (int k = 0; k < g.n_inst; ++k) {
            const int a = k * NT + tid;
            const long long s = W0 + 4LL * sample_granule(g, a);
            uint32_t w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long si = s + u;
                w[u] = (si >= g.lo_valid && si < g.hi_valid) ? src[si] : 0u;
            }
            if (a < g.a_gran) lds4[a] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    __syncthreads();

    float2 acc[R];
    walk<4, R>(g, lds4, gtab, tid, acc);
    if (g.swap_iq) {                       // Q-first input: real taps, so swapping the sums is exact
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = make_float2(acc[r].y, acc[r].x);
    }

    const long long m0 = M0 + (long long)tid * R;
    float2 ynext = make_float2(0.f, 0.f);
    if (g.mode == SDR_MODE_FM) {
        float2 *ybuf = (float2 *)(smem + (size_t)g.a_gran * 16);
        ybuf[tid] = acc[0];
        __syncthreads();
        ynext = ybuf[tid + 1];                 // the last lane's is unused: it emits nothing
    }
    store_outputs<R>(g, outv, ch, m0, acc, ynext, M0 + g.per_tile);
}

// ---------------------------------------------------------------------------------------------
// fir_planar_kernel: int16 IQ without mixer, converted ONCE while the tile is staged and kept as
// two float planes (all I, all Q) in LDS.  On gfx950 only plain f32 FMA/MUL/ADD issue in 2 cycles
// per wave64; v_cvt_f32_i32 (and every packed, SDWA or DPP form) takes 4, and in the raw-tile
// kernel each sample is converted by every lane whose window covers it (~4 for 64 taps / 10).
// Here a sample is converted by the one lane that stages it and the walk is FMA only.  Planes
// rather than interleaved float2: the lane stride stays D*R*4 bytes, the same bank pattern (and
// the same granule geometry, 4 samples per 16 B) as the raw tile, at twice the LDS footprint.
// LDS: [I plane (a_gran float4) | Q plane (a_gran float4) | ybuf (NT+2 float2)].
// ---------------------------------------------------------------------------------------------
template <int U, bool PAD>
__device__ __forceinline__ void pwalk_load(float4 (&vr)[U], float4 (&vi)[U], const float4 *re,
                                           const float4 *im, int c, const FirGeom &g)
{
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const int idx = walk_index<PAD>(c + k, g);
        vr[k] = re[idx];
        vi[k] = im[idx];
    }
}

template <int U, int N>
__device__ __forceinline__ void pwalk_touch(const float4 (&vr)[U], const float4 (&vi)[U],
                                            const float (&t)[N])
{
#pragma unroll
    for (int k = 0; k < U; ++k) {
        asm volatile("" ::"v"(vr[k].x), "v"(vr[k].y), "v"(vr[k].z), "v"(vr[k].w));
        asm volatile("" ::"v"(vi[k].x), "v"(vi[k].y), "v"(vi[k].z), "v"(vi[k].w));
    }
#pragma unroll
    for (int k = 0; k < N; k += 8)
        asm volatile("" ::"s"(t[k]), "s"(t[k + 1]), "s"(t[k + 2]), "s"(t[k + 3]), "s"(t[k + 4]),
                     "s"(t[k + 5]), "s"(t[k + 6]), "s"(t[k + 7]));
}

template <int R, int U>
__device__ __forceinline__ void pwalk_fma(const float4 (&vr)[U], const float4 (&vi)[U],
                                          const float (&t)[4 * R * U], float (&ar)[R], float (&ai)[R])
{
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const float xi[4] = {vr[k].x, vr[k].y, vr[k].z, vr[k].w};
        const float xq[4] = {vi[k].x, vi[k].y, vi[k].z, vi[k].w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float h = t[(k * 4 + u) * R + r];
                ar[r] = __fmaf_rn(h, xi[u], ar[r]);
                ai[r] = __fmaf_rn(h, xq[u], ai[r]);
            }
    }
}

template <int R, bool PAD>
__device__ __forceinline__ void pwalk_p(const FirGeom &g, const float4 *re4, const float4 *im4,
                                        const float *__restrict__ gtab, int tid, float2 (&acc)[R])
{
    constexpr int U = Walk<4, R>::U;
    constexpr int N = Walk<4, R>::N;
    float ar[R], ai[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { ar[r] = 0.f; ai[r] = 0.f; }
    const int lane_off = tid * (g.q + (PAD ? 1 : 0));
    const float4 *re = re4 + lane_off, *im = im4 + lane_off;
    const int nstep = g.ngran / U;           // even (host pads)
    float4 var[U], vai[U], vbr[U], vbi[U];
    float ta[N], tb[N];
    pwalk_load<U, PAD>(var, vai, re, im, 0, g);
    walk_taps<N>(ta, gtab);
    for (int s = 0; s < nstep; s += 2) {
        pwalk_touch<U, N>(var, vai, ta);
        __builtin_amdgcn_sched_barrier(0);
        pwalk_load<U, PAD>(vbr, vbi, re, im, (s + 1) * U, g);
        walk_taps<N>(tb, gtab + (size_t)(s + 1) * N);
        __builtin_amdgcn_sched_barrier(0);
        pwalk_fma<R, U>(var, vai, ta, ar, ai);
        pwalk_touch<U, N>(vbr, vbi, tb);
        __builtin_amdgcn_sched_barrier(0);
        pwalk_load<U, PAD>(var, vai, re, im, (s + 2) * U, g);      // look-ahead step (zero taps at the end)
        walk_taps<N>(ta, gtab + (size_t)(s + 2) * N);
        __builtin_amdgcn_sched_barrier(0);
        pwalk_fma<R, U>(vbr, vbi, tb, ar, ai);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float2(ar[r], ai[r]);
}

constexpr int kPlanarSlots = 8;     // granules one lane stages per round of loads

// Stage one tile as two float planes: every lane fetches its granules (4 int16 pairs each), converts
// them once and writes I and Q to re4[] / im4[].  The loads are branch-free: each granule comes from
// an address clamped into the stream's valid samples, and only tiles that touch an end of the
// stream (wave-uniform test) repair their granules in registers afterwards.  (Loads on two paths
// into the same registers make hipcc wait for vmcnt(0) before every load, one HBM round trip each.)
__device__ __forceinline__ void stage_planar(const FirGeom &g, const uint32_t *__restrict__ src,
                                             long long W0, float4 *re4, float4 *im4, int tid, int NT)
{
    const long long n = 4LL * g.s_gran;
    long long lo_l = g.lo_valid - W0, hi_l = g.hi_valid - W0;
    lo_l = lo_l < 0 ? 0 : (lo_l > n ? n : lo_l);
    hi_l = hi_l > n ? n : (hi_l < lo_l ? lo_l : hi_l);
    const int lo = (int)lo_l, hi = (int)hi_l;                // valid samples of the tile: [lo, hi)
    const uint32_t *base = src + W0;                         // only clamped offsets are dereferenced
    if (hi - lo < 4) {
        // fewer than four valid samples under the tile: no 16-byte load fits, fetch sample by sample
        for (int a = tid; a < g.a_gran; a += NT) {
            int sg = sample_granule(g, a);
            sg = sg < g.s_gran ? sg : 0;
            float fi[4], fq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int si = 4 * sg + u;
                const uint32_t e = (si >= lo && si < hi) ? base[si] : 0u;
                fi[u] = lo16(e);
                fq[u] = hi16(e);
            }
            re4[a] = make_float4(fi[0], fi[1], fi[2], fi[3]);
            im4[a] = make_float4(fq[0], fq[1], fq[2], fq[3]);
        }
        return;
    }
    const bool edge = lo > 0 || hi < (int)n;
    const int top = hi - 4;
    for (int k0 = 0; k0 < g.n_inst; k0 += kPlanarSlots) {
        uint4 w[kPlanarSlots];
        int rel[kPlanarSlots];
        // all loads of the round first, unconditionally (slots past the tile repeat its last
        // round-robin row)
#pragma unroll
        for (int k = 0; k < kPlanarSlots; ++k) {
            const int kk = (k0 + k < g.n_inst) ? k0 + k : g.n_inst - 1;
            int sg = sample_granule(g, kk * NT + tid);
            sg = sg < g.s_gran ? sg : 0;
            rel[k] = 4 * sg;
            const int at = rel[k] < lo ? lo : (rel[k] > top ? top : rel[k]);
            w[k] = *(const uint4 *)(base + at);
        }
#pragma unroll
        for (int k = 0; k < kPlanarSlots; ++k) {
            const int a = (k0 + k) * NT + tid;
            if (k0 + k < g.n_inst && a < g.a_gran) {
                uint32_t e[4] = {w[k].x, w[k].y, w[k].z, w[k].w};
                if (edge) {                                  // wave-uniform
                    const int at = rel[k] < lo ? lo : (rel[k] > top ? top : rel[k]);
                    uint32_t o[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int si = rel[k] + u, idx = si - at;
                        uint32_t v = e[0];
                        v = idx == 1 ? e[1] : v;
                        v = idx == 2 ? e[2] : v;
                        v = idx == 3 ? e[3] : v;
                        o[u] = (si >= lo && si < hi && idx >= 0 && idx < 4) ? v : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) e[u] = o[u];
                }
                re4[a] = make_float4(lo16(e[0]), lo16(e[1]), lo16(e[2]), lo16(e[3]));
                im4[a] = make_float4(hi16(e[0]), hi16(e[1]), hi16(e[2]), hi16(e[3]));
            }
        }
    }
}

template <int R>
__global__ void __launch_bounds__(256)
fir_planar_kernel(const uint32_t *__restrict__ in, void *__restrict__ outv,
                  const float *__restrict__ gtab, const FirGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *re4 = (float4 *)smem;
    float4 *im4 = re4 + g.a_gran;
    const int tid = threadIdx.x;
    const int NT = blockDim.x;
    const long long ch = blockIdx.y;
    const int tile = xcd_tile(g);
    if (tile >= g.tiles_per_ch) return;
    const long long M0 = g.tile0 + (long long)tile * g.per_tile;
    const long long W0 = M0 * g.D - g.Tback;                 // first staged sample (multiple of 4)
    const uint32_t *src = in + ch * g.in_stride;             // sample 0 of this channel

    stage_planar(g, src, W0, re4, im4, tid, NT);
    __syncthreads();
    if (g.swap_iq) { float4 *t = re4; re4 = im4; im4 = t; }      // Q-first input

    float2 acc[R];
    if (g.pad) pwalk_p<R, true>(g, re4, im4, gtab, tid, acc);
    else pwalk_p<R, false>(g, re4, im4, gtab, tid, acc);

    const long long m0 = M0 + (long long)tid * R;
    float2 ynext = make_float2(0.f, 0.f);
    if (g.mode == SDR_MODE_FM) {
        float2 *ybuf = (float2 *)(smem + (size_t)g.a_gran * 32);
        ybuf[tid] = acc[0];
        __syncthreads();
        ynext = ybuf[tid + 1];                 // the last lane's is unused: it emits nothing
    }
    store_outputs<R>(g, outv, ch, m0, acc, ynext, M0 + g.per_tile);
}

// ---------------------------------------------------------------------------------------------
// fir_mfma_kernel: long filters as a Toeplitz GEMM on the matrix cores, full float32
// (v_mfma_f32_16x16x4_f32).  Same planar float tile as fir_planar_kernel; the window walk becomes
//   Y[i][j] = sum_k H[i][k] * X[k][j],     H[i][k] = h[D*i + Tb - k]   (Toeplitz, 16 x W)
// with i = output within a group of 16 consecutive outputs, j = (group, component): 8 groups x
// {I, Q} per wave, k = offset in the group's window (W = Tb + 15*D + 1 rounded up to 16).  A wave
// owns 128 outputs, a 256-lane workgroup 512.  The useful fraction of the MACs is T / W (511 taps /
// 10: 76 %).  One k-chunk of 16 is four MFMAs whose operands are one 16-byte read each: lane
// (j, kq) takes samples 16q + 4kq + 0..3 of its column from LDS and lane (i, kq) the four taps that
// go with them from a table laid out for exactly this (atab, global memory, L1/L2 resident).
// Accumulators go through LDS (ybuf) to the common epilogue, which wants outputs 2*tid, 2*tid+1.
// LDS: [I plane | Q plane | ybuf (513 float2)].
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMfmaAhead = 4;      // operand prefetch distance, in k-chunks of 16

__global__ void __launch_bounds__(256)
fir_mfma_kernel(const uint32_t *__restrict__ in, void *__restrict__ outv,
                const float4 *__restrict__ atab, const FirGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *re4 = (float4 *)smem;
    float4 *im4 = re4 + g.a_gran;
    float2 *ybuf = (float2 *)(smem + (size_t)g.a_gran * 32);
    const int tid = threadIdx.x;
    constexpr int NT = 256, R = 2;
    const long long ch = blockIdx.y;
    const int tile = xcd_tile(g);
    if (tile >= g.tiles_per_ch) return;
    const long long M0 = g.tile0 + (long long)tile * g.per_tile;
    const long long W0 = M0 * g.D - g.Tback;
    const uint32_t *src = in + ch * g.in_stride;
    stage_planar(g, src, W0, re4, im4, tid, NT);
    __syncthreads();
    if (g.swap_iq) { float4 *t = re4; re4 = im4; im4 = t; }

    const int wave = tid >> 6, lane = tid & 63;
    const int j = lane & 15, kq = lane >> 4;
    const int grp = 8 * wave + (j >> 1);                   // group of 16 outputs within the tile
    const float *plane = (j & 1) ? (const float *)im4 : (const float *)re4;
    const float4 *bsrc = (const float4 *)(plane + g.D * 16 * grp + 4 * kq);
    const float4 *asrc = atab + lane;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // Operands are requested kMfmaAhead chunks (4 MFMAs = 128 cycles each) before they are used:
    // the tap table comes from L2 at several hundred cycles, the samples from LDS.  The host pads
    // the chunk count to a multiple of kMfmaAhead and both sources by kMfmaAhead zero chunks.
    const int nq = g.ngran;
    float4 a[kMfmaAhead], b[kMfmaAhead];
#pragma unroll
    for (int u = 0; u < kMfmaAhead; ++u) { a[u] = asrc[64 * u]; b[u] = bsrc[4 * u]; }
    for (int q = 0; q < nq; q += kMfmaAhead) {
#pragma unroll
        for (int u = 0; u < kMfmaAhead; ++u) {
            // "use" the oldest operand set so that hipcc waits for it here (counted vmcnt/lgkmcnt),
            // then request the set kMfmaAhead chunks ahead into the same slot, then multiply; the
            // sched_barriers keep hipcc from sinking the loads next to their use
            const float4 av = a[u], bv = b[u];
            asm volatile("" ::"v"(av.x), "v"(av.y), "v"(av.z), "v"(av.w), "v"(bv.x), "v"(bv.y), "v"(bv.z), "v"(bv.w));
            __builtin_amdgcn_sched_barrier(0);
            a[u] = asrc[64 * (q + kMfmaAhead + u)];
            b[u] = bsrc[4 * (q + kMfmaAhead + u)];
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // lane (j, kq) holds component j&1 of outputs 4*kq + v of its group
    {
        float *yb = (float *)ybuf + 2 * (16 * grp + 4 * kq) + (j & 1);
        yb[0] = acc0[0] + acc1[0];
        yb[2] = acc0[1] + acc1[1];
        yb[4] = acc0[2] + acc1[2];
        yb[6] = acc0[3] + acc1[3];
    }
    __syncthreads();
    const float2 y[R] = {ybuf[2 * tid], ybuf[2 * tid + 1]};
    const float2 ynext = ybuf[2 * tid + 2];                // FM only; the last lane's is unused
    store_outputs<R>(g, outv, ch, M0 + 2LL * tid, y, ynext, M0 + g.per_tile);
}

// ---------------------------------------------------------------------------------------------
// fir_mfma_i8_kernel: the same Toeplitz GEMM on the int8 matrix pipe (v_mfma_i32_16x16x64_i8: twice
// the bf16 rate, 32x the float32 MFMA rate) in EXACT integer arithmetic.  Both operands are
// split into signed bytes:
//   int16 sample  s = 256*hi + (lo' + 128)       hi = s >> 8,  lo' = (s & 255) - 128: the two raw
//                                                 bytes of the sample, the low one with bit 7 flipped
//   tap           h ~ q * 2^-F,  q = d3*2^24 + d2*2^16 + d1*2^8 + d0    balanced digits in [-128, 127]
//                                                 (q has 31 bits: finer than the float32 taps of the other kernels)
// Seven byte products per 64 window offsets (all but lowest digit x low byte, which lies below 2^-30
// of full scale) accumulate in four int32 accumulators, one per weight 2^(8w), w = digit + (hi ? 1 : 0),
// over the WHOLE window.  The four exact integers of an output are combined in float32 once; the +128
// of the low bytes comes back as 128 * (sum of the row's digits), combined the same way.  No rounding
// before that step, zero samples give exact zeros, and an output's bits depend neither on the tile
// it falls into nor on how the stream is cut into launches or shards.  (Discriminator plans whose tap
// set has a tap below the fixed-point resolution look at their exactly-zero outputs again, with
// float32 taps: i8_zero_fix.)
// Staging is eight byte permutes per four samples -- no conversions.
//
// 512 lanes, two waves per SIMD with 256 registers each, in two roles.  Waves 0-3 ("matrix waves")
// keep the tap digits of the whole window in registers (12 steps x 4 digits x 16 B per lane = 192
// VGPRs), take a quarter of the tile's column blocks each (a block = 16 columns = 8 groups of 16
// outputs x {I, Q}; NCB = 8, a tile of 1024 outputs, when the LDS holds it, else 4) and run nothing
// but k loops: 84 MFMAs per column block, then one float32 combination.  Waves 4-7 ("helper waves")
// fetch the raw samples two tiles ahead, split them into byte planes and stage the next tile into
// the other LDS buffer, then finish the tile before the last (discriminator / envelope, stores):
//   phase p:  matrix: k loops of tile p-1       | helpers: tile p -> buffer p&1, request tile p+2,
//             (buffer (p-1)&1) -> ybuf[(p-1)&1]  |          finish tile p-2 from ybuf[p&1]
//             barrier
// Why this split (tools/ubench/mfma_valu_overlap.hip): while one wave streams MFMAs, another wave of
// the same SIMD issues about two plain 32-bit vector instructions per MFMA in its shadow (packed
// float32 instructions: none -- they share the matrix data path), so a tile costs the MFMAs plus
// whatever vector work does not fit that shadow.  An earlier form (1024 lanes; the window dealt to
// four matrix waves per column block, 48 registers of digits each) paid the combination four times
// per block plus a four-way sum in the finishing waves: 626 vector instructions per SIMD and tile
// against a shadow of 366, and the tile took twice the matrix time.  This form needs ~390.
//   A (taps):    lane (i = lane % 16, kq = lane / 16) holds digit d of H[i][64*ks + 16*kq + 0..15],
//                H[i][k] = q[D*i + Tb - k] (the Toeplitz skew lives here, so sample reads stay aligned)
//   B (samples): lane (j, kq) holds bytes 16*D*grp + 64*ks + 16*kq + 0..15 of plane (j & 1, hi | lo)
//   C:           lane (j, kq) holds outputs 4*kq + 0..3 of group grp = 8*cb + j/2, component j & 1
// (the order of the 64 k values inside a step only has to be the same in A and B).
// LDS: 2 x [I hi | Q hi | I lo | Q lo] byte planes | 2 x ybuf (NCB blocks x 66 float4).
// ---------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int kI8MaxSteps = 12;    // k steps of 64: windows up to 768 samples (192 registers of tap digits)
constexpr int kI8PartSlots = 66;   // float4 slots of one column block's outputs (2 components x 32, padded)
// octets (8 samples) a staging lane fetches ahead per tile: 4 (6 for the 1024-output tile) x 256 x 8 samples

// SDR_MEASUREMENT_BUILD only (sdr_internal.h): 1 no k loop, 2 no staging, 4 no epilogue, 8 per-wave
// phase clocks behind the outputs (tools/i8_stamps.py), 16 no accumulator combination, 32 no sample
// reads inside the k loop
#ifdef SDR_I8_ABLATE
#define SDR_I8_ABL SDR_I8_ABLATE
#else
#define SDR_I8_ABL 0
#endif

// bit 8: shader cycles each wave spends working / parked at the per-tile barrier
struct I8Clock {
#if SDR_I8_ABL & 8
    long long busy = 0, wait = 0, t = 0, sec[3] = {0, 0, 0}, ts = 0;
    __device__ __forceinline__ void start() { t = ts = (long long)__builtin_amdgcn_s_memtime(); }
    // helper waves: time since the last mark (or the barrier) goes to section i
    __device__ __forceinline__ void mark(int i) { const long long n = (long long)__builtin_amdgcn_s_memtime(); sec[i] += n - ts; ts = n; }
    __device__ __forceinline__ void arrive() { const long long n = (long long)__builtin_amdgcn_s_memtime(); busy += n - t; t = n; }
    __device__ __forceinline__ void leave() { const long long n = (long long)__builtin_amdgcn_s_memtime(); wait += n - t; t = ts = n; }
    __device__ __forceinline__ void flush(void *outv, long long n_out, int phases) const
    {
        if ((threadIdx.x & 63) == 0) {
            uint32_t *d = (uint32_t *)outv + n_out + 64 + 8 * ((long long)blockIdx.x * 8 + (threadIdx.x >> 6));
            d[0] = (uint32_t)busy; d[1] = (uint32_t)wait; d[2] = (uint32_t)phases; d[3] = 0x51u;
            d[4] = (uint32_t)sec[0]; d[5] = (uint32_t)sec[1]; d[6] = (uint32_t)sec[2]; d[7] = 0;
        }
    }
#else
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void arrive() {}
    __device__ __forceinline__ void leave() {}
    __device__ __forceinline__ void flush(void *, long long, int) const {}
#endif
};

__device__ __forceinline__ i32x4 as_i32x4(uint4 v) { return __builtin_bit_cast(i32x4, v); }

// Workgroup barrier that orders LDS traffic only (__syncthreads() would also drain vmcnt, i.e. wait
// for the helper waves' prefetch of the next tile).
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Staging fetches are branch-free: a quad (samples 4*quad .. 4*quad+3 of the tile) is always loaded
// from an address clamped into the valid samples [lo, hi - 4] (tile-relative), so no load sits in
// divergent control flow -- hipcc answers loads on two paths into the same registers with
// s_waitcnt vmcnt(0) before every load, which serialises the HBM round trips.  Only tiles that
// touch an end of the stream (wave-uniform test) then repair the quads in registers: samples
// outside the stream read as zero.
// n-th tile of this workgroup.  A round is gridDim.x consecutive tiles; within a round XCD x (the
// workgroups with blockIdx.x % 8 == x) takes the x-th contiguous eighth, so neighbouring tiles --
// whose windows overlap by the filter length -- meet in one L2.
__device__ __forceinline__ int i8_tile_index(int n)
{
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const int slot = (G & 7) ? b : (b & 7) * (G >> 3) + (b >> 3);
    return n * G + slot;
}

// channel of tile t (one scalar compare instead of a division when the launch has a single channel)
__device__ __forceinline__ int i8_tile_channel(const FirGeom &g, int t)
{
    return g.total_tiles == g.tiles_per_ch ? 0 : t / g.tiles_per_ch;
}

// Tile hand-out.  Static: the n-th tile of a workgroup is i8_tile_index(n).  Dynamic (the DYN
// instantiations, on a grid of a multiple of eight workgroups): the workgroups of XCD x draw the tiles of that same sequence
// -- the x-th eighth of every round, in order -- from one counter per XCD, so a compute unit that runs
// slow (the memory channels are not equally near to all of them) takes fewer tiles instead of holding
// the launch up; a bare persistent reader gains 3.5 % that way (tools/ubench/stream_read.hip).  Results
// do not depend on who computes a tile.  Counters: one per 128-byte line, qctr[32 * x], and the count
// of finished workgroups at qctr[32 * 8]; the last workgroup to finish zeroes them for the next launch.
constexpr int kI8QueueWords = 32 * 9;
// one lane: `count` consecutive places in this XCD's sequence (dynamic) -- returns the first
__device__ __forceinline__ unsigned i8_draw(unsigned *qctr, unsigned count)
{
    return atomicAdd(&qctr[32 * (blockIdx.x & 7)], count);
}
// tile at place q of this XCD's sequence, or -1 past the end (G = gridDim.x, x = blockIdx.x & 7: kept in
// scalar registers by the caller -- read where they are used, the dispatch packet is loaded again in
// every phase, and the wait for that load also waits for whatever LDS read is in flight)
__device__ __forceinline__ int i8_tile_at(const FirGeom &g, unsigned G, unsigned x, unsigned q)
{
    const unsigned G8 = G >> 3, r = q / G8;
    const long long t = (long long)r * G + x * G8 + (q - r * G8);
    return t < g.total_tiles ? (int)t : -1;
}

struct I8Tile {
    const unsigned char *vbase;   // the tile's first valid sample (inside the caller's buffer; wave-uniform)
    int lo, hi;                   // valid samples of the tile, tile-relative: [lo, hi)
    int span;                     // loads start 0 .. span bytes behind vbase: the last one ends at sample hi
    bool edge;                    // some staged sample is outside [lo, hi)
};

// clamp of a 64-bit difference into [0, n] with scalar 32-bit pieces (a 64-bit signed compare would go
// through the vector unit)
__device__ __forceinline__ int i8_clamp_diff(long long d, int n)
{
    const int top = (int)(d >> 32);
    const unsigned low = (unsigned)d;
    return top < 0 ? 0 : ((top > 0 || low > (unsigned)n) ? n : (int)low);
}

// S8 = the samples are int8 pairs (2 bytes, SDR_IN_I8_IQ) instead of int16 pairs (4 bytes): a 16-byte
// load then holds eight samples instead of four
template <bool S8>
__device__ __forceinline__ I8Tile i8_tile(const FirGeom &g, const uint32_t *__restrict__ in, int t)
{
    constexpr int BPS = S8 ? 2 : 4, SPL = 16 / BPS;
    const int ch = i8_tile_channel(g, t);
    const long long W0 = (g.tile0 + (long long)(t - ch * g.tiles_per_ch) * g.per_tile) * g.D - g.Tback;
    const int n = 4 * g.s_gran;
    const int lo = i8_clamp_diff(g.lo_valid - W0, n);
    int hi = i8_clamp_diff(g.hi_valid - W0, n);
    hi = hi < lo ? lo : hi;
    I8Tile r;
    r.vbase = (const unsigned char *)in + BPS * ((long long)ch * g.in_stride + W0 + lo);
    r.lo = lo;
    r.hi = hi;
    r.span = BPS * max(hi - SPL - lo, 0);                 // the host refuses streams shorter than one load
    r.edge = lo > 0 || hi < n;
    return r;
}

__device__ __forceinline__ int i8_med3(int x, int lo, int hi) { return min(max(x, lo), hi); }   // v_med3_i32 (lo <= hi)

// The 16 bytes from tile-relative sample SPL*unit, read at the nearest position inside the valid
// samples: two vector instructions and a load with a scalar base.
template <bool S8>
__device__ __forceinline__ uint4 i8_fetch(const I8Tile &tl, int unit)
{
    constexpr int BPS = S8 ? 2 : 4;
    const int off = i8_med3(16 * unit - BPS * tl.lo, 0, tl.span);
    return *(const uint4 *)(tl.vbase + (unsigned)off);
}

// edge tiles: w was loaded at the clamped position; pick / zero the samples the unit asked for
template <bool S8>
__device__ __forceinline__ uint4 i8_repair(const I8Tile &tl, int unit, uint4 w)
{
    constexpr int BPS = S8 ? 2 : 4, SPL = 16 / BPS;
    const int rel = SPL * unit;
    const int at = tl.lo + i8_med3(16 * unit - BPS * tl.lo, 0, tl.span) / BPS;
    const uint32_t e[4] = {w.x, w.y, w.z, w.w};
    uint32_t o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < SPL; ++u) {
        const int si = rel + u, idx = si - at;              // sample idx of the loaded 16 bytes, if it is in there
        const int dw = S8 ? idx >> 1 : idx;
        uint32_t v = e[0];
        v = dw == 1 ? e[1] : v;
        v = dw == 2 ? e[2] : v;
        v = dw == 3 ? e[3] : v;
        if (S8) v = (idx & 1) ? v >> 16 : v & 0xffffu;
        v = (si >= tl.lo && si < tl.hi && idx >= 0 && idx < SPL) ? v : 0u;
        if (S8) o[u >> 1] |= v << (16 * (u & 1));
        else o[u] = v;
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// eight int8 samples (I | Q << 8 each) -> two dwords per byte plane: {I 0..3, I 4..7}, {Q 0..3, Q 4..7}
__device__ __forceinline__ void i8_split8(uint4 w, uint2 &pi, uint2 &pq)
{
    pi = make_uint2(__builtin_amdgcn_perm(w.y, w.x, 0x06040200u), __builtin_amdgcn_perm(w.w, w.z, 0x06040200u));
    pq = make_uint2(__builtin_amdgcn_perm(w.y, w.x, 0x07050301u), __builtin_amdgcn_perm(w.w, w.z, 0x07050301u));
}

// four samples (I | Q << 16 each) -> one dword per byte plane: {I lo', I hi, Q lo', Q hi}
__device__ __forceinline__ void i8_split4(uint4 w, uint32_t &il, uint32_t &ih, uint32_t &ql, uint32_t &qh)
{
    // v_perm_b32(s0, s1, sel): byte n of {s0 (7..4), s1 (3..0)}
    const uint32_t a01 = __builtin_amdgcn_perm(w.y, w.x, 0x05010400u);   // I0l I1l I0h I1h
    const uint32_t a23 = __builtin_amdgcn_perm(w.w, w.z, 0x05010400u);
    const uint32_t c01 = __builtin_amdgcn_perm(w.y, w.x, 0x07030602u);   // Q0l Q1l Q0h Q1h
    const uint32_t c23 = __builtin_amdgcn_perm(w.w, w.z, 0x07030602u);
    il = __builtin_amdgcn_perm(a23, a01, 0x05040100u) ^ 0x80808080u;
    ih = __builtin_amdgcn_perm(a23, a01, 0x07060302u);
    ql = __builtin_amdgcn_perm(c23, c01, 0x05040100u) ^ 0x80808080u;
    qh = __builtin_amdgcn_perm(c23, c01, 0x07060302u);
}

// One filter output y[m] with float32 taps, computed by a whole wave (every argument wave-uniform, all
// 64 lanes active): lane l takes taps l, l + 64, ..., the partial sums meet in a butterfly.  One fixed
// summation order per output, so the value depends neither on the tile nor on the shard that asks.
// This is the arithmetic of the vector-unit kernels (float32 taps, Q11 / Q7 folded in), and what
// fir_mfma_i8_kernel falls back to for the few outputs its fixed-point taps cannot decide (below).
// src = sample 0 of the channel; samples outside [lo_valid, hi_valid) read as zero.
struct I8Chan {
    const void *src;
    const float *taps;
    long long lo_valid, hi_valid;
    double fr;              // channelizer plans: the channel's NCO frequency (the sample is mixed as the reference
                            // mixes it, phase = fr * n reduced in double, bladerf_apt_demod.cpp:488); else 0
    int D, T, swap_iq;
};

template <bool S8>
__device__ __forceinline__ float2 i8_y_f32(const I8Chan &c, long long m, int lane)
{
    // one sample as (I, Q) floats: an int16 pair (either order) or an int8 pair
    auto sample = [&](long long s) {
        if constexpr (S8) {
            const unsigned w = ((const unsigned short *)c.src)[s];
            return make_float2((float)(signed char)(w & 255u), (float)(signed char)(w >> 8));
        } else {
            const uint32_t w = ((const uint32_t *)c.src)[s];
            return c.swap_iq ? make_float2(hi16(w), lo16(w)) : make_float2(lo16(w), hi16(w));
        }
    };
    float2 a = make_float2(0.f, 0.f);
    const long long n0 = m * c.D;
    for (int k = lane; k < c.T; k += 64) {
        const float h = c.taps[k];
        const long long s = n0 - k;
        if (s >= c.lo_valid && s < c.hi_valid) {
            float2 x = sample(s);
            if (c.fr != 0.0) x = mix(x, c.fr, s);
            a.x = __fmaf_rn(h, x.x, a.x);
            a.y = __fmaf_rn(h, x.y, a.y);
        }
    }
    return wave_sum(a);
}

template <bool S8>
__device__ __forceinline__ I8Chan i8_chan(const uint32_t *__restrict__ in, const float *__restrict__ taps, const FirGeom &g, long long ch)
{
    I8Chan c;
    c.src = (const unsigned char *)in + (S8 ? 2 : 4) * (ch * g.in_stride);
    c.taps = taps;
    c.lo_valid = g.lo_valid; c.hi_valid = g.hi_valid;
    c.fr = g.chan_fr ? g.chan_fr[chan_index(g, ch)] : 0.0;
    c.D = g.D; c.T = g.T; c.swap_iq = g.swap_iq;
    return c;
}

// The discriminator outputs of a stream's start-up transient, computed with float32 taps by the
// helper waves of fir_mfma_i8_kernel while the pipeline fills (they have nothing to finish in
// its first two phases).  There |y| is small and the taps below the matrix kernel's fixed-point
// resolution (2^-31 of the largest tap: the numerically-zero taps of a windowed sinc) decide the
// angle of an output whose other y is exactly zero, which the reference reports as 0 or +-pi by
// the signs of its zeros.  One wave per output.  A few dozen outputs per stream start.
template <bool S8>
__device__ __forceinline__ void i8_startup_fm_output(const uint32_t *__restrict__ in, float *__restrict__ out,
                                                     const float *__restrict__ taps, const FirGeom &g,
                                                     long long m, long long ch, int lane)
{
    const I8Chan c = i8_chan<S8>(in, taps, g, ch);
    const float2 a0 = i8_y_f32<S8>(c, m, lane);
    const float2 a1 = i8_y_f32<S8>(c, m + 1, lane);
    if (lane == 0) out[ch * g.out_stride + (m - g.out_begin)] = discriminate(a1, a0, g.scale);
}

// Mid-stream twin of the start-up case: after a run of >= T zero samples the first output whose window
// meets a non-zero sample only through taps below the fixed-point resolution is an exact 0 here and a
// tiny vector in the reference -- and the discriminator reports the ANGLE of that vector (and, one
// output earlier, 0 or +-pi by its signs).  Every y that came out of the matrix pipe as an exact zero
// and feeds a stored discriminator output is therefore looked at again: if the samples under the wave's
// outputs are not all zero, the flagged y are recomputed with float32 taps (i8_y_f32), exactly the
// value the start-up path and the vector-unit kernels give.  An all-zero window gives +0 either way, so
// long runs of silence cost one scan of their samples (L2 hits: the tile was staged two phases ago)
// and no recomputation.  Lane F of the wave holds y[m0 + r], r < RH (m0 = the wave's first output +
// RH*F) and their successor y[m0 + RH] (slot RH).  Stored outputs: store_from <= j < out_end.
// A real function, not inlined: it runs once in a blue moon, and inlined its registers (and the scalar
// registers of its loops) crowd the matrix waves' tap digits -- 75 scalar spills in the headline kernel.
// Complex taps (CPX, the channelizer): the matrix pipe yields z[m] = sum_i (h[i] e^{-jwi}) x[mD - i], i.e.
// y[m] without the mixer's phasor e^{jwmD} of the output's own sample, and the fast path turns the
// discriminator's product by the constant e^{jwD} instead of turning every y.  A lane that holds a flagged
// zero leaves that shortcut: its slots come back as the reference's y -- recomputed (mixed sample by sample,
// as the reference mixes) where flagged, turned by their own phasor elsewhere -- and ydom tells the caller
// to take the plain discriminator for that lane.
template <int RH> struct I8Ys { float2 y[RH + 1]; int ydom; };

template <int RH, bool S8, bool CPX>
__device__ __noinline__ I8Ys<RH> i8_zero_fix(I8Chan c, long long m0, long long store_from, long long out_end, I8Ys<RH> v)
{
    v.ydom = 0;
    const int lane = threadIdx.x & 63;
    bool flag[RH + 1];
    bool any = false;
#pragma unroll
    for (int r = 0; r <= RH; ++r) {
        // d[j] = angle(y[j+1] conj y[j]): slot r serves d[m0 + r] (r < RH) and d[m0 + r - 1] (r > 0)
        const long long j = m0 + r;
        const bool used = (r < RH && j >= store_from && j < out_end) || (r > 0 && j - 1 >= store_from && j - 1 < out_end);
        flag[r] = used && v.y[r].x == 0.f && v.y[r].y == 0.f;
        any |= flag[r];
    }
    if (__ballot(any) == 0ULL) return v;
    // the samples under all of the wave's outputs: [D*m_first - (T-1), D*(m_first + 64*RH)], valid part
    const long long m_first = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(m0 >> 32)) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)m0));
    long long s_lo = m_first * c.D - (c.T - 1), s_hi = (m_first + 64 * RH) * c.D + 1;
    s_lo = s_lo < c.lo_valid ? c.lo_valid : s_lo;
    s_hi = s_hi > c.hi_valid ? c.hi_valid : s_hi;
    unsigned seen = 0u;
    if constexpr (S8) {
        for (long long s = s_lo + lane; s < s_hi; s += 64) seen |= ((const unsigned short *)c.src)[s];
    } else {
        for (long long s = s_lo + lane; s < s_hi; s += 64) seen |= ((const uint32_t *)c.src)[s];
    }
    if (__ballot(seen != 0u) == 0ULL) return v;
#pragma unroll
    for (int r = 0; r <= RH; ++r) {
        unsigned long long todo = __ballot(flag[r]);
        while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            const float2 yv = i8_y_f32<S8>(c, m_first + (long long)RH * l + r, lane);
            if (lane == l) v.y[r] = yv;
        }
    }
    if constexpr (CPX) {
        if (any) {
            v.ydom = 1;
#pragma unroll
            for (int r = 0; r <= RH; ++r) {
                if (flag[r]) continue;                     // recomputed above: already the reference's y
                const double ph = c.fr * (double)((m0 + r) * c.D);
                float2 P;
                sincospif(2.0f * (float)(ph - rint(ph)), &P.y, &P.x);
                v.y[r] = cmulf(v.y[r], P);
            }
        }
    }
    return v;
}

// Finishing of tile t (computed in the previous phase, outputs in ybuf[par]):
// finishing lane F takes outputs RH*F .. RH*F + RH-1 of the tile and, for the discriminator, their
// successor.  A column block (128 outputs) is kI8PartSlots float4 slots, component 0's outputs in order
// from slot 0, component 1's from slot 33.
template <int RH, int NCB, bool S8>
__device__ __forceinline__ void i8_finish(const FirGeom &g, void *__restrict__ outv, const float *ybuf, int par, int F, int t,
                                          const uint32_t *__restrict__ in, const float *__restrict__ ftaps)
{
    constexpr int TILE = 128 * NCB;
    const int o0 = RH * F, o2 = o0 + RH < TILE ? o0 + RH : TILE - 1;
    const float *pa = ybuf + par * NCB * (4 * kI8PartSlots) + (o0 >> 7) * (4 * kI8PartSlots) + (o0 & 127);
    const float *pb = ybuf + par * NCB * (4 * kI8PartSlots) + (o2 >> 7) * (4 * kI8PartSlots) + (o2 & 127);
    const int ch = i8_tile_channel(g, t);
    const long long M0 = g.tile0 + (long long)(t - ch * g.tiles_per_ch) * g.per_tile;
    if (RH > 2 && g.mode == SDR_MODE_IQ) {
        // complex outputs need no neighbour: lane F takes the pairs 2*F + j * (2 * lanes), so that every
        // store instruction of a wave writes 1 KB of consecutive bytes (RH consecutive outputs per lane
        // are 8 * RH bytes apart between lanes, and those partial lines cost 2x on the int8 kernel's
        // 4-output lanes: 0.43 against 0.21 ms per 2^28 samples at 64 taps / 10)
        constexpr int NL = TILE / RH;
#pragma unroll
        for (int jp = 0; jp < RH / 2; ++jp) {
            const int o = 2 * NL * jp + 2 * F;
            const float *pp = ybuf + par * NCB * (4 * kI8PartSlots) + (o >> 7) * (4 * kI8PartSlots) + (o & 127);
            const float2 qi = *(const float2 *)pp, qq = *(const float2 *)(pp + 4 * 33);
            const float2 yy[2] = {make_float2(qi.x, qq.x), make_float2(qi.y, qq.y)};
            store_outputs<2>(g, outv, ch, M0 + (long long)o, yy, yy[1], M0 + g.per_tile);
        }
        return;
    }
    float2 y[RH], ynext;
    if constexpr (RH == 4) {
        const float4 qi = *(const float4 *)pa, qq = *(const float4 *)(pa + 4 * 33);
        y[0] = make_float2(qi.x, qq.x); y[1] = make_float2(qi.y, qq.y);
        y[2] = make_float2(qi.z, qq.z); y[3] = make_float2(qi.w, qq.w);
    } else if constexpr (RH == 2) {
        const float2 qi = *(const float2 *)pa, qq = *(const float2 *)(pa + 4 * 33);
        y[0] = make_float2(qi.x, qq.x); y[1] = make_float2(qi.y, qq.y);
    } else {
        y[0] = make_float2(pa[0], pa[4 * 33]);
    }
    ynext = make_float2(pb[0], pb[4 * 33]);
    // Plans whose taps include one that the fixed point rounds to zero (FirGeom::zfix, decided at plan
    // creation) look for exact zeros among the y this lane's outputs read: see i8_zero_fix.  The outputs
    // are stored first and stored again in the rare case that a y changes (the same wave's stores to one
    // address keep their order): a branch in front of the discriminator would split the block the
    // scheduler overlaps the LDS reads and the arctangents in.
    store_outputs<RH>(g, outv, ch, M0 + (long long)o0, y, ynext, M0 + g.per_tile);
    if (g.zfix) {
        // the matrix waves' combination never yields -0, so "both components zero" is "both words zero":
        // one OR per output, one unsigned minimum over three of them (in this kernel every vector
        // instruction of a tile is paid for, whichever wave issues it)
        unsigned mn = __float_as_uint(ynext.x) | __float_as_uint(ynext.y);
#pragma unroll
        for (int r = 0; r < RH; ++r) mn = min(mn, __float_as_uint(y[r].x) | __float_as_uint(y[r].y));
        if (__builtin_expect(__ballot(mn == 0u) != 0ULL, 0)) {
            const long long tile_end = M0 + g.per_tile;
            I8Ys<RH> v;
#pragma unroll
            for (int r = 0; r < RH; ++r) v.y[r] = y[r];
            v.y[RH] = ynext;
            v = i8_zero_fix<RH, S8, false>(i8_chan<S8>(in, ftaps, g, ch), M0 + (long long)o0, g.store_from,
                                    tile_end < g.out_end ? tile_end : g.out_end, v);
            bool changed = v.y[RH].x != ynext.x || v.y[RH].y != ynext.y;
#pragma unroll
            for (int r = 0; r < RH; ++r) { changed |= v.y[r].x != y[r].x || v.y[r].y != y[r].y; y[r] = v.y[r]; }
            ynext = v.y[RH];
            if (__ballot(changed) != 0ULL) store_outputs<RH>(g, outv, ch, M0 + (long long)o0, y, ynext, M0 + g.per_tile);
        }
    }
}

// Finishing for complex taps (the channelizer; see I8Ys): ybuf holds z[m], the filter output without the
// mixer's phasor of its own sample, y[m] = e^{jw mD} z[m].
//   envelope       |y| = |z|: nothing to do
//   discriminator  y[m+1] conj(y[m]) = e^{jwD} z[m+1] conj(z[m]): one constant turn E = e^{jwD} per channel
//                  on the product (four multiply-adds per output) instead of a phasor per y; the product
//                  of two exact zeros stays (+0, +0) (the turn's -0 is washed out by adding +0); a lane with
//                  ONE exact zero goes through i8_zero_fix, which brings back the reference's y
//   IQ, real part  y itself is the output: the lane's first phasor from its phase (reduced in double), the
//                  following ones by E
template <int RH, int NCB, bool S8>
__device__ __forceinline__ void i8_finish_cpx(const FirGeom &g, void *__restrict__ outv, const float *ybuf, int par, int F, int t,
                                              const uint32_t *__restrict__ in, const float *__restrict__ ftaps)
{
    constexpr int TILE = 128 * NCB;
    const int o0 = RH * F, o2 = o0 + RH < TILE ? o0 + RH : TILE - 1;
    const float *pa = ybuf + par * NCB * (4 * kI8PartSlots) + (o0 >> 7) * (4 * kI8PartSlots) + (o0 & 127);
    const float *pb = ybuf + par * NCB * (4 * kI8PartSlots) + (o2 >> 7) * (4 * kI8PartSlots) + (o2 & 127);
    const int ch = i8_tile_channel(g, t);
    const long long M0 = g.tile0 + (long long)(t - ch * g.tiles_per_ch) * g.per_tile;
    const long long m0 = M0 + (long long)o0;
    const long long tile_end = M0 + g.per_tile;
    const long long out_end = tile_end < g.out_end ? tile_end : g.out_end;
    const float2 E = g.chan_tab[chan_index(g, ch)];
    float2 y[RH], ynext;
    if constexpr (RH == 4) {
        const float4 qi = *(const float4 *)pa, qq = *(const float4 *)(pa + 4 * 33);
        y[0] = make_float2(qi.x, qq.x); y[1] = make_float2(qi.y, qq.y);
        y[2] = make_float2(qi.z, qq.z); y[3] = make_float2(qi.w, qq.w);
    } else {
        static_assert(RH == 2, "complex-tap tiles have 1024 outputs");
        const float2 qi = *(const float2 *)pa, qq = *(const float2 *)(pa + 4 * 33);
        y[0] = make_float2(qi.x, qq.x); y[1] = make_float2(qi.y, qq.y);
    }
    ynext = make_float2(pb[0], pb[4 * 33]);
    if (g.mode == SDR_MODE_ENV) {
        float v[RH];
#pragma unroll
        for (int r = 0; r < RH; ++r) v[r] = sqrtf(__fmaf_rn(y[r].x, y[r].x, y[r].y * y[r].y));
        store_values<RH>(g, outv, ch, m0, v, out_end);
        return;
    }
    if (g.mode == SDR_MODE_IQ || g.mode == SDR_MODE_REAL) {
        const double ph = g.chan_fr[chan_index(g, ch)] * (double)(m0 * g.D);
        float2 P;
        sincospif(2.0f * (float)(ph - rint(ph)), &P.y, &P.x);
        float2 *dst = (float2 *)outv + (long long)ch * g.out_stride + (m0 - g.out_begin);
        float re[RH];
#pragma unroll
        for (int r = 0; r < RH; ++r) {
            const float2 o = cmulf(y[r], P);
            re[r] = o.x;
            if (g.mode == SDR_MODE_IQ && m0 + r >= g.store_from && m0 + r < out_end) nt_store((float *)(dst + r), o.x, o.y);
            P = cmulf(P, E);
        }
        if (g.mode == SDR_MODE_REAL) store_values<RH>(g, outv, ch, m0, re, out_end);
        return;
    }
    // discriminator
    auto turned = [&](float2 y1, float2 y0) {
        const float a = y1.x, b = y1.y, c = y0.x, e = -y0.y;
        const float re = __fmaf_rn(a, c, -(b * e)), im = __fmaf_rn(a, e, b * c);
        const float tr = __fmaf_rn(re, E.x, -(im * E.y)) + 0.0f, ti = __fmaf_rn(re, E.y, im * E.x) + 0.0f;
        return g.scale * atan2_finite(ti, tr);
    };
    float v[RH];
#pragma unroll
    for (int r = 0; r < RH; ++r) v[r] = turned(r + 1 < RH ? y[r + 1] : ynext, y[r]);
    store_values<RH>(g, outv, ch, m0, v, out_end);
    unsigned mn = __float_as_uint(ynext.x) | __float_as_uint(ynext.y);
#pragma unroll
    for (int r = 0; r < RH; ++r) mn = min(mn, __float_as_uint(y[r].x) | __float_as_uint(y[r].y));
    if (__builtin_expect(__ballot(mn == 0u) != 0ULL, 0)) {
        I8Ys<RH> w;
#pragma unroll
        for (int r = 0; r < RH; ++r) w.y[r] = y[r];
        w.y[RH] = ynext;
        w = i8_zero_fix<RH, S8, true>(i8_chan<S8>(in, ftaps, g, ch), m0, g.store_from, out_end, w);
        if (__ballot(w.ydom != 0) != 0ULL) {
#pragma unroll
            for (int r = 0; r < RH; ++r)
                v[r] = w.ydom ? discriminate(w.y[r + 1], w.y[r], g.scale) : v[r];
            store_values<RH>(g, outv, ch, m0, v, out_end);
        }
    }
}

// NS = k steps of the window, rounded up to a multiple of four (steps past the window carry zero taps
// and are skipped; the staged tile covers NS steps, so no sample read leaves it).
// PRE = octets (8 samples) per helper lane and tile: 256 * PRE * 8 samples cover the staged tile.
// S8 = int8 sample pairs (SDR_IN_I8_IQ): one byte plane per component, four byte products per step
// (every tap digit with the sample byte; weights 2^0 .. 2^24, no low-byte bias), one 16-byte load per octet.
// DYN = tiles are drawn from the per-XCD counters at qctr (see i8_draw) instead of the static order.
// CPX = complex taps, one set per channel (the channelizer: the per-channel mixer folded into the taps,
// g[i] = h[i] e^{-jwi}; see i8_finish_cpx): the digits of Re g and Im g are two A operands, each step runs
// its byte products twice into two sets of accumulators, and the combination subtracts / adds the
// neighbouring column's "imaginary" sums (real output = Gr*Xi - Gi*Xq, imaginary = Gr*Xq + Gi*Xi; the I
// and Q columns of a group sit in adjacent lanes).  A workgroup takes a CONTIGUOUS run of tiles, so it
// changes channel -- and reloads 32 KB of digits -- a handful of times per launch.  Windows of up to
// four steps (NS = 4), tiles of 1024 outputs, static order.
template <int NS, int NCB, int PRE, bool S8, bool DYN, bool CPX = false>
__global__ void __launch_bounds__(512, 1)
fir_mfma_i8_kernel(const uint32_t *__restrict__ in, void *__restrict__ outv,
                   const uint4 *__restrict__ atab, const uint4 *__restrict__ btab, const FirGeom g,
                   const int plane_bytes, const float yscale, const float *__restrict__ ftaps,
                   unsigned *__restrict__ qctr, unsigned *__restrict__ q_done, const unsigned q_gen)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const bool matrix = tid < 256;
    const int buf_bytes = 4 * plane_bytes;
    f32x4 *ybuf = (f32x4 *)(smem + 2 * buf_bytes);             // [tile parity][column block][kI8PartSlots]
    // Phases: this workgroup's n-th tile is staged in phase n, multiplied in n+1, finished in n+2; phases
    // run in pairs, so that the helper waves' loop over their two register sets has no conditional half,
    // and every wave ends on the same test of the same tile numbers (tile_of: -1 past the last tile).
    // Static order: tile_of is arithmetic.  Dynamic order: the number is written to tq[n & 7] four phases
    // ahead by matrix wave 0 (that wave has no loads in flight an atomic's return would have to queue
    // behind), and every wave works from a snapshot of the eight entries (lane i holds tq[i & 7]) that
    // it took at the start of the phase before -- read in the phase that uses them, the LDS round trip
    // would sit in front of every phase's address arithmetic.  The snapshot of phase p-1 holds the tiles
    // up to p+2, all that phase p needs; the entry written meanwhile (tile p+3) is not among them.
    int *tq = (int *)(smem + 2 * buf_bytes + 2 * NCB * kI8PartSlots * 16 + (NS > 8 ? kI8MaxSteps * 1024 : 0));
    unsigned q_drawn = 0;                                      // (lane 0 of wave 0) the place drawn in the previous phase
    unsigned G = gridDim.x, xcd = blockIdx.x & 7;
    static_assert(!CPX || (NS == 4 && NCB == 8 && !DYN), "complex taps: four steps, 1024-output tiles, static order");
    // static order: tile n of this workgroup is n * tstep + slot0 -- every G-th tile, or (complex taps) a
    // contiguous run of ceil(total / G) tiles
    const int run = (g.total_tiles + (int)G - 1) / (int)G;
    const int tstep = CPX ? 1 : (int)G;
    int slot0 = CPX ? (int)blockIdx.x * run : i8_tile_index(0);
    int tq_snap = 0;
    const int n_t = CPX ? max(0, min(run, g.total_tiles - slot0))
                        : (g.total_tiles - slot0 + (int)G - 1) / (int)G;   // static order: this workgroup's tiles ...
    // ... and phases.  (Dynamic order: an upper bound, left early: hipcc schedules the loops with a trip
    // count visibly better than the same loops without one -- 7 % on short phases.)
    const int n_ph_s = DYN ? (g.total_tiles + 4) & ~1 : (n_t + 3) & ~1;
    if constexpr (DYN) {
        asm volatile("" : "+s"(G), "+s"(xcd));                 // (opaque: kept in registers, not loaded again in every phase)
        if (tid == 0) {
            const unsigned q0 = i8_draw(qctr, 5u);
#pragma unroll
            for (int n = 0; n < 8; ++n) tq[n] = n >= 4 ? -1 : i8_tile_at(g, G, xcd, q0 + n);
            q_drawn = q0 + 4;
        }
        __syncthreads();
        tq_snap = tq[tid & 7];
    }
    auto tile_of = [&](int n) -> int {                         // n in p-2 .. p+2
        if constexpr (DYN) return n < 0 ? -1 : __builtin_amdgcn_readlane(tq_snap, n & 7);
        else return (n >= 0 && n < n_t) ? n * tstep + slot0 : -1;
    };
    // who finishes: 0 the matrix waves, 1 the helper waves, 2 both (one half of the tile each; see the matrix waves)
    constexpr int FIN = (NCB == 8 && PRE <= (S8 ? 6 : 4)) ? 2 : 0;
    I8Clock clk;
    int n_ph = 0;

    if (matrix) {
        // four matrix waves, one per SIMD; wave w takes column blocks CBW*w .. CBW*w + CBW-1 of every tile
        // through the whole window, so an output is one exact integer sum: its value depends neither
        // on the tile it falls into nor on the shard
        constexpr int CBW = NCB / 4;
        const int wave = tid >> 6, lane = tid & 63;
        const int j = lane & 15, kq = lane >> 4;
        // the tap digits of the whole window, resident for the whole kernel.  Twelve steps x four digits
        // would be 192 registers and leave no room for the epilogue: the lowest digit, which only one of
        // the seven products reads, then lives in LDS and is fetched with the samples (144 registers).
        constexpr bool A0_LDS = NS > 8;
        constexpr int NP = CPX ? 2 : 1;                        // tap parts: real only | real, imaginary
        i32x4 a[NS][NP][4];
        uint4 *a0lds = (uint4 *)(ybuf + 2 * NCB * kI8PartSlots);
        int ch_loaded = CPX ? -1 : 0;                          // complex taps: the channel whose digits are resident
        if constexpr (!CPX) {
#pragma unroll
            for (int ks = 0; ks < NS; ++ks)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    if (A0_LDS && d == 0) { if (wave == 0) a0lds[ks * 64 + lane] = atab[(ks * 4) * 64 + lane]; }
                    else a[ks][0][d] = as_i32x4(atab[(ks * 4 + d) * 64 + lane]);
                }
        } else {
#pragma unroll
            for (int ks = 0; ks < NS; ++ks)
#pragma unroll
                for (int pp = 0; pp < NP; ++pp)
#pragma unroll
                    for (int d = 0; d < 4; ++d) a[ks][pp][d] = i32x4{0, 0, 0, 0};
        }
        // dynamic order: the digits are waited for here, once -- left to their first use inside the loop,
        // the wait (s_waitcnt vmcnt(0) between two MFMAs) stays in every phase of that loop and then also
        // waits for the acknowledgement of the stores the wave has just issued (the counted loop of the
        // static order is peeled by hipcc, and the wait stays in its first round)
        if constexpr (DYN) {
#pragma unroll
            for (int ks = 0; ks < NS; ++ks)
#pragma unroll
                for (int d = (A0_LDS ? 1 : 0); d < 4; ++d) asm volatile("" ::"v"(a[ks][0][d]));
        }
        // 128 * (sum of the row's digits), combined over the digits exactly as the accumulators are
        // below: it undoes the -128 of the low bytes, and because the combination is the same sequence
        // of float32 operations on the negated integers, zero samples still give an exact zero
        f32x4 bias = (S8 || CPX) ? f32x4{0.f, 0.f, 0.f, 0.f} : __builtin_bit_cast(f32x4, btab[lane]);
        if (DYN && !S8) asm volatile("" ::"v"(bias));          // (waited for here, like the digits above)
        // hi plane of component j&1, this lane's group within a column block, step 0
        const int bofs = (j & 1) * plane_bytes + 16 * g.D * (j >> 1) + 16 * kq + CBW * wave * (16 * g.D * 8);
        const int cb_bytes = 16 * g.D * 8;                     // between column blocks
        // each matrix wave also finishes its own quarter of the tile before the last (R outputs per lane).
        // Measured alternatives, all within 2 %: finishing in the helper waves (they are then the longer
        // role), half in each, and its arithmetic interleaved with the MFMAs by sched_group_barrier --
        // MFMA and vector time of a SIMD add up whichever wave issues them.
        // Narrow tiles leave the helper waves little to stage: there each role finishes one half of the
        // tile (FIN = 2: 512 finishing lanes).  int16 samples: tiles of up to four octets per helper
        // lane (D <= 7; same box, 2^28 samples: 32/2 0.600 -> 0.557 ms, 255/4 0.408 -> 0.392, 700/4
        // 0.616 -> 0.591; at D = 10 it would lose 8 %).  int8 samples -- four products per step instead
        // of seven, half the bytes to split, both roles short: up to six octets (D <= 10; 32/2 0.565 ->
        // 0.495, 255/4 0.341 -> 0.306, 511/10 0.194 -> 0.183; wider tiles lose 5-10 % that way).
        constexpr int R = NCB / 2;
        clk.start();
        int p = 0;
        for (; p < n_ph_s; ++p) {
            const int t_fin = tile_of(p - 2);
            if (DYN && !(p & 1) && p >= 2 && t_fin < 0) break; // both of this pair's tiles lie past the end
            int tq_in = 0;
            if constexpr (DYN) {
                tq_in = tq[tid & 7];
                // the tile staged four phases from now: its place was drawn a phase ago, behind that phase's
                // stores, so the atomic's return has had a whole phase to arrive (a wave's memory operations
                // are waited for in order: asked for at the end of the phase it was drawn in, the return
                // would wait for the stores' acknowledgements too: +7 % on the headline)
                if (wave == 0) {                               // (scalar arithmetic on the drawn place)
                    const int t_new = i8_tile_at(g, G, xcd, (unsigned)__builtin_amdgcn_readfirstlane((int)q_drawn));
                    if (lane == 0) tq[(p + 4) & 7] = t_new;
                }
            }
            // the tile before the last
            if (FIN != 1 && t_fin >= 0 && !(SDR_I8_ABL & 4))
            {
                if constexpr (CPX) i8_finish_cpx<(FIN == 2 ? R / 2 : R), NCB, S8>(g, outv, (const float *)ybuf, p & 1, tid, t_fin, in, ftaps);
                else i8_finish<(FIN == 2 ? R / 2 : R), NCB, S8>(g, outv, (const float *)ybuf, p & 1, tid, t_fin, in, ftaps);
            }
            if (DYN && wave == 0 && lane == 0) q_drawn = i8_draw(qctr, 1u);
            clk.mark(2);
            const bool work = tile_of(p - 1) >= 0 && !(SDR_I8_ABL & 1);
            if constexpr (CPX) {
                // this tile's channel: its digits (and the low-byte bias that goes with them) become resident
                if (work) {
                    const int chm = (int)chan_index(g, i8_tile_channel(g, tile_of(p - 1)));
                    if (chm != ch_loaded) {
                        ch_loaded = chm;
                        const uint4 *ac = atab + (size_t)chm * (NS * NP * 4 * 64) + lane;
#pragma unroll
                        for (int ks = 0; ks < NS; ++ks)
#pragma unroll
                            for (int pp = 0; pp < NP; ++pp)
#pragma unroll
                                for (int d = 0; d < 4; ++d) a[ks][pp][d] = as_i32x4(ac[((ks * NP + pp) * 4 + d) * 64]);
                        if (!S8) bias = __builtin_bit_cast(f32x4, btab[(size_t)chm * 64 + lane]);
                    }
                }
            }
            if (work) {
                const unsigned char *bbase = smem + ((p - 1) & 1) * buf_bytes + bofs;
                // units of (column block, step), samples requested PD units ahead (int8 samples: a unit
                // is four MFMAs, 64 cycles, so the LDS round trip needs the longer lead)
                constexpr int UNITS = CBW * NS;
                constexpr int PD = CPX ? (S8 ? 2 : 1) : (S8 ? 4 : 2), NB = PD + 1;
                i32x4 b[NB][3];                                // [buffer][hi, lo, lowest tap digit]
#pragma unroll
                for (int u = 0; u < PD && u < UNITS; ++u) {
                    const unsigned char *bp = bbase + (u / NS) * cb_bytes + 64 * (u % NS);
                    b[u][0] = as_i32x4(*(const uint4 *)bp);
                    if (!S8) b[u][1] = as_i32x4(*(const uint4 *)(bp + 2 * plane_bytes));
                    if (A0_LDS) b[u][2] = as_i32x4(a0lds[(u % NS) * 64 + lane]);
                }
                i32x4 acc[4];                                  // weights 2^8 .. 2^32
                i32x4 acci[CPX ? 4 : 1];                       // complex taps: the imaginary part's sums
#pragma unroll
                for (int u = 0; u < UNITS; ++u) {
                    const int cb = u / NS, ks = u % NS;
                    if (u + PD < UNITS && !(SDR_I8_ABL & 32)) {
                        const unsigned char *bp = bbase + ((u + PD) / NS) * cb_bytes + 64 * ((u + PD) % NS);
                        b[(u + PD) % NB][0] = as_i32x4(*(const uint4 *)bp);
                        if (!S8) b[(u + PD) % NB][1] = as_i32x4(*(const uint4 *)(bp + 2 * plane_bytes));
                        if (A0_LDS) b[(u + PD) % NB][2] = as_i32x4(a0lds[((u + PD) % NS) * 64 + lane]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks == 0) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) acc[w] = i32x4{0, 0, 0, 0};
                        if constexpr (CPX) {
#pragma unroll
                            for (int w = 0; w < 4; ++w) acci[w] = i32x4{0, 0, 0, 0};
                        }
                    }
                    if (ks < NS - 3 || ks < g.ngran) {         // wave-uniform
                        const i32x4 xh = b[u % NB][0];
                        const i32x4 a0 = A0_LDS ? b[u % NB][2] : a[ks][0][0];
                        if constexpr (S8) {
                            // one byte per sample: digit w x sample, accumulator w
                            acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, xh, acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks][0][1], xh, acc[1], 0, 0, 0);
                            acc[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks][0][2], xh, acc[2], 0, 0, 0);
                            acc[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks][0][3], xh, acc[3], 0, 0, 0);
                            if constexpr (CPX) {
#pragma unroll
                                for (int w = 0; w < 4; ++w) acci[w] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks][NP - 1][w], xh, acci[w], 0, 0, 0);
                            }
                        } else {
                        const i32x4 xl = b[u % NB][1];
                        // weight w = digit + (hi ? 1 : 0), accumulator w - 1; the weight-0 product (lowest
                        // digit x low byte, < 2^-30 of full scale) is left out: seven products, not eight.
                        // No accumulator twice in a row.
                        acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks][0][1], xl, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks][0][2], xl, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks][0][3], xl, acc[2], 0, 0, 0);
                        acc[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks][0][3], xh, acc[3], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, xh, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks][0][1], xh, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks][0][2], xh, acc[2], 0, 0, 0);
                        if constexpr (CPX) {
                            acci[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks][NP - 1][1], xl, acci[0], 0, 0, 0);
                            acci[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks][NP - 1][2], xl, acci[1], 0, 0, 0);
                            acci[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks][NP - 1][3], xl, acci[2], 0, 0, 0);
                            acci[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks][NP - 1][3], xh, acci[3], 0, 0, 0);
                            acci[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks][NP - 1][0], xh, acci[0], 0, 0, 0);
                            acci[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks][NP - 1][1], xh, acci[1], 0, 0, 0);
                            acci[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks][NP - 1][2], xh, acci[2], 0, 0, 0);
                        }
                        }
                    }
                    if (ks == NS - 1) {
                        // y = yscale * sum of 2^(8w) * acc[w] (the common 2^8 sits in yscale): the only
                        // floating-point step, a fixed sequence on exact integers.  Balanced digits: the
                        // top term carries the value, the rest is a correction -- nothing cancels.
                        if constexpr (CPX) {
                            // real output (the I column, even lane) = Gr*Xi - Gi*Xq, imaginary (the Q column,
                            // odd lane) = Gr*Xq + Gi*Xi: the other column's "imaginary" sums come from the
                            // adjacent lane (quad_perm 1,0,3,2), exact in int32
#pragma unroll
                            for (int w = 0; w < 4; ++w)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int other = __builtin_amdgcn_mov_dpp(acci[w][r], 0xB1, 0xF, 0xF, true);
                                    acc[w][r] = (j & 1) ? acc[w][r] + other : acc[w][r] - other;
                                }
                        }
                        f32x4 rv;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (SDR_I8_ABL & 16) { rv[r] = __int_as_float(acc[0][r] ^ acc[1][r] ^ acc[2][r] ^ acc[3][r]); continue; }
                            const float lo = __fmaf_rn((float)acc[1][r], 256.0f, (float)acc[0][r]);
                            const float hi = __fmaf_rn((float)acc[3][r], 256.0f, (float)acc[2][r]);
                            rv[r] = (__fmaf_rn(hi, 65536.0f, lo) + bias[r]) * yscale;
                        }
                        // within a column block (66 slots of 16 bytes) the 128 outputs of each component are
                        // stored in output order, component 1 starting at slot 33: the helper lanes read
                        // consecutive slots, and the odd component offset keeps these stores down to a
                        // two-way bank conflict, which hides behind the store's register transfer
                        ybuf[(((p - 1) & 1) * NCB + CBW * wave + cb) * kI8PartSlots + (j & 1) * 33 + (j >> 1) * 4 + kq] = rv;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (DYN) tq_snap = tq_in;
            clk.arrive();
            lds_only_barrier();
            clk.leave();
        }
        n_ph = p;
    } else {
        // ================= helper waves: stage tile p, request tile p+2 ===========================
        const int st = tid - 256;                              // 0..255
        const int cI = g.swap_iq ? 1 : 0;                      // Q-first input swaps the component planes
        const int n_oct = (g.s_gran + 1) / 2;                  // octets per tile
        // raw samples of two tiles travel at any time: set p&1 holds tile p until it is staged in
        // phase p and is then refilled with tile p+2
        constexpr int kI8Pre = PRE;                            // octets per lane and tile held in registers
        constexpr int LPO = S8 ? 1 : 2;                        // 16-byte loads per octet
        uint4 w0[kI8Pre][LPO], w1[kI8Pre][LPO];
        // n-th tile of this workgroup, and its descriptor kept (in scalar registers) until it is staged.
        // Unconditional (past the end: the launch's first tile, unused): a skipped request would make the
        // registers conditionally defined and hipcc would copy them right after the loads
        auto request = [&](int t, uint4 (&w)[kI8Pre][LPO], I8Tile &tl) {
            tl = i8_tile<S8>(g, in, t < 0 ? 0 : t);
#pragma unroll
            for (int k = 0; k < kI8Pre; ++k)
#pragma unroll
                for (int h = 0; h < LPO; ++h) w[k][h] = i8_fetch<S8>(tl, LPO * (256 * k + st) + h);
        };
        auto phase = [&](int p, uint4 (&w)[kI8Pre][LPO], I8Tile &tl) {
            const int t_req = tile_of(p + 2), t_fin = tile_of(p - 2);
            int tq_in = 0;
            if constexpr (DYN) tq_in = tq[tid & 7];
            // (past the last tile: the last one again, into a buffer nobody reads any more -- a skipped
            // staging would leave the set's loads pending on one path into the request below, and hipcc
            // would then wait for every outstanding load, the other set's included, on all paths)
            if (!(SDR_I8_ABL & 2)) {
                unsigned char *buf = smem + (p & 1) * buf_bytes;
                auto put = [&](int oct, uint4 q0, uint4 q1) {
                    unsigned char *dst = buf + 8 * oct;
                    if constexpr (S8) {                        // q0 holds the octet; the bytes are the planes
                        uint2 pi, pq;
                        i8_split8(q0, pi, pq);
                        *(uint2 *)dst = pi;
                        *(uint2 *)(dst + plane_bytes) = pq;
                        return;
                    }
                    uint32_t il[2], ih[2], ql[2], qh[2];
                    i8_split4(q0, il[0], ih[0], ql[0], qh[0]);
                    i8_split4(q1, il[1], ih[1], ql[1], qh[1]);
                    *(uint2 *)(dst + cI * plane_bytes) = make_uint2(ih[0], ih[1]);
                    *(uint2 *)(dst + (1 - cI) * plane_bytes) = make_uint2(qh[0], qh[1]);
                    *(uint2 *)(dst + (2 + cI) * plane_bytes) = make_uint2(il[0], il[1]);
                    *(uint2 *)(dst + (3 - cI) * plane_bytes) = make_uint2(ql[0], ql[1]);
                };
                // two copies of the loop, not one with a conditional repair inside: a value that is either
                // the loaded quad or its repaired form becomes a phi of single registers, which cannot
                // share the load's register tuple, and hipcc then copies every quad (after waiting for
                // it) at the end of each phase -- the end of all prefetching
                if (tl.edge) {                                 // wave-uniform
#pragma unroll
                    for (int k = 0; k < kI8Pre; ++k)
                        if (256 * k + st < n_oct)
                            put(256 * k + st, i8_repair<S8>(tl, LPO * (256 * k + st), w[k][0]),
                                i8_repair<S8>(tl, LPO * (256 * k + st) + LPO - 1, w[k][LPO - 1]));
                } else {
#pragma unroll
                    for (int k = 0; k < kI8Pre; ++k)
                        if (256 * k + st < n_oct) put(256 * k + st, w[k][0], w[k][LPO - 1]);
                }
            }
            clk.mark(0);
            request(t_req, w, tl);
            clk.mark(1);
            if (FIN != 0 && t_fin >= 0 && !(SDR_I8_ABL & 4))
            {
                if constexpr (CPX) i8_finish_cpx<(FIN == 2 ? NCB / 4 : NCB / 2), NCB, S8>(g, outv, (const float *)ybuf, p & 1, FIN == 2 ? tid : st, t_fin, in, ftaps);
                else i8_finish<(FIN == 2 ? NCB / 4 : NCB / 2), NCB, S8>(g, outv, (const float *)ybuf, p & 1, FIN == 2 ? tid : st, t_fin, in, ftaps);
            }
            if constexpr (DYN) tq_snap = tq_in;
            clk.arrive();
            lds_only_barrier();
            clk.leave();
        };
        I8Tile tl0, tl1;
        request(tile_of(0), w0, tl0);
        request(tile_of(1), w1, tl1);
        if (ftaps && g.store_from > g.out_begin) {
            // outputs [out_begin, store_from) of every channel: see i8_startup_fm_output; dealt over the
            // helper waves of the whole grid, done behind the first two requests (their loads are in flight meanwhile)
            const long long n_su = g.store_from - g.out_begin, total = n_su * (g.total_tiles / g.tiles_per_ch);
            for (long long idx = 4LL * blockIdx.x + (st >> 6); idx < total; idx += 4LL * gridDim.x) {
                const long long ch = idx / n_su;
                i8_startup_fm_output<S8>(in, (float *)outv, ftaps, g, g.out_begin + (idx - ch * n_su), ch, st & 63);
            }
        }
        clk.start();
        int p = 0;
        for (; p < n_ph_s && (!DYN || p < 2 || tile_of(p - 2) >= 0); p += 2) {
            phase(p, w0, tl0);
            phase(p + 1, w1, tl1);
        }
        n_ph = p;
    }
    // the last workgroup to get here zeroes the counters for the next launch (every draw of a workgroup
    // precedes its own count here) and then tells the host, through a word of pinned host memory, that
    // this launch (generation q_gen) has let go of the block: only then is the block handed to a launch
    // on another stream (launch_r)
    if (DYN && tid == 0) {
        __threadfence();
        if (atomicAdd(&qctr[32 * 8], 1u) == gridDim.x - 1) {
#pragma unroll
            for (int x = 0; x < 9; ++x) atomicExch(&qctr[32 * x], 0u);
            __threadfence_system();
            __hip_atomic_store(q_done, q_gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    clk.flush(outv, g.out_end - g.out_begin, n_ph);
}

}  // namespace

namespace sdr {
#ifdef SDR_MEASUREMENT_BUILD
int measurement_flags_fir() { return SDR_I8_ABL ? SDR_I8_ABL : -1; }
#else
int measurement_flags_fir() { return 0; }
#endif
}  // namespace sdr

// =============================================================================================
// host side
// =============================================================================================
struct sdr_fir_plan {
    int input, mode, D, T, R, NT;
    int count_rule = SDR_COUNT_FLOOR;   // sdr_count_rule
    bool raw;               // raw-tile geometry (4 samples per granule): fir_raw_kernel or fir_planar_kernel
    bool planar = false;    // fir_planar_kernel (float planes, convert once)
    bool mfma = false;      // fir_mfma_kernel (float planes, Toeplitz GEMM on the matrix cores)
    float4 *d_atab = nullptr;   // its A-operand table
    bool ksplit = false;    // fir_ksplit_kernel (real input, short streams: taps split over the four waves of a workgroup)
    bool mfma_i8 = false;   // fir_mfma_i8_kernel (byte-split Toeplitz GEMM on the int8 matrix pipe, persistent)
    bool cpx = false;       // ... with complex taps per channel (the channelizer's mixers folded into the taps)
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

namespace {

// the dynamic LDS limit is a property of the kernel, not of the plan: raise it to the whole CU's
// 160 KB so that plans with different tiles can share a kernel in any order
constexpr int kLdsPerCu = 160 * 1024;

int roundup(int v, int m) { return (v + m - 1) / m * m; }

struct Geometry { int Tback, Lw4, U, ngran, look, q, pad, s_gran, a_gran, n_inst, lds_bytes; };

// Tile geometry for (T, D, R, NT) with spg samples per LDS granule.  Returns false when the lane
// stride is not a whole number of granules.
// entries of the coarse mixer phasor table for a tile of s_gran 2-sample granules (even count)
int ph_hi_entries(int s_gran) { return ((2 * s_gran) / 64 + 2 + 1) & ~1; }

bool make_geometry(int T, int D, int R, int NT, int spg, bool stream, Geometry *o, bool planar = false, bool mixer = false,
                   int step_mult = 2)
{
    if (((long long)D * R) % spg) return false;
    o->Tback = roundup(T - 1, 4);
    o->Lw4 = roundup(o->Tback + D * (R - 1) + 1, 4);
    o->U = (spg * R >= 16) ? 1 : 16 / (spg * R);
    const int steps = roundup((o->Lw4 / spg + o->U - 1) / o->U, step_mult);
    o->ngran = steps * o->U;                  // granules walked (padded with zero taps)
    o->look = o->U;                           // one look-ahead step is loaded, never used
    o->q = D * R / spg;
    o->pad = (o->q % 2 == 0) ? 1 : 0;
    const long long s_need = (long long)D * R * (NT - 1) + (long long)(o->ngran + o->look) * spg;
    const long long s_tile = s_need;
    if (s_tile > (1LL << 22)) return false;
    o->s_gran = (int)((s_tile + spg - 1) / spg);
    const int a = o->s_gran + (o->pad ? o->s_gran / o->q + 1 : 0);
    // the persistent streaming kernel stages whole rounds of NT granules; the one-tile kernels
    // mask the last round, so their image is exact (64 taps / 10: 22.8 KB, a seventh workgroup
    // per CU fits in the 160 KB)
    o->a_gran = stream ? roundup(a, NT) : a;
    o->n_inst = (o->a_gran + NT - 1) / NT;
    o->lds_bytes = o->a_gran * (planar ? 32 : 16) + (NT + 2) * 8;
    if (mixer) o->lds_bytes += (ph_hi_entries(o->s_gran) + 64) * 8;
    return true;
}

size_t in_elem_bytes(int input) { return sdr::in_elem_bytes(input); }
size_t out_elem_bytes(int mode) { return mode == SDR_MODE_IQ ? 8 : 4; }

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

template <int R>
hipError_t launch_r(const sdr_fir_plan *p, hipStream_t st, const void *in, void *out,
                    const FirGeom &g, unsigned tiles, unsigned n_ch, const char **what)
{
    *what = "fir kernel launch";
    if (p->mfma_i8) {
        const unsigned nb = (unsigned)(g.total_tiles < p->stream_blocks ? g.total_tiles : p->stream_blocks);
        // the helper lanes hold a whole staged tile in registers: 4, 6 or 8 octets each (plan creation
        // refuses tiles beyond 256 * 8 octets)
        const int pre = ((p->s_gran + 1) / 2 + 255) / 256;
        // Dynamic tile hand-out for long launches (128 tiles per workgroup and more: the static order's
        // imbalance grows with the launch, 0 / 2 / 5 / 9 % at 2^27 / 2^28 / 2^29 / 2^30 samples of the
        // headline shape) of tiles of 1024 outputs and more than four octets per helper lane (D >= 8;
        // shorter phases do not hide the queue's upkeep) with windows of up to four steps (256 samples:
        // the shapes the sample stream bounds.  Where the matrix pipe is the bound every compute unit
        // takes the same time per tile and the queue only costs: +2 % at 511 taps / 10, -1.5 .. +3 % at
        // 200 / 10 from box to box; int8 sample pairs, half the bytes per tile: +-1 %), on a grid of whole
        // XCD groups.  int16 shapes inside the rule, 2^30 samples, same box: 64/8 -8 %, 96/9 -10 %, 64/10
        // -9 %, 100/10 -9 %, 64/11 -6 %, 64/12 -7 %, 40/14 -5 %, 30/15 0 %.  Launches under
        // stream capture (a replayed graph may run on any stream, beside anything) and launches that find
        // the next counter block of the plan's ring still in use by another stream (sdr_fir_plan::d_i8q)
        // keep the static order.
        unsigned *qctr = nullptr, *q_done = nullptr;
        unsigned q_gen = 0;
        static const bool static_order = std::getenv("SDRHIP_STATIC_TILES") != nullptr;   // A/B measurements
        if (!static_order && p->R == 4 && pre > 4 && p->ngran <= 4 && p->input != SDR_IN_I8_IQ && (nb & 7) == 0 && g.total_tiles >= 128LL * nb) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusActive; }
            if (cs == hipStreamCaptureStatusNone) {
                std::lock_guard<std::mutex> lk(p->q_mu);
                if (!p->d_i8q && !p->q_failed) {
                    const size_t qb = (size_t)sdr_fir_plan::kQueueBlocks * kI8QueueWords * sizeof(unsigned);
                    unsigned *dq = nullptr, *hd = nullptr;
                    hipError_t ea = sdr::dev_malloc((void **)&dq, qb);
                    if (ea == hipSuccess) ea = hipMemset(dq, 0, qb);
                    if (ea == hipSuccess) ea = sdr::host_malloc((void **)&hd, sdr_fir_plan::kQueueBlocks * sizeof(unsigned), hipHostMallocMapped);
                    if (ea == hipSuccess) {
                        std::memset(hd, 0, sdr_fir_plan::kQueueBlocks * sizeof(unsigned));
                        p->d_i8q = dq; p->h_q_done = hd;
                    } else {
                        (void)hipGetLastError();
                        if (dq) (void)hipFree(dq);
                        p->q_failed = true;
                    }
                }
                if (p->d_i8q) {
                    const int b = p->q_ring;
                    const bool done = p->q_gen[b] == 0 ||
                                      __atomic_load_n((const volatile unsigned *)&p->h_q_done[b], __ATOMIC_ACQUIRE) == p->q_gen[b];
                    const bool ordinary = st != hipStreamPerThread && st != hipStreamLegacy;
                    if (done || (ordinary && p->q_handle[b] == st)) {
                        p->q_ring = (b + 1) % sdr_fir_plan::kQueueBlocks;
                        if (++p->q_next_gen == 0) ++p->q_next_gen;
                        p->q_gen[b] = q_gen = p->q_next_gen;
                        p->q_handle[b] = st;
                        qctr = p->d_i8q + (size_t)b * kI8QueueWords;
                        q_done = p->h_q_done + b;
                    }
                }
            }
        }
#define SDR_I8K(NS, CB, PRE, S8, DYN, ...)                                                       \
    do {                                                                                         \
        SDR_OPT_IN((fir_mfma_i8_kernel<NS, CB, PRE, S8, DYN, ##__VA_ARGS__>));                   \
        fir_mfma_i8_kernel<NS, CB, PRE, S8, DYN, ##__VA_ARGS__><<<nb, 512, p->lds_bytes, st>>>((const uint32_t *)in, out, p->d_atab8, p->d_btab8, g, \
                                                                  p->plane_bytes, p->yscale,                 \
                                                                  p->d_ftaps, qctr, q_done, q_gen); \
    } while (0)
        // (the dynamic order is instantiated where the rule above can choose it)
#define SDR_I8Q(NS, CB, PRE, S8)                                                                 \
    do {                                                                                         \
        if constexpr (CB == 8 && PRE > 4 && NS == 4 && !S8) { if (qctr) { SDR_I8K(NS, CB, PRE, S8, true); break; } } \
        SDR_I8K(NS, CB, PRE, S8, false);                                                         \
    } while (0)
#define SDR_I8P(NS, CB, PRE) do { if (p->input == SDR_IN_I8_IQ) SDR_I8Q(NS, CB, PRE, true); else SDR_I8Q(NS, CB, PRE, false); } while (0)
#define SDR_I8(NS, CB) do { if (pre <= 4 && CB == 8) SDR_I8P(NS, 8, 4); else if (pre <= 6) SDR_I8P(NS, CB, 6); else SDR_I8P(NS, CB, 8); } while (0)
        // 512-output tiles are taken for D >= 16 only (the larger tile then exceeds the LDS or the helper
        // lanes' registers): their windows have more than four steps and their tiles more than four
        // octets per lane, so those instantiations do not exist (a wider one would do the same work)
        const int ns = (p->ngran + 3) / 4;
        if (p->cpx) {
            // complex taps per channel: four steps, tiles of 1024 outputs, contiguous runs of tiles
#define SDR_I8C(PRE, S8) SDR_I8K(4, 8, PRE, S8, false, true)
            if (p->input == SDR_IN_I8_IQ) { if (pre <= 4) SDR_I8C(4, true); else if (pre <= 6) SDR_I8C(6, true); else SDR_I8C(8, true); }
            else { if (pre <= 4) SDR_I8C(4, false); else if (pre <= 6) SDR_I8C(6, false); else SDR_I8C(8, false); }
#undef SDR_I8C
        } else
        if (p->R == 4) { if (ns <= 1) SDR_I8(4, 8); else if (ns == 2) SDR_I8(8, 8); else SDR_I8(12, 8); }
        else if (ns <= 2) SDR_I8(8, 4);
        else SDR_I8(12, 4);
#undef SDR_I8
#undef SDR_I8P
#undef SDR_I8Q
#undef SDR_I8K
        return hipGetLastError();
    }
    if (p->ksplit) {
        if constexpr (R == 4) {
            SDR_OPT_IN(fir_ksplit_kernel<4>);
            fir_ksplit_kernel<4><<<dim3(tiles, n_ch), 256, p->lds_bytes, st>>>((const float *)in, out, p->d_gtab, g);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;               // plans of this kind always have four outputs per lane
        }
    }
    if (p->mfma) {
        SDR_OPT_IN(fir_mfma_kernel);
        fir_mfma_kernel<<<dim3(8u * (unsigned)g.tiles_per_xcd, n_ch), 256, p->lds_bytes, st>>>(
            (const uint32_t *)in, out, p->d_atab, g);
        return hipGetLastError();
    }
    if (p->planar) {
        SDR_OPT_IN(fir_planar_kernel<R>);
        fir_planar_kernel<R><<<dim3(8u * (unsigned)g.tiles_per_xcd, n_ch), p->NT, p->lds_bytes, st>>>((const uint32_t *)in, out,
                                                                           p->d_gtab, g);
        return hipGetLastError();
    }
    if (p->raw) {
        SDR_OPT_IN(fir_raw_kernel<R>);
        fir_raw_kernel<R><<<dim3(8u * (unsigned)g.tiles_per_xcd, n_ch), p->NT, p->lds_bytes, st>>>((const uint32_t *)in, out,
                                                                        p->d_gtab, g);
        return hipGetLastError();
    }
    const unsigned nb = (unsigned)(g.total_tiles < p->stream_blocks ? g.total_tiles : p->stream_blocks);
#define SDR_STREAM(INK, SL)                                                                      \
    do {                                                                                         \
        SDR_OPT_IN(fir_stream_kernel<INK, R, SL>);                                               \
        fir_stream_kernel<INK, R, SL><<<nb, p->NT, p->lds_bytes, st>>>(in, out, p->d_gtab, g); \
    } while (0)
    const bool small = p->n_inst <= kSlotsSmall, mid = p->n_inst <= kSlotsReal;
    if (p->input == SDR_IN_I16_IQ || p->input == SDR_IN_I16_QI) { if (small) SDR_STREAM(IN_I16, kSlotsSmall); else if (mid) SDR_STREAM(IN_I16, kSlotsReal); else SDR_STREAM(IN_I16, kSlots); }
    else if (p->input == SDR_IN_I8_IQ) { if (small) SDR_STREAM(IN_I8, kSlotsSmall); else if (mid) SDR_STREAM(IN_I8, kSlotsReal); else SDR_STREAM(IN_I8, kSlots); }
    else if (p->input == SDR_IN_CF32) { if (small) SDR_STREAM(IN_CF32, kSlotsSmall); else if (mid) SDR_STREAM(IN_CF32, kSlotsReal); else SDR_STREAM(IN_CF32, kSlots); }
    else { if (p->n_inst <= kSlotsReal) SDR_STREAM(IN_REAL, kSlotsReal); else SDR_STREAM(IN_REAL, kSlots); }
#undef SDR_STREAM
    return hipGetLastError();
}
#undef SDR_OPT_IN

// Enqueue one launch.  `in` points at stream sample index in_first of channel 0.
int launch(sdr_fir_plan *p, const void *in, long long in_first, long long n_resident,
           long long n_total, long long out_begin, long long out_end, long long in_stride,
           size_t n_ch, void *out, long long out_stride, hipStream_t st, long long chan_first = 0)
{
    if (out_end <= out_begin || n_ch == 0) return SDR_OK;
    if (n_ch > 65535) { sdr::set_error("n_channels > 65535"); return SDR_EINVAL; }
    if (int rc = sdr::same_device(p->device)) return rc;
    const size_t ia = in_elem_bytes(p->input);
    if (((uintptr_t)in) % ia) { sdr::set_error("input pointer not sample aligned"); return SDR_EALIGN; }
    if (((uintptr_t)out) % out_elem_bytes(p->mode)) { sdr::set_error("output pointer misaligned"); return SDR_EALIGN; }
    FirGeom g;
    g.in_stride = in_stride;
    g.out_stride = out_stride;
    g.lo_valid = in_first > 0 ? in_first : 0;
    g.hi_valid = (in_first + n_resident < n_total) ? in_first + n_resident : n_total;
    if (p->mfma_i8 && g.hi_valid - g.lo_valid < (p->input == SDR_IN_I8_IQ ? 8 : 4)) {
        // its staging loads are 16 bytes wide (4 int16 or 8 int8 samples) and clamped into the valid range
        if (p->tiny) return launch(p->tiny, in, in_first, n_resident, n_total, out_begin, out_end, in_stride, n_ch, out, out_stride, st);
        sdr::set_error("pipeline 5 needs at least %d resident samples", p->input == SDR_IN_I8_IQ ? 8 : 4);
        return SDR_EINVAL;
    }
    g.tile0 = out_begin & ~3LL;
    g.out_begin = out_begin;
    g.out_end = out_end;
    g.store_from = out_begin;
    if (p->mfma_i8 && p->d_ftaps && p->mode == SDR_MODE_FM && out_begin * p->D < p->T - 1) {
        // the start-up outputs are computed with float32 taps (i8_startup_fm_output), not by the tiles
        const long long split = (p->T - 1 + p->D - 1) / p->D + 1;
        g.store_from = split < out_end ? split : out_end;
    }
    g.mix_fr = p->mix_fr;
    g.D = p->D; g.T = p->T; g.Tback = p->Tback; g.ngran = p->ngran; g.q = p->q; g.pad = p->pad;
    g.magic = (unsigned)((0x100000000ULL + (unsigned)p->q) / (unsigned)(p->q + 1));
    g.magic_q = (unsigned)((0x100000000ULL + (unsigned)p->q - 1) / (unsigned)p->q);
    g.s_gran = p->s_gran; g.a_gran = p->a_gran; g.n_inst = p->n_inst; g.mode = p->mode;
    g.swap_iq = (p->input == SDR_IN_I16_QI) ? 1 : 0;
    g.ph_off = p->ph_off; g.n_ph_hi = p->n_ph_hi;
    g.zfix = (p->mfma_i8 && p->mode == SDR_MODE_FM && p->zfix) ? 1 : 0;
    g.chan_fr = nullptr; g.chan_tab = nullptr; g.chan_first = 0; g.chan_tab_stride = 0; g.chan_bcast = p->chan_bcast ? 1 : 0;
    if (p->chan_bcast) chan_first = 0;
    if (p->n_chan_mix > 0) {
        if (!p->chan_bcast && (chan_first < 0 || chan_first + (long long)n_ch > p->n_chan_mix)) {
            sdr::set_error("channels [%lld, %lld) but the plan has %d channel mixers", chan_first, chan_first + (long long)n_ch, p->n_chan_mix);
            return SDR_EINVAL;
        }
        g.chan_fr = p->d_chan_fr; g.chan_tab = p->cpx ? p->d_chan_e : p->d_chan_tab; g.chan_first = (int)chan_first; g.chan_tab_stride = p->chan_tab_stride;
    }
    g.scale = (float)p->scale;
    // with the discriminator a tile's last lane only supplies its neighbour's successor y
    const long long per_tile = (long long)(p->mode == SDR_MODE_FM ? p->NT - 1 : p->NT) * p->R;
    g.per_tile = (int)per_tile;
    const long long tiles = (out_end - g.tile0 + per_tile - 1) / per_tile;
    if (tiles * (long long)n_ch > 0x7fffffffLL) { sdr::set_error("too many tiles"); return SDR_EINVAL; }
    g.tiles_per_ch = (int)tiles;
    g.tiles_per_xcd = (int)((tiles + 7) / 8);
    g.total_tiles = (int)(tiles * (long long)n_ch);
    // virtual pointer to stream sample 0 of channel 0 (only dereferenced inside the valid window)
    const void *base = (const unsigned char *)in - (size_t)in_first * in_elem_bytes(p->input);
    hipError_t e;
    const char *what = "fir kernel launch";
    switch (p->R) {
    case 1: e = launch_r<1>(p, st, base, out, g, (unsigned)tiles, (unsigned)n_ch, &what); break;
    case 2: e = launch_r<2>(p, st, base, out, g, (unsigned)tiles, (unsigned)n_ch, &what); break;
    case 4: e = launch_r<4>(p, st, base, out, g, (unsigned)tiles, (unsigned)n_ch, &what); break;
    default: e = launch_r<8>(p, st, base, out, g, (unsigned)tiles, (unsigned)n_ch, &what); break;
    }
    if (e != hipSuccess) return sdr::hip_fail(e, what);
    p->lds_opt_in.store(true, std::memory_order_relaxed);
    return SDR_OK;
}

// number of final outputs for a stream of n samples
long long out_count(const sdr_fir_plan *p, long long n)
{
    // y[m] = filtered sample D*m: m < floor(n/D) as polyphase_decimate (bladerf_apt_demod.cpp:282), or
    // every index 0, D, 2D, ... <= n-1 as x3(af::seq(0, n, D)) picks them (bladerf_fm_demod.cpp:212,294;
    // the seq's inclusive end n itself is past the block and never read here)
    const long long ny = (p->count_rule == SDR_COUNT_AF_SEQ) ? (n > 0 ? (n - 1) / p->D + 1 : 0) : n / p->D;
    if (p->mode == SDR_MODE_FM) return ny > 0 ? ny - 1 : 0;
    return ny;
}

}  // namespace

extern "C" int sdr_fir_plan_create(const sdr_fir_desc *d, sdr_fir_plan **out)
{
    if (!d || !out) { sdr::set_error("null argument"); return SDR_EINVAL; }
    *out = nullptr;
    // the reference throws std::invalid_argument for D <= 0 (bladerf_apt_demod.cpp:254-255)
    if (d->decim <= 0) { sdr::set_error("Decimation factor must be positive."); return SDR_EINVAL; }
    if (d->n_taps <= 0 || !d->taps) { sdr::set_error("n_taps must be positive"); return SDR_EINVAL; }
    if (d->n_taps > 16384) { sdr::set_error("n_taps > 16384 unsupported"); return SDR_EINVAL; }
    if (d->input < SDR_IN_I16_IQ || d->input > SDR_IN_I8_IQ) { sdr::set_error("bad input kind"); return SDR_EINVAL; }
    if (d->mode < SDR_MODE_IQ || d->mode > SDR_MODE_REAL) { sdr::set_error("bad mode"); return SDR_EINVAL; }
    if (d->pipeline < 0 || d->pipeline > 6) { sdr::set_error("bad pipeline"); return SDR_EINVAL; }
    if (d->count_rule != SDR_COUNT_FLOOR && d->count_rule != SDR_COUNT_AF_SEQ) { sdr::set_error("bad count_rule"); return SDR_EINVAL; }
    int rc = sdr::ensure_device();
    if (rc != SDR_OK) return rc;

    // One mixer on integer input (frequency_shift in front of the filter): when the window fits
    // the complex-tap matrix kernel (256 samples), the plan is built as a channelizer whose one table entry
    // every channel uses -- 0.31 against 0.55 ms per 2^28 samples at 64 taps / 10 on the float-tile kernel,
    // and exact integer sums, so shards and blocks reproduce the one-piece result bit for bit.
    if (d && out && d->mix_fr != 0.0 && d->n_mix_channels == 0 && (d->pipeline == 0 || d->pipeline == 5) && !d->outputs_per_lane &&
        !d->threads && d->taps && d->decim > 0 && d->n_taps >= 6 && d->n_taps <= 256 &&
        (d->input == SDR_IN_I16_IQ || d->input == SDR_IN_I16_QI || d->input == SDR_IN_I8_IQ) &&
        (d->n_taps - 1 + 3) / 4 * 4 + 15 * d->decim + 1 <= 256 && (d->size_hint == 0 || d->size_hint >= (1LL << 22))) {
        sdr_fir_desc dd = *d;
        const double fr1 = d->mix_fr;
        dd.mix_fr = 0.0; dd.n_mix_channels = 1; dd.mix_fr_channels = &fr1; dd.pipeline = 5;
        sdr_fir_plan *q = nullptr;
        if (sdr_fir_plan_create(&dd, &q) == SDR_OK && q) {
            q->chan_bcast = true;
            q->mix_fr = d->mix_fr;
            if (q->tiny) { q->tiny->chan_bcast = true; q->tiny->mix_fr = d->mix_fr; }
            *out = q;
            return SDR_OK;
        }
        if (d->pipeline == 5) return SDR_EINVAL;       // (the explicit choice does not fit: the error text is set)
    }
    const bool chan_mix = d->n_mix_channels > 0;
    if (d->n_mix_channels < 0 || (chan_mix && !d->mix_fr_channels)) { sdr::set_error("n_mix_channels without mix_fr_channels"); return SDR_EINVAL; }
    if (chan_mix && d->mix_fr != 0.0) { sdr::set_error("mix_fr and mix_fr_channels are alternatives"); return SDR_EINVAL; }
    if (chan_mix && d->input == SDR_IN_F32_REAL) { sdr::set_error("per-channel mixers need complex or integer input"); return SDR_EINVAL; }
    if (chan_mix && d->n_mix_channels > 65535) { sdr::set_error("n_mix_channels > 65535"); return SDR_EINVAL; }

    sdr_fir_plan *p = new sdr_fir_plan();
    p->input = d->input; p->mode = d->mode; p->D = d->decim; p->T = d->n_taps;
    p->scale = d->scale; p->mix_fr = d->mix_fr; p->count_rule = d->count_rule;
    (void)hipGetDevice(&p->device);
    const bool raw_possible = ((d->input == SDR_IN_I16_IQ || d->input == SDR_IN_I16_QI) && d->mix_fr == 0.0 && !chan_mix);
    // the int8 matrix kernel also takes int8 sample pairs (one byte plane per component)
    const bool int_input = d->input == SDR_IN_I16_IQ || d->input == SDR_IN_I16_QI || d->input == SDR_IN_I8_IQ;
    // (channelizer plans: the per-channel mixer folded into complex taps, windows of up to 256 samples)
    const bool i8_possible = int_input && d->mix_fr == 0.0;
    if ((d->pipeline == 1 || d->pipeline == 3 || d->pipeline == 4) && !raw_possible) { delete p; sdr::set_error("pipelines 1, 3 and 4 need int16 input without mixer"); return SDR_EINVAL; }
    if (d->pipeline == 5 && !i8_possible) { delete p; sdr::set_error("pipeline 5 needs int16 or int8 input without mixer"); return SDR_EINVAL; }
    const bool ksplit_possible = d->input == SDR_IN_F32_REAL && d->mode != SDR_MODE_FM;
    if (d->pipeline == 6 && !ksplit_possible) { delete p; sdr::set_error("pipeline 6 needs real float input and a mode other than FM"); return SDR_EINVAL; }
    // The planar kernel (float planes, convert once) is an explicit choice only: beyond the int8
    // matrix kernel's window it is 8-12 % ahead of the raw tile at D = 8 and 10 but 1.4-1.8x behind
    // at D = 16 and 20 (T = 640..2047, tools/sweep_shapes.py), so the automatic choice there is the raw tile.
    const bool planar = (d->pipeline == 3);

    // ---- choose kernel, outputs per lane (R) and workgroup size (NT) ------------------------
    // Raw-tile / planar kernels (int16, no mixer), from geometry sweeps on MI355X
    // (tools/tune_fir.py; D = 2..20, T = 32..511): two outputs per lane halve the conversions per
    // output and stay at an odd lane stride for most D; four when D is small (a lane should own
    // >= 16 input samples); one or eight are never the best (one: a conversion pair per MAC pair;
    // eight: 2-3x slower, tap SGPR sets and accumulators crowd the wave).  The largest workgroup
    // whose tile leaves two workgroups per CU wins within an R.
    // The streaming kernel (float inputs, mixer) keeps the instruction-count model below.
    const int Rs[4] = {1, 2, 4, 8};
    const int NTs[3] = {256, 128, 64};
    Geometry best{};
    int bestR = 0, bestNT = 0;
    bool best_raw = false;
    double best_cost = 0;
    if (d->pipeline == 4) {
        // Toeplitz GEMM geometry: 256 lanes, 512 outputs per tile, groups of 16 outputs
        if (d->outputs_per_lane && d->outputs_per_lane != 2) { delete p; sdr::set_error("pipeline 4 has 2 outputs per lane"); return SDR_EINVAL; }
        if (d->threads && d->threads != 256) { delete p; sdr::set_error("pipeline 4 has 256 threads"); return SDR_EINVAL; }
        Geometry gm{};
        gm.Tback = roundup(p->T - 1, 4);
        const int W = roundup(gm.Tback + 15 * p->D + 1, 16);
        const int nq = roundup(W / 16, kMfmaAhead);
        gm.ngran = nq;                                      // k-chunks of 16
        const long long s_a = (long long)p->D * 16 * 31 + 16LL * (nq + kMfmaAhead);   // last group + look-ahead chunks
        const long long s_tile = s_a;
        gm.s_gran = (int)((s_tile + 3) / 4);
        gm.a_gran = gm.s_gran;
        gm.n_inst = (gm.a_gran + 255) / 256;
        gm.q = 1; gm.pad = 0;
        gm.lds_bytes = gm.a_gran * 32 + 513 * 8 + 8;
        if (gm.lds_bytes > 160 * 1024) { sdr::set_error("pipeline 4: tile does not fit in LDS (decim=%d taps=%d)", p->D, p->T); delete p; return SDR_EINVAL; }
        best = gm; bestR = 2; bestNT = 256; best_raw = true;
        p->mfma = true;
    }
    // Automatic choice between the vector-unit kernels and the int8 matrix kernel, from a sweep of
    // fifteen shapes on MI355X (T = 64..700, D = 4..25), in ms per 2^28 samples:
    //   vector unit:  max(0.256 [HBM], 0.17 + 0.0125 * T/D)
    //   matrix pipe:  max(HBM at 5.4 TB/s, tiles per CU * time per tile); a tile of 1024 outputs takes
    //                 0.7 + 0.112 * (k steps of the window) + 0.085 * D microseconds (matrix waves: MFMAs
    //                 per step; helper waves: samples per tile), a tile of 512 outputs (when the LDS or
    //                 the helper lanes' registers do not hold the larger one) 0.55 of that
    // The matrix kernel's time follows the number of tiles, not the filter length; it loses where
    // tiles are many and windows short (D <= 4 with few taps).
    bool auto_i8 = false;
    const bool s8 = d->input == SDR_IN_I8_IQ;
    if (d->pipeline == 0 && i8_possible && !d->outputs_per_lane && !d->threads && d->n_taps >= (s8 ? 6 : 2)) {
        // int8 sample pairs (tools/sweep_int8_input.py, eighteen shapes): the float-tile kernel is the
        // alternative, max(0.30, 0.25 + 0.02 * T/D, 0.2 + 0.85 / D); a matrix tile takes
        // 0.78 + 0.06 * steps + 0.045 * D microseconds (four byte products per step instead of seven,
        // half the bytes per sample)
        const double tpd = (double)d->n_taps / (double)d->decim;
        const double t_valu = s8 ? std::fmax(std::fmax(0.30, 0.25 + 0.02 * tpd), 0.2 + 0.85 / d->decim)
                                 : std::fmax(0.256, 0.17 + 0.0125 * tpd);
        const int Wn = roundup(d->n_taps - 1, 4) + 15 * d->decim + 1;
        const int nks = (Wn + 63) / 64, steps = (nks + 3) / 4;
        const long long s_big = 16LL * d->decim * 63 + 256LL * steps;
        const bool big = s_big <= 8 * 256 * 8 &&
                         2 * 4 * (roundup((int)s_big, 256) + 240) + 2 * 8 * kI8PartSlots * 16 + (steps > 2 ? 12 * 1024 : 0) + 64 <= 160 * 1024;
        const double tile_us = (s8 ? 0.78 + 0.06 * nks + 0.045 * d->decim : 0.7 + 0.112 * nks + 0.085 * d->decim) * (big ? 1.0 : 0.55);
        const double tiles_per_cu = (big ? 1024.0 : 2048.0) / (double)d->decim;
        const double t_hbm = ((s8 ? 2.0 : 4.0) + 4.0 / d->decim) * 268435456.0 / 5.4e12 * 1e3;
        const double t_i8 = std::fmax(t_hbm, tiles_per_cu * tile_us * 1e-3);
        auto_i8 = t_i8 < t_valu || chan_mix;     // channelizer: 2.7x the float-tile kernel with per-channel tables
        if (d->size_hint > 0 && !chan_mix) {
            // a launch of the persistent matrix kernel costs ~13 us before its first output (tap digits
            // into registers, three pipeline stages), one of the tile kernels ~4 us: short launches
            // (below 2^26 samples at the headline shape, 2^22 at 511 / 10) stay on the vector unit
            const double f = (double)d->size_hint / 268435456.0;
            auto_i8 = 0.013 + t_i8 * f < 0.004 + t_valu * f;
        }
    }
    if (d->pipeline == 5 && d->n_taps < (s8 ? 6 : 2)) {
        // no history halo: a tile at the stream's end could hold fewer than the four (int8 input: eight)
        // valid samples its 16-byte staging loads need, and a one-tap "filter" has nothing for the matrix
        // pipe anyway
        delete p;
        sdr::set_error("pipeline 5 needs at least %d taps", s8 ? 6 : 2);
        return SDR_EINVAL;
    }
    if (d->pipeline == 5 || auto_i8) {
        bool fits = true;
        // byte-split Toeplitz GEMM: 1024 or 512 outputs per tile, k steps of 64 dealt to 4 matrix waves
        if (d->outputs_per_lane || d->threads) { delete p; sdr::set_error("pipeline 5 chooses its own geometry (512 lanes): leave outputs_per_lane and threads 0"); return SDR_EINVAL; }
        Geometry gm{};
        gm.Tback = roundup(p->T - 1, 4);
        const int W = gm.Tback + 15 * p->D + 1;
        const int nks = (W + 63) / 64;
        if (chan_mix && nks > 4) {
            // complex taps double the digit registers: four steps (256 samples) at most
            if (!auto_i8) { sdr::set_error("pipeline 5 with per-channel mixers: window of %d samples exceeds 256 (decim=%d taps=%d)", W, p->D, p->T); delete p; return SDR_EINVAL; }
            fits = false;
        }
        if (nks > kI8MaxSteps) {
            if (!auto_i8) { sdr::set_error("pipeline 5: window of %d samples exceeds %d (decim=%d taps=%d)", W, 64 * kI8MaxSteps, p->D, p->T); delete p; return SDR_EINVAL; }
            fits = false;
        }
        gm.ngran = nks;                                     // k steps of 64
        const int ns = (nks + 3) / 4;                       // per matrix wave
        // tile of 1024 outputs (8 column blocks, 4 outputs per finishing lane) when the LDS holds it,
        // else 512 outputs
        int ncb = 0;
        for (int cand = 8; cand >= 4 && !ncb; cand -= 4) {
            const long long s_tile = (long long)p->D * 16 * (8 * cand - 1) + 64LL * 4 * ns;   // last group's base + every wave's steps
            // byte planes, 1 B per sample; the offset between the I and Q planes is chosen so that the
            // 16-lane groups of the matrix waves' ds_read_b128 (MI355X_MICROARCH.md, LDS) meet no bank twice
            static const int grp[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                           {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                           {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                           {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
            const int base = roundup((int)s_tile, 256);
            int best_skew = 0, best_cost = 1 << 30;
            for (int skew = 0; skew < 256; skew += 16) {
                int cost = 0;
                for (int gi = 0; gi < 4; ++gi) {
                    int per_slot[16] = {0}, worst = 0;
                    for (int li = 0; li < 16; ++li) {
                        const int lane = grp[gi][li], jj = lane & 15, kq = lane >> 4;
                        const int addr = (jj & 1) * (base + skew) + 16 * p->D * (jj >> 1) + 16 * kq;
                        const int slot = (addr / 16) & 15;
                        if (++per_slot[slot] > worst) worst = per_slot[slot];
                    }
                    cost += worst;
                }
                if (cost < best_cost) { best_cost = cost; best_skew = skew; }
            }
            const int plane = base + best_skew;
            // two tile buffers + two tiles of outputs + the lowest tap digits of a 12-step window
            const int lds = 2 * 4 * plane + 2 * cand * kI8PartSlots * 16 + (ns > 2 ? kI8MaxSteps * 1024 : 0) + 64;   // + the tile queue
            // the helper lanes hold the staged tile in registers, at most 8 octets of samples each
            const bool wide = (s_tile / 4 + 1) / 2 > 256 * 8;
            if ((lds > 160 * 1024 || wide) && cand > 4) continue;
            ncb = cand;
            p->plane_bytes = plane;
            gm.lds_bytes = wide ? 161 * 1024 : lds;         // (reported as "does not fit" below)
            gm.s_gran = (int)(s_tile / 4);                  // quads of samples staged
        }
        gm.a_gran = gm.s_gran;
        gm.n_inst = (gm.a_gran + 255) / 256;
        gm.q = 1; gm.pad = 0;
        if (gm.lds_bytes > 160 * 1024) {
            if (!auto_i8) { sdr::set_error("pipeline 5: tile does not fit in LDS (decim=%d taps=%d)", p->D, p->T); delete p; return SDR_EINVAL; }
            fits = false;
        }
        if (fits && chan_mix && ncb != 8) {
            if (!auto_i8) { sdr::set_error("pipeline 5 with per-channel mixers needs tiles of 1024 outputs (decim=%d taps=%d)", p->D, p->T); delete p; return SDR_EINVAL; }
            fits = false;
        }
        if (fits) {
            best = gm; bestR = ncb / 2; bestNT = 256; best_raw = true;   // 256 finishing lanes, two or four outputs each
            p->mfma_i8 = true;
            p->cpx = chan_mix;
        }
    }
    // Short real-input streams (the plan's size hint): when the streaming kernel would put fewer than two
    // waves on a SIMD its launch lasts as long as one lane's walk over the whole filter; the K-split
    // kernel cuts that chain in four (APT stage 2 at cfg5's size: 32 -> 12 us)
    bool want_ksplit = d->pipeline == 6;
    if (d->pipeline == 0 && ksplit_possible && d->size_hint > 0 && !d->outputs_per_lane && !d->threads && d->n_taps >= 64) {
        hipDeviceProp_t prop;
        int ncu = 256;
        if (hipGetDeviceProperties(&prop, p->device) == hipSuccess) ncu = prop.multiProcessorCount;
        const long long chains = d->size_hint / d->decim / 256;            // waves of 64 lanes x 4 outputs
        want_ksplit = chains < 8LL * ncu;
    }
    if (want_ksplit) {
        if ((d->outputs_per_lane && d->outputs_per_lane != 4) || (d->threads && d->threads != 256)) { delete p; sdr::set_error("pipeline 6 has 4 outputs per lane and 256 threads"); return SDR_EINVAL; }
        Geometry gm;
        bool ok = make_geometry(p->T, p->D, 4, 64, 4, false, &gm, false, false, 8);
        if (ok) {
            gm.lds_bytes = gm.a_gran * 16 + 4 * 64 * 4 * 8;
            ok = gm.lds_bytes <= 160 * 1024;
        }
        if (ok) { best = gm; bestR = 4; bestNT = 64; best_raw = false; p->ksplit = true; }
        else if (d->pipeline == 6) { delete p; sdr::set_error("pipeline 6: tile does not fit in LDS (decim=%d taps=%d)", p->D, p->T); return SDR_EINVAL; }
    }
    if (!bestR && raw_possible && d->pipeline != 2) {
        const int pref_small[4] = {4, 2, 1, 8}, pref_large[4] = {2, 4, 1, 8};
        const int *pref = (p->D * 2 >= 16) ? pref_large : pref_small;
        for (int ri = 0; ri < 4 && !bestR; ++ri) {
            const int R = pref[ri];
            if (d->outputs_per_lane && d->outputs_per_lane != R) continue;
            for (int ni = 0; ni < 3; ++ni) {
                const int NT = NTs[ni];
                if (d->threads && d->threads != NT) continue;
                Geometry gm;
                if (!make_geometry(p->T, p->D, R, NT, 4, false, &gm, planar)) break;
                const bool forced = d->threads != 0;
                const int budget = (forced || ni == 2) ? 160 * 1024 : 56 * 1024;
                if (gm.lds_bytes > budget) continue;
                best = gm; bestR = R; bestNT = NT; best_raw = true;
                break;
            }
        }
    }
    // second pass: shapes that fit no preferred tile (large decimation factors, very long filters)
    // take any tile the LDS holds, and more granules per lane than the kernel prefetches
    for (int pass = 0; pass < 2 && !bestR; ++pass) {
        if (d->pipeline == 1 || d->pipeline == 3) continue;
        const bool raw = false;
        const int spg = (d->input == SDR_IN_F32_REAL) ? 4 : 2;     // real input: four samples per LDS granule
        for (int ri = 0; ri < 4; ++ri) {
            const int R = Rs[ri];
            if (d->outputs_per_lane && d->outputs_per_lane != R) continue;
            for (int ni = 0; ni < 3; ++ni) {
                const int NT = NTs[ni];
                if (d->threads && d->threads != NT) continue;
                Geometry gm;
                if (!make_geometry(p->T, p->D, R, NT, spg, !raw, &gm, false, (d->mix_fr != 0.0 || chan_mix) && spg == 2)) break;
                if (gm.n_inst > kSlots && pass == 0) continue;
                // channelizer: a lane carries at most two table entries across the walk (fir_stream_kernel)
                if (chan_mix && ph_hi_entries(gm.s_gran) + 64 > 2 * NT) continue;
                const bool forced = d->threads != 0;
                // automatic choice: float tiles <= 80 KB (two workgroups per CU; three below 53 KB)
                const int budget = (forced || ni == 2 || pass == 1) ? 160 * 1024 : 80 * 1024;
                if (gm.lds_bytes > budget) continue;
                // SIMD cycles per output: one packed FMA per (sample, output), one ds_read_b128
                // per granule
                const double walked = (double)gm.ngran * spg;                  // samples per lane
                double cost = walked * 4.0 + walked / spg * 8.0 / R;         // a 16-byte LDS read per wave: 8 cycles of a pipe four SIMDs share
                if (gm.pad) cost *= 1.1;                                       // padded image: extra index work
                // smaller tiles re-read more halo; real tiles (half the bytes per sample) run best
                // from 64-lane workgroups all the same (nine of twelve swept shapes, up to 25 %)
                if (spg == 4 ? ni != 2 : ni > 0) cost *= 1.05;
                if (R == 8) cost *= 2.0;
                // Tiles with more than 20 granules per lane go to the 32-slot instantiation, whose
                // prefetch set (128 VGPRs for complex float input) spills: measured 1.6-2.7x slower
                // than the best 12- or 20-slot geometry of the same shape (complex float input,
                // T = 64..511, D = 3..20; tools/sweep_float_tiles.py), so it is a last resort.
                // Two workgroups per CU instead of three cost little on long filters.
                if (gm.n_inst > kSlotsReal) cost *= 2.0;
                if (gm.lds_bytes > 53 * 1024) cost *= 1.08;
                if (gm.n_inst > kSlots) cost *= 1.0 + 0.05 * (gm.n_inst - kSlots);   // unprefetched rest of the tile
                if (!bestR || cost < best_cost) { best = gm; bestR = R; bestNT = NT; best_raw = raw; best_cost = cost; }
                if (spg != 4) break;          // complex tiles: the largest workgroup that fits; real tiles: compare all
            }
        }
    }
    if (!bestR) {
        delete p;
        sdr::set_error("no kernel geometry for decim=%d taps=%d (outputs_per_lane=%d threads=%d pipeline=%d)",
                       d->decim, d->n_taps, d->outputs_per_lane, d->threads, d->pipeline);
        return SDR_EINVAL;
    }
    p->R = bestR; p->NT = bestNT; p->raw = best_raw; p->planar = best_raw && planar && !p->mfma && !p->mfma_i8; p->spg = (best_raw || p->input == SDR_IN_F32_REAL) ? 4 : 2;
    p->Tback = best.Tback; p->Lw4 = best.Lw4; p->ngran = best.ngran; p->q = best.q; p->pad = best.pad;
    p->s_gran = best.s_gran; p->a_gran = best.a_gran; p->n_inst = best.n_inst; p->lds_bytes = best.lds_bytes;
    {
        hipDeviceProp_t prop;
        int ncu = 256;
        if (hipGetDeviceProperties(&prop, p->device) == hipSuccess) ncu = prop.multiProcessorCount;
        int per_cu = (160 * 1024) / p->lds_bytes;
        const int wave_cap = 32 / (p->NT / 64);
        if (per_cu > wave_cap) per_cu = wave_cap;
        if (per_cu > 8) per_cu = 8;
        if (per_cu < 1) per_cu = 1;
        if (p->mfma_i8) per_cu = 1;                           // 512 lanes with 256 registers each, one workgroup per CU
        p->stream_blocks = ncu * per_cu;
    }

    // ---- tap tables ---------------------------------------------------------------------------
    // taps rounded to float32; for integer input the exact power-of-two scale (2^-11 int16, 2^-7 int8) is folded in
    const float fold = (p->input == SDR_IN_I16_IQ || p->input == SDR_IN_I16_QI) ? sdr::kQ11
                       : (p->input == SDR_IN_I8_IQ ? sdr::kQ7 : 1.0f);
    std::vector<float> hf(p->T);
    for (int i = 0; i < p->T; ++i) hf[i] = (float)d->taps[i] * fold;
    // gtab[j][r] = h[D*r + Tback - j]  (tap that window offset j contributes to output r);
    // rows past the window (padding + look-ahead step) stay zero
    const size_t rows = (size_t)(best.ngran + best.look) * p->spg;
    const bool real_tile = !p->raw && p->input == SDR_IN_F32_REAL;
    std::vector<float> gt(rows * p->R * (real_tile ? 2 : 1), 0.0f);
    for (size_t j = 0; j < rows; ++j)
        for (int r = 0; r < p->R; ++r) {
            const long long i = (long long)p->D * r + p->Tback - (long long)j;
            if (i < 0 || i >= p->T) continue;
            if (!real_tile) { gt[j * p->R + r] = hf[i]; continue; }
            // complex taps h[i] * exp(-j*2*pi*fr*i), phase reduced in double
            const double ph = p->mix_fr * (double)i, f = ph - std::rint(ph);
            gt[2 * (j * p->R + r)] = (float)(d->taps[i] * std::cos(2.0 * M_PI * f));
            gt[2 * (j * p->R + r) + 1] = (float)(-d->taps[i] * std::sin(2.0 * M_PI * f));
        }
    // mixer phasor tables (see fir_stream_kernel), phases reduced in double: HI[j] = the phasor of offset
    // 64*j, LO[i] of offset i
    auto phasor_tables = [](double fr, int n_hi, std::vector<float> &dst) {
        auto phasor = [&](long long i) {
            const double ph = fr * (double)i;
            const double f = ph - std::rint(ph);
            dst.push_back((float)std::cos(2.0 * M_PI * f));
            dst.push_back((float)std::sin(2.0 * M_PI * f));
        };
        for (int j = 0; j < n_hi; ++j) phasor(64LL * j);
        for (int i = 0; i < 64; ++i) phasor(i);
    };
    if (!p->raw && p->mix_fr != 0.0 && !real_tile) {
        while (gt.size() % 4) gt.push_back(0.0f);                 // 16-byte aligned
        p->ph_off = (int)gt.size();
        p->n_ph_hi = ph_hi_entries(p->s_gran);
        phasor_tables(p->mix_fr, p->n_ph_hi, gt);
    }
    std::vector<float> chan_tab;
    if (chan_mix) p->n_chan_mix = d->n_mix_channels;
    if (chan_mix && !p->cpx) {
        // the channelizer: one pair of tables per channel (frequency_shift / create_freq_rotation per
        // channel, bladerf_apt_demod.cpp:476-495, sdr_functions.h:63-74), ~1 KB each
        p->n_ph_hi = ph_hi_entries(p->s_gran);
        p->chan_tab_stride = p->n_ph_hi + 64;
        p->n_chan_mix = d->n_mix_channels;
        chan_tab.reserve((size_t)p->n_chan_mix * p->chan_tab_stride * 2);
        for (int c = 0; c < p->n_chan_mix; ++c) phasor_tables(d->mix_fr_channels[c], p->n_ph_hi, chan_tab);
    }
    std::vector<float4> at;
    if (p->mfma) {
        // atab[q][lane]: the four taps lane (i = lane % 16, kq = lane / 16) multiplies with window
        // samples 16q + 4kq + 0..3 of its group: H[i][k] = h[D*i + Tb - k]; zero look-ahead chunks at the end
        at.assign((size_t)(best.ngran + kMfmaAhead) * 64, make_float4(0.f, 0.f, 0.f, 0.f));
        for (int q = 0; q < best.ngran; ++q)
            for (int lane = 0; lane < 64; ++lane) {
                float v[4];
                for (int c = 0; c < 4; ++c) {
                    const long long t = (long long)p->D * (lane & 15) + p->Tback - (16LL * q + 4 * (lane >> 4) + c);
                    v[c] = (t >= 0 && t < p->T) ? hf[t] : 0.0f;
                }
                at[(size_t)q * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
            }
    }
    std::vector<uint4> at8, bt8;
    std::vector<float> chan_e;
    if (p->mfma_i8) {
        // taps as 31-bit fixed point q = round(h * 2^F), F from the largest tap, in four balanced
        // base-256 digits; atab8[wave][s][digit][lane] = sixteen digits of H[i][64*ks + 16*kq + c],
        // ks = wave + 4*s, H[i][k] = q[D*i + Tb - k]
        double hmax = 0.0;
        for (int i = 0; i < p->T; ++i) hmax = std::fmax(hmax, std::fabs(d->taps[i]));
        int e2 = 0;
        if (hmax > 0.0) (void)std::frexp(hmax, &e2);           // hmax = m * 2^e2, 0.5 <= m < 1
        const int F = 30 - e2;                                 // |q| <= 2^30
        std::vector<long long> q(p->T);
        for (int i = 0; i < p->T; ++i) q[i] = std::llround(std::ldexp(d->taps[i], F));
        // A tap that is not zero but vanishes here -- it rounds to zero (below 2^-31 of the largest: the
        // numerical zeros of a windowed sinc where fc*t is an integer), or it has a lowest digit only
        // (below 2^-24) and meets a sample of 0..255, whose one byte product is the one left out -- is the
        // only way an output can be an exact 0 in this arithmetic and a tiny vector in the reference's
        // (short of an exact cancellation over the whole window).  Only plans with such a tap pay for
        // i8_zero_fix's look at their zeros.  (SDRHIP_ZFIX_ALWAYS=1 turns the look on for every plan:
        // A/B measurements of its cost; it cannot change a result.)
        static const bool zfix_always = std::getenv("SDRHIP_ZFIX_ALWAYS") != nullptr;
        p->zfix = zfix_always;
        for (int i = 0; i < p->T; ++i) p->zfix |= (std::llabs(q[i]) < 128 && d->taps[i] != 0.0);
        // Q11 samples, the kernel's sums start at weight 2^8 (int16) | Q7 samples, sums from weight 2^0 (int8)
        p->yscale = (float)std::ldexp(1.0, p->input == SDR_IN_I8_IQ ? -(F + 7) : -(F + 11) + 8);
        // sixteen-digit groups of one tap set: dst[(ks * ks_stride + digit) * 64 + lane], row sums added to rowsum[digit][row]
        auto digits_of = [&](const std::vector<long long> &qq, int n_steps, size_t ks_stride, uint4 *dst, int *rowsum) {
            for (int ks = 0; ks < n_steps; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    unsigned char dg[4][16];
                    for (int c = 0; c < 16; ++c) {
                        const long long k = 64LL * ks + 16 * (lane >> 4) + c;
                        const long long t = (long long)p->D * (lane & 15) + p->Tback - k;
                        long long r = (t >= 0 && t < p->T) ? qq[t] : 0;
                        for (int di = 0; di < 4; ++di) {
                            const long long dgt = ((r + 128) & 255) - 128;       // balanced digit in [-128, 127]
                            dg[di][c] = (unsigned char)(dgt & 255);
                            rowsum[(size_t)di * 16 + (lane & 15)] += (int)dgt;
                            r = (r - dgt) / 256;
                        }
                    }
                    for (int di = 0; di < 4; ++di) {
                        uint32_t v[4];
                        for (int u = 0; u < 4; ++u)
                            v[u] = dg[di][4 * u] | ((uint32_t)dg[di][4 * u + 1] << 8) | ((uint32_t)dg[di][4 * u + 2] << 16) | ((uint32_t)dg[di][4 * u + 3] << 24);
                        dst[((size_t)ks * ks_stride + di) * 64 + lane] = make_uint4(v[0], v[1], v[2], v[3]);
                    }
                }
        };
        // bias of lane (j, kq): rows 4*kq + 0..3; digit d meets the low bytes in accumulator d-1 (weight 0,
        // digit 0, is not computed): 128 * rowsum, combined like the kernel's accumulators
        auto bias_of = [](const int *rs, int lane) {
            float bf[4];
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (lane >> 4) + r;
                const float lo = std::fmaf((float)(128 * rs[32 + row]), 256.0f, (float)(128 * rs[16 + row]));
                const float hi = std::fmaf(0.0f, 256.0f, (float)(128 * rs[48 + row]));
                bf[r] = std::fmaf(hi, 65536.0f, lo);
            }
            uint4 u;
            std::memcpy(&u, bf, 16);
            return u;
        };
        if (!p->cpx) {
            std::vector<int> rowsum((size_t)4 * 16, 0);            // [digit][row]
            at8.assign((size_t)kI8MaxSteps * 4 * 64, make_uint4(0u, 0u, 0u, 0u));
            digits_of(q, kI8MaxSteps, 4, at8.data(), rowsum.data());
            bt8.assign((size_t)64, make_uint4(0u, 0u, 0u, 0u));
            for (int lane = 0; lane < 64; ++lane) bt8[lane] = bias_of(rowsum.data(), lane);
        } else {
            // the channelizer: per channel the digits of Re g and Im g, g[i] = h[i] * exp(-j*2*pi*fr*i) (phase
            // reduced in double), laid out [channel][step][part][digit][lane]; the bias of an even lane (real
            // output = Gr*Xi - Gi*Xq) comes from the row sums' difference, of an odd lane from their sum
            constexpr int NSC = 4;
            at8.assign((size_t)p->n_chan_mix * NSC * 2 * 4 * 64, make_uint4(0u, 0u, 0u, 0u));
            bt8.assign((size_t)p->n_chan_mix * 64, make_uint4(0u, 0u, 0u, 0u));
            chan_e.resize((size_t)p->n_chan_mix * 2);
            std::vector<long long> qr(p->T), qi(p->T);
            p->zfix = true;      // the discriminator's shortcut (a constant turn of the product) needs every exact zero looked at
            for (int c = 0; c < p->n_chan_mix; ++c) {
                const double fr = d->mix_fr_channels[c];
                for (int i = 0; i < p->T; ++i) {
                    const double ph = fr * (double)i, f = ph - std::rint(ph);
                    qr[i] = std::llround(std::ldexp(d->taps[i] * std::cos(2.0 * M_PI * f), F));
                    qi[i] = std::llround(std::ldexp(-d->taps[i] * std::sin(2.0 * M_PI * f), F));
                }
                std::vector<int> rs_r(64, 0), rs_i(64, 0), rs(64, 0);
                uint4 *dst = at8.data() + (size_t)c * NSC * 2 * 4 * 64;
                digits_of(qr, NSC, 8, dst, rs_r.data());
                digits_of(qi, NSC, 8, dst + 4 * 64, rs_i.data());
                for (int lane = 0; lane < 64; ++lane) {
                    for (int k = 0; k < 64; ++k) rs[k] = (lane & 1) ? rs_r[k] + rs_i[k] : rs_r[k] - rs_i[k];
                    bt8[(size_t)c * 64 + lane] = bias_of(rs.data(), lane);
                }
                const double phd = fr * (double)p->D, fd = phd - std::rint(phd);
                chan_e[2 * (size_t)c] = (float)std::cos(2.0 * M_PI * fd);
                chan_e[2 * (size_t)c + 1] = (float)std::sin(2.0 * M_PI * fd);
            }
        }
    }
    hipError_t e = sdr::dev_malloc((void **)&p->d_gtab, gt.size() * sizeof(float));
    if (e == hipSuccess && p->mfma_i8) e = sdr::dev_malloc((void **)&p->d_atab8, at8.size() * sizeof(uint4));
    if (e == hipSuccess && p->mfma_i8) e = hipMemcpy(p->d_atab8, at8.data(), at8.size() * sizeof(uint4), hipMemcpyHostToDevice);
    if (e == hipSuccess && p->mfma_i8) e = sdr::dev_malloc((void **)&p->d_btab8, bt8.size() * sizeof(uint4));
    if (e == hipSuccess && p->mfma_i8) e = hipMemcpy(p->d_btab8, bt8.data(), bt8.size() * sizeof(uint4), hipMemcpyHostToDevice);
    if (e == hipSuccess && p->mfma_i8 && p->mode == SDR_MODE_FM) e = sdr::dev_malloc((void **)&p->d_ftaps, hf.size() * sizeof(float));
    if (e == hipSuccess && p->d_ftaps) e = hipMemcpy(p->d_ftaps, hf.data(), hf.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && p->mfma) e = sdr::dev_malloc((void **)&p->d_atab, at.size() * sizeof(float4));
    if (e == hipSuccess && p->mfma) e = hipMemcpy(p->d_atab, at.data(), at.size() * sizeof(float4), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_gtab, gt.data(), gt.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && chan_mix && !p->cpx) e = sdr::dev_malloc((void **)&p->d_chan_tab, chan_tab.size() * sizeof(float));
    if (e == hipSuccess && chan_mix && !p->cpx) e = hipMemcpy(p->d_chan_tab, chan_tab.data(), chan_tab.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && p->cpx) e = sdr::dev_malloc((void **)&p->d_chan_e, chan_e.size() * sizeof(float));
    if (e == hipSuccess && p->cpx) e = hipMemcpy(p->d_chan_e, chan_e.data(), chan_e.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && chan_mix) e = sdr::dev_malloc((void **)&p->d_chan_fr, (size_t)p->n_chan_mix * sizeof(double));
    if (e == hipSuccess && chan_mix) e = hipMemcpy(p->d_chan_fr, d->mix_fr_channels, (size_t)p->n_chan_mix * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        sdr_fir_plan_destroy(p);
        return sdr::hip_fail(e, "plan upload");
    }
    if (p->mfma_i8 && (d->pipeline == 0 || p->cpx)) {
        sdr_fir_desc d2 = *d;
        d2.pipeline = (d->input == SDR_IN_I8_IQ || chan_mix) ? 2 : 1;        // float tiles | raw tile
        if (sdr_fir_plan_create(&d2, &p->tiny) != SDR_OK) p->tiny = nullptr;
    }
    *out = p;
    return SDR_OK;
}

extern "C" int sdr_fir_plan_destroy(sdr_fir_plan *p)
{
    if (!p) return SDR_OK;
    if (p->d_gtab) (void)hipFree(p->d_gtab);
    if (p->d_atab) (void)hipFree(p->d_atab);
    if (p->tiny) (void)sdr_fir_plan_destroy(p->tiny);
    if (p->d_atab8) (void)hipFree(p->d_atab8);
    if (p->d_btab8) (void)hipFree(p->d_btab8);
    if (p->d_i8q) (void)hipFree(p->d_i8q);
    if (p->h_q_done) (void)hipHostFree(p->h_q_done);
    if (p->d_ftaps) (void)hipFree(p->d_ftaps);
    if (p->d_chan_tab) (void)hipFree(p->d_chan_tab);
    if (p->d_chan_fr) (void)hipFree(p->d_chan_fr);
    if (p->d_chan_e) (void)hipFree(p->d_chan_e);
    if (p->d_in) (void)hipFree(p->d_in);
    if (p->d_out) (void)hipFree(p->d_out);
    delete p;
    return SDR_OK;
}

extern "C" size_t sdr_fir_out_count(const sdr_fir_plan *p, size_t n_in)
{
    return p ? (size_t)out_count(p, (long long)n_in) : 0;
}

extern "C" int sdr_fir_plan_info(const sdr_fir_plan *p, int *R, int *NT, int *lds)
{
    if (!p) return SDR_EINVAL;
    if (R) *R = p->R;
    if (NT) *NT = p->mfma_i8 ? 512 : p->NT;       // (pipeline 5: NT is its 256 finishing lanes; the workgroup has 512)
    if (lds) *lds = p->lds_bytes;
    return SDR_OK;
}

extern "C" int sdr_fir_plan_shape(const sdr_fir_plan *p, int *input, int *mode, int *decim, int *n_taps)
{
    if (!p) return SDR_EINVAL;
    if (input) *input = p->input;
    if (mode) *mode = p->mode;
    if (decim) *decim = p->D;
    if (n_taps) *n_taps = p->T;
    return SDR_OK;
}

extern "C" int sdr_fir_plan_kernel(const sdr_fir_plan *p)
{
    return p ? (p->ksplit ? 6 : (p->mfma_i8 ? 5 : (p->mfma ? 4 : (p->planar ? 3 : (p->raw ? 1 : 2))))) : 0;
}

extern "C" int sdr_fir_execute_dev(sdr_fir_plan *p, const void *d_in, size_t n_in, void *d_out,
                                   size_t out_capacity, size_t *n_out, void *stream)
{
    if (!p || (!d_in && n_in)) { sdr::set_error("null argument"); return SDR_EINVAL; }
    const long long cnt = out_count(p, (long long)n_in);
    if (n_out) *n_out = (size_t)cnt;
    if ((size_t)cnt > out_capacity) { sdr::set_error("output capacity %zu < %lld", out_capacity, cnt); return SDR_ECAPACITY; }
    if (cnt == 0) return SDR_OK;
    if (!d_out) { sdr::set_error("null output"); return SDR_EINVAL; }
    return launch(p, d_in, 0, (long long)n_in, (long long)n_in, 0, cnt, 0, 1, d_out, 0,
                  (hipStream_t)stream);
}

extern "C" int sdr_fir_execute_range_dev(sdr_fir_plan *p, const void *d_in, uint64_t in_first,
                                         uint64_t n_resident, uint64_t n_total,
                                         uint64_t out_begin, uint64_t out_end, void *d_out,
                                         void *stream)
{
    if (!p || !d_in || !d_out) { sdr::set_error("null argument"); return SDR_EINVAL; }
    const long long cnt = out_count(p, (long long)n_total);
    if (out_begin > out_end || (long long)out_end > cnt) { sdr::set_error("output range outside [0, %lld]", cnt); return SDR_EINVAL; }
    if (out_begin == out_end) return SDR_OK;
    // samples the range needs: history halo on the left, one more output on the right for FM
    long long need_lo = (long long)out_begin * p->D - (p->T - 1);
    if (need_lo < 0) need_lo = 0;
    long long need_hi = ((long long)out_end - 1 + (p->mode == SDR_MODE_FM ? 1 : 0)) * p->D + 1;
    if (need_hi > (long long)n_total) need_hi = (long long)n_total;
    if (need_lo < (long long)in_first || need_hi > (long long)(in_first + n_resident)) {
        sdr::set_error("range needs samples [%lld, %lld) but [%llu, %llu) are resident", need_lo,
                       need_hi, (unsigned long long)in_first, (unsigned long long)(in_first + n_resident));
        return SDR_ERANGE;
    }
    return launch(p, d_in, (long long)in_first, (long long)n_resident, (long long)n_total,
                  (long long)out_begin, (long long)out_end, 0, 1, d_out, 0, (hipStream_t)stream);
}

extern "C" int sdr_fir_execute_batched_dev(sdr_fir_plan *p, const void *d_in, size_t n_in,
                                           size_t in_stride, size_t n_channels, void *d_out,
                                           size_t out_stride, void *stream)
{
    if (!p || !d_in || !d_out) { sdr::set_error("null argument"); return SDR_EINVAL; }
    const long long cnt = out_count(p, (long long)n_in);
    if (n_channels > 1 && (in_stride < n_in || out_stride < (size_t)cnt)) { sdr::set_error("channel stride smaller than channel length"); return SDR_EINVAL; }
    return launch(p, d_in, 0, (long long)n_in, (long long)n_in, 0, cnt, (long long)in_stride,
                  n_channels, d_out, (long long)out_stride, (hipStream_t)stream);
}

extern "C" int sdr_fir_execute_channels_dev(sdr_fir_plan *p, const void *d_in, size_t n_in, size_t in_stride,
                                            size_t chan_first, size_t n_channels, void *d_out, size_t out_stride,
                                            void *stream)
{
    if (!p || !d_in || !d_out) { sdr::set_error("null argument"); return SDR_EINVAL; }
    if (p->n_chan_mix <= 0 || p->chan_bcast) { sdr::set_error("the plan has no per-channel mixers (sdr_fir_desc.n_mix_channels)"); return SDR_EINVAL; }
    const long long cnt = out_count(p, (long long)n_in);
    // in_stride 0: every channel reads the same wideband buffer
    if (n_channels > 1 && ((in_stride != 0 && in_stride < n_in) || out_stride < (size_t)cnt)) { sdr::set_error("channel stride smaller than channel length"); return SDR_EINVAL; }
    return launch(p, d_in, 0, (long long)n_in, (long long)n_in, 0, cnt, (long long)in_stride,
                  n_channels, d_out, (long long)out_stride, (hipStream_t)stream, (long long)chan_first);
}

extern "C" int sdr_fir_execute_host(sdr_fir_plan *p, const void *in, size_t n_in, void *out,
                                    size_t out_capacity, size_t *n_out)
{
    if (!p || (!in && n_in)) { sdr::set_error("null argument"); return SDR_EINVAL; }
    const long long cnt = out_count(p, (long long)n_in);
    if (n_out) *n_out = (size_t)cnt;
    if ((size_t)cnt > out_capacity) { sdr::set_error("output capacity %zu < %lld", out_capacity, cnt); return SDR_ECAPACITY; }
    if (cnt == 0) return SDR_OK;
    if (!out) { sdr::set_error("null output"); return SDR_EINVAL; }
    if (int rc = sdr::same_device(p->device)) return rc;
    const size_t bin = n_in * in_elem_bytes(p->input), bout = (size_t)cnt * out_elem_bytes(p->mode);
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
    SDR_HIP(hipMemcpyAsync(p->d_in, in, bin, hipMemcpyHostToDevice, 0));
    int rc = launch(p, p->d_in, 0, (long long)n_in, (long long)n_in, 0, cnt, 0, 1, p->d_out, 0, 0);
    if (rc != SDR_OK) return rc;
    SDR_HIP(hipMemcpyAsync(out, p->d_out, bout, hipMemcpyDeviceToHost, 0));
    SDR_HIP(hipStreamSynchronize(0));
    return SDR_OK;
}

extern "C" int sdr_fir_time_dev(sdr_fir_plan *p, const void *d_in, size_t n_in, size_t in_stride,
                                size_t n_channels, void *d_out, size_t out_stride, int iters,
                                void *stream, float *ms_per_launch)
{
    if (!p || !d_in || !d_out || iters <= 0 || !ms_per_launch) { sdr::set_error("bad argument"); return SDR_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const long long cnt = out_count(p, (long long)n_in);
    int rc = SDR_OK;
    float ms = 0.f;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, st);
    for (int i = 0; e == hipSuccess && i < iters && rc == SDR_OK; ++i)
        rc = launch(p, d_in, 0, (long long)n_in, (long long)n_in, 0, cnt, (long long)in_stride,
                    n_channels ? n_channels : 1, d_out, (long long)out_stride, st);
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) return sdr::hip_fail(e, "timing events");
    *ms_per_launch = ms / (float)iters;
    return rc;
}
