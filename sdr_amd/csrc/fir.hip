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
// Two kernels share that scheme:
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

#include <cmath>
#include <cstring>
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
};

// kernel-side input kinds (Q-first int16 is IN_I16 with FirGeom::swap_iq)
enum { IN_I16 = SDR_IN_I16_IQ, IN_CF32 = SDR_IN_CF32, IN_REAL = SDR_IN_F32_REAL, IN_I8 = SDR_IN_I8_IQ };

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
__device__ __forceinline__ void store_outputs(const FirGeom &g, void *__restrict__ outv,
                                              long long ch, long long m0, const float2 (&y)[R],
                                              float2 ynext, long long tile_end)
{
    const long long out_end = tile_end < g.out_end ? tile_end : g.out_end;   // this tile's share ends here
    if (g.mode == SDR_MODE_IQ) {
        float2 *out = (float2 *)outv + ch * g.out_stride;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long j = m0 + r;
            if (j >= g.out_begin && j < out_end) nt_store((float *)(out + (j - g.out_begin)), y[r].x, y[r].y);
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
    float *out = (float *)outv + ch * g.out_stride;
    float *dst = out + (m0 - g.out_begin);
    const bool whole = (m0 >= g.out_begin) && (m0 + R <= out_end);
    if (R >= 2 && whole && (((uintptr_t)dst) & (R >= 4 ? 15 : 7)) == 0) {
        if (R == 2) {
            nt_store(dst, v[0], v[1]);
        } else {
#pragma unroll
            for (int r = 0; r < R; r += 4)
                nt_store(dst + r, v[r], v[(r + 1) % R], v[(r + 2) % R], v[(r + 3) % R]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long j = m0 + r;
            if (j >= g.out_begin && j < out_end) nt_store(dst + r, v[r]);
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

template <int SPG, int R, int U>
__device__ __forceinline__ void walk_fma(const uint4 (&v)[U], const float (&t)[SPG * R * U],
                                         float2 (&acc)[R])
{
#pragma unroll
    for (int k = 0; k < U; ++k) {
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

template <int SPG, int R, bool PAD>
__device__ __forceinline__ void walk_p(const FirGeom &g, const uint4 *lds4,
                                       const float *__restrict__ gtab, int tid, float2 (&acc)[R])
{
    constexpr int U = Walk<SPG, R>::U;
    constexpr int N = Walk<SPG, R>::N;
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float2(0.f, 0.f);
    const uint4 *lane_base = lds4 + tid * (g.q + (PAD ? 1 : 0));
    const int nstep = g.ngran / U;           // even (host pads)
    uint4 va[U], vb[U];
    float ta[N], tb[N];
    walk_load<U, PAD>(va, lane_base, 0, g);
    walk_taps<N>(ta, gtab);
    for (int s = 0; s < nstep; s += 2) {
        walk_touch<U, N>(va, ta);
        __builtin_amdgcn_sched_barrier(0);
        walk_load<U, PAD>(vb, lane_base, (s + 1) * U, g);
        walk_taps<N>(tb, gtab + (size_t)(s + 1) * N);
        __builtin_amdgcn_sched_barrier(0);
        walk_fma<SPG, R, U>(va, ta, acc);
        walk_touch<U, N>(vb, tb);
        __builtin_amdgcn_sched_barrier(0);
        walk_load<U, PAD>(va, lane_base, (s + 2) * U, g);      // look-ahead step (zero taps at the end)
        walk_taps<N>(ta, gtab + (size_t)(s + 2) * N);
        __builtin_amdgcn_sched_barrier(0);
        walk_fma<SPG, R, U>(vb, tb, acc);
    }
}

template <int SPG, int R>
__device__ __forceinline__ void walk(const FirGeom &g, const uint4 *lds4,
                                     const float *__restrict__ gtab, int tid, float2 (&acc)[R])
{
    if (g.pad) walk_p<SPG, R, true>(g, lds4, gtab, tid, acc);
    else walk_p<SPG, R, false>(g, lds4, gtab, tid, acc);
}

// ---------------------------------------------------------------------------------------------
// fir_stream_kernel: float2 tiles, convert once, persistent workgroups with register prefetch.
// LDS: [tile (a_gran float4) | ybuf (NT+2 float2)].
// ---------------------------------------------------------------------------------------------
template <int IN> struct RawPair;                        // two consecutive input samples
template <> struct RawPair<IN_I16> { typedef uint2 type; };
template <> struct RawPair<IN_CF32> { typedef float4 type; };
template <> struct RawPair<IN_REAL> { typedef float2 type; };
template <> struct RawPair<IN_I8> { typedef uint32_t type; };

// bytes of one RawPair
template <int IN> struct PairBytes { static constexpr unsigned value = (IN == IN_CF32) ? 16u : (IN == IN_I8 ? 4u : 8u); };

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
        float2 w; memcpy(&w, &p, sizeof(w));
        x0 = make_float2(w.x, 0.f);
        x1 = make_float2(w.y, 0.f);
    }
    if (fr != 0.0) { x0 = mix(x0, fr, s); x1 = mix(x1, fr, s + 1); }
    return make_float4(x0.x, x0.y, x1.x, x1.y);
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
    r.fast = (r.W0 >= g.lo_valid) && (r.W0 + 2LL * g.s_gran <= g.hi_valid) &&
             ((((uintptr_t)r.src) & (IN == IN_CF32 ? 15 : (IN == IN_I8 ? 3 : 7))) == 0);
    return r;
}

constexpr int kSlotsSmall = 12, kSlots = 32;   // 2-sample granules one lane stages per tile

template <int IN, int R, int SLOTS>
__global__ void __launch_bounds__(256, (SLOTS <= kSlotsSmall ? 3 : 2))
fir_stream_kernel(const void *__restrict__ in, void *__restrict__ outv,
                  const float *__restrict__ gtab, const FirGeom g)
{
    typedef typename RawPair<IN>::type P;
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
                pre[k] = fetch_pair_checked<IN>(g, tr.src, tr.W0 + 2LL * sgk[k]);
        }
    };

    int t = blockIdx.x;
    if (t < g.total_tiles) fetch_tile(tile_ref<IN>(g, in, t, g.per_tile));
    for (; t < g.total_tiles; t += gridDim.x) {
        const TileRef tr = tile_ref<IN>(g, in, t, g.per_tile);
        __syncthreads();                         // every lane has finished with the previous tile
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            const int a = k * NT + tid;
            const float4 x = convert_pair<IN>(pre[k], g.mix_fr, tr.W0 + 2LL * sgk[k], g.swap_iq);
            if (a < g.a_gran) tile[a] = x;
        }
        __syncthreads();
        // request the next tile now: its HBM latency is covered by the walk below
        {
            const int tn = t + gridDim.x;
            if (tn < g.total_tiles) fetch_tile(tile_ref<IN>(g, in, tn, g.per_tile));
        }
        __builtin_amdgcn_sched_barrier(0);

        float2 acc[R];
        walk<2, R>(g, (const uint4 *)tile, gtab, tid, acc);

        const long long m0 = tr.M0 + (long long)tid * R;
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
// them once and writes I and Q to re4[] / im4[].
__device__ __forceinline__ void stage_planar(const FirGeom &g, const uint32_t *__restrict__ src,
                                             long long W0, float4 *re4, float4 *im4, int tid, int NT)
{
    const bool interior = (W0 >= g.lo_valid) && (W0 + 4LL * g.s_gran <= g.hi_valid);
    for (int k0 = 0; k0 < g.n_inst; k0 += kPlanarSlots) {
        uint4 w[kPlanarSlots];
        // all loads of the round first, unconditionally (slots past the tile repeat its last
        // round-robin row: a branch per load would serialise the HBM round trips)
#pragma unroll
        for (int k = 0; k < kPlanarSlots; ++k) {
            const int kk = (k0 + k < g.n_inst) ? k0 + k : g.n_inst - 1;
            int sg = sample_granule(g, kk * NT + tid);
            sg = sg < g.s_gran ? sg : 0;
            const long long s = W0 + 4LL * sg;
            if (interior) {
                w[k] = *(const uint4 *)(src + s);
            } else {
                uint32_t e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long si = s + u;
                    e[u] = (si >= g.lo_valid && si < g.hi_valid) ? src[si] : 0u;
                }
                w[k] = make_uint4(e[0], e[1], e[2], e[3]);
            }
        }
#pragma unroll
        for (int k = 0; k < kPlanarSlots; ++k) {
            const int a = (k0 + k) * NT + tid;
            if (k0 + k < g.n_inst && a < g.a_gran) {
                re4[a] = make_float4(lo16(w[k].x), lo16(w[k].y), lo16(w[k].z), lo16(w[k].w));
                im4[a] = make_float4(hi16(w[k].x), hi16(w[k].y), hi16(w[k].z), hi16(w[k].w));
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
// fir_mfma_bf16_kernel: the same Toeplitz GEMM on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16,
// 16x the float32 MFMA rate) WITHOUT giving up float32 accuracy: both operands are split into
// bf16 pieces that represent them exactly,
//   int16 sample  s = 256*(s >> 8) + (s & 255)        two pieces, each an 8-bit integer (x 2^8)
//   float32 tap   h = h1 + h2 + h3                    three pieces of 8 significant bits
// so the six piece products are exact in float32 and only the matrix unit's float32 accumulation
// rounds -- as in the float32 kernels.  Six MFMAs of 16 cycles replace thirty-two of 32.
//
// Work split: a persistent 256-lane workgroup walks tiles of 512 outputs = 32 groups of 16 =
// four column blocks of 16 columns (8 groups x {I, Q}).  The k range (window offsets, steps of
// 32) is dealt to the four waves round-robin: wave w owns steps w, w+4, ... (at most 6) for ALL
// column blocks, so its share of the tap matrix (6 steps x 3 pieces x 16 B per lane = 72 VGPRs)
// is loaded once per kernel and the k loop reads only samples: one ds_read_b128 per (step, block,
// piece) feeding three MFMAs.  The four partial sums meet in LDS (aliased onto the tile).
//   A (taps):    lane (i = lane % 16, kq = lane / 16) holds H[i][32*ks + 8*kq + 0..7],
//                H[i][k] = h[D*i + Tb - k] (Toeplitz skew lives here, so sample reads stay aligned)
//   B (samples): lane (j, kq) holds samples 16*D*grp + 32*ks + 8*kq + 0..7 of plane (j & 1, piece)
//   C:           lane (j, kq) holds outputs 4*kq + 0..3 of group grp = 8*cb + j/2, component j & 1
// LDS: [I hi | Q hi | I lo | Q lo] bf16 planes (plane stride = 32 mod 256 bytes: the I and Q lanes
// of one read land on different banks) | ybuf (513 float2).
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kBfSteps = 6;        // k steps per wave (24 steps of 32 = windows up to 768 samples)

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

// two exact floats -> their bf16 bit patterns packed (a low, b high)
__device__ __forceinline__ uint32_t pack_bf16(float a, float b)
{
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}

__global__ void __launch_bounds__(256, 2)
fir_mfma_bf16_kernel(const uint32_t *__restrict__ in, void *__restrict__ outv,
                     const uint4 *__restrict__ atab, const FirGeom g, const int plane_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int j = lane & 15, kq = lane >> 4;
    const int nks = g.ngran;                                   // k steps of 32 in the window
    const int region = 4 * plane_bytes > 16384 ? 4 * plane_bytes : 16384;
    float2 *ybuf = (float2 *)(smem + region);
    // plane p = 2*piece + component; Q-first input swaps the component planes
    const int cI = g.swap_iq ? 1 : 0, cQ = 1 - cI;

    // this wave's share of the tap matrix, resident for the whole kernel
    bf16x8 a[kBfSteps][3];
#pragma unroll
    for (int s = 0; s < kBfSteps; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) a[s][p] = as_bf16x8(atab[((wave * kBfSteps + s) * 3 + p) * 64 + lane]);

    const int bofs = (j & 1) * plane_bytes + 2 * (16 * g.D * (j >> 1) + 8 * kq);   // hi plane of component j&1

    for (int t = blockIdx.x; t < g.total_tiles; t += gridDim.x) {
        const long long ch = t / g.tiles_per_ch;
        const int tile = t - (int)ch * g.tiles_per_ch;
        const long long M0 = g.tile0 + (long long)tile * g.per_tile;
        const long long W0 = M0 * g.D - g.Tback;
        const uint32_t *src = in + ch * g.in_stride;

        // ---- stage: 4 samples per lane and round, split into the four planes ----------------
        const bool interior = (W0 >= g.lo_valid) && (W0 + 4LL * g.s_gran <= g.hi_valid);
        for (int k0 = 0; k0 < g.s_gran; k0 += 256 * 4) {
            uint4 w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int sg = k0 + 256 * k + tid;
                sg = sg < g.s_gran ? sg : g.s_gran - 1;
                const long long sidx = W0 + 4LL * sg;
                if (interior) {
                    w[k] = *(const uint4 *)(src + sidx);
                } else {
                    uint32_t e[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const long long si = sidx + u;
                        e[u] = (si >= g.lo_valid && si < g.hi_valid) ? src[si] : 0u;
                    }
                    w[k] = make_uint4(e[0], e[1], e[2], e[3]);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sg = k0 + 256 * k + tid;
                if (sg < g.s_gran) {
                    const uint32_t e[4] = {w[k].x, w[k].y, w[k].z, w[k].w};
                    float ih[4], il[4], qh[4], ql[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        ih[u] = (float)(short)(e[u] & 0xff00u);              // 256 * (I >> 8), exact
                        il[u] = (float)(e[u] & 0xffu);                       // I & 255
                        qh[u] = (float)((int)(e[u] & 0xff000000u) >> 16);    // 256 * (Q >> 8)
                        ql[u] = (float)((e[u] >> 16) & 0xffu);               // Q & 255
                    }
                    unsigned char *dst = smem + 8 * sg;
                    *(uint2 *)(dst + cI * plane_bytes) = make_uint2(pack_bf16(ih[0], ih[1]), pack_bf16(ih[2], ih[3]));
                    *(uint2 *)(dst + cQ * plane_bytes) = make_uint2(pack_bf16(qh[0], qh[1]), pack_bf16(qh[2], qh[3]));
                    *(uint2 *)(dst + (2 + cI) * plane_bytes) = make_uint2(pack_bf16(il[0], il[1]), pack_bf16(il[2], il[3]));
                    *(uint2 *)(dst + (2 + cQ) * plane_bytes) = make_uint2(pack_bf16(ql[0], ql[1]), pack_bf16(ql[2], ql[3]));
                }
            }
        }
        __syncthreads();

        // ---- k loop: this wave's steps, all four column blocks -------------------------------
        f32x4 acc[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < kBfSteps; ++s) {
            const int ks = wave + 4 * s;
            if (ks < nks) {                                    // wave-uniform
                bf16x8 bh[4], bl[4];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const unsigned char *bp = smem + bofs + 2 * (16 * g.D * 8 * cb + 32 * ks);
                    bh[cb] = as_bf16x8(*(const uint4 *)bp);
                    bl[cb] = as_bf16x8(*(const uint4 *)(bp + 2 * plane_bytes));
                }
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][p], bh[cb], acc[cb], 0, 0, 0);
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][p], bl[cb], acc[cb], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                       // every wave is done with the planes

        // ---- meet the four partial sums (aliased onto the tile), hand outputs to ybuf --------
        f32x4 *part = (f32x4 *)smem;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) part[(wave * 4 + cb) * 64 + lane] = acc[cb];
        __syncthreads();
        {
            const int cb = wave;                               // lane (cb, lane) sums block cb
            f32x4 v = part[cb * 64 + lane];
#pragma unroll
            for (int w2 = 1; w2 < 4; ++w2) {
                const f32x4 u = part[(w2 * 4 + cb) * 64 + lane];
                v += u;
            }
            float *yb = (float *)ybuf + 2 * (16 * (8 * cb + (j >> 1)) + 4 * kq) + (j & 1);
            yb[0] = v[0];
            yb[2] = v[1];
            yb[4] = v[2];
            yb[6] = v[3];
        }
        __syncthreads();
        const float2 y[2] = {ybuf[2 * tid], ybuf[2 * tid + 1]};
        const float2 ynext = ybuf[2 * tid + 2];                // FM only; the last lane's is unused
        store_outputs<2>(g, outv, ch, M0 + 2LL * tid, y, ynext, M0 + g.per_tile);
        // the next tile's staging overwrites the partial sums (read before the last barrier) and,
        // three barriers later, ybuf
    }
}

}  // namespace

// =============================================================================================
// host side
// =============================================================================================
struct sdr_fir_plan {
    int input, mode, D, T, R, NT;
    bool raw;               // raw-tile geometry (4 samples per granule): fir_raw_kernel or fir_planar_kernel
    bool planar = false;    // fir_planar_kernel (float planes, convert once)
    bool mfma = false;      // fir_mfma_kernel (float planes, Toeplitz GEMM on the matrix cores)
    float4 *d_atab = nullptr;   // its A-operand table
    bool mfma_bf16 = false; // fir_mfma_bf16_kernel (split-bf16 Toeplitz GEMM, persistent)
    uint4 *d_atab16 = nullptr;  // its A-operand table (bf16 pieces)
    int plane_bytes = 0;
    double scale, mix_fr;
    int spg;                // samples per LDS granule: 4 (raw) / 2 (stream)
    int Tback, Lw4, ngran, q, pad, s_gran, a_gran, n_inst, lds_bytes;
    int stream_blocks = 0;  // persistent grid: CUs x resident workgroups per CU
    float *d_gtab = nullptr;
    // staging buffers for the *_host entry point
    void *d_in = nullptr, *d_out = nullptr;
    size_t cap_in = 0, cap_out = 0;
    int device = 0;
};

namespace {

int roundup(int v, int m) { return (v + m - 1) / m * m; }

struct Geometry { int Tback, Lw4, U, ngran, look, q, pad, s_gran, a_gran, n_inst, lds_bytes; };

// Tile geometry for (T, D, R, NT) with spg samples per LDS granule.  Returns false when the lane
// stride is not a whole number of granules.
bool make_geometry(int T, int D, int R, int NT, int spg, bool stream, Geometry *o, bool planar = false)
{
    if (((long long)D * R) % spg) return false;
    o->Tback = roundup(T - 1, 4);
    o->Lw4 = roundup(o->Tback + D * (R - 1) + 1, 4);
    o->U = (spg * R >= 16) ? 1 : 16 / (spg * R);
    const int steps = roundup((o->Lw4 / spg + o->U - 1) / o->U, 2);
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
    return true;
}

size_t in_elem_bytes(int input) { return sdr::in_elem_bytes(input); }
size_t out_elem_bytes(int mode) { return mode == SDR_MODE_IQ ? 8 : 4; }

template <int R>
hipError_t launch_r(const sdr_fir_plan *p, hipStream_t st, const void *in, void *out,
                    const FirGeom &g, unsigned tiles, unsigned n_ch)
{
    if (p->mfma_bf16) {
        if (p->lds_bytes > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)fir_mfma_bf16_kernel,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, p->lds_bytes);
        const unsigned nb = (unsigned)(g.total_tiles < p->stream_blocks ? g.total_tiles : p->stream_blocks);
        fir_mfma_bf16_kernel<<<nb, 256, p->lds_bytes, st>>>((const uint32_t *)in, out, p->d_atab16, g, p->plane_bytes);
        return hipGetLastError();
    }
    if (p->mfma) {
        if (p->lds_bytes > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)fir_mfma_kernel,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, p->lds_bytes);
        fir_mfma_kernel<<<dim3(8u * (unsigned)g.tiles_per_xcd, n_ch), 256, p->lds_bytes, st>>>(
            (const uint32_t *)in, out, p->d_atab, g);
        return hipGetLastError();
    }
    if (p->planar) {
        if (p->lds_bytes > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)fir_planar_kernel<R>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, p->lds_bytes);
        fir_planar_kernel<R><<<dim3(8u * (unsigned)g.tiles_per_xcd, n_ch), p->NT, p->lds_bytes, st>>>((const uint32_t *)in, out,
                                                                           p->d_gtab, g);
        return hipGetLastError();
    }
    if (p->raw) {
        if (p->lds_bytes > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)fir_raw_kernel<R>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, p->lds_bytes);
        fir_raw_kernel<R><<<dim3(8u * (unsigned)g.tiles_per_xcd, n_ch), p->NT, p->lds_bytes, st>>>((const uint32_t *)in, out,
                                                                        p->d_gtab, g);
        return hipGetLastError();
    }
    const unsigned nb = (unsigned)(g.total_tiles < p->stream_blocks ? g.total_tiles : p->stream_blocks);
#define SDR_STREAM(INK, SL)                                                                      \
    do {                                                                                         \
        if (p->lds_bytes > 64 * 1024)                                                            \
            (void)hipFuncSetAttribute((const void *)fir_stream_kernel<INK, R, SL>,               \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, p->lds_bytes); \
        fir_stream_kernel<INK, R, SL><<<nb, p->NT, p->lds_bytes, st>>>(in, out, p->d_gtab, g); \
    } while (0)
    const bool small = p->n_inst <= kSlotsSmall;
    if (p->input == SDR_IN_I16_IQ || p->input == SDR_IN_I16_QI) { if (small) SDR_STREAM(IN_I16, kSlotsSmall); else SDR_STREAM(IN_I16, kSlots); }
    else if (p->input == SDR_IN_I8_IQ) { if (small) SDR_STREAM(IN_I8, kSlotsSmall); else SDR_STREAM(IN_I8, kSlots); }
    else if (p->input == SDR_IN_CF32) { if (small) SDR_STREAM(IN_CF32, kSlotsSmall); else SDR_STREAM(IN_CF32, kSlots); }
    else { if (small) SDR_STREAM(IN_REAL, kSlotsSmall); else SDR_STREAM(IN_REAL, kSlots); }
#undef SDR_STREAM
    return hipGetLastError();
}

// Enqueue one launch.  `in` points at stream sample index in_first of channel 0.
int launch(sdr_fir_plan *p, const void *in, long long in_first, long long n_resident,
           long long n_total, long long out_begin, long long out_end, long long in_stride,
           size_t n_ch, void *out, long long out_stride, hipStream_t st)
{
    if (out_end <= out_begin || n_ch == 0) return SDR_OK;
    if (n_ch > 65535) { sdr::set_error("n_channels > 65535"); return SDR_EINVAL; }
    const size_t ia = in_elem_bytes(p->input);
    if (((uintptr_t)in) % ia) { sdr::set_error("input pointer not sample aligned"); return SDR_EALIGN; }
    if (((uintptr_t)out) % out_elem_bytes(p->mode)) { sdr::set_error("output pointer misaligned"); return SDR_EALIGN; }
    FirGeom g;
    g.in_stride = in_stride;
    g.out_stride = out_stride;
    g.lo_valid = in_first > 0 ? in_first : 0;
    g.hi_valid = (in_first + n_resident < n_total) ? in_first + n_resident : n_total;
    g.tile0 = out_begin & ~3LL;
    g.out_begin = out_begin;
    g.out_end = out_end;
    g.mix_fr = p->mix_fr;
    g.D = p->D; g.T = p->T; g.Tback = p->Tback; g.ngran = p->ngran; g.q = p->q; g.pad = p->pad;
    g.magic = (unsigned)((0x100000000ULL + (unsigned)p->q) / (unsigned)(p->q + 1));
    g.magic_q = (unsigned)((0x100000000ULL + (unsigned)p->q - 1) / (unsigned)p->q);
    g.s_gran = p->s_gran; g.a_gran = p->a_gran; g.n_inst = p->n_inst; g.mode = p->mode;
    g.swap_iq = (p->input == SDR_IN_I16_QI) ? 1 : 0;
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
    switch (p->R) {
    case 1: e = launch_r<1>(p, st, base, out, g, (unsigned)tiles, (unsigned)n_ch); break;
    case 2: e = launch_r<2>(p, st, base, out, g, (unsigned)tiles, (unsigned)n_ch); break;
    case 4: e = launch_r<4>(p, st, base, out, g, (unsigned)tiles, (unsigned)n_ch); break;
    default: e = launch_r<8>(p, st, base, out, g, (unsigned)tiles, (unsigned)n_ch); break;
    }
    if (e != hipSuccess) return sdr::hip_fail(e, "fir kernel launch");
    return SDR_OK;
}

// number of final outputs for a stream of n samples
long long out_count(const sdr_fir_plan *p, long long n)
{
    const long long ny = n / p->D;
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
    if (d->pipeline < 0 || d->pipeline > 5) { sdr::set_error("bad pipeline"); return SDR_EINVAL; }
    int rc = sdr::ensure_device();
    if (rc != SDR_OK) return rc;

    sdr_fir_plan *p = new sdr_fir_plan();
    p->input = d->input; p->mode = d->mode; p->D = d->decim; p->T = d->n_taps;
    p->scale = d->scale; p->mix_fr = d->mix_fr;
    (void)hipGetDevice(&p->device);
    const bool raw_possible = ((d->input == SDR_IN_I16_IQ || d->input == SDR_IN_I16_QI) && d->mix_fr == 0.0);
    if ((d->pipeline == 1 || d->pipeline >= 3) && !raw_possible) { delete p; sdr::set_error("pipelines 1, 3, 4 and 5 need int16 input without mixer"); return SDR_EINVAL; }
    // Long filters (>= 32 MACs per input sample) are bound by VALU issue and, sustained, by the
    // power the conversions burn: the planar kernel converts once and runs T = 511 / D = 10 27 %
    // faster when the chip is warm (0.230 vs 0.292 ms per 2^26 samples); short filters are bound
    // by the HBM stream and want the raw tile's seven workgroups per CU.
    const bool planar = (d->pipeline == 3) || (d->pipeline == 0 && raw_possible && d->n_taps >= 32 * d->decim);

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
        if (gm.lds_bytes > 160 * 1024) { delete p; sdr::set_error("pipeline 4: tile does not fit in LDS (decim=%d taps=%d)", p->D, p->T); return SDR_EINVAL; }
        best = gm; bestR = 2; bestNT = 256; best_raw = true;
        p->mfma = true;
    }
    if (d->pipeline == 5) {
        // split-bf16 Toeplitz GEMM: 256 lanes, 512 outputs per tile, k steps of 32 dealt to 4 waves
        if (d->outputs_per_lane && d->outputs_per_lane != 2) { delete p; sdr::set_error("pipeline 5 has 2 outputs per lane"); return SDR_EINVAL; }
        if (d->threads && d->threads != 256) { delete p; sdr::set_error("pipeline 5 has 256 threads"); return SDR_EINVAL; }
        Geometry gm{};
        gm.Tback = roundup(p->T - 1, 4);
        const int W = gm.Tback + 15 * p->D + 1;
        const int nks = (W + 31) / 32;
        if (nks > 4 * kBfSteps) { delete p; sdr::set_error("pipeline 5: window of %d samples exceeds %d (decim=%d taps=%d)", W, 32 * 4 * kBfSteps, p->D, p->T); return SDR_EINVAL; }
        gm.ngran = nks;                                     // k steps of 32
        const long long s_tile = (long long)p->D * 16 * 31 + 32LL * nks;   // last group's base + its window
        gm.s_gran = (int)(s_tile / 4);
        gm.a_gran = gm.s_gran;
        gm.n_inst = (gm.a_gran + 255) / 256;
        gm.q = 1; gm.pad = 0;
        p->plane_bytes = roundup((int)(2 * s_tile), 256) + 32;
        const int region = 4 * p->plane_bytes > 16384 ? 4 * p->plane_bytes : 16384;
        gm.lds_bytes = region + 513 * 8 + 8;
        if (gm.lds_bytes > 160 * 1024) { delete p; sdr::set_error("pipeline 5: tile does not fit in LDS (decim=%d taps=%d)", p->D, p->T); return SDR_EINVAL; }
        best = gm; bestR = 2; bestNT = 256; best_raw = true;
        p->mfma_bf16 = true;
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
    for (int pass = 0; pass < 1 && !bestR; ++pass) {
        if (d->pipeline == 1 || d->pipeline == 3) continue;
        const bool raw = false;
        const int spg = 2;
        for (int ri = 0; ri < 4; ++ri) {
            const int R = Rs[ri];
            if (d->outputs_per_lane && d->outputs_per_lane != R) continue;
            for (int ni = 0; ni < 3; ++ni) {
                const int NT = NTs[ni];
                if (d->threads && d->threads != NT) continue;
                Geometry gm;
                if (!make_geometry(p->T, p->D, R, NT, spg, !raw, &gm, false)) break;
                if (gm.n_inst > kSlots) continue;
                const bool forced = d->threads != 0;
                // automatic choice: float tiles <= 53 KB (3 workgroups per CU)
                const int budget = (forced || ni == 2) ? 160 * 1024 : 53 * 1024;
                if (gm.lds_bytes > budget) continue;
                // SIMD cycles per output: one packed FMA per (sample, output), one ds_read_b128
                // per granule
                const double walked = (double)gm.ngran * spg;                  // samples per lane
                double cost = walked * 4.0 + walked / spg * 4.0 / R;
                if (gm.pad) cost *= 1.1;                                       // padded image: extra index work
                if (ni > 0) cost *= 1.05;                                      // smaller tiles: more halo
                if (R == 8) cost *= 2.0;
                if (!bestR || cost < best_cost) { best = gm; bestR = R; bestNT = NT; best_raw = raw; best_cost = cost; }
                break;
            }
        }
    }
    if (!bestR) {
        delete p;
        sdr::set_error("no kernel geometry for decim=%d taps=%d (outputs_per_lane=%d threads=%d pipeline=%d)",
                       d->decim, d->n_taps, d->outputs_per_lane, d->threads, d->pipeline);
        return SDR_EINVAL;
    }
    p->R = bestR; p->NT = bestNT; p->raw = best_raw; p->planar = best_raw && planar && !p->mfma && !p->mfma_bf16; p->spg = best_raw ? 4 : 2;
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
        if (p->mfma_bf16 && per_cu > 3) per_cu = 3;          // 164 VGPRs (72 of them taps): three waves per SIMD
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
    std::vector<float> gt(rows * p->R, 0.0f);
    for (size_t j = 0; j < rows; ++j)
        for (int r = 0; r < p->R; ++r) {
            const long long i = (long long)p->D * r + p->Tback - (long long)j;
            if (i >= 0 && i < p->T) gt[j * p->R + r] = hf[i];
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
    std::vector<uint4> at16;
    if (p->mfma_bf16) {
        // atab16[wave][s][piece][lane]: eight bf16 taps H[i][32*ks + 8*kq + c], ks = wave + 4*s,
        // H[i][k] = h[D*i + Tb - k]; piece 0..2 of the exact split h = h1 + h2 + h3 (each the
        // round-to-nearest bf16 of what the previous pieces left: 8 + 8 + 8 significant bits)
        auto bf16_rne = [](float f) -> uint16_t {
            uint32_t u; std::memcpy(&u, &f, 4);
            if ((u & 0x7f800000u) == 0) return (uint16_t)(u >> 16 & 0x8000u);      // denormal -> signed zero
            u += 0x7fffu + ((u >> 16) & 1u);
            return (uint16_t)(u >> 16);
        };
        auto bf16_val = [](uint16_t b) -> float { uint32_t u = (uint32_t)b << 16; float f; std::memcpy(&f, &u, 4); return f; };
        at16.assign((size_t)4 * kBfSteps * 3 * 64, make_uint4(0u, 0u, 0u, 0u));
        for (int w = 0; w < 4; ++w)
            for (int sidx = 0; sidx < kBfSteps; ++sidx)
                for (int lane = 0; lane < 64; ++lane) {
                    uint16_t pc[3][8];
                    for (int c = 0; c < 8; ++c) {
                        const long long k = 32LL * (w + 4 * sidx) + 8 * (lane >> 4) + c;
                        const long long t = (long long)p->D * (lane & 15) + p->Tback - k;
                        float r = (t >= 0 && t < p->T) ? hf[t] : 0.0f;
                        for (int pi = 0; pi < 3; ++pi) {
                            pc[pi][c] = bf16_rne(r);
                            r -= bf16_val(pc[pi][c]);                        // exact: r loses its top 8 bits
                        }
                    }
                    for (int pi = 0; pi < 3; ++pi) {
                        uint4 v;
                        v.x = pc[pi][0] | ((uint32_t)pc[pi][1] << 16);
                        v.y = pc[pi][2] | ((uint32_t)pc[pi][3] << 16);
                        v.z = pc[pi][4] | ((uint32_t)pc[pi][5] << 16);
                        v.w = pc[pi][6] | ((uint32_t)pc[pi][7] << 16);
                        at16[(((size_t)w * kBfSteps + sidx) * 3 + pi) * 64 + lane] = v;
                    }
                }
    }
    hipError_t e = hipMalloc((void **)&p->d_gtab, gt.size() * sizeof(float));
    if (e == hipSuccess && p->mfma_bf16) e = hipMalloc((void **)&p->d_atab16, at16.size() * sizeof(uint4));
    if (e == hipSuccess && p->mfma_bf16) e = hipMemcpy(p->d_atab16, at16.data(), at16.size() * sizeof(uint4), hipMemcpyHostToDevice);
    if (e == hipSuccess && p->mfma) e = hipMalloc((void **)&p->d_atab, at.size() * sizeof(float4));
    if (e == hipSuccess && p->mfma) e = hipMemcpy(p->d_atab, at.data(), at.size() * sizeof(float4), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_gtab, gt.data(), gt.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        sdr_fir_plan_destroy(p);
        return sdr::hip_fail(e, "plan upload");
    }
    *out = p;
    return SDR_OK;
}

extern "C" int sdr_fir_plan_destroy(sdr_fir_plan *p)
{
    if (!p) return SDR_OK;
    if (p->d_gtab) (void)hipFree(p->d_gtab);
    if (p->d_atab) (void)hipFree(p->d_atab);
    if (p->d_atab16) (void)hipFree(p->d_atab16);
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
    if (NT) *NT = p->NT;
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
    return p ? (p->mfma_bf16 ? 5 : (p->mfma ? 4 : (p->planar ? 3 : (p->raw ? 1 : 2)))) : 0;
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

extern "C" int sdr_fir_execute_host(sdr_fir_plan *p, const void *in, size_t n_in, void *out,
                                    size_t out_capacity, size_t *n_out)
{
    if (!p || (!in && n_in)) { sdr::set_error("null argument"); return SDR_EINVAL; }
    const long long cnt = out_count(p, (long long)n_in);
    if (n_out) *n_out = (size_t)cnt;
    if ((size_t)cnt > out_capacity) { sdr::set_error("output capacity %zu < %lld", out_capacity, cnt); return SDR_ECAPACITY; }
    if (cnt == 0) return SDR_OK;
    if (!out) { sdr::set_error("null output"); return SDR_EINVAL; }
    const size_t bin = n_in * in_elem_bytes(p->input), bout = (size_t)cnt * out_elem_bytes(p->mode);
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
    hipEvent_t e0, e1;
    SDR_HIP(hipEventCreate(&e0));
    SDR_HIP(hipEventCreate(&e1));
    const long long cnt = out_count(p, (long long)n_in);
    int rc = SDR_OK;
    SDR_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters && rc == SDR_OK; ++i)
        rc = launch(p, d_in, 0, (long long)n_in, (long long)n_in, 0, cnt, (long long)in_stride,
                    n_channels ? n_channels : 1, d_out, (long long)out_stride, st);
    SDR_HIP(hipEventRecord(e1, st));
    SDR_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    SDR_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = ms / (float)iters;
    return rc;
}
