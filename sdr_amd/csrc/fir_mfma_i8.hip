// fir_mfma_i8.hip -- the FIR + decimate (+ discriminator | envelope) chain as a Toeplitz GEMM on the int8 matrix
// pipe of gfx950, in exact integer arithmetic (plan `pipeline` = 5; the automatic choice for int16 / int8 samples
// wherever its window fits).  Replaces the same reference loop as fir_tiles.hip:
//   int16 -> float scale, polyphase_decimate, polar_discriminator, std::abs envelope
//   bladerf/apt_demod/src/bladerf_apt_demod.cpp:728-732, :249-313, :405-425, :750-754
//   per-channel mixers (create_freq_rotation, bladerf/common/include/sdr_functions.h:63-74) folded into complex taps

#include <cstdlib>

#include "fir_common.h"

using namespace sdr::fir;

namespace {

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

template <int RH, bool S8>
__device__ __forceinline__ void i8_finish_zfix(const FirGeom &g, void *__restrict__ outv, float2 (&y)[RH], float2 ynext, int ch,
                                               long long M0, int o0, const uint32_t *__restrict__ in, const float *__restrict__ ftaps);

// The discriminator of a tile's outputs, written operation by operation ACROSS the lane's RH outputs.  A lone wave
// issues a dependent vector instruction every 6-9 cycles, and hipcc keeps the source order of independent chains:
// written as RH calls of discriminate() the lane's arctangents ran one after the other (four dependent chains of
// ~36 instructions: 1 450 cycles per tile in the matrix waves of a long-filter phase, a quarter of it, with the
// matrix pipe idle); operation-major they overlap in the vector pipeline.  The arithmetic is discriminate() /
// atan2_finite() operation for operation (fir_common.h), so the bits are those of every other kernel.
// (Tried in round 3 and not kept: issuing these operations between the MFMAs of the next tile's k loop -- one
// step per unit of seven MFMAs, sched_group_barrier -- gains nothing on windows of up to eight steps, which the
// sample stream bounds, and spills 43 registers on twelve-step windows, whose matrix waves have none to spare.)
// TURN (complex taps, i8_finish_cpx): the product y[m+1] conj(y[m]) is turned by the channel's constant E = e^{jwD}
// before the arctangent.
template <int RH, int NCB, bool TURN = false>
struct I8FinFM {
    static constexpr int kOps = 26;
    float Ex = 1.f, Ey = 0.f;
    float yr[RH + 1], yi[RH + 1];          // y[0..RH-1] and the successor of the last
    float re[RH], im[RH], mx[RH], mn[RH], s[RH], q[RH], u[RH];
    int ch, o0;
    long long M0;

    __device__ __forceinline__ void load(const FirGeom &g, const float *ybuf, int par, int F, int t)
    {
        constexpr int TILE = 128 * NCB;
        o0 = RH * F;
        const int o2 = o0 + RH < TILE ? o0 + RH : TILE - 1;
        const float *pa = ybuf + par * NCB * (4 * kI8PartSlots) + (o0 >> 7) * (4 * kI8PartSlots) + (o0 & 127);
        const float *pb = ybuf + par * NCB * (4 * kI8PartSlots) + (o2 >> 7) * (4 * kI8PartSlots) + (o2 & 127);
        ch = i8_tile_channel(g, t);
        M0 = g.tile0 + (long long)(t - ch * g.tiles_per_ch) * g.per_tile;
        if constexpr (RH == 4) {
            const float4 qi = *(const float4 *)pa, qq = *(const float4 *)(pa + 4 * 33);
            yr[0] = qi.x; yr[1] = qi.y; yr[2] = qi.z; yr[3] = qi.w;
            yi[0] = qq.x; yi[1] = qq.y; yi[2] = qq.z; yi[3] = qq.w;
        } else {
            static_assert(RH == 2, "two or four outputs per finishing lane");
            const float2 qi = *(const float2 *)pa, qq = *(const float2 *)(pa + 4 * 33);
            yr[0] = qi.x; yr[1] = qi.y; yi[0] = qq.x; yi[1] = qq.y;
        }
        yr[RH] = pb[0]; yi[RH] = pb[4 * 33];
    }

    // operation I of the chain, for all RH outputs (the order of discriminate() and atan2_finite())
    template <int I>
    __device__ __forceinline__ void op()
    {
#pragma unroll
        for (int r = 0; r < RH; ++r) {
            // y1 = y[r+1] = (a, b), y0 = y[r] = (c, -e)
            const float a = yr[r + 1], b = yi[r + 1], c = yr[r], e = -yi[r];
            if constexpr (I == 0) q[r] = b * e;
            else if constexpr (I == 1) re[r] = __fmaf_rn(a, c, -q[r]);
            else if constexpr (I == 2) q[r] = b * c;
            else if constexpr (I == 3) im[r] = __fmaf_rn(a, e, q[r]);
            else if constexpr (I == 4) mx[r] = fmaxf(fabsf(re[r]), fabsf(im[r]));
            else if constexpr (I == 5) mn[r] = fminf(fabsf(re[r]), fabsf(im[r]));
            else if constexpr (I == 6) q[r] = (mx[r] < 1e-30f) ? 1.8446744e19f : 1.0f;
            else if constexpr (I == 7) mx[r] *= q[r];
            else if constexpr (I == 8) mn[r] *= q[r];
            else if constexpr (I == 9) q[r] = __builtin_amdgcn_rcpf(mx[r]);
            else if constexpr (I == 10) s[r] = mn[r] * q[r];
            else if constexpr (I == 11) s[r] = __fmaf_rn(__fmaf_rn(-mx[r], s[r], mn[r]), q[r], s[r]);
            else if constexpr (I == 12) s[r] = (mx[r] == 0.0f) ? 0.0f : s[r];
            else if constexpr (I == 13) q[r] = s[r] * s[r];                                   // t
            else if constexpr (I == 14) u[r] = __fmaf_rn(0.00282363896258175373077393f, q[r], -0.0159569028764963150024414f);
            else if constexpr (I == 15) u[r] = __fmaf_rn(u[r], q[r], 0.0425049886107444763183594f);
            else if constexpr (I == 16) u[r] = __fmaf_rn(u[r], q[r], -0.0748900920152664184570312f);
            else if constexpr (I == 17) u[r] = __fmaf_rn(u[r], q[r], 0.106347933411598205566406f);
            else if constexpr (I == 18) u[r] = __fmaf_rn(u[r], q[r], -0.142027363181114196777344f);
            else if constexpr (I == 19) u[r] = __fmaf_rn(u[r], q[r], 0.199926957488059997558594f);
            else if constexpr (I == 20) u[r] = __fmaf_rn(u[r], q[r], -0.333331018686294555664062f);
            else if constexpr (I == 21) s[r] = __fmaf_rn(s[r] * q[r], u[r], s[r]);            // atan(mn/mx)
            else if constexpr (I == 22) s[r] = (fabsf(im[r]) > fabsf(re[r])) ? 1.57079632679489661923f - s[r] : s[r];
            else if constexpr (I == 23) s[r] = (__float_as_uint(re[r]) >> 31) ? 3.14159265358979323846f - s[r] : s[r];
            else if constexpr (I == 24) s[r] = __uint_as_float(__float_as_uint(s[r]) | (__float_as_uint(im[r]) & 0x80000000u));
            // (I == 25: the scale, in store())
        }
    }

    __device__ __forceinline__ void run()
    {
        static_for_ops<0, 4>();
        if constexpr (TURN) {
            // tr = fma(re, E.x, -(im * E.y)) + 0, ti = fma(re, E.y, im * E.x) + 0 (the + 0 washes a -0 out: the product
            // of two exact zeros stays (+0, +0), see i8_finish_cpx)
#pragma unroll
            for (int r = 0; r < RH; ++r) q[r] = im[r] * Ey;
#pragma unroll
            for (int r = 0; r < RH; ++r) mx[r] = __fmaf_rn(re[r], Ex, -q[r]) + 0.0f;
#pragma unroll
            for (int r = 0; r < RH; ++r) q[r] = im[r] * Ex;
#pragma unroll
            for (int r = 0; r < RH; ++r) im[r] = __fmaf_rn(re[r], Ey, q[r]) + 0.0f;
#pragma unroll
            for (int r = 0; r < RH; ++r) re[r] = mx[r];
        }
        static_for_ops<4, kOps - 1>();
    }
    // the scaled angles (after run())
    __device__ __forceinline__ void values(float scale, float (&v)[RH]) const
    {
#pragma unroll
        for (int r = 0; r < RH; ++r) v[r] = scale * s[r];
    }
    template <int I, int N>
    __device__ __forceinline__ void static_for_ops()
    {
        if constexpr (I < N) { op<I>(); static_for_ops<I + 1, N>(); }
    }

    template <bool S8>
    __device__ __forceinline__ void store(const FirGeom &g, void *__restrict__ outv, const uint32_t *__restrict__ in,
                                          const float *__restrict__ ftaps)
    {
        float v[RH];
#pragma unroll
        for (int r = 0; r < RH; ++r) v[r] = g.scale * s[r];
        const long long tile_end = M0 + g.per_tile;
        store_values<RH>(g, outv, ch, M0 + (long long)o0, v, tile_end < g.out_end ? tile_end : g.out_end);
        float2 y[RH];
#pragma unroll
        for (int r = 0; r < RH; ++r) y[r] = make_float2(yr[r], yi[r]);
        i8_finish_zfix<RH, S8>(g, outv, y, make_float2(yr[RH], yi[RH]), ch, M0, o0, in, ftaps);
    }
};

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
    if constexpr (RH >= 2) {
        if (g.mode == SDR_MODE_FM) {
            I8FinFM<RH, NCB> fs;
            fs.load(g, ybuf, par, F, t);
            fs.run();
            fs.template store<S8>(g, outv, in, ftaps);
            return;
        }
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
    i8_finish_zfix<RH, S8>(g, outv, y, ynext, ch, M0, o0, in, ftaps);
}

// (see i8_finish) the second look at exactly-zero y behind the stores
template <int RH, bool S8>
__device__ __forceinline__ void i8_finish_zfix(const FirGeom &g, void *__restrict__ outv, float2 (&y)[RH], float2 ynext, int ch,
                                               long long M0, int o0, const uint32_t *__restrict__ in, const float *__restrict__ ftaps)
{
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
// The finishing waves' view of the channel they are finishing: the turn E = e^{jwD} and the NCO frequency, fetched
// when the channel CHANGES (a workgroup walks a contiguous run of tiles: a handful of times per launch) and waited
// for there.  Read where they are used -- once per tile -- the wait for the two values is a wait for everything the
// wave has in flight: in the helper waves the samples just requested, a whole memory round trip per tile.
struct I8ChanE {
    int ch = -1;            // table entry (chan_index) the values belong to
    float ex = 1.f, ey = 0.f;
    double fr = 0.0;
    __device__ __forceinline__ void select(const FirGeom &g, int launch_ch)
    {
        const int c = __builtin_amdgcn_readfirstlane((int)chan_index(g, launch_ch));
        if (c != ch) {
            ch = c;
            const float2 e = g.chan_tab[c];
            const double f = g.chan_fr[c];
            // (through readfirstlane: the values have arrived when the branch ends, and live in scalar registers)
            ex = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(e.x)));
            ey = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(e.y)));
            fr = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(f)), __builtin_amdgcn_readfirstlane(__double2loint(f)));
        }
    }
};

// ROWS: ybuf holds (re, im) pairs in output order (the row layout of complex taps, CPX = 2 below) instead of one
// run of 128 values per component and column block.
template <int RH, int NCB, bool S8, bool ROWS>
__device__ __forceinline__ void i8_finish_cpx(const FirGeom &g, void *__restrict__ outv, const float *ybuf, int par, int F, int t,
                                              const uint32_t *__restrict__ in, const float *__restrict__ ftaps, I8ChanE &ce)
{
    constexpr int TILE = 128 * NCB;
    constexpr int PER = ROWS ? 2 : 1;                           // floats per output within a column block's run
    const int o0 = RH * F, o2 = o0 + RH < TILE ? o0 + RH : TILE - 1;
    const float *pa = ybuf + par * NCB * (4 * kI8PartSlots) + (o0 >> 7) * (4 * kI8PartSlots) + PER * (o0 & 127);
    const float *pb = ybuf + par * NCB * (4 * kI8PartSlots) + (o2 >> 7) * (4 * kI8PartSlots) + PER * (o2 & 127);
    const int ch = i8_tile_channel(g, t);
    const long long M0 = g.tile0 + (long long)(t - ch * g.tiles_per_ch) * g.per_tile;
    const long long m0 = M0 + (long long)o0;
    const long long tile_end = M0 + g.per_tile;
    const long long out_end = tile_end < g.out_end ? tile_end : g.out_end;
    ce.select(g, ch);
    const float2 E = make_float2(ce.ex, ce.ey);
    float2 y[RH], ynext;
    static_assert(RH == 2 || RH == 4, "complex-tap tiles have 1024 outputs");
    if constexpr (ROWS) {
        const float4 q0 = *(const float4 *)pa;
        y[0] = make_float2(q0.x, q0.y); y[1] = make_float2(q0.z, q0.w);
        if constexpr (RH == 4) {
            const float4 q1 = *(const float4 *)(pa + 4);
            y[2] = make_float2(q1.x, q1.y); y[3] = make_float2(q1.z, q1.w);
        }
        ynext = *(const float2 *)pb;
    } else if constexpr (RH == 4) {
        const float4 qi = *(const float4 *)pa, qq = *(const float4 *)(pa + 4 * 33);
        y[0] = make_float2(qi.x, qq.x); y[1] = make_float2(qi.y, qq.y);
        y[2] = make_float2(qi.z, qq.z); y[3] = make_float2(qi.w, qq.w);
        ynext = make_float2(pb[0], pb[4 * 33]);
    } else {
        const float2 qi = *(const float2 *)pa, qq = *(const float2 *)(pa + 4 * 33);
        y[0] = make_float2(qi.x, qq.x); y[1] = make_float2(qi.y, qq.y);
        ynext = make_float2(pb[0], pb[4 * 33]);
    }
    if (g.mode == SDR_MODE_ENV) {
        float v[RH];
#pragma unroll
        for (int r = 0; r < RH; ++r) v[r] = sqrtf(__fmaf_rn(y[r].x, y[r].x, y[r].y * y[r].y));
        store_values<RH>(g, outv, ch, m0, v, out_end);
        return;
    }
    if (g.mode == SDR_MODE_IQ || g.mode == SDR_MODE_REAL) {
        const double ph = ce.fr * (double)(m0 * g.D);
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
    {
        // operation-major over the lane's outputs (I8FinFM): the same arithmetic as `turned`, chains overlapped
        I8FinFM<RH, NCB, true> fs;
        fs.Ex = E.x; fs.Ey = E.y;
#pragma unroll
        for (int r = 0; r < RH; ++r) { fs.yr[r] = y[r].x; fs.yi[r] = y[r].y; }
        fs.yr[RH] = ynext.x; fs.yi[RH] = ynext.y;
        fs.run();
        fs.values(g.scale, v);
    }
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
// CPX = 2 (even decimation factors): the same complex taps with the output's component in the ROWS of the matrix
// product instead of its columns.  Row i of a block = (output i / 2 of a group of EIGHT outputs, component i % 2), the 64
// k values of a step = 32 window offsets x (I plane | Q plane), and the taps carry the signs:
//   A[(o, re)][(k, I)] = Gr[t]   A[(o, re)][(k, Q)] = -Gi[t]   A[(o, im)][(k, I)] = Gi[t]   A[(o, im)][(k, Q)] = Gr[t],  t = D o + Tb - k
//   B (samples): lane (j, kq) holds bytes 8 D grp + 32 ks + 16 (kq / 2) + 0..15 of plane (kq % 2, hi | lo), grp = 16 cb + j
//   C:           lane (j, kq) holds (re, im) of outputs 2 kq and 2 kq + 1 of group grp
// One set of accumulators, no exchange between lanes, and a group's window is Tb + 7 D + 1 samples instead of Tb + 15 D
// + 1: at 64 taps / 10 five steps of seven products per column block (35) where the column form runs four steps of
// fourteen (56) and then 64 cross-lane additions.  (Odd D: the 16-byte sample reads at 8 D grp would be misaligned.)
template <int NS, int NCB, int PRE, bool S8, bool DYN, int CPX = 0>
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
    static_assert(!CPX || (NCB == 8 && !DYN), "complex taps: 1024-output tiles, static order");
    static_assert(CPX != 1 || NS == 4, "complex taps in column form: four steps");
    static_assert(CPX != 2 || NS == 4 || NS == 8, "complex taps in row form: up to eight steps of 32 window offsets");
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
    // Measured (same box, 2^28 samples, FIN 0 / 1 / 2 in every instantiation, tools/ab_libs.py): wide tiles of windows
    // the sample stream bounds (64/10, 200/10, 101/5) lose 6-13 % to any finishing in the helper waves; twelve-step
    // windows of int16 samples gain 3-4.5 % from halves (511/10) when the finishing comes BEFORE the helper waves' next
    // request (FIN_FIRST: whatever wait the finishing holds then does not cover the samples just requested).  Complex
    // taps in row form, 64/10: int16 pairs 0.263 / 0.272 / 0.271 ms with FIN 0 / 1 / 2, int8 pairs 0.180 / 0.172 / 0.169.
    // A wave issues one instruction in four cycles whatever its kind, so a role's time is its instruction count, scalar
    // bookkeeping included, and every wave that finishes pays the finishing's fixed part (~270 instructions) again.
    constexpr bool LONG16 = !CPX && NS == 12 && !S8 && NCB == 8 && PRE == 6;     // (with eight octets per helper lane the halves spill)
    constexpr int FIN = CPX == 2 ? (S8 ? 2 : 0) : (NCB == 8 && (PRE <= (S8 ? 6 : 4) || LONG16)) ? 2 : 0;
    constexpr bool FIN_FIRST = CPX == 2 || LONG16;
    I8Clock clk;
    int n_ph = 0;
    I8ChanE chan_e;                                            // (complex taps) the channel being finished

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
        constexpr int NP = CPX == 1 ? 2 : 1;                   // tap parts: real only | real, imaginary (column form)
        constexpr int KSB = CPX == 2 ? 32 : 64;                // sample bytes of a plane per step
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
        const int cb_bytes = 16 * g.D * 8;                     // between column blocks
        const int bofs = (CPX == 2 ? (kq & 1) * plane_bytes + 8 * g.D * j + 16 * (kq >> 1)
                                   : (j & 1) * plane_bytes + 16 * g.D * (j >> 1) + 16 * kq) + CBW * wave * cb_bytes;
        // each matrix wave also finishes its own quarter of the tile before the last (R outputs per lane),
        // unless FIN (above) gives the helper waves a half or all of it.  Also measured: the finishing's
        // arithmetic interleaved with the MFMAs by sched_group_barrier (no gain up to eight steps, spills at twelve).
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
                if constexpr (CPX) i8_finish_cpx<(FIN == 2 ? R / 2 : R), NCB, S8, CPX == 2>(g, outv, (const float *)ybuf, p & 1, tid, t_fin, in, ftaps, chan_e);
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
                constexpr int PD = CPX == 1 ? (S8 ? 2 : 1) : (S8 ? 4 : 2), NB = PD + 1;
                i32x4 b[NB][3];                                // [buffer][hi, lo, lowest tap digit]
#pragma unroll
                for (int u = 0; u < PD && u < UNITS; ++u) {
                    const unsigned char *bp = bbase + (u / NS) * cb_bytes + KSB * (u % NS);
                    b[u][0] = as_i32x4(*(const uint4 *)bp);
                    if (!S8) b[u][1] = as_i32x4(*(const uint4 *)(bp + 2 * plane_bytes));
                    if (A0_LDS) b[u][2] = as_i32x4(a0lds[(u % NS) * 64 + lane]);
                }
                i32x4 acc[4];                                  // weights 2^8 .. 2^32
                i32x4 acci[CPX == 1 ? 4 : 1];                  // complex taps in column form: the imaginary part's sums
#pragma unroll
                for (int u = 0; u < UNITS; ++u) {
                    const int cb = u / NS, ks = u % NS;
                    if (u + PD < UNITS && !(SDR_I8_ABL & 32)) {
                        const unsigned char *bp = bbase + ((u + PD) / NS) * cb_bytes + KSB * ((u + PD) % NS);
                        b[(u + PD) % NB][0] = as_i32x4(*(const uint4 *)bp);
                        if (!S8) b[(u + PD) % NB][1] = as_i32x4(*(const uint4 *)(bp + 2 * plane_bytes));
                        if (A0_LDS) b[(u + PD) % NB][2] = as_i32x4(a0lds[((u + PD) % NS) * 64 + lane]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks == 0) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) acc[w] = i32x4{0, 0, 0, 0};
                        if constexpr (CPX == 1) {
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
                            if constexpr (CPX == 1) {
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
                        if constexpr (CPX == 1) {
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
                        if constexpr (CPX == 1) {
                            // real output (the I column, even lane) = Gr*Xi - Gi*Xq, imaginary (the Q column,
                            // odd lane) = Gr*Xq + Gi*Xi: the other column's "imaginary" sums come from the
                            // adjacent lane (quad_perm 1,0,3,2), exact in int32
#pragma unroll
                            for (int w = 0; w < 4; ++w)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    int other = __builtin_amdgcn_mov_dpp(acci[w][r], 0xB1, 0xF, 0xF, true);
                                    asm volatile("" : "+v"(other));
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
                        // (row form: (re, im) pairs in output order, 64 consecutive slots per column block)
                        ybuf[(((p - 1) & 1) * NCB + CBW * wave + cb) * kI8PartSlots + (CPX == 2 ? j * 4 + kq : (j & 1) * 33 + (j >> 1) * 4 + kq)] = rv;
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
            auto finish = [&]() {
                if (FIN != 0 && t_fin >= 0 && !(SDR_I8_ABL & 4))
                {
                    if constexpr (CPX) i8_finish_cpx<(FIN == 2 ? NCB / 4 : NCB / 2), NCB, S8, CPX == 2>(g, outv, (const float *)ybuf, p & 1, FIN == 2 ? tid : st, t_fin, in, ftaps, chan_e);
                    else i8_finish<(FIN == 2 ? NCB / 4 : NCB / 2), NCB, S8>(g, outv, (const float *)ybuf, p & 1, FIN == 2 ? tid : st, t_fin, in, ftaps);
                }
            };
            if constexpr (FIN_FIRST) finish();
            request(t_req, w, tl);
            clk.mark(1);
            if constexpr (!FIN_FIRST) finish();
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

hipError_t sdr::fir::launch_mfma_i8(const sdr_fir_plan *p, hipStream_t st, const void *in, void *out, const FirGeom &g,
                                    const char **what)
{
    *what = "fir kernel launch";
    {
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
            // complex taps per channel: tiles of 1024 outputs, contiguous runs of tiles; row form (even D) with up to
            // four or eight steps of 32 window offsets, column form with four steps of 64
#define SDR_I8C(NS, PRE, S8, FORM) SDR_I8K(NS, 8, PRE, S8, false, FORM)
#define SDR_I8CP(NS, S8, FORM) do { if (pre <= 4) SDR_I8C(NS, 4, S8, FORM); else if (pre <= 6) SDR_I8C(NS, 6, S8, FORM); else SDR_I8C(NS, 8, S8, FORM); } while (0)
#define SDR_I8CF(NS, FORM) do { if (p->input == SDR_IN_I8_IQ) SDR_I8CP(NS, true, FORM); else SDR_I8CP(NS, false, FORM); } while (0)
            if (p->cpx == 2) { if (p->ngran <= 4) SDR_I8CF(4, 2); else SDR_I8CF(8, 2); }
            else SDR_I8CF(4, 1);
#undef SDR_I8CF
#undef SDR_I8CP
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
}
