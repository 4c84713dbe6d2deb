// fir_tiles.hip -- fused FIR + decimate (+ NCO mix, + FM discriminator | envelope) for gfx950: the tile kernels
// (shared device code: fir_common.h; the int8 matrix kernel: fir_mfma_i8.hip; plans and the C ABI: fir_plan.hip).
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

#include "fir_common.h"

using namespace sdr::fir;

namespace {

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
// LDS: the tile (a_gran float4), then -- in the same memory -- the partial sums (4 x 64*R float2): a smaller
// workgroup image puts eight workgroups on a CU instead of six (APT stage 2 at cfg5's size: 1 747 tiles in one
// round of 2 048 slots instead of one and a bit of 1 536).  No discriminator (the successor of a
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
    float2 *part = (float2 *)smem;                 // the partial sums reuse the tile's memory (behind a barrier)
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
    __syncthreads();                               // every wave has finished reading the tile
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

// IN = IN_I16: raw int16 pairs, converted after each LDS read (the kernel described above).
// IN = IN_CF32 / IN_REAL (plan `pipeline` = 7): float input without mixer needs no conversion at all, so the
// DMA'd image IS the float tile the walk reads (two complex or four real samples per 16-byte granule) --
// no register prefetch, no LDS stores, one tile per workgroup under the hardware dispatcher.
// CT (complex float input only): complex taps -- the plan's mixer folded into them, g[i] = h[i] e^{-jwi}; the output
// takes the mixer's phasor at its own sample, as for real input.
template <int R, int IN, bool CT = false>
__global__ void __launch_bounds__(256)
fir_raw_kernel(const void *__restrict__ in, void *__restrict__ outv,
               const float *__restrict__ gtab, const FirGeom g)
{
    constexpr int SPG = (IN == IN_CF32) ? 2 : 4;             // samples per 16-byte granule
    constexpr int ESZ = (IN == IN_CF32) ? 8 : 4;             // bytes per sample
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *lds4 = (uint4 *)smem;
    const int tid = threadIdx.x;
    const int NT = blockDim.x;
    const long long ch = blockIdx.y;
    const int tile = xcd_tile(g);
    if (tile >= g.tiles_per_ch) return;
    const long long M0 = g.tile0 + (long long)tile * g.per_tile;
    const long long W0 = M0 * g.D - g.Tback;                 // first staged sample (multiple of 4)
    const unsigned char *src = (const unsigned char *)in + ch * g.in_stride * ESZ;   // sample 0 of this channel

    const bool interior = (W0 >= g.lo_valid) && (W0 + (long long)SPG * g.s_gran <= g.hi_valid);
    if (interior) {
        const unsigned char *wsrc = src + W0 * ESZ;
        for (int k = 0; k < g.n_inst; ++k) {
            const int a = k * NT + tid;
            int sg = sample_granule(g, a);
            sg = sg < g.s_gran ? sg : 0;
            if (a < g.a_gran)              // last round: lanes past the image stay masked
                dma16(wsrc + 16 * sg, __builtin_amdgcn_readfirstlane(lds_addr(lds4 + k * NT + (tid & ~63))));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (int k = 0; k < g.n_inst; ++k) {
            const int a = k * NT + tid;
            const long long s = W0 + (long long)SPG * sample_granule(g, a);
            uint4 v;
            if constexpr (IN == IN_I16) {
                uint32_t w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long si = s + u;
                    w[u] = (si >= g.lo_valid && si < g.hi_valid) ? ((const uint32_t *)src)[si] : 0u;
                }
                v = make_uint4(w[0], w[1], w[2], w[3]);
            } else {
                const float4 f = fetch_pair_checked<IN>(g, src, s);
                v = __builtin_bit_cast(uint4, f);
            }
            if (a < g.a_gran) lds4[a] = v;
        }
    }
    __syncthreads();

    float2 acc[R];
    walk<SPG, R, (IN == IN_REAL ? 1 : (CT ? 2 : 0))>(g, lds4, gtab, tid, acc);
    if (IN == IN_I16 && g.swap_iq) {       // Q-first input: real taps, so swapping the sums is exact
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = make_float2(acc[r].y, acc[r].x);
    }

    const long long m0 = M0 + (long long)tid * R;
    if ((IN == IN_REAL || CT) && g.mix_fr != 0.0 && g.mode != SDR_MODE_ENV) {
        // input with a mixer: the taps are complex (h[i] e^{-jwi}, built by the plan) and the output takes the
        // mixer's phasor at its own sample, phase reduced in double (as fir_stream_kernel does; none for the envelope)
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


template <int R>
hipError_t launch_r(const sdr_fir_plan *p, hipStream_t st, const void *in, void *out,
                    const FirGeom &g, unsigned tiles, unsigned n_ch, const char **what)
{
    *what = "fir kernel launch";
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
#define SDR_RAW(INK)                                                                             \
    do {                                                                                         \
        SDR_OPT_IN((fir_raw_kernel<R, INK>));                                                    \
        fir_raw_kernel<R, INK><<<dim3(8u * (unsigned)g.tiles_per_xcd, n_ch), p->NT, p->lds_bytes, st>>>(in, out, p->d_gtab, g); \
    } while (0)
        if (!p->rawf) SDR_RAW(IN_I16);
        else if (p->input == SDR_IN_CF32) {
            if (p->mix_fr != 0.0) {
                SDR_OPT_IN((fir_raw_kernel<R, IN_CF32, true>));
                fir_raw_kernel<R, IN_CF32, true><<<dim3(8u * (unsigned)g.tiles_per_xcd, n_ch), p->NT, p->lds_bytes, st>>>(in, out, p->d_gtab, g);
            } else SDR_RAW(IN_CF32);
        } else SDR_RAW(IN_REAL);
#undef SDR_RAW
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

}  // namespace

hipError_t sdr::fir::launch_tile_kernels(const sdr_fir_plan *p, hipStream_t st, const void *in, void *out, const FirGeom &g,
                                         unsigned tiles, unsigned n_ch, const char **what)
{
    switch (p->R) {
    case 1: return launch_r<1>(p, st, in, out, g, tiles, n_ch, what);
    case 2: return launch_r<2>(p, st, in, out, g, tiles, n_ch, what);
    case 4: return launch_r<4>(p, st, in, out, g, tiles, n_ch, what);
    default: return launch_r<8>(p, st, in, out, g, tiles, n_ch, what);
    }
}
