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
// outputs; lane l owns outputs m0 = M0 + l*R .. m0+R-1.  The input window of the tile
//   samples W0 = D*M0 - Tback  ..  D*(M0+NT*R) + 1      (Tback = (T-1) rounded up to 4)
// i.e. the tile's own D*NT*R samples plus the overlap-save halo of T-1 history samples (and D
// more on the right for the discriminator's successor), is staged once in LDS:
//   * int16 IQ without mixer: the raw 4-byte samples are copied HBM -> LDS by LDS-DMA
//     (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPR round trip) and converted
//     after the LDS read (the 2^-11 scale is folded into the taps, exact);
//   * every other input (complex float, real float, any input with the NCO mixer on) is
//     converted / mixed once while staging and kept as float2.
// Lane l then walks its window [D*R*l, D*R*l + Lw) with 16-byte LDS reads.  One loaded sample
// feeds all R outputs of the lane: output r uses tap D*r + Tback - j for window offset j, which
// the host has laid out as a zero-padded table gtab[j][r]; the table index is wave-uniform, so
// taps arrive through scalar loads and cost no LDS or VALU work.  Lane stride in LDS is
// q = D*R/(samples per 16 B) granules; when q is even the tile is stored with one pad granule
// per q (stride q+1, odd) so that the 16-lane groups of ds_read_b128 hit distinct banks.
//
// Numerics: float32 fmaf accumulation in ascending window order; decimation indexing is exact
// (output m <-> input D*m); zero pre-history (samples < 0 read as 0) as the reference.

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
    int ngran;              // 16-byte granules in one lane's window
    int q;                  // lane stride in sample granules
    int pad;                // 1: one pad granule after every q sample granules
    unsigned magic;         // ceil(2^32/(q+1)) for the staging division
    int s_gran;             // sample granules staged per tile
    int a_gran;             // LDS granules per tile (pads included, rounded up to NT)
    int mode;               // sdr_mode
    int input;              // sdr_input
    float scale;            // discriminator scale
};

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

__device__ __forceinline__ int lds_granule(const FirGeom &g, int sg)
{
    // LDS granule that holds sample granule sg
    return g.pad ? sg + sg / g.q : sg;
}

// sample granule held by LDS granule a (pad granules duplicate their left neighbour)
__device__ __forceinline__ int sample_granule(const FirGeom &g, int a)
{
    if (!g.pad) return a;
    const int blk = (int)__umulhi((unsigned)a, g.magic);        // a / (q+1)
    const int rem = a - blk * (g.q + 1);
    return blk * g.q + (rem < g.q ? rem : g.q - 1);
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

// polar_discriminator, bladerf_apt_demod.cpp:419-421: tmp = y1 * conj(y0); scale*atan2(im, re).
// Written as libstdc++ evaluates the product, (a*c - b*e, a*e + b*c) with e = -d, so that a
// zero y0 produces the same signed zeros (and therefore the same 0 / +-pi) as the reference.
__device__ __forceinline__ float discriminate(float2 y1, float2 y0, float scale)
{
    const float a = y1.x, b = y1.y, c = y0.x, e = -y0.y;
    const float re = __fmaf_rn(a, c, -(b * e));
    const float im = __fmaf_rn(a, e, b * c);
    return scale * atan2f(im, re);
}

template <int R>
__device__ __forceinline__ void store_outputs(const FirGeom &g, void *__restrict__ outv,
                                              long long ch, long long m0, const float2 (&y)[R],
                                              float2 ynext)
{
    if (g.mode == SDR_MODE_IQ) {
        float2 *out = (float2 *)outv + ch * g.out_stride;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long j = m0 + r;
            if (j >= g.out_begin && j < g.out_end) out[j - g.out_begin] = y[r];
        }
        return;
    }
    float v[R];
    if (g.mode == SDR_MODE_FM) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = discriminate(r + 1 < R ? y[r + 1] : ynext, y[r], g.scale);
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = hypotf(y[r].x, y[r].y);
    }
    float *out = (float *)outv + ch * g.out_stride;
    float *dst = out + (m0 - g.out_begin);
    const bool whole = (m0 >= g.out_begin) && (m0 + R <= g.out_end);
    if (R >= 2 && whole && (((uintptr_t)dst) & (R >= 4 ? 15 : 7)) == 0) {
        if (R == 2) {
            *(float2 *)dst = make_float2(v[0], v[1]);
        } else {
#pragma unroll
            for (int r = 0; r < R; r += 4)
                *(float4 *)(dst + r) = make_float4(v[r], v[(r + 1) % R], v[(r + 2) % R], v[(r + 3) % R]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long j = m0 + r;
            if (j >= g.out_begin && j < g.out_end) dst[r] = v[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// int16 IQ, no mixer: raw samples in LDS (4 B each, 4 per granule), LDS-DMA staging.
// ---------------------------------------------------------------------------------------------
template <int R>
__global__ void __launch_bounds__(256)
fir_raw_kernel(const uint32_t *__restrict__ in, void *__restrict__ outv,
               const float *__restrict__ gtab, const float *__restrict__ taps, const FirGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *lds4 = (uint4 *)smem;
    const int tid = threadIdx.x;
    const int NT = blockDim.x;
    const long long ch = blockIdx.y;
    const long long M0 = g.tile0 + (long long)blockIdx.x * (NT * R);
    const long long W0 = M0 * g.D - g.Tback;                 // first staged sample (multiple of 4)
    const uint32_t *src = in + ch * g.in_stride;             // sample 0 of this channel

    // ---- stage the tile ----------------------------------------------------------------
    const int n_inst = g.a_gran / NT;
    const bool interior = (W0 >= g.lo_valid) && (W0 + 4LL * g.s_gran <= g.hi_valid);
    if (interior) {
        const uint32_t *wsrc = src + W0;
        for (int k = 0; k < n_inst; ++k) {
            const int a = k * NT + tid;
            int sg = sample_granule(g, a);
            sg = sg < g.s_gran ? sg : 0;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wsrc + 4 * sg),
                                             (lds_ptr_t)(lds4 + k * NT + (tid & ~63)), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (int k = 0; k < n_inst; ++k) {
            const int a = k * NT + tid;
            const int sg = sample_granule(g, a);
            const long long s = W0 + 4LL * sg;
            uint32_t w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long si = s + u;
                w[u] = (si >= g.lo_valid && si < g.hi_valid) ? src[si] : 0u;
            }
            lds4[a] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    __syncthreads();

    // ---- FIR: lane window walk -----------------------------------------------------------
    float2 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float2(0.f, 0.f);

    const int lane_g = tid * (g.q + g.pad);
    int padoff = 0, cq = 0;
    for (int c = 0; c < g.ngran; ++c) {
        const uint4 v = lds4[lane_g + c + padoff];
        if (g.pad) { if (++cq == g.q) { cq = 0; ++padoff; } }
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        const float *gt = gtab + (size_t)c * (4 * R);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float xi = (float)(short)(w[u] & 0xffffu);
            const float xq = (float)((int)w[u] >> 16);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float h = gt[u * R + r];
                acc[r].x = __fmaf_rn(h, xi, acc[r].x);
                acc[r].y = __fmaf_rn(h, xq, acc[r].y);
            }
        }
    }

    // ---- epilogue --------------------------------------------------------------------------
    const long long m0 = M0 + (long long)tid * R;
    float2 ynext = make_float2(0.f, 0.f);
    if (g.mode == SDR_MODE_FM) {
        float2 *ybuf = (float2 *)(smem + (size_t)g.a_gran * 16);
        ybuf[tid] = acc[0];
        if (tid < 64) {
            // successor of the tile's last output, y[M0 + NT*R], summed across the wave
            float2 part = make_float2(0.f, 0.f);
            const uint32_t *lds1 = (const uint32_t *)smem;
            const int sbase = g.D * NT * R + g.Tback;      // window offset of sample D*(M0+NT*R)
            for (int i = tid; i < g.T; i += 64) {
                const int sidx = sbase - i;
                const uint32_t wv = lds1[lds_granule(g, sidx >> 2) * 4 + (sidx & 3)];
                const float h = taps[i];
                part.x = __fmaf_rn(h, (float)(short)(wv & 0xffffu), part.x);
                part.y = __fmaf_rn(h, (float)((int)wv >> 16), part.y);
            }
            part = wave_sum(part);
            if (tid == 0) ybuf[NT] = part;
        }
        __syncthreads();
        ynext = ybuf[tid + 1];
    }
    store_outputs<R>(g, outv, ch, m0, acc, ynext);
}

// ---------------------------------------------------------------------------------------------
// every other input: float2 samples in LDS (8 B each, 2 per granule), staged through registers
// with the optional NCO mixer applied once per sample.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 load_sample(const FirGeom &g, const void *__restrict__ src,
                                              long long si)
{
    if (si < g.lo_valid || si >= g.hi_valid) return make_float2(0.f, 0.f);
    float2 x;
    if (g.input == SDR_IN_I16_IQ) {
        const uint32_t w = ((const uint32_t *)src)[si];
        x = make_float2((float)(short)(w & 0xffffu), (float)((int)w >> 16));
    } else if (g.input == SDR_IN_CF32) {
        x = ((const float2 *)src)[si];
    } else {
        x = make_float2(((const float *)src)[si], 0.f);
    }
    if (g.mix_fr != 0.0) {
        // exp(+j*2*pi*fr*n): the phase in revolutions is reduced in double, so it is exact to
        // ~1e-16*n*fr before the float sincospi
        const double ph = g.mix_fr * (double)si;
        const float fr = (float)(ph - rint(ph));
        float sn, cs;
        sincospif(2.0f * fr, &sn, &cs);
        x = make_float2(__fmaf_rn(x.x, cs, -(x.y * sn)), __fmaf_rn(x.x, sn, x.y * cs));
    }
    return x;
}

template <int R>
__global__ void __launch_bounds__(256)
fir_f2_kernel(const void *__restrict__ inv, void *__restrict__ outv,
              const float *__restrict__ gtab, const float *__restrict__ taps, const FirGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *lds4 = (float4 *)smem;
    const int tid = threadIdx.x;
    const int NT = blockDim.x;
    const long long ch = blockIdx.y;
    const long long M0 = g.tile0 + (long long)blockIdx.x * (NT * R);
    const long long W0 = M0 * g.D - g.Tback;
    const size_t esz = g.input == SDR_IN_CF32 ? 8 : 4;
    const void *src = (const unsigned char *)inv + (size_t)(ch * g.in_stride) * esz;

    const int n_inst = g.a_gran / NT;
    for (int k = 0; k < n_inst; ++k) {
        const int a = k * NT + tid;
        const int sg = sample_granule(g, a);
        const long long s = W0 + 2LL * sg;
        const float2 x0 = load_sample(g, src, s);
        const float2 x1 = load_sample(g, src, s + 1);
        lds4[a] = make_float4(x0.x, x0.y, x1.x, x1.y);
    }
    __syncthreads();

    float2 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float2(0.f, 0.f);

    const int lane_g = tid * (g.q + g.pad);
    int padoff = 0, cq = 0;
    for (int c = 0; c < g.ngran; ++c) {
        const float4 v = lds4[lane_g + c + padoff];
        if (g.pad) { if (++cq == g.q) { cq = 0; ++padoff; } }
        const float *gt = gtab + (size_t)c * (2 * R);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float h0 = gt[r], h1 = gt[R + r];
            acc[r].x = __fmaf_rn(h0, v.x, acc[r].x);
            acc[r].y = __fmaf_rn(h0, v.y, acc[r].y);
            acc[r].x = __fmaf_rn(h1, v.z, acc[r].x);
            acc[r].y = __fmaf_rn(h1, v.w, acc[r].y);
        }
    }

    const long long m0 = M0 + (long long)tid * R;
    float2 ynext = make_float2(0.f, 0.f);
    if (g.mode == SDR_MODE_FM) {
        float2 *ybuf = (float2 *)(smem + (size_t)g.a_gran * 16);
        ybuf[tid] = acc[0];
        if (tid < 64) {
            float2 part = make_float2(0.f, 0.f);
            const float2 *lds2 = (const float2 *)smem;
            const int sbase = g.D * NT * R + g.Tback;
            for (int i = tid; i < g.T; i += 64) {
                const int sidx = sbase - i;
                const float2 x = lds2[lds_granule(g, sidx >> 1) * 2 + (sidx & 1)];
                const float h = taps[i];
                part.x = __fmaf_rn(h, x.x, part.x);
                part.y = __fmaf_rn(h, x.y, part.y);
            }
            part = wave_sum(part);
            if (tid == 0) ybuf[NT] = part;
        }
        __syncthreads();
        ynext = ybuf[tid + 1];
    }
    store_outputs<R>(g, outv, ch, m0, acc, ynext);
}

}  // namespace

// =============================================================================================
// host side
// =============================================================================================
struct sdr_fir_plan {
    int input, mode, D, T, R, NT;
    bool raw;               // raw int16 LDS kernel (int16 input, mixer off)
    double scale, mix_fr;
    int Tback, Lw4, ngran, q, pad, s_gran, a_gran, lds_bytes, spg;   // spg: samples per granule
    unsigned magic;
    float *d_gtab = nullptr, *d_taps = nullptr;
    // staging buffers for the *_host entry point
    void *d_in = nullptr, *d_out = nullptr;
    size_t cap_in = 0, cap_out = 0;
    int device = 0;
};

namespace {

int roundup(int v, int m) { return (v + m - 1) / m * m; }

bool r_allowed(int D, int R, int spg) { return ((long long)D * R) % spg == 0; }

int window_len(int T, int D, int R) { return roundup(roundup(T - 1, 4) + D * (R - 1) + 1, 4); }

// LDS bytes of one workgroup for a given geometry (tile + discriminator hand-off buffer)
void tile_geometry(int T, int D, int R, int NT, int spg, int *q, int *pad, int *s_gran,
                   int *a_gran, int *lds_bytes)
{
    const int Tback = roundup(T - 1, 4);
    const int Lw4 = window_len(T, D, R);
    *q = D * R / spg;
    *pad = (*q % 2 == 0) ? 1 : 0;
    const long long s_need = (long long)D * R * (NT - 1) + Lw4;
    const long long s_fm = (long long)D * R * NT + Tback + 1;
    const long long s_tile = s_need > s_fm ? s_need : s_fm;
    *s_gran = (int)((s_tile + spg - 1) / spg);
    int a = *s_gran + (*pad ? *s_gran / *q + 1 : 0);
    *a_gran = roundup(a, NT);
    *lds_bytes = *a_gran * 16 + (NT + 1) * 8;
}

template <int R>
hipError_t launch_r(const sdr_fir_plan *p, dim3 grid, hipStream_t st, const void *in, void *out,
                    const FirGeom &g)
{
    if (p->raw) {
        if (p->lds_bytes > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)fir_raw_kernel<R>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, p->lds_bytes);
        fir_raw_kernel<R><<<grid, p->NT, p->lds_bytes, st>>>((const uint32_t *)in, out, p->d_gtab,
                                                            p->d_taps, g);
    } else {
        if (p->lds_bytes > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)fir_f2_kernel<R>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, p->lds_bytes);
        fir_f2_kernel<R><<<grid, p->NT, p->lds_bytes, st>>>(in, out, p->d_gtab, p->d_taps, g);
    }
    return hipGetLastError();
}

size_t in_elem_bytes(int input) { return input == SDR_IN_CF32 ? 8 : 4; }
size_t out_elem_bytes(int mode) { return mode == SDR_MODE_IQ ? 8 : 4; }

// Enqueue one launch.  `in` points at stream sample index in_first of channel 0.
int launch(sdr_fir_plan *p, const void *in, long long in_first, long long n_resident,
           long long n_total, long long out_begin, long long out_end, long long in_stride,
           size_t n_ch, void *out, long long out_stride, hipStream_t st)
{
    if (out_end <= out_begin || n_ch == 0) return SDR_OK;
    if (n_ch > 65535) { sdr::set_error("n_channels > 65535"); return SDR_EINVAL; }
    const size_t ia = in_elem_bytes(p->input) == 8 ? 8 : 4;
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
    g.magic = p->magic; g.s_gran = p->s_gran; g.a_gran = p->a_gran; g.mode = p->mode;
    g.input = p->input; g.scale = (float)p->scale;
    const long long per_tile = (long long)p->NT * p->R;
    const long long tiles = (out_end - g.tile0 + per_tile - 1) / per_tile;
    if (tiles > 0x7fffffffLL) { sdr::set_error("too many tiles"); return SDR_EINVAL; }
    // virtual pointer to stream sample 0 of channel 0 (only dereferenced inside the valid window)
    const void *base = (const unsigned char *)in - (size_t)in_first * in_elem_bytes(p->input);
    dim3 grid((unsigned)tiles, (unsigned)n_ch);
    hipError_t e;
    switch (p->R) {
    case 1: e = launch_r<1>(p, grid, st, base, out, g); break;
    case 2: e = launch_r<2>(p, grid, st, base, out, g); break;
    case 4: e = launch_r<4>(p, grid, st, base, out, g); break;
    default: e = launch_r<8>(p, grid, st, base, out, g); break;
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
    if (d->input < SDR_IN_I16_IQ || d->input > SDR_IN_F32_REAL) { sdr::set_error("bad input kind"); return SDR_EINVAL; }
    if (d->mode < SDR_MODE_IQ || d->mode > SDR_MODE_ENV) { sdr::set_error("bad mode"); return SDR_EINVAL; }
    int rc = sdr::ensure_device();
    if (rc != SDR_OK) return rc;

    sdr_fir_plan *p = new sdr_fir_plan();
    p->input = d->input; p->mode = d->mode; p->D = d->decim; p->T = d->n_taps;
    p->scale = d->scale; p->mix_fr = d->mix_fr;
    p->raw = (d->input == SDR_IN_I16_IQ && d->mix_fr == 0.0);
    p->spg = p->raw ? 4 : 2;
    (void)hipGetDevice(&p->device);

    // ---- choose outputs per lane (R) and workgroup size (NT) --------------------------------
    const int Rs[4] = {1, 2, 4, 8};
    const int NTs[3] = {256, 128, 64};
    int bestR = 0, bestNT = 0;
    double best_cost = 0;
    for (int ri = 0; ri < 4; ++ri) {
        const int R = Rs[ri];
        if (d->outputs_per_lane && d->outputs_per_lane != R) continue;
        if (!r_allowed(p->D, R, p->spg)) continue;
        const int Lw4 = window_len(p->T, p->D, R);
        // VALU lane-ops per output: conversions (raw only) + 2 FMAs per (sample, output)
        const double cost = (double)Lw4 * ((p->raw ? 2.0 : 0.0) + 2.0 * R) / R;
        for (int ni = 0; ni < 3; ++ni) {
            const int NT = NTs[ni];
            if (d->threads && d->threads != NT) continue;
            int q, pad, sg, ag, lds;
            tile_geometry(p->T, p->D, R, NT, p->spg, &q, &pad, &sg, &ag, &lds);
            const int budget = d->threads || ni == 2 ? 160 * 1024 : 52 * 1024;   // >= 3 WGs per CU
            if (lds > budget) continue;
            if (!bestR || cost < best_cost * 0.97) { bestR = R; bestNT = NT; best_cost = cost; }
            break;
        }
    }
    if (!bestR) {
        delete p;
        sdr::set_error("no kernel geometry for decim=%d taps=%d (outputs_per_lane=%d threads=%d)",
                       d->decim, d->n_taps, d->outputs_per_lane, d->threads);
        return SDR_EINVAL;
    }
    p->R = bestR; p->NT = bestNT;
    p->Tback = roundup(p->T - 1, 4);
    p->Lw4 = window_len(p->T, p->D, p->R);
    p->ngran = p->Lw4 / p->spg;
    tile_geometry(p->T, p->D, p->R, p->NT, p->spg, &p->q, &p->pad, &p->s_gran, &p->a_gran,
                  &p->lds_bytes);
    p->magic = (unsigned)((0x100000000ULL + (unsigned)p->q) / (unsigned)(p->q + 1));  // ceil(2^32/(q+1))

    // ---- tap tables ---------------------------------------------------------------------------
    // taps rounded to float32; for int16 input the exact 2^-11 Q11 scale is folded in
    const float fold = (p->input == SDR_IN_I16_IQ) ? sdr::kQ11 : 1.0f;
    std::vector<float> hf(p->T);
    for (int i = 0; i < p->T; ++i) hf[i] = (float)d->taps[i] * fold;
    // gtab[j][r] = h[D*r + Tback - j]  (tap that window offset j contributes to output r)
    std::vector<float> gt((size_t)p->Lw4 * p->R, 0.0f);
    for (int j = 0; j < p->Lw4; ++j)
        for (int r = 0; r < p->R; ++r) {
            const long long i = (long long)p->D * r + p->Tback - j;
            if (i >= 0 && i < p->T) gt[(size_t)j * p->R + r] = hf[i];
        }
    hipError_t e = hipMalloc((void **)&p->d_gtab, gt.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_taps, hf.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->d_gtab, gt.data(), gt.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_taps, hf.data(), hf.size() * sizeof(float), hipMemcpyHostToDevice);
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
    if (p->d_taps) (void)hipFree(p->d_taps);
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

extern "C" int sdr_fir_execute_dev(sdr_fir_plan *p, const void *d_in, size_t n_in, void *d_out,
                                   size_t out_capacity, size_t *n_out, void *stream)
{
    if (!p || (!d_in && n_in) ) { sdr::set_error("null argument"); return SDR_EINVAL; }
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
    if (in_elem_bytes(p->input) == 8 ? false : false) {}
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
