// fft_common.h -- device helpers shared by the FFT translation units (fft.hip, fft_wave.hip): packed
// float32 complex arithmetic, the small in-register DFTs, and the launch geometry.  Not part of the ABI.
#ifndef SDR_FFT_COMMON_H_
#define SDR_FFT_COMMON_H_

#include "sdr_internal.h"

namespace sdr {
struct FftGeom {
    long long n_in;      // samples available
    long long frames;    // frames to compute
    long long chunk;     // consecutive frames one (workgroup, slot) pair transforms
    float *dummy;        // one spare row: slots past the last frame store there
    int hop;
};
}  // namespace sdr

namespace {
using sdr::FftGeom;

// Complex values are register pairs and every complex operation is packed float32 arithmetic
// (VOP3P: one 4-cycle wave64 instruction computes both halves; a plain v_add_f32 takes the same 4
// cycles for one): a +- b is one v_pk_add_f32, a -+ j*b one v_pk_add_f32 whose op_sel swaps the
// halves of b, a complex product one v_pk_mul_f32 + one v_pk_fma_f32.  The half selection and
// sign modifiers have no C++ spelling, hence the asm.
typedef float cf __attribute__((ext_vector_type(2)));

__device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__device__ __forceinline__ cf csub(cf a, cf b)
{
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a - j*b = (a.x + b.y, a.y - b.x)  and  a + j*b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ cf csub_j(cf a, cf b)
{
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cf cadd_j(cf a, cf b)
{
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// r = a * (p + j*q) with p = (PN ? - : +) W[PI], q = (QN ? - : +) W[QI] taken from the halves of the
// register pair W (NQN = !QN):  t = (a.x*p, a.x*q);  r = (t.x - a.y*q, t.y + a.y*p)
#define SDR_CMUL(r, a, W, PI, PN, QI, QN, NQN)                                                     \
    do {                                                                                           \
        cf t_;                                                                                     \
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0," #PI "] op_sel_hi:[0," #QI "] neg_lo:[0," #PN      \
            "] neg_hi:[0," #QN "]" : "=v"(t_) : "v"(a), "v"(W));                                   \
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1," #QI ",0] op_sel_hi:[1," #PI ",1] neg_lo:[0,"  \
            #NQN ",0] neg_hi:[0," #PN ",0]" : "=v"(r) : "v"(a), "v"(W), "v"(t_));                  \
    } while (0)

__device__ __forceinline__ cf cmul(cf a, cf w)
{
    cf r;
    SDR_CMUL(r, a, w, 0, 0, 1, 0, 1);
    return r;
}

// forward 4-point DFT (kernel e^{-j 2 pi nk/4}), natural order in and out.  ROT2: x2 stands for
// -j*x2 (the rotation folds into the first butterfly instead of costing an instruction).
template <bool ROT2 = false>
__device__ __forceinline__ void dft4(cf &x0, cf &x1, cf &x2, cf &x3)
{
    const cf t0 = ROT2 ? csub_j(x0, x2) : cadd(x0, x2), t1 = ROT2 ? cadd_j(x0, x2) : csub(x0, x2);
    const cf t2 = cadd(x1, x3), t3 = csub(x1, x3);
    x0 = cadd(t0, t2);
    x2 = csub(t0, t2);
    x1 = csub_j(t1, t3);
    x3 = cadd_j(t1, t3);
}

template <int R>
__device__ __forceinline__ void dft(cf (&x)[R]);

template <>
__device__ __forceinline__ void dft<2>(cf (&x)[2])
{
    const cf a = x[0], b = x[1];
    x[0] = cadd(a, b);
    x[1] = csub(a, b);
}

template <>
__device__ __forceinline__ void dft<4>(cf (&x)[4]) { dft4(x[0], x[1], x[2], x[3]); }

// 8 = 2 x 4:  n = a + 2m, k = b + 4c:  X[b + 4c] = sum_a W8^{ab} (-1)^{ac} DFT4_m(x[a+2m])[b]
template <>
__device__ __forceinline__ void dft<8>(cf (&x)[8])
{
    const cf W2 = {0.70710678118654752f, 0.70710678118654752f};
    cf u0[4] = {x[0], x[2], x[4], x[6]};
    cf u1[4] = {x[1], x[3], x[5], x[7]};
    dft4(u0[0], u0[1], u0[2], u0[3]);
    dft4(u1[0], u1[1], u1[2], u1[3]);
    SDR_CMUL(u1[1], u1[1], W2, 0, 0, 1, 1, 0);              // * (s - j s)
    SDR_CMUL(u1[3], u1[3], W2, 0, 1, 1, 1, 0);              // * (-s - j s)
    x[0] = cadd(u0[0], u1[0]);  x[4] = csub(u0[0], u1[0]);
    x[1] = cadd(u0[1], u1[1]);  x[5] = csub(u0[1], u1[1]);
    x[2] = csub_j(u0[2], u1[2]);  x[6] = cadd_j(u0[2], u1[2]);   // u1[2] * (-j)
    x[3] = cadd(u0[3], u1[3]);  x[7] = csub(u0[3], u1[3]);
}

// 16 = 4 x 4:  n = a + 4m, k = b + 4c:  X[b + 4c] = DFT4_a( W16^{ab} * DFT4_m(x[a+4m])[b] )[c]
template <>
__device__ __forceinline__ void dft<16>(cf (&x)[16])
{
    const cf W1 = {0.92387953251128674f, 0.38268343236508977f};     // (c1, s1)
    const cf W2 = {0.70710678118654752f, 0.70710678118654752f};     // (c2, c2)
#pragma unroll
    for (int a = 0; a < 4; ++a) dft4(x[a], x[a + 4], x[a + 8], x[a + 12]);
    // after this, x[a + 4b] = u[a][b]; twiddle W16^{ab} = exp(-j*2*pi*a*b/16)
    SDR_CMUL(x[1 + 4], x[1 + 4], W1, 0, 0, 1, 1, 0);        // ab = 1:  c1 - j s1
    SDR_CMUL(x[2 + 4], x[2 + 4], W2, 0, 0, 1, 1, 0);        // 2:  c2 - j c2
    SDR_CMUL(x[3 + 4], x[3 + 4], W1, 1, 0, 0, 1, 0);        // 3:  s1 - j c1
    SDR_CMUL(x[1 + 8], x[1 + 8], W2, 0, 0, 1, 1, 0);        // 2
    /* x[2 + 8] * (-j), ab = 4: folded into the b = 2 butterfly below */
    SDR_CMUL(x[3 + 8], x[3 + 8], W2, 0, 1, 1, 1, 0);        // 6: -c2 - j c2
    SDR_CMUL(x[1 + 12], x[1 + 12], W1, 1, 0, 0, 1, 0);      // 3
    SDR_CMUL(x[2 + 12], x[2 + 12], W2, 0, 1, 1, 1, 0);      // 6
    SDR_CMUL(x[3 + 12], x[3 + 12], W1, 0, 1, 1, 0, 1);      // 9: -c1 + j s1
    dft4(x[0], x[1], x[2], x[3]);
    dft4(x[4], x[5], x[6], x[7]);
    dft4<true>(x[8], x[9], x[10], x[11]);
    dft4(x[12], x[13], x[14], x[15]);
    // now x[4b + c] = X[b + 4c]: transpose to natural order
    cf y[16];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) y[b + 4 * c] = x[4 * b + c];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = y[i];
}

__device__ __forceinline__ int pidx(int i) { return i + (i >> 4); }

// SDR_MEASUREMENT_BUILD only (sdr_internal.h; tools/fft_ablate.sh; the results are wrong by construction):
// bit 0 = no LDS exchanges (later passes take the lane's own registers), bit 1 = no butterflies,
// bit 2 = no log, bit 3 = every workgroup stores to one row of its own (the stores stay in L2), bit 4 = the
// sample loads read one cached row.  The shipped library is built with 0.
#ifndef SDR_FFT_ABLATE
#define SDR_FFT_ABLATE 0
#endif
#ifndef SDR_FFT_NT_MIN
#define SDR_FFT_NT_MIN 256
#endif


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


}  // namespace

namespace sdr {
// fft_wave.hip: the 4096-point / hop-2048 kernel with one frame per wavefront; `two` = eight waves per
// CU (two per SIMD) instead of four
int launch_fft4096_wave(const uint32_t *iq, const float *win, const cf *tw, float *out, const FftGeom &g,
                        int n_cu, bool two, hipStream_t st);
}  // namespace sdr

#endif  // SDR_FFT_COMMON_H_
