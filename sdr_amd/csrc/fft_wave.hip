// fft_wave.hip -- the waterfall kernel for 4096-point frames at hop 2048 (BASELINE config 4): one frame
// per wavefront, no workgroup barrier in the frame loop.  Replaces, like fft.hip,
// bladerf/rx_example/src/bladerf_main_v1.cpp:183-196 (int16 -> complex float * 1/2048, fftInPlace(x, 1/N),
// 20*log10(shift(abs))) plus the window and the 50 % overlap.  Its own translation unit because it is
// built with the max-ILP scheduling strategy (sdr_amd/capi.py EXTRA_FLAGS): the kernel is bound by
// vector-instruction issue, and the default strategy leaves more dependent-issue stalls (0.723 vs
// 0.734 ms per 2^28 samples).

#include <atomic>
#include <type_traits>

#include "fft_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// fft4096_wave_kernel: ONE 4096-point frame per WAVEFRONT, no workgroup barrier in the frame loop.
// 4096 = 64 x 64: sample n = 64*n1 + n2, bin k = k1 + 64*k2,
//   X[k1 + 64 k2] = sum_n2 W64^(n2 k2) * W4096^(n2 k1) * sum_n1 x[64 n1 + n2] W64^(n1 k1).
//   pass 1: lane n2 holds its 64 samples (n1 = register index) and runs a 64-point DFT over n1 in
//           registers; the results are twiddled by W4096^(n2 k1) and written to the wave's own LDS
//           image at [k1][n2];
//   one transpose through LDS: lane k1 reads row [k1][0..63] back.  The wave's LDS operations
//           execute in order, so nothing but the data dependency (s_waitcnt lgkmcnt) separates the
//           two sides -- no s_barrier, and the four waves of a workgroup never wait for each other;
//   pass 2: lane k1 runs the 64-point DFT over n2; register k2 then holds bin k1 + 64 k2, so every
//           store instruction writes 64 consecutive floats of the fft-shifted dB row.
// Image rows are 66 complex values (528 B): 8-byte stores of consecutive lanes and 16-byte loads at
// a lane stride of 132 dwords (= 4 mod 64) are both bank-conflict free.
// The 64-point DFT is 16 x 4 on the packed-float32 butterflies above: four dft<16>, constant W64
// twiddles (eight constant pairs cover every W64^p through half selection and sign modifiers), sixteen
// dft4.  The per-lane twiddles W4096^(lane*k1), k1 = 8a + b, are applied as A[a] * B[b] from fourteen
// register pairs (63 would not fit).
// One wave per SIMD (64 points per lane are 128 registers): a persistent workgroup of four waves per
// CU, each wave walking its own run of consecutive frames with the overlapped half frame kept in
// registers and the next frame's new half requested a whole frame ahead.
// LDS: window table (16 KB) + 4 images (33 KB each) = 148 KB.
// ---------------------------------------------------------------------------------------------
// r = a * (p + j q), p = (PN ? - : +) W[PI], q = (QN ? - : +) W[QI], W a constant pair in SGPRs
#define SDR_CMULS(r, a, W, PI, PN, QI, QN, NQN)                                                    \
    do {                                                                                           \
        cf t_;                                                                                     \
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0," #PI "] op_sel_hi:[0," #QI "] neg_lo:[0," #PN      \
            "] neg_hi:[0," #QN "]" : "=v"(t_) : "v"(a), "s"(W));                                   \
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1," #QI ",0] op_sel_hi:[1," #PI ",1] neg_lo:[0,"  \
            #NQN ",0] neg_hi:[0," #PN ",0]" : "=v"(r) : "v"(a), "s"(W), "v"(t_));                  \
    } while (0)

template <int PI, int PN, int QI, int QN>
__device__ __forceinline__ cf cmul_sel(cf a, cf W)
{
    cf r;
#define SDR_SEL(pi, pn, qi, qn, nqn) \
    if constexpr (PI == pi && PN == pn && QI == qi && QN == qn) SDR_CMULS(r, a, W, pi, pn, qi, qn, nqn);
    SDR_SEL(0, 0, 0, 0, 1) SDR_SEL(0, 0, 0, 1, 0) SDR_SEL(0, 0, 1, 0, 1) SDR_SEL(0, 0, 1, 1, 0)
    SDR_SEL(0, 1, 0, 0, 1) SDR_SEL(0, 1, 0, 1, 0) SDR_SEL(0, 1, 1, 0, 1) SDR_SEL(0, 1, 1, 1, 0)
    SDR_SEL(1, 0, 0, 0, 1) SDR_SEL(1, 0, 0, 1, 0) SDR_SEL(1, 0, 1, 0, 1) SDR_SEL(1, 0, 1, 1, 0)
    SDR_SEL(1, 1, 0, 0, 1) SDR_SEL(1, 1, 0, 1, 0) SDR_SEL(1, 1, 1, 0, 1) SDR_SEL(1, 1, 1, 1, 0)
#undef SDR_SEL
    return r;
}

// the octant table: W64O[r] = (cos, sin)(2*pi*r/64), r = 1..8 (entry 0 unused)
struct W64Table { cf w[9]; };
__device__ __forceinline__ W64Table w64_table()
{
    W64Table t;
    t.w[0] = cf{1.f, 0.f};
    t.w[1] = cf{0.99518472667219688624f, 0.09801714032956060199f};
    t.w[2] = cf{0.98078528040323044913f, 0.19509032201612826785f};
    t.w[3] = cf{0.95694033573220886494f, 0.29028467725446236764f};
    t.w[4] = cf{0.92387953251128675613f, 0.38268343236508977173f};
    t.w[5] = cf{0.88192126434835502971f, 0.47139673682599764856f};
    t.w[6] = cf{0.83146961230254523708f, 0.55557023301960222474f};
    t.w[7] = cf{0.77301045336273696081f, 0.63439328416364549822f};
    t.w[8] = cf{0.70710678118654752440f, 0.70710678118654752440f};
    return t;
}

// a * W64^P = a * exp(-j*2*pi*P/64), 0 <= P < 48: P = 16 Q + r; W^r = (c, -s) with (c, s) from the
// octant table (halves swapped for r > 8); (-j)^Q turns (re, im) into (im, -re)
template <int P>
__device__ __forceinline__ cf tw64(cf a, const W64Table &T)
{
    constexpr int Q = P / 16, r = P % 16;
    if constexpr (r == 0) {
        if constexpr (Q == 0) return a;
        cf o;
        if constexpr (Q == 1) asm("v_pk_mul_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(o) : "v"(a));   // (a.y, -a.x)
        else asm("v_pk_mul_f32 %0, %1, 1.0 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]" : "=v"(o) : "v"(a));      // -a
        return o;
    } else {
        constexpr int idx = r <= 8 ? r : 16 - r;
        constexpr int ch = r <= 8 ? 0 : 1, sh = 1 - ch;          // halves that hold cos / sin of the angle
        // Q = 0: (c, -s);  Q = 1: (-s, -c);  Q = 2: (-c, s)
        if constexpr (Q == 0) return cmul_sel<ch, 0, sh, 1>(a, T.w[idx]);
        else if constexpr (Q == 1) return cmul_sel<sh, 1, ch, 1>(a, T.w[idx]);
        else return cmul_sel<ch, 1, sh, 0>(a, T.w[idx]);
    }
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// 64-point forward DFT, natural order in and out: n = 4 n1 + n2, k = k1 + 16 k2
__device__ __forceinline__ void dft64(cf (&x)[64], const W64Table &T)
{
    cf u[4][16];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
        cf t[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) t[n1] = x[4 * n1 + n2];
        dft<16>(t);
#pragma unroll
        for (int k = 0; k < 16; ++k) u[n2][k] = t[k];
    }
    static_for<0, 16>([&](auto K) {
        constexpr int k1 = decltype(K)::value;
        cf a0 = u[0][k1];
        cf a1 = tw64<k1>(u[1][k1], T);
        cf a2 = tw64<2 * k1>(u[2][k1], T);
        cf a3 = tw64<3 * k1>(u[3][k1], T);
        dft4(a0, a1, a2, a3);
        x[k1] = a0; x[k1 + 16] = a1; x[k1 + 32] = a2; x[k1 + 48] = a3;
    });
}

// The last radix-4 stage of pass 2 leaves its outputs as (re, re) and (im, im) pairs of two bins -- the same four
// v_pk_add_f32 with other half selections -- so that |X|^2 of two bins is one v_pk_mul_f32 + one v_pk_fma_f32
// (instead of two v_pk_mul_f32 + two v_add_f32 on (re, im) pairs: 64 vector instructions less per frame, in a
// kernel bound by vector-instruction issue).  p02 = |y0|^2, |y2|^2;  p13 = |y1|^2, |y3|^2.
__device__ __forceinline__ void dft4_power(cf x0, cf x1, cf x2, cf x3, cf &p02, cf &p13)
{
    const cf t0 = cadd(x0, x2), t1 = csub(x0, x2), t2 = cadd(x1, x3), t3 = csub(x1, x3);
    cf re02, im02, re13, im13;
    // y0 = t0 + t2, y2 = t0 - t2
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(re02) : "v"(t0), "v"(t2));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(im02) : "v"(t0), "v"(t2));
    // y1 = t1 - j t3 = (t1.x + t3.y, t1.y - t3.x), y3 = t1 + j t3 = (t1.x - t3.y, t1.y + t3.x)
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(re13) : "v"(t1), "v"(t3));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(im13) : "v"(t1), "v"(t3));
    p02 = re02 * re02;
    p13 = re13 * re13;
    asm("v_pk_fma_f32 %0, %1, %1, %2" : "=v"(p02) : "v"(im02), "v"(p02));
    asm("v_pk_fma_f32 %0, %1, %1, %2" : "=v"(p13) : "v"(im13), "v"(p13));
}

// dft64 whose last stage is dft4_power: pw[k1] = |X|^2 of bins k1, k1 + 32; pw[16 + k1] = bins k1 + 16, k1 + 48
__device__ __forceinline__ void dft64_power(cf (&x)[64], const W64Table &T, cf (&pw)[32])
{
    cf u[4][16];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
        cf t[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) t[n1] = x[4 * n1 + n2];
        dft<16>(t);
#pragma unroll
        for (int k = 0; k < 16; ++k) u[n2][k] = t[k];
    }
    static_for<0, 16>([&](auto K) {
        constexpr int k1 = decltype(K)::value;
        dft4_power(u[0][k1], tw64<k1>(u[1][k1], T), tw64<2 * k1>(u[2][k1], T), tw64<3 * k1>(u[3][k1], T), pw[k1], pw[16 + k1]);
    });
}

// TWO = false: four waves per CU (one per SIMD), whole 64 x 66 images, A/B twiddles in registers.
// TWO = true:  eight waves per CU (two per SIMD, 256 registers each): the image is half as tall, the
//              transpose runs in two rounds plus one half-wave register swap (see there), and the A/B
//              twiddles come from two small LDS tables.
template <bool TWO>
__global__ void __launch_bounds__(TWO ? 512 : 256, TWO ? 2 : 1)
fft4096_wave_kernel(const uint32_t *__restrict__ iq, const float *__restrict__ win,
                    const cf *__restrict__ tw, float *__restrict__ out, const FftGeom g)
{
    constexpr int N = 4096, ROW = 66, NW = TWO ? 8 : 4, NTH = 64 * NW;
    constexpr int IMG_BYTES = (TWO ? 32 : 64) * ROW * 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
    float *lwin = (float *)dyn_smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    cf *img = (cf *)(dyn_smem + N * sizeof(float) + wave * IMG_BYTES);
    cf *abtab = (cf *)(dyn_smem + N * sizeof(float) + NW * IMG_BYTES);            // TWO: [14][64]
    // TWO: frames done by each wave, for the priority hand-over between the two waves of a SIMD
    volatile int *progress = (volatile int *)(dyn_smem + N * sizeof(float) + NW * IMG_BYTES + 14 * 64 * sizeof(cf));
    if constexpr (TWO) { if (threadIdx.x < NW) progress[threadIdx.x] = 0; }

    for (int i = threadIdx.x; i < N; i += NTH) lwin[i] = win[i];
    // per-lane twiddles W4096^(lane * (8a + b)) = A[a] * B[b]
    cf A[7], B[7];
    if constexpr (TWO) {
        for (int i = threadIdx.x; i < 14 * 64; i += NTH) {
            const int r = i >> 6, l = i & 63;
            abtab[i] = r < 7 ? tw[(8 * l * (r + 1)) & (N - 1)] : tw[(l * (r - 6)) & (N - 1)];
        }
    } else {
#pragma unroll
        for (int a = 1; a < 8; ++a) {
            A[a - 1] = tw[(8 * lane * a) & (N - 1)];
            B[a - 1] = tw[(lane * a) & (N - 1)];
        }
    }
    const W64Table T = w64_table();
    __syncthreads();                       // tables filled (the only workgroup barrier)
    // measurement builds, bit 5: every wave leaves its start and end time (100 MHz counter) behind the
    // output rows, at out[(frames + 1) * N + 2 * wave_id + {0, 1}] (the tool allocates that row)
    unsigned long long t_start = 0, c_start = 0;
    if constexpr (SDR_FFT_ABLATE & 32) { t_start = __builtin_amdgcn_s_memrealtime(); c_start = __builtin_amdgcn_s_memtime(); }

    const long long f0 = ((long long)blockIdx.x * NW + wave) * g.chunk;       // this wave's run of frames
    long long f_end = f0 + g.chunk;
    if (f_end > g.frames) f_end = g.frames;
    const long long last = g.frames - 1;

    // raw samples: keep[] = first half of the coming frame (already in registers), nxt[] = its second
    // half.  Loads and stores are unconditional (a run past the last frame re-reads the last frame and
    // stores to the spare row), so no load sits behind a branch.
    uint32_t keep[32], nxt[32];
    {
        const uint32_t *src = iq + (f0 < last ? f0 : last) * g.hop + lane;
#pragma unroll
        for (int t = 0; t < 32; ++t) { keep[t] = src[64 * t]; nxt[t] = src[64 * (t + 32)]; }
    }
    for (long long it = 0; it < g.chunk; ++it) {
        const long long f = f0 + it;
        if constexpr (TWO) {
            // Waves w and w + 4 share a SIMD, and with equal priority the older one wins every issue
            // slot both want: measured, waves 0-3 finished their runs after 447 us and waves 4-7 after
            // 680 us, the last third of the kernel running one wave per SIMD.  Whoever is behind its
            // partner therefore takes the higher priority for the coming frame.
            if (lane == 0) progress[wave] = (int)it;
            const int other = __builtin_amdgcn_readfirstlane(progress[wave ^ 4]);
            if ((int)it < other) __builtin_amdgcn_s_setprio(2);
            else if ((int)it > other) __builtin_amdgcn_s_setprio(0);
            else __builtin_amdgcn_s_setprio(1);
        }
        float *row = (f < f_end ? out + f * N : g.dummy) + lane;
        cf x[64];
        // ---- window + int16 -> float: x[n1] = w[64 n1 + lane] * sample(64 n1 + lane) -------------------
#pragma unroll
        for (int t = 0; t < 64; t += 2) {
            const cf wv = {lwin[64 * t + lane], lwin[64 * (t + 1) + lane]};
            const uint32_t r0 = t < 32 ? keep[t] : nxt[t - 32], r1 = t < 32 ? keep[t + 1] : nxt[t - 31];
            const cf a = {(float)(short)(r0 & 0xffffu), (float)((int)r0 >> 16)};
            const cf b = {(float)(short)(r1 & 0xffffu), (float)((int)r1 >> 16)};
            asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(x[t]) : "v"(a), "v"(wv));
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(x[t + 1]) : "v"(b), "v"(wv));
            // two waves per SIMD have 256 registers each: keep hipcc from hoisting all 64 window reads and
            // conversions above the first multiply (it spills otherwise)
            if constexpr (TWO) { if ((t & 14) == 14) __builtin_amdgcn_sched_barrier(0); }
        }
#pragma unroll
        for (int t = 0; t < 32; ++t) keep[t] = nxt[t];
        auto prefetch = [&]() {            // request the next frame's new half
            const uint32_t *src = iq + (f + 1 < last ? f + 1 : last) * g.hop + lane;
#pragma unroll
            for (int t = 0; t < 32; ++t) nxt[t] = src[64 * (t + 32)];
        };
        // one wave per SIMD: a whole frame of lead; two (256 registers each): after pass 1, whose
        // temporaries need the room -- the transposes, pass 2 and the partner wave's share of the SIMD
        // still give the loads several microseconds
        if constexpr (!TWO) prefetch();
        // ---- pass 1 -------------------------------------------------------------------------------
        if constexpr (!(SDR_FFT_ABLATE & 2)) dft64(x, T);
        if constexpr (TWO) {
            prefetch();
#pragma unroll
            for (int r = 0; r < 7; ++r) { A[r] = abtab[r * 64 + lane]; B[r] = abtab[(r + 7) * 64 + lane]; }
        }
        if constexpr (!(SDR_FFT_ABLATE & 2)) {
#pragma unroll
            for (int k1 = 1; k1 < 64; ++k1) {
                if (k1 & 7) x[k1] = cmul(x[k1], B[(k1 & 7) - 1]);
                if (k1 >> 3) x[k1] = cmul(x[k1], A[(k1 >> 3) - 1]);
            }
        }
        // ---- transpose through the wave's own image ----------------------------------------------
        if constexpr (SDR_FFT_ABLATE & 1) {
        } else if constexpr (TWO) {
            // Two rounds through an image of 32 rows.  Round 1 holds rows k1 < 32, round 2 rows k1 >= 32;
            // in each, EVERY lane reads 32 values: lane L takes row (L & 31) of the round, columns
            // 32*(L >> 5) .. +31.  Lane r then holds the left halves of rows r and 32 + r, lane r + 32
            // the right halves; one v_permlane32_swap per register trades lane r's half of row 32 + r
            // for lane r + 32's half of row r, and every lane L owns the whole row L.  Each round refills
            // the 64 registers it has just emptied: the transpose needs no registers beyond x[].
            const cf *rd = img + (lane & 31) * ROW + 32 * (lane >> 5);
            cf P[32], Q[32];
#pragma unroll
            for (int k1 = 0; k1 < 32; ++k1) img[k1 * ROW + lane] = x[k1];
#pragma unroll
            for (int j = 0; j < 32; j += 2) {
                const float4 v = *(const float4 *)(rd + j);
                P[j] = cf{v.x, v.y};
                P[j + 1] = cf{v.z, v.w};
            }
#pragma unroll
            for (int k1 = 32; k1 < 64; ++k1) img[(k1 - 32) * ROW + lane] = x[k1];
#pragma unroll
            for (int j = 0; j < 32; j += 2) {
                const float4 v = *(const float4 *)(rd + j);
                Q[j] = cf{v.x, v.y};
                Q[j + 1] = cf{v.z, v.w};
            }
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                // lanes 32-63 of P <-> lanes 0-31 of Q
                const auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(P[j].x), __float_as_uint(Q[j].x), false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(P[j].y), __float_as_uint(Q[j].y), false, false);
                x[j] = cf{__uint_as_float(sx[0]), __uint_as_float(sy[0])};
                x[32 + j] = cf{__uint_as_float(sx[1]), __uint_as_float(sy[1])};
            }
        } else {
#pragma unroll
            for (int k1 = 0; k1 < 64; ++k1) img[k1 * ROW + lane] = x[k1];
#pragma unroll
            for (int n2 = 0; n2 < 64; n2 += 2) {
                const float4 v = *(const float4 *)(img + lane * ROW + n2);
                x[n2] = cf{v.x, v.y};
                x[n2 + 1] = cf{v.z, v.w};
            }
        }
        // ---- pass 2 -------------------------------------------------------------------------------
        cf pw[32];
        if constexpr (!(SDR_FFT_ABLATE & 2)) dft64_power(x, T, pw);
        else {
#pragma unroll
            for (int k = 0; k < 32; ++k) pw[k] = x[k] * x[k + 32];
        }
        // the prefetched half frame has had the whole frame to arrive; waiting for it here, BEFORE this
        // frame's stores are issued, keeps the wait from covering the stores (vmcnt counts in issue order)
#pragma unroll
        for (int t = 0; t < 32; ++t) asm volatile("" ::"v"(nxt[t]));
        // ---- |X|^2 -> dB, fft-shifted row: pw[i] holds bins lane + 64 i and lane + 64 (i + 32)
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const cf p = cf{__log2f(pw[i].x), __log2f(pw[i].y)} * 3.01029995663981195f;
            __builtin_nontemporal_store(p.x, &row[64 * (i + 32)]);       // bin lane + 64 i
            __builtin_nontemporal_store(p.y, &row[64 * i]);              // bin lane + 64 (i + 32)
        }
    }
    if constexpr (SDR_FFT_ABLATE & 32) {
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime(), c_end = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned *st = (unsigned *)(out + (g.frames + 1) * N) + 2 * (blockIdx.x * NW + wave);
            st[0] = (unsigned)t_start;
            st[1] = (unsigned)t_end;
            // shader cycles of the wave's life, one row further on (clock = cycles / time)
            ((unsigned *)(out + (g.frames + 2) * N))[blockIdx.x * NW + wave] = (unsigned)(c_end - c_start);
        }
    }
}

}  // namespace

namespace sdr {

#ifdef SDR_MEASUREMENT_BUILD
int measurement_flags_fft_wave() { return SDR_FFT_ABLATE ? SDR_FFT_ABLATE : -1; }
#else
int measurement_flags_fft_wave() { return 0; }
#endif

int launch_fft4096_wave(const uint32_t *iq, const float *win, const cf *tw, float *out, const FftGeom &g,
                        int n_cu, bool two, hipStream_t st)
{
    // the kernels' dynamic LDS limit is raised once per DEVICE (the attribute belongs to the function as
    // loaded on one device; a process may drive several GPUs); the flag is set after a successful launch
    constexpr int kMaxDevices = 64;
    static std::atomic<bool> opted_dev[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = -1;
    FftGeom gw = g;
    const int nw = two ? 8 : 4;
    const long long waves = (long long)nw * n_cu;
    gw.chunk = (g.frames + waves - 1) / waves;
    const unsigned nb = (unsigned)(((g.frames + gw.chunk - 1) / gw.chunk + nw - 1) / nw);
    const size_t lds = 4096 * sizeof(float) + (two ? 8 * (size_t)(32 * 66 * 8) + 14 * 64 * 8 + 64 : 4 * (size_t)(64 * 66 * 8));
    if (dev < 0 || !opted_dev[dev].load(std::memory_order_relaxed)) {
        hipError_t ea = hipFuncSetAttribute((const void *)fft4096_wave_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ea == hipSuccess) ea = hipFuncSetAttribute((const void *)fft4096_wave_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ea != hipSuccess) return hip_fail(ea, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) of the fft wave kernel");
    }
    if (two) fft4096_wave_kernel<true><<<nb, 512, lds, st>>>(iq, win, tw, out, gw);
    else fft4096_wave_kernel<false><<<nb, 256, lds, st>>>(iq, win, tw, out, gw);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "fft wave kernel launch");
    if (dev >= 0) opted_dev[dev].store(true, std::memory_order_relaxed);
    return SDR_OK;
}

}  // namespace sdr
