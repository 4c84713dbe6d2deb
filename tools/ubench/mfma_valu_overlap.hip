// Micro-benchmark: do vector-ALU instructions of one wave issue while another wave of the same SIMD
// keeps the matrix pipe busy?  A workgroup of 8 (or 12) waves on one CU: waves 0-3 (one per SIMD) run
// a stream of v_mfma_i32_16x16x64_i8 (16 cycles each), waves 4-7 a stream of independent v_fma_f32
// (4 cycles each, four per MFMA: the same time alone), optional waves 8-11 a stream of ds_read_b128.
// Each role alone, then together, with the vector waves younger (default launch order), and with
// s_setprio raised on them.  If the pipes overlap, "together" takes as long as the slower role
// alone; if the SIMD issues one instruction at a time regardless of the pipe, it takes the sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));

// mask bit 0: matrix waves work, bit 1: vector waves work, bit 2: LDS waves work
// prio: s_setprio of the vector (and LDS) waves
template <int PRIO, bool SAME_WAVE>
__global__ void __launch_bounds__(768, 1) overlap(uint32_t *out, int iters, int mask)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 768) ((uint32_t *)lds)[i] = i * 2654435761u;
    __syncthreads();
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    uint32_t sink = 0;
    if (wave < 4) {
        if (mask & 1) {
            i32x4 a = {lane, lane + 1, lane + 2, lane + 3}, b = {lane * 3, lane * 5, lane * 7, lane * 9};
            i32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            float f[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) f[k] = (float)(lane + k);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
                    if (SAME_WAVE) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) f[k] = __fmaf_rn(f[k], 1.0001f, 0.5f);
                    }
                    c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
                    if (SAME_WAVE) {
#pragma unroll
                        for (int k = 3; k < 6; ++k) f[k] = __fmaf_rn(f[k], 1.0001f, 0.5f);
                    }
                    c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
                    if (SAME_WAVE) {
#pragma unroll
                        for (int k = 6; k < 9; ++k) f[k] = __fmaf_rn(f[k], 1.0001f, 0.5f);
                    }
                    c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
                    if (SAME_WAVE) {
#pragma unroll
                        for (int k = 9; k < 12; ++k) f[k] = __fmaf_rn(f[k], 1.0001f, 0.5f);
                    }
                }
            }
            sink = (uint32_t)(c0[0] ^ c1[1] ^ c2[2] ^ c3[3]);
            if (SAME_WAVE)
#pragma unroll
                for (int k = 0; k < 12; ++k) sink ^= __float_as_uint(f[k]);
        }
    } else if (wave < 8) {
        if (mask & 2) {
            __builtin_amdgcn_s_setprio(PRIO);
            float f[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) f[k] = (float)(lane + k);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u)            // 64 fma per iteration = 16 MFMA-times
#pragma unroll
                    for (int k = 0; k < 16; ++k) f[k] = __fmaf_rn(f[k], 1.0001f, 0.5f);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) sink ^= __float_as_uint(f[k]);
        }
    } else {
        if (mask & 4) {
            __builtin_amdgcn_s_setprio(PRIO);
            const uint4 *src = (const uint4 *)lds + lane;
            uint4 acc = {0, 0, 0, 0};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {          // 16 reads of 1 KB per iteration: 8 cycles each at 128 B/clk
                    const uint4 v = src[((u * 67 + it) & 31) * 64];
                    acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
                }
            }
            sink = acc.x ^ acc.y ^ acc.z ^ acc.w;
        }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if (lane == 0) {
        out[(blockIdx.x * 12 + wave) * 2] = (uint32_t)(t1 - t0);
        out[(blockIdx.x * 12 + wave) * 2 + 1] = sink;
    }
}


// second experiment: the matrix wave leaves the issue port alone for a while after each MFMA
// (s_nop NOP-1, or NV independent v_fma_f32 of its own); do the younger vector waves get in?
template <int NV, int NOP, int PRIO>
__global__ void __launch_bounds__(512, 1) gaps(uint32_t *out, int iters, int mask)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    uint32_t sink = 0;
    long long done = 0;
    if (wave < 4) {
        if (mask & 1) {
            i32x4 a = {lane, lane + 1, lane + 2, lane + 3}, b = {lane * 3, lane * 5, lane * 7, lane * 9};
            i32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            float f[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) f[k] = (float)(lane + k);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    c[u & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[u & 3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < NV; ++k) f[(u * NV + k) & 15] = __fmaf_rn(f[(u * NV + k) & 15], 1.0001f, 0.5f);
                    if constexpr (NOP > 0) asm volatile("s_nop %0" ::"n"(NOP - 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            sink = (uint32_t)(c[0][0] ^ c[1][1] ^ c[2][2] ^ c[3][3]);
#pragma unroll
            for (int k = 0; k < 16; ++k) sink ^= __float_as_uint(f[k]);
        }
    } else if (mask & 2) {
        __builtin_amdgcn_s_setprio(PRIO);
        float f[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) f[k] = (float)(lane + k);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int k = 0; k < 16; ++k) f[k] = __fmaf_rn(f[k], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) sink ^= __float_as_uint(f[k]);
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if (lane == 0) {
        out[(blockIdx.x * 12 + wave) * 2] = (uint32_t)(t1 - t0);
        out[(blockIdx.x * 12 + wave) * 2 + 1] = sink;
    }
}

template <int NV, int NOP, int PRIO>
static void run_gaps(uint32_t *d_out, int iters)
{
    static uint32_t h[256 * 12 * 2];
    double m_alone = 0, m_both = 0, v_both = 0, v_alone = 0;
    for (int mask = 1; mask <= 3; ++mask) {
        for (int w = 0; w < 2; ++w) gaps<NV, NOP, PRIO><<<256, 512>>>(d_out, iters, mask);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
        double cyc[2] = {0, 0};
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < 8; ++w) cyc[w / 4] += h[(b * 12 + w) * 2];
        if (mask == 1) m_alone = cyc[0] / 1024 / iters / 16;
        if (mask == 2) v_alone = cyc[1] / 1024 / iters / 64;
        if (mask == 3) { m_both = cyc[0] / 1024 / iters / 16; v_both = cyc[1] / 1024 / iters / 64; }
    }
    printf("matrix wave: MFMA + %d fma + s_nop x%-2d  vector prio %d | cycles per MFMA alone %5.1f, with vector waves %5.1f | cycles per "
           "vector fma alone %4.2f, with matrix waves %5.2f (while the matrix wave runs: %.2f fma per MFMA)\n",
           NV, NOP, PRIO, m_alone, m_both, v_alone, v_both, v_both * 64 > m_both * 16 ? (m_both * 16) / (v_both * 64 - v_alone * 64 + 1e-9) * 0 + (64.0 - (v_both * 64 - m_both * 16) / v_alone) / 16.0 : 4.0);
}


// third experiment: which vector instructions issue in the shadow of another wave's MFMA stream?
// Vector waves run 2 x 64 instructions of one kind per iteration (inline asm, 16 independent registers).
template <int OP>
__device__ __forceinline__ void vop(float &x, float2 &xx)
{
    if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(x));
    if constexpr (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(xx));
    if constexpr (OP == 2) asm volatile("v_perm_b32 %0, %0, %0, %0" : "+v"(x));
    if constexpr (OP == 3) asm volatile("v_lshl_add_u32 %0, %0, 8, %0" : "+v"(x));
    if constexpr (OP == 4) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x));
    if constexpr (OP == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
    if constexpr (OP == 6) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x));
    if constexpr (OP == 7) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(xx));
    if constexpr (OP == 8) asm volatile("v_mov_b32 %0, %0" : "+v"(x));
    if constexpr (OP == 9) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(xx));
    if constexpr (OP == 10) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(x));
    if constexpr (OP == 11) asm volatile("v_xor_b32 %0, %0, %0" : "+v"(x));
    if constexpr (OP == 12) asm volatile("v_max_f32 %0, %0, %0" : "+v"(x));
    if constexpr (OP == 13) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(x));
    if constexpr (OP == 14) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(x));
    if constexpr (OP == 15) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(xx));
}

template <int OP, int MW = 1>
__global__ void __launch_bounds__(256 + 256 * MW, 1) kinds(uint32_t *out, int iters, int mask)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    uint32_t sink = 0;
    if (wave < 4 * MW) {
        if (mask & 1) {
            i32x4 a = {lane, lane + 1, lane + 2, lane + 3}, b = {lane * 3, lane * 5, lane * 7, lane * 9};
            i32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[u & 3], 0, 0, 0);
            }
            sink = (uint32_t)(c[0][0] ^ c[1][1] ^ c[2][2] ^ c[3][3]);
        }
    } else if (mask & 2) {
        float f[16];
        float2 ff[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { f[k] = (float)(lane + k); ff[k] = make_float2(f[k], 1.0f); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) vop<OP>(f[k], ff[k]);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) sink ^= __float_as_uint(f[k]) ^ __float_as_uint(ff[k].x);
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if (lane == 0) {
        out[(blockIdx.x * 12 + wave) * 2] = (uint32_t)(t1 - t0);
        out[(blockIdx.x * 12 + wave) * 2 + 1] = sink;
    }
}

template <int OP, int MW = 1>
static void run_kinds(const char *name, uint32_t *d_out, int iters)
{
    static uint32_t h[256 * 12 * 2];
    double m_alone = 0, m_both = 0, v_both = 0, v_alone = 0;
    for (int mask = 1; mask <= 3; ++mask) {
        for (int w = 0; w < 2; ++w) kinds<OP, MW><<<256, 256 + 256 * MW>>>(d_out, iters, mask);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
        double cyc[2] = {0, 0};
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < 4 + 4 * MW; ++w) cyc[w < 4 * MW ? 0 : 1] += h[(b * 12 + w) * 2];
        // MW matrix waves per SIMD share the pipe: cycles per MFMA of the SIMD
        if (mask == 1) m_alone = cyc[0] / (1024 * MW) / iters / MW;
        if (mask == 2) v_alone = cyc[1] / 1024 / iters;
        if (mask == 3) { m_both = cyc[0] / (1024 * MW) / iters / MW; v_both = cyc[1] / 1024 / iters; }
    }
    if (MW > 1) { m_alone *= MW; m_both *= MW; }     // back to "time of the matrix phase per 16 MFMAs of one wave"
    // vector work done while the matrix stream ran = 1 - (what was left afterwards) / (all of it alone)
    const double frac = v_both > m_both ? 1.0 - (v_both - m_both) / v_alone : 1.0;
    printf("%-16s alone %5.2f cycles each | MFMA %5.2f cycles alone, %5.2f with these waves | issued per MFMA in its shadow: %.2f%s\n",
           name, v_alone / 128, m_alone / 16, m_both / 16, frac * 128 / 16 * (m_alone / m_both), v_both > m_both ? "" : " (or more: finished first)");
}

template <int PRIO, bool SAME>
static void run(const char *name, int mask, uint32_t *d_out, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) overlap<PRIO, SAME><<<256, 768>>>(d_out, iters, mask);
    CK(hipEventRecord(e0));
    overlap<PRIO, SAME><<<256, 768>>>(d_out, iters, mask);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    static uint32_t h[256 * 12 * 2];
    CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    double cyc[3] = {0, 0, 0};
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < 12; ++w) cyc[w / 4] += h[(b * 12 + w) * 2];
    printf("%-44s %8.3f ms | cycles per iteration: matrix waves %7.1f  vector waves %7.1f  lds waves %7.1f\n", name, ms,
           cyc[0] / 1024 / iters, cyc[1] / 1024 / iters, cyc[2] / 1024 / iters);
}

int main()
{
    uint32_t *d_out;
    CK(hipMalloc(&d_out, 256 * 12 * 2 * 4));
    const int iters = 20000;
    printf("per iteration: 16 MFMA (16 cycles each = 256) | 64 v_fma_f32 (4 cycles each = 256) | 16 ds_read_b128\n");
    run<0, false>("matrix alone", 1, d_out, iters);
    run<0, false>("vector alone", 2, d_out, iters);
    run<0, false>("lds alone", 4, d_out, iters);
    run<0, false>("matrix + vector, equal priority", 3, d_out, iters);
    run<3, false>("matrix + vector, vector at priority 3", 3, d_out, iters);
    run<0, false>("matrix + lds", 5, d_out, iters);
    run<0, false>("vector + lds", 6, d_out, iters);
    run<0, false>("all three, equal priority", 7, d_out, iters);
    run<3, false>("all three, helpers at priority 3", 7, d_out, iters);
    run<0, true>("one wave: 16 MFMA + 48 fma interleaved", 1, d_out, iters);
    run<0, true>("the same + vector waves", 3, d_out, iters);
    run_kinds<0>("v_fma_f32", d_out, iters);
    run_kinds<0, 2>("v_fma_f32 2mw", d_out, iters);
    run_kinds<2, 2>("v_perm_b32 2mw", d_out, iters);
    run_kinds<1>("v_pk_fma_f32", d_out, iters);
    run_kinds<7>("v_pk_add_f32", d_out, iters);
    run_kinds<9>("v_pk_mul_f32", d_out, iters);
    run_kinds<6>("v_add_f32", d_out, iters);
    run_kinds<10>("v_mul_f32", d_out, iters);
    run_kinds<12>("v_max_f32", d_out, iters);
    run_kinds<2>("v_perm_b32", d_out, iters);
    run_kinds<3>("v_lshl_add_u32", d_out, iters);
    run_kinds<11>("v_xor_b32", d_out, iters);
    run_kinds<8>("v_mov_b32", d_out, iters);
    run_kinds<13>("v_cndmask_b32", d_out, iters);
    run_kinds<4>("v_cvt_f32_i32", d_out, iters);
    run_kinds<5>("v_rcp_f32", d_out, iters);
    run_kinds<14>("v_mul_lo_u32", d_out, iters);
    run_kinds<15>("v_fma_f64", d_out, iters);
    run_gaps<0, 0, 0>(d_out, iters);
    run_gaps<1, 0, 0>(d_out, iters);
    run_gaps<2, 0, 0>(d_out, iters);
    run_gaps<3, 0, 0>(d_out, iters);
    run_gaps<4, 0, 0>(d_out, iters);
    run_gaps<0, 4, 0>(d_out, iters);
    run_gaps<0, 8, 0>(d_out, iters);
    run_gaps<0, 12, 0>(d_out, iters);
    run_gaps<0, 16, 0>(d_out, iters);
    run_gaps<0, 8, 3>(d_out, iters);
    run_gaps<0, 12, 3>(d_out, iters);
    return 0;
}
