// Micro-test: are packed-float32 results of one wave correct while ANOTHER wave of the same SIMD streams MFMAs?
// Found through fir_mfma_i8_kernel (round 3): with the SLP vectoriser on, its helper waves' discriminator used
// v_pk_mul_f32 / v_pk_fma_f32, and the LOW half of those results came out wrong in lanes 48-63 -- only while the
// matrix wave of the same SIMD was inside its k loop.  This test separates the hardware from the kernel: waves
// 0-3 run v_mfma_i32_16x16x64_i8 back to back (or idle), waves 4-7 evaluate a packed product chain whose inputs
// are written by plain v_mov just before (the sequence of the kernel) and compare every result with the same
// arithmetic done by unpacked instructions.  Prints mismatches per lane quarter and half.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int VARIANT>
__global__ void __launch_bounds__(512, 1) test(uint32_t *out, int iters, int matrix_on)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        uint32_t sink = 0;
        if (matrix_on) {
            i32x4 a = {lane, lane + 1, lane + 2, lane + 3}, b = {lane * 3, lane * 5, lane * 7, lane * 9};
            i32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            for (int it = 0; it < iters * 8; ++it) {
                c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
            }
            sink = (uint32_t)(c0[0] ^ c1[1] ^ c2[2] ^ c3[3]);
        }
        if (sink == 0x12345678u) out[0] = sink;
        return;
    }
    // vector waves: bad[half] counts mismatches of the low / high half of packed results
    uint32_t bad_lo = 0, bad_hi = 0;
    float s = 0.37f + 0.01f * lane;
    for (int it = 0; it < iters; ++it) {
        float a = s, b = s * 1.7f - 0.3f, c = 0.9f - s, d = s * 0.5f + 0.11f, e = 1.3f - s * 0.2f, f = s + 0.77f;
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
        // reference with unpacked instructions
        float r0, r1;
        asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %5\n\t"
                     "v_fma_f32 %0, %6, %4, %0\n\tv_fma_f32 %1, %7, %5, %1"
                     : "=&v"(r0), "=&v"(r1) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f));
        v2f p;
        if (VARIANT == 1) {
            v2f x = {a, b}, y = {c, d}, z = {e, f};
            asm volatile("v_pk_mul_f32 %0, %1, %2\n\tv_pk_fma_f32 %0, %3, %2, %0" : "=&v"(p) : "v"(x), "v"(y), "v"(z));
        } else {
            // the kernel's sequence around the reciprocal: two v_rcp_f32 (transcendental unit, quarter rate) into a
            // register pair, one unrelated instruction, then a packed product that reads the pair
            float q0, q1;
            asm volatile("v_rcp_f32 %0, %2\n\tv_rcp_f32 %1, %3\n\ts_nop 7\n\ts_nop 7\n\tv_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %5"
                         : "=&v"(q0), "=&v"(q1) : "v"(a), "v"(f), "v"(c), "v"(d));
            r0 = q0; r1 = q1;
            float o0, o1;
            asm volatile("v_mov_b32 v22, %4\n\tv_mov_b32 v23, %5\n\t"
                         "v_rcp_f32 v20, %2\n\tv_rcp_f32 v21, %3\n\t"
                         "v_cmp_neq_f32 vcc, 0, %3\n\t"
                         "v_pk_mul_f32 v[24:25], v[22:23], v[20:21]\n\t"
                         "v_mov_b32 %0, v24\n\tv_mov_b32 %1, v25"
                         : "=&v"(o0), "=&v"(o1) : "v"(a), "v"(f), "v"(c), "v"(d) : "v20", "v21", "v22", "v23", "v24", "v25", "vcc");
            p.x = o0; p.y = o1;
        }
        bad_lo += __float_as_uint(p.x) != __float_as_uint(r0);
        bad_hi += __float_as_uint(p.y) != __float_as_uint(r1);
        s = s * 1.0001f + 1e-4f;
        if (s > 3.0f) s -= 2.5f;
    }
    uint32_t *o = out + 16 + ((blockIdx.x * 4 + (wave - 4)) * 64 + lane) * 2;
    o[0] = bad_lo; o[1] = bad_hi;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, blocks = 256;
    uint32_t *d;
    const size_t words = 16 + (size_t)blocks * 4 * 64 * 2;
    CK(hipMalloc(&d, words * 4));
    uint32_t *h = (uint32_t *)malloc(words * 4);
    for (int var = 1; var <= 2; ++var)
        for (int mat = 0; mat <= 1; ++mat) {
            CK(hipMemset(d, 0, words * 4));
            if (var == 1) test<1><<<blocks, 512>>>(d, iters, mat); else test<2><<<blocks, 512>>>(d, iters, mat);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, d, words * 4, hipMemcpyDeviceToHost));
            unsigned long long q[4][2] = {{0}};
            for (int i = 0; i < blocks * 4 * 64; ++i) { q[(i & 63) >> 4][0] += h[16 + 2 * i]; q[(i & 63) >> 4][1] += h[16 + 2 * i + 1]; }
            printf("variant %d (1 = packed mul + fma, 2 = v_rcp_f32 x2 -> one instruction -> v_pk_mul_f32), matrix waves %s: mismatches of packed results (low half | high half) by lane quarter: ", var, mat ? "streaming MFMAs" : "idle");
            for (int k = 0; k < 4; ++k) printf(" lanes %2d-%2d: %llu | %llu ", 16 * k, 16 * k + 15, q[k][0], q[k][1]);
            printf(" of %llu each\n", (unsigned long long)blocks * 4 * 16 * iters);
        }
    return 0;
}
