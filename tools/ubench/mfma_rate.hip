// Micro-benchmark: sustained rate of v_mfma_f32_16x16x32_bf16 and v_mfma_i32_16x16x64_i8 on every
// SIMD of the chip (one wave per SIMD and two), operands in registers.  Prints ns per MFMA per
// SIMD and the clock that would make it 16 cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_bf16(float *out, int iters)
{
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i - 3); }
    f32x4 c[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[u & 3], 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
}

__global__ void __launch_bounds__(256) k_i8(int *out, int iters)
{
    i32x4 a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)threadIdx.x, 8};
    i32x4 c[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[u & 3], 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
}

int main()
{
    float *o; CK(hipMalloc(&o, 4 * 256 * 2048));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep)
    for (int wg : {256, 512}) {
        for (int kind = 0; kind < 2; ++kind) {
            const int iters = 20000;
            for (int w = 0; w < 2; ++w) {
                CK(hipEventRecord(e0));
                if (kind == 0) k_bf16<<<wg, 256>>>(o, iters); else k_i8<<<wg, 256>>>((int *)o, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double per_simd = (double)iters * 16 * (wg / 256);      // MFMAs each SIMD executed
            const double ns = ms * 1e6 / per_simd;
            printf("%s  %d wave(s)/SIMD  %.3f ms  %.2f ns per MFMA per SIMD  -> 16 cycles at %.2f GHz\n",
                   kind ? "i32_16x16x64_i8 " : "f32_16x16x32_bf16", wg / 256, ms, ns, 16.0 / ns);
        }
    }
    return 0;
}
