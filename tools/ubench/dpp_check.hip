// Micro-benchmark + semantics check for DPP-modified VOP2 FMAs on gfx950:
//   v_fmac_f32_dpp acc, x, tap row_shr:k   ==   acc += x[lane-k] * tap  (within rows of 16 lanes)
// Prints (1) the values seen by a few lanes so the shift direction / out-of-row behaviour is on
// record, (2) cycles per wave-instruction next to plain v_fmac_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void sem(float *out)
{
    const int l = threadIdx.x;
    float x = (float)l, one = 1.0f;
    float a = 1000.f, b = 1000.f, c = 1000.f, d = 1000.f;
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a) : "v"(x), "v"(one));
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_shr:3 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(x), "v"(one));
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(c) : "v"(x), "v"(one));
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(d) : "v"(x), "v"(one));
    out[l] = a; out[64 + l] = b; out[128 + l] = c; out[192 + l] = d;
}

#define ITER 2000
template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, float s)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    float x0 = threadIdx.x * 0.5f, x1 = s;
    unsigned w = threadIdx.x * 0x10003u + 77u;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x0), "v"(x1));
            if (KIND == 1) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a[i]) : "v"(x0), "v"(x1));
            if (KIND == 2) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a[i]) : "v"(x0), "v"(x1));
            if (KIND == 3) asm volatile("v_mov_b32_dpp %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(a[i]) : "v"(x0));
            if (KIND == 4) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(w), "s"(0xffffu), "v"(0x4b000000u));
            if (KIND == 5) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(w), "v"(0x4b000000u), "s"(0x03020706u));
            if (KIND == 6) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x1));
            if (KIND == 7) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(w));
            if (KIND == 8) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(w));
            if (KIND == 9) asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a[i]) : "v"(x0), "v"(x1));
            if (KIND == 10) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a[i & 3]) : "v"(x0), "v"(x1));
            if (KIND == 11) asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w), "v"(x1));
            if (KIND == 12) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(w));
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
void run(const char *name, int wps)
{
    const int blocks = 256 * wps;
    float *out;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(out, 2.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(out, 2.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)ITER * 16 * wps;
    printf("%-40s W=%d  %.3f ms  %.3f ns per wave-instr per SIMD\n", name, wps, ms, ms * 1e6 / n);
    hipFree(out);
}

int main()
{
    float *d; hipMalloc(&d, 256 * 4);
    sem<<<1, 64>>>(d);
    std::vector<float> h(256);
    hipMemcpy(h.data(), d, 256 * 4, hipMemcpyDeviceToHost);
    const char *nm[4] = {"row_shr:3 bound_ctrl:0", "row_shr:3 (no bound_ctrl)", "row_shl:1 bound_ctrl:0", "wave_shr:1 bound_ctrl:0"};
    for (int t = 0; t < 4; ++t) {
        printf("%-28s:", nm[t]);
        for (int l : {0, 1, 2, 3, 4, 15, 16, 17, 18, 19, 31, 32, 47, 48, 63}) printf(" l%d=%g", l, h[t * 64 + l] - 1000.f);
        printf("\n");
    }
    for (int w : {1, 4}) {
        run<0>("v_fmac_f32", w);
        run<1>("v_fmac_f32_dpp row_shr:1", w);
        run<2>("v_fmac_f32_dpp row_shr:4", w);
        run<9>("v_fmac_f32_dpp wave_shr:1", w);
        run<10>("v_fmac_f32_dpp row_shr:1, 4 accumulators", w);
        run<3>("v_mov_b32_dpp row_shr:2", w);
        run<4>("v_and_or_b32", w);
        run<5>("v_perm_b32", w);
        run<6>("v_sub_f32", w);
        run<7>("v_cvt_f32_f16", w);
        run<8>("v_cvt_f32_ubyte1", w);
        run<11>("v_mad_i32_i16", w);
        run<12>("v_cvt_f32_i32", w);
        printf("\n");
    }
    return 0;
}
