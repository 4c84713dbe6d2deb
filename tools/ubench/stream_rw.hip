// Micro-benchmark: what the waterfall's byte mix (4 B read per sample, 8 B written: N floats per
// hop of N/2 samples) reaches on this GPU without any FFT work.  Variants: write only / read+write,
// 4-byte and 16-byte accesses per lane, plain and non-temporal stores, grid-stride and persistent
// row-walking (the FFT kernel's pattern: a workgroup writes whole 16 KB rows, 256 B per wave-store).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT>
__device__ __forceinline__ void st4(float4 *p, float4 v)
{
    if constexpr (NT) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
                        __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w); }
    else *p = v;
}

// write only, 16 B per lane, grid-stride
template <bool NT>
__global__ void __launch_bounds__(256) fill16(float4 *out, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride)
        st4<NT>(out + i, make_float4(1.f, 2.f, 3.f, (float)i));
}

// write only, 4 B per lane, contiguous chunk per workgroup
__global__ void __launch_bounds__(256) fill4_rows(float *out, size_t rows, size_t chunk)
{
    size_t r0 = (size_t)blockIdx.x * chunk, r1 = r0 + chunk;
    if (r1 > rows) r1 = rows;
    for (size_t r = r0; r < r1; ++r) {
        float *row = out + r * 4096;
#pragma unroll
        for (int t = 0; t < 16; ++t) row[threadIdx.x + 256 * t] = (float)t;
    }
}

// read 16 B, write 32 B per lane, grid-stride
template <bool NT>
__global__ void __launch_bounds__(256) copy_1to2(const uint4 *in, float4 *out, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const uint4 a = in[i];
        st4<NT>(out + 2 * i, make_float4((float)a.x, (float)a.y, (float)a.z, (float)a.w));
        st4<NT>(out + 2 * i + 1, make_float4((float)(a.x >> 16), (float)(a.y >> 16), (float)(a.z >> 16), (float)(a.w >> 16)));
    }
}

// the FFT kernel's pattern: persistent workgroup walks `chunk` rows; per row reads 8 x 4 B per lane
// (2048 new samples) with the next row requested first, and writes 16 x 4 B per lane
template <int W>   // W = floats per lane per store (1 or 4)
__global__ void __launch_bounds__(256) rows_rw(const uint32_t *in, float *out, size_t rows, size_t chunk)
{
    size_t r0 = (size_t)blockIdx.x * chunk, r1 = r0 + chunk;
    if (r1 > rows) r1 = rows;
    if (r0 >= r1) return;
    uint32_t nxt[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) nxt[t] = in[r0 * 2048 + threadIdx.x + 256 * t];
    for (size_t r = r0; r < r1; ++r) {
        uint32_t cur[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) cur[t] = nxt[t];
        const size_t rn = r + 1 < r1 ? r + 1 : r;
#pragma unroll
        for (int t = 0; t < 8; ++t) nxt[t] = in[rn * 2048 + threadIdx.x + 256 * t];
        float *row = out + r * 4096;
        if constexpr (W == 1) {
#pragma unroll
            for (int t = 0; t < 16; ++t) row[threadIdx.x + 256 * t] = (float)(cur[t & 7] >> (t & 8));
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                ((float4 *)row)[threadIdx.x + 256 * t] = make_float4((float)cur[2 * t], (float)cur[2 * t + 1],
                                                                     (float)(cur[2 * t] >> 8), (float)(cur[2 * t + 1] >> 8));
        }
    }
}

// persistent, interleaved: run k of workgroup b is rows [(k*G + m(b))*L, +L); m = b or the XCD-contiguous
// order; FULL: every row reads all 16 x 4 B per lane (no register reuse between consecutive rows)
template <bool FULL, bool XCD>
__global__ void __launch_bounds__(256) rows_il(const uint32_t *in, float *out, size_t rows, int L)
{
    const size_t G = gridDim.x;
    const size_t m = XCD ? (size_t)(blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    for (size_t run = m * L; run < rows; run += G * L) {
        size_t r1 = run + L; if (r1 > rows) r1 = rows;
        uint32_t keep[8];
        if (!FULL) {
#pragma unroll
            for (int t = 0; t < 8; ++t) keep[t] = in[run * 2048 + threadIdx.x + 256 * t];
        }
        for (size_t r = run; r < r1; ++r) {
            uint32_t cur[16];
            if (FULL) {
#pragma unroll
                for (int t = 0; t < 16; ++t) cur[t] = in[r * 2048 + threadIdx.x + 256 * t];
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) { cur[t] = keep[t]; cur[t + 8] = in[(r + 1) * 2048 + threadIdx.x + 256 * t]; keep[t] = cur[t + 8]; }
            }
            float *row = out + r * 4096;
#pragma unroll
            for (int t = 0; t < 16; ++t) row[threadIdx.x + 256 * t] = (float)(cur[t] >> 3);
        }
    }
}

template <class F>
static float timeit(F f, int iters = 20)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main()
{
    const size_t n = (size_t)1 << 28;                // samples
    const size_t rows = n / 2048;                    // output rows of 4096 floats
    const size_t bin = n * 4, bout = rows * 4096 * 4;
    uint32_t *in; float *out;
    CK(hipMalloc(&in, bin)); CK(hipMalloc(&out, bout));
    CK(hipMemset(in, 1, bin)); CK(hipMemset(out, 0, bout));
    const double gb_w = bout / 1e9, gb_rw = (bin + bout) / 1e9;
    float ms;
    for (int wg : {256 * 4, 256 * 8, 256 * 16}) {
        ms = timeit([&] { fill16<false><<<wg, 256>>>((float4 *)out, bout / 16); });
        printf("fill16      grid %5d   %.3f ms  %.0f GB/s (write only)\n", wg, ms, gb_w / ms * 1e3);
        ms = timeit([&] { fill16<true><<<wg, 256>>>((float4 *)out, bout / 16); });
        printf("fill16 nt   grid %5d   %.3f ms  %.0f GB/s (write only)\n", wg, ms, gb_w / ms * 1e3);
    }
    ms = timeit([&] { CK(hipMemsetAsync(out, 0, bout)); });
    printf("hipMemset                 %.3f ms  %.0f GB/s (write only)\n", ms, gb_w / ms * 1e3);
    for (int per : {3, 4, 8}) {
        const size_t groups = 256 * per, chunk = (rows + groups - 1) / groups;
        ms = timeit([&] { fill4_rows<<<groups, 256>>>(out, rows, chunk); });
        printf("fill4 rows  %d wg/CU       %.3f ms  %.0f GB/s (write only)\n", per, ms, gb_w / ms * 1e3);
    }
    for (int wg : {256 * 4, 256 * 8, 256 * 16}) {
        ms = timeit([&] { copy_1to2<false><<<wg, 256>>>((const uint4 *)in, (float4 *)out, bin / 16); });
        printf("copy 1:2    grid %5d   %.3f ms  %.0f GB/s\n", wg, ms, gb_rw / ms * 1e3);
        ms = timeit([&] { copy_1to2<true><<<wg, 256>>>((const uint4 *)in, (float4 *)out, bin / 16); });
        printf("copy 1:2 nt grid %5d   %.3f ms  %.0f GB/s\n", wg, ms, gb_rw / ms * 1e3);
    }
    for (int per : {3, 4, 8}) {
        const size_t groups = 256 * per, chunk = (rows + groups - 1) / groups;
        ms = timeit([&] { rows_rw<1><<<groups, 256>>>(in, out, rows, chunk); });
        printf("rows rw 4B  %d wg/CU       %.3f ms  %.0f GB/s\n", per, ms, gb_rw / ms * 1e3);
        ms = timeit([&] { rows_rw<4><<<groups, 256>>>(in, out, rows, chunk); });
        printf("rows rw 16B %d wg/CU       %.3f ms  %.0f GB/s\n", per, ms, gb_rw / ms * 1e3);
    }
    for (int per : {3, 4}) for (int L : {1, 2, 4, 8, 32}) {
        const int G = 256 * per;
        ms = timeit([&] { rows_il<false, false><<<G, 256>>>(in, out, rows - 1, L); });
        printf("rows il reuse     %d wg/CU L=%2d  %.3f ms  %.0f GB/s\n", per, L, ms, gb_rw / ms * 1e3);
        ms = timeit([&] { rows_il<false, true><<<G, 256>>>(in, out, rows - 1, L); });
        printf("rows il reuse xcd %d wg/CU L=%2d  %.3f ms  %.0f GB/s\n", per, L, ms, gb_rw / ms * 1e3);
        if (L > 1) continue;
        ms = timeit([&] { rows_il<true, false><<<G, 256>>>(in, out, rows - 1, L); });
        printf("rows il full      %d wg/CU L=%2d  %.3f ms  %.0f GB/s\n", per, L, ms, gb_rw / ms * 1e3);
        ms = timeit([&] { rows_il<true, true><<<G, 256>>>(in, out, rows - 1, L); });
        printf("rows il full xcd  %d wg/CU L=%2d  %.3f ms  %.0f GB/s\n", per, L, ms, gb_rw / ms * 1e3);
    }
    // one row per workgroup, rows handed out in launch order (non-persistent)
    ms = timeit([&] { rows_rw<1><<<rows, 256>>>(in, out, rows, 1); });
    printf("rows rw 4B  1 row per wg     %.3f ms  %.0f GB/s\n", ms, gb_rw / ms * 1e3);
    ms = timeit([&] { rows_rw<4><<<rows, 256>>>(in, out, rows, 1); });
    printf("rows rw 16B 1 row per wg     %.3f ms  %.0f GB/s\n", ms, gb_rw / ms * 1e3);
    return 0;
}
