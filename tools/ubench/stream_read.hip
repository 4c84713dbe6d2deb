// Micro-benchmark: what a read-dominated stream with the FIR chain's byte ratio (4 B in per
// sample, 0.4 B out) reaches on this GPU, without any of the FIR's work: every lane reads 16-byte
// pieces, xors them together and one lane in ten stores a float.  Variants: plain vector loads
// (grid-stride and one-tile-per-workgroup) and LDS-DMA tiles as the FIR kernel stages them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(256) read_gridstride(const uint4 *in, float *out, size_t n16)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
        if ((threadIdx.x % 10) < 4) out[(i * 4) / 10] = __uint_as_float(acc);
    }
}

// one 20 KB tile per workgroup (5 x 16 B per lane), like the FIR kernel's tiles
__global__ void __launch_bounds__(256) read_tiles(const uint4 *in, float *out, size_t n16)
{
    const size_t base = (size_t)blockIdx.x * 1280;
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const uint4 a = in[base + k * 256 + threadIdx.x];
        acc ^= a.x ^ a.y ^ a.z ^ a.w;
    }
    // 512 floats out per tile
    ((float2 *)out)[(size_t)blockIdx.x * 256 + threadIdx.x] = make_float2(__uint_as_float(acc), 1.f);
}

__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_byte)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}

__global__ void __launch_bounds__(256) read_tiles_dma(const uint4 *in, float *out, size_t n16)
{
    __shared__ uint4 tile[1280 + 256];
    const size_t base = (size_t)blockIdx.x * 1280;
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)(tile + k * 256 + (tid & ~63));
        dma16(in + base + k * 256 + tid, __builtin_amdgcn_readfirstlane(lds));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint4 a = tile[(tid * 5) % 1280];
    ((float2 *)out)[(size_t)blockIdx.x * 256 + tid] = make_float2(__uint_as_float(a.x ^ a.y ^ a.z ^ a.w), 1.f);
}

// variants of the tile kernel: K pieces per lane, optional non-temporal loads, optional XCD-contiguous
// tile order (workgroup b runs on XCD b % 8; give every XCD one contiguous eighth of the stream)
template <int K, bool NT, bool XCD>
__global__ void __launch_bounds__(256) read_tiles_v(const uint4 *in, float *out, size_t n16)
{
    size_t tile = blockIdx.x;
    if (XCD) tile = (size_t)(blockIdx.x & 7) * (gridDim.x / 8) + (blockIdx.x >> 3);
    const size_t base = tile * (256 * K);
    unsigned acc = 0;
    uint4 a[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint4 *p = in + base + k * 256 + threadIdx.x;
        if (NT) {
            a[k].x = __builtin_nontemporal_load(&p->x); a[k].y = __builtin_nontemporal_load(&p->y);
            a[k].z = __builtin_nontemporal_load(&p->z); a[k].w = __builtin_nontemporal_load(&p->w);
        } else a[k] = *p;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) acc ^= a[k].x ^ a[k].y ^ a[k].z ^ a[k].w;
    if (K >= 5) {
#pragma unroll
        for (int j = 0; j < K / 5; ++j)
            ((float2 *)out)[(tile * (K / 5) + j) * 256 + threadIdx.x] = make_float2(__uint_as_float(acc), 1.f);
    }
}

// store policy variants of the 20 KB tile kernel: 0 = no stores at all, 1 = non-temporal stores,
// 2 = one wave of the workgroup stores the whole 2 KB row (dwordx4 per lane)
template <int POLICY>
__global__ void __launch_bounds__(256) read_tiles_store(const uint4 *in, float *out, size_t n16)
{
    __shared__ float row[512];
    const size_t tile = (size_t)(blockIdx.x & 7) * (gridDim.x / 8) + (blockIdx.x >> 3);
    const size_t base = tile * 1280;
    unsigned acc = 0;
    uint4 a[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) a[k] = in[base + k * 256 + threadIdx.x];
#pragma unroll
    for (int k = 0; k < 5; ++k) acc ^= a[k].x ^ a[k].y ^ a[k].z ^ a[k].w;
    const float v = __uint_as_float(acc);
    if (POLICY == 0) {
        if (acc == 0x12345678u) out[tile] = v;
    } else if (POLICY == 5) {          // same bytes, 8x larger bursts: every 8th tile writes 16 KB
        if ((tile & 7) == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ((float2 *)out)[(tile + j) * 256 + threadIdx.x] = make_float2(v, 1.f);
        } else if (acc == 0x12345678u) out[tile] = v;
    } else if (POLICY == 6 || POLICY == 7 || POLICY == 8) {      // cache-policy bits on the store
        float2 *p = (float2 *)out + tile * 256 + threadIdx.x;
        const float2 val = make_float2(v, 1.f);
        if (POLICY == 6) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(val) : "memory");
        if (POLICY == 7) asm volatile("global_store_dwordx2 %0, %1, off nt sc0 sc1" ::"v"(p), "v"(val) : "memory");
        if (POLICY == 8) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(val) : "memory");
    } else if (POLICY == 3) {          // stores that stay in L2: every workgroup of an XCD rewrites the same 2 KB
        ((float2 *)out)[(size_t)(blockIdx.x & 7) * 4096 + threadIdx.x] = make_float2(v, 1.f);
    } else if (POLICY == 4) {          // same bytes to HBM, but a 64 KB-strided (scattered) destination
        ((float2 *)out)[((tile * 37) % (gridDim.x)) * 256 + threadIdx.x] = make_float2(v, 1.f);
    } else if (POLICY == 1) {
        float *p = out + (tile * 256 + threadIdx.x) * 2;
        __builtin_nontemporal_store(v, p);
        __builtin_nontemporal_store(1.f, p + 1);
    } else {
        row[2 * threadIdx.x] = v; row[2 * threadIdx.x + 1] = 1.f;
        __syncthreads();
        if (threadIdx.x < 128) ((float4 *)out)[tile * 128 + threadIdx.x] = ((float4 *)row)[threadIdx.x];
    }
}

// K consecutive 20 KB tiles per workgroup, one after the other: the store of tile k is in flight
// while tile k+1 is fetched, so only the last store's acknowledge holds the workgroup's slot
template <int K, bool NT>
__global__ void __launch_bounds__(256) read_tiles_seq(const uint4 *in, float *out, size_t n16)
{
    const size_t grp = (size_t)(blockIdx.x & 7) * (gridDim.x / 8) + (blockIdx.x >> 3);
    for (int t = 0; t < K; ++t) {
        const size_t tile = grp * K + t;
        const size_t base = tile * 1280;
        unsigned acc = 0;
        uint4 a[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) a[k] = in[base + k * 256 + threadIdx.x];
#pragma unroll
        for (int k = 0; k < 5; ++k) acc ^= a[k].x ^ a[k].y ^ a[k].z ^ a[k].w;
        float *p = out + (tile * 256 + threadIdx.x) * 2;
        if (NT) { __builtin_nontemporal_store(__uint_as_float(acc), p); __builtin_nontemporal_store(1.f, p + 1); }
        else *(float2 *)p = make_float2(__uint_as_float(acc), 1.f);
    }
}

// 40 KB tile per workgroup, XCD-contiguous, non-temporal stores.  PAIR: a lane reads the adjacent pieces
// 2*i and 2*i+1 with two instructions (each instruction then covers a 2 KB span at a 32-byte lane stride,
// as the int8 FIR kernel's helper lanes do); otherwise every instruction reads 1 KB of consecutive bytes.
template <bool PAIR>
__global__ void __launch_bounds__(256) read_tiles_pair(const uint4 *in, float *out, size_t n16)
{
    const size_t tile = (size_t)(blockIdx.x & 7) * (gridDim.x / 8) + (blockIdx.x >> 3);
    const size_t base = tile * 2560;
    unsigned acc = 0;
    uint4 a[10];
#pragma unroll
    for (int k = 0; k < 10; ++k)
        a[k] = PAIR ? in[base + 2 * ((k >> 1) * 256 + threadIdx.x) + (k & 1)] : in[base + k * 256 + threadIdx.x];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc ^= a[k].x ^ a[k].y ^ a[k].z ^ a[k].w;
    float *p = out + (tile * 512 + threadIdx.x * 2) * 2;
    __builtin_nontemporal_store(__uint_as_float(acc), p);
    __builtin_nontemporal_store(1.f, p + 1);
    __builtin_nontemporal_store(2.f, p + 2);
    __builtin_nontemporal_store(3.f, p + 3);
}

// in-flight depth: persistent workgroups (one per CU, 256 loader lanes, as the int8 FIR kernel's helper
// waves) that keep DEPTH 40 KB tiles in flight: tile t+DEPTH is requested when tile t has been consumed
template <int DEPTH>
__global__ void __launch_bounds__(256, 1) read_persistent(const uint4 *in, float *out, size_t n16)
{
    const int G = gridDim.x, b = blockIdx.x;
    const size_t slot = (size_t)(b & 7) * (G >> 3) + (b >> 3);
    const size_t n_tiles = n16 / 2560;
    const int n_t = (int)((n_tiles - slot + G - 1) / G);
    uint4 a[DEPTH][10];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const size_t tile = (size_t)(d < n_t ? d : n_t - 1) * G + slot;
#pragma unroll
        for (int k = 0; k < 10; ++k) a[d][k] = in[tile * 2560 + k * 256 + threadIdx.x];
    }
    for (int t = 0; t < n_t; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const size_t tile = (size_t)(t + d) * G + slot;
            unsigned acc = 0;
#pragma unroll
            for (int k = 0; k < 10; ++k) acc ^= a[d][k].x ^ a[d][k].y ^ a[d][k].z ^ a[d][k].w;
            const int nx = t + d + DEPTH < n_t ? t + d + DEPTH : n_t - 1;
            const size_t tn = (size_t)nx * G + slot;
#pragma unroll
            for (int k = 0; k < 10; ++k) a[d][k] = in[tn * 2560 + k * 256 + threadIdx.x];
            if (t + d < n_t) {
                float *p = out + (tile * 512 + threadIdx.x * 2) * 2;
                __builtin_nontemporal_store(__uint_as_float(acc), p);
                __builtin_nontemporal_store(1.f, p + 1);
                __builtin_nontemporal_store(2.f, p + 2);
                __builtin_nontemporal_store(3.f, p + 3);
            }
        }
    }
}

// the same persistent reader, two tiles in flight, but tiles handed out by an atomic counter instead of
// a static round-robin: MODE 0 = one global counter, MODE 1 = one counter per XCD over its contiguous eighth
template <int MODE>
__global__ void __launch_bounds__(256, 1) read_persistent_dyn(const uint4 *in, float *out, size_t n16, unsigned *ctr)
{
    __shared__ unsigned nxt[4];
    const int G = gridDim.x, b = blockIdx.x, x = b & 7;
    const unsigned n_tiles = (unsigned)(n16 / 2560) & ~7u;
    const unsigned per_x = n_tiles / 8;
    auto grab = [&]() -> unsigned {            // all lanes return the same tile (or >= n_tiles)
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned v = atomicAdd(&ctr[MODE ? 32 * x : 0], 1u);
            nxt[0] = MODE ? (v < per_x ? x * per_x + v : n_tiles) : v;
        }
        __syncthreads();
        return nxt[0];
    };
    uint4 a[2][10];
    unsigned tl[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        tl[d] = grab();
        const size_t tile = tl[d] < n_tiles ? tl[d] : 0;
#pragma unroll
        for (int k = 0; k < 10; ++k) a[d][k] = in[tile * 2560 + k * 256 + threadIdx.x];
    }
    while (tl[0] < n_tiles || tl[1] < n_tiles) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const unsigned tile = tl[d];
            unsigned acc = 0;
#pragma unroll
            for (int k = 0; k < 10; ++k) acc ^= a[d][k].x ^ a[d][k].y ^ a[d][k].z ^ a[d][k].w;
            tl[d] = grab();
            const size_t tn = tl[d] < n_tiles ? tl[d] : 0;
#pragma unroll
            for (int k = 0; k < 10; ++k) a[d][k] = in[tn * 2560 + k * 256 + threadIdx.x];
            if (tile < n_tiles) {
                float *p = out + ((size_t)tile * 512 + threadIdx.x * 2) * 2;
                __builtin_nontemporal_store(__uint_as_float(acc), p);
                __builtin_nontemporal_store(1.f, p + 1);
                __builtin_nontemporal_store(2.f, p + 2);
                __builtin_nontemporal_store(3.f, p + 3);
            }
        }
    }
}

int main()
{
    const size_t n = (size_t)1 << 30;          // samples, 4 B each
    const size_t n16 = n / 4;
    uint4 *in; float *out;
    hipMalloc(&in, n * 4 + 65536);
    hipMalloc(&out, n / 10 * 4 + (1 << 22));
    hipMemset(in, 1, n * 4);
    unsigned *ctr; hipMalloc(&ctr, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = 4.4 * (double)n;
    for (int variant = 9; variant < 30; ++variant) {
        if (variant >= 10 && variant <= 21) continue;
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 20; ++it) {
                if (variant == 0) read_gridstride<<<256 * 8, 256>>>(in, out, n16);
                if (variant == 1) read_gridstride<<<256 * 32, 256>>>(in, out, n16);
                if (variant == 2) read_tiles<<<(unsigned)(n16 / 1280), 256>>>(in, out, n16);
                if (variant == 3) read_tiles_dma<<<(unsigned)(n16 / 1280), 256>>>(in, out, n16);
                if (variant == 4) read_tiles_v<10, false, false><<<(unsigned)(n16 / 2560), 256>>>(in, out, n16);
                if (variant == 5) read_tiles_v<5, true, false><<<(unsigned)(n16 / 1280), 256>>>(in, out, n16);
                if (variant == 6) read_tiles_v<5, false, true><<<(unsigned)(n16 / 1280) & ~7u, 256>>>(in, out, n16);
                if (variant == 7) read_tiles_v<20, false, false><<<(unsigned)(n16 / 5120), 256>>>(in, out, n16);
                if (variant == 9) read_tiles_store<0><<<(unsigned)(n16 / 1280) & ~7u, 256>>>(in, out, n16);
                if (variant == 10) read_tiles_store<1><<<(unsigned)(n16 / 1280) & ~7u, 256>>>(in, out, n16);
                if (variant == 11) read_tiles_store<2><<<(unsigned)(n16 / 1280) & ~7u, 256>>>(in, out, n16);
                if (variant == 12) read_tiles_seq<2, false><<<(unsigned)(n16 / 1280 / 2) & ~7u, 256>>>(in, out, n16);
                if (variant == 13) read_tiles_seq<4, false><<<(unsigned)(n16 / 1280 / 4) & ~7u, 256>>>(in, out, n16);
                if (variant == 14) read_tiles_seq<8, false><<<(unsigned)(n16 / 1280 / 8) & ~7u, 256>>>(in, out, n16);
                if (variant == 15) read_tiles_seq<4, true><<<(unsigned)(n16 / 1280 / 4) & ~7u, 256>>>(in, out, n16);
                if (variant == 16) read_tiles_store<3><<<(unsigned)(n16 / 1280) & ~7u, 256>>>(in, out, n16);
                if (variant == 17) read_tiles_store<4><<<(unsigned)(n16 / 1280) & ~7u, 256>>>(in, out, n16);
                if (variant == 18) read_tiles_store<5><<<(unsigned)(n16 / 1280) & ~7u, 256>>>(in, out, n16);
                if (variant == 19) read_tiles_store<6><<<(unsigned)(n16 / 1280) & ~7u, 256>>>(in, out, n16);
                if (variant == 20) read_tiles_store<7><<<(unsigned)(n16 / 1280) & ~7u, 256>>>(in, out, n16);
                if (variant == 21) read_tiles_store<8><<<(unsigned)(n16 / 1280) & ~7u, 256>>>(in, out, n16);
                if (variant == 22) read_tiles_pair<false><<<(unsigned)(n16 / 2560) & ~7u, 256>>>(in, out, n16);
                if (variant == 23) read_tiles_pair<true><<<(unsigned)(n16 / 2560) & ~7u, 256>>>(in, out, n16);
                if (variant == 24) read_persistent<1><<<256, 256>>>(in, out, n16);
                if (variant == 25) read_persistent<2><<<256, 256>>>(in, out, n16);
                if (variant == 26) read_persistent<3><<<256, 256>>>(in, out, n16);
                if (variant == 27) read_persistent<4><<<256, 256>>>(in, out, n16);
                if (variant == 28) { hipMemsetAsync(ctr, 0, 4096, 0); read_persistent_dyn<0><<<256, 256>>>(in, out, n16, ctr); }
                if (variant == 29) { hipMemsetAsync(ctr, 0, 4096, 0); read_persistent_dyn<1><<<256, 256>>>(in, out, n16, ctr); }
                if (variant == 8) read_tiles_v<10, true, true><<<(unsigned)(n16 / 2560) & ~7u, 256>>>(in, out, n16);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
            if (rep && ms < best) best = ms;
        }
        const char *nm[30] = {"grid-stride 8 wg/CU", "grid-stride 32 wg/CU", "20 KB tile per wg, vector loads", "20 KB tile per wg, LDS-DMA", "40 KB tile", "20 KB tile, nontemporal", "20 KB tile, XCD-contiguous", "80 KB tile", "40 KB nontemporal XCD-contiguous", "20 KB XCD-contiguous, NO stores", "20 KB XCD-contiguous, nontemporal stores", "20 KB XCD-contiguous, dwordx4 stores from 2 waves", "2 tiles in sequence per wg", "4 tiles in sequence per wg", "8 tiles in sequence per wg", "4 tiles in sequence, nontemporal stores", "stores that stay in L2 (same 2 KB per XCD)", "stores to a permuted tile order", "every 8th tile stores 16 KB", "stores sc0 sc1", "stores nt sc0 sc1", "stores sc1", "40 KB tile, 1 KB per load instruction", "40 KB tile, adjacent pieces per lane (32 B lane stride)", "persistent, 1 x 40 KB in flight per CU", "persistent, 2 x 40 KB in flight per CU", "persistent, 3 x 40 KB in flight per CU", "persistent, 4 x 40 KB in flight per CU", "persistent, tiles from one atomic counter", "persistent, tiles from a counter per XCD"};
        printf("%-34s %.4f ms  %.0f GB/s (4.4 B/sample)\n", nm[variant], best, bytes / best / 1e6);
    }
    return 0;
}
