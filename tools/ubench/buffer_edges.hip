// Micro-check: raw buffer loads of 16 bytes that straddle the ends of the buffer -- is the range
// check per dword (out-of-range dwords read as zero, the others as data), also for offsets that
// wrap below zero?  Decides whether the int8 kernel's staging can drop its address clamps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k(const uint32_t *in, uint4 *out, int n_bytes, int shift)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, n_bytes, 0x00020000);
    auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)threadIdx.x * 4 - shift, 0, 0);
    out[threadIdx.x] = __builtin_bit_cast(uint4, v);
}

int main()
{
    uint32_t h[64], *d_in;
    uint4 *d_out, o[64];
    for (int i = 0; i < 64; ++i) h[i] = 0x1000 + i;
    CK(hipMalloc(&d_in, 256 + 64));
    CK(hipMalloc(&d_out, 64 * 16));
    CK(hipMemcpy(d_in + 1, h, 256, hipMemcpyHostToDevice));      // base 4-byte aligned only
    // buffer = 5 dwords (20 bytes) starting at h[0]; lane L loads dwords L-3 .. L
    k<<<1, 64>>>(d_in + 1, d_out, 20, 12);
    CK(hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int L = 0; L < 12; ++L) {
        printf("lane %2d (dwords %3d..%3d):", L, L - 3, L);
        const uint32_t got[4] = {o[L].x, o[L].y, o[L].z, o[L].w};
        for (int c = 0; c < 4; ++c) {
            const int idx = L - 3 + c;
            const uint32_t want = (idx >= 0 && idx < 5) ? h[idx] : 0;
            printf(" %5x%s", got[c], got[c] == want ? "" : "!");
            bad += got[c] != want;
        }
        printf("\n");
    }
    printf(bad ? "MISMATCH: %d dwords\n" : "per-dword range check holds (%d mismatches)\n", bad);
    return 0;
}
