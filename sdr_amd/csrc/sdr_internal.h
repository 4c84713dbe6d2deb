// Internal helpers shared by the libsdrhip translation units (not part of the ABI).
#ifndef SDR_INTERNAL_H_
#define SDR_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "sdr_hip.h"

namespace sdr {

void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);
bool device_ready();          // true once the current device has been verified to be gfx950
int ensure_device();          // SDR_OK or SDR_ENODEV / SDR_EHIP
int same_device(int plan_device);   // SDR_OK when the calling thread's current device is the plan's, else SDR_EINVAL

// Every device / pinned allocation of the library goes through these two, which count: tests hold the
// steady-state paths (a second shim call with the same filter, a block through the hand-off, the
// allocation-free FFT entry point) to "no allocation" by the counter (sdr_alloc_count()).
void note_alloc();
inline hipError_t dev_malloc(void **p, size_t bytes) { note_alloc(); return hipMalloc(p, bytes); }
inline hipError_t host_malloc(void **p, size_t bytes, unsigned flags) { note_alloc(); return hipHostMalloc(p, bytes, flags); }

#define SDR_HIP(call)                                         \
    do {                                                      \
        hipError_t e_ = (call);                               \
        if (e_ != hipSuccess) return ::sdr::hip_fail(e_, #call); \
    } while (0)

// Measurement switches (SDR_I8_ABLATE in fir_mfma_i8.hip; SDR_FFT_ABLATE, SDR_FFT_PADLDS, SDR_FFT_MINB in
// fft.hip) compile parts of a kernel out to time the rest: their results are wrong by construction.
// They are honoured only under -DSDR_MEASUREMENT_BUILD, which capi.build() never sets, and a
// library that contains such a translation unit says so: sdr_build_info() names it and sdr_init()
// prints a warning.
#ifndef SDR_MEASUREMENT_BUILD
#if defined(SDR_I8_ABLATE) || defined(SDR_FFT_ABLATE) || defined(SDR_FFT_PADLDS) || defined(SDR_FFT_MINB) || defined(SDR_FFT_NT_MIN)
#error "kernel ablation switches need -DSDR_MEASUREMENT_BUILD: such a library computes wrong results on purpose"
#endif
#endif
int fft_static_scratch_init(void **ptr_out);   // fft.hip: address of the no-alloc scratch on the current device
int measurement_flags_fir();  // 0 in the product; the ablation mask of fir_mfma_i8.hip in a measurement build (-1: none set)
int measurement_flags_fft();  // the same for fft.hip
int measurement_flags_fft_wave();  // and for fft_wave.hip

// 2^-11: the Q11 scale of BLADERF_FORMAT_SC16_Q11 samples (bladerf_sdr.h:274)
constexpr float kQ11 = 1.0f / 2048.0f;
// 2^-7: full scale of the HackRF's int8 samples
constexpr float kQ7 = 1.0f / 128.0f;

// bytes of one input sample of kind `input` (sdr_input)
inline size_t in_elem_bytes(int input)
{
    switch (input) {
    case SDR_IN_CF32: return 8;
    case SDR_IN_I8_IQ: return 2;
    default: return 4;
    }
}

}  // namespace sdr

#endif
