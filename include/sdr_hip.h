/*
 * sdr_hip.h -- C ABI of libsdrhip: the MI355X (gfx950) receive-side DSP hot path of
 * davemers0160/SDR (int16 IQ -> FIR + decimate -> FM discriminator | envelope | IQ, NCO mix,
 * windowed FFT -> dB).
 *
 * The reference has no FFI for this path: its boundary is a set of header-only C++ free
 * functions over std::vector.  Each entry point below names the reference
 * interface it replaces (file:line relative to the reference root).  include/sdr_hip_shims.hpp
 * re-creates those C++ signatures on top of this ABI; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain pointers and sizes only; every function returns an int status (SDR_OK == 0) except
 *     the *_strerror / *_count helpers.  Where the reference throws std::invalid_argument
 *     (bladerf_apt_demod.cpp:254-259) the ABI returns SDR_EINVAL.
 *   - IQ input is interleaved little-endian int16 (I then Q, 4 B per sample, Q11), i.e. the
 *     memory of std::vector<std::complex<int16_t>> / std::vector<int16_t> IQIQ...
 *     (bladerf_sdr.h:273,281; bladerf_main_v1.cpp:128,175).  Complex float is interleaved
 *     (re, im) float32 == std::complex<float>.
 *   - *_host entry points take host pointers (they stage through device memory owned by the
 *     plan); *_dev entry points take device pointers and a hipStream_t passed as void*
 *     (NULL = the default stream) and only enqueue work.
 *   - there is NO CPU fallback: without a gfx950 device sdr_init() and every plan call fail
 *     with SDR_ENODEV.
 *   - threads and devices: a plan lives on the device that was current when it was created and
 *     every execute call checks that the calling thread's current device is that one
 *     (SDR_EINVAL otherwise).  Plans are immutable after creation except for the staging /
 *     scratch buffers of the *_host entry points and of FFT frames longer than 4096 points:
 *     those calls are one at a time per plan; *_dev calls of one plan may be enqueued on several
 *     streams (long launches of the int8 matrix kernel take one block of tile counters each from
 *     a ring of 64 per plan; a block goes to a launch on another stream only after its previous
 *     launch has reported completion, so launches that may overlap never share one, whatever their
 *     stream handles -- hipStreamPerThread from several threads included; with no block free, and
 *     under HIP stream capture, a launch runs the static tile order: same results, a few per cent
 *     slower).  The last-error
 *     text is per thread.
 */
#ifndef SDR_HIP_H_
#define SDR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum sdr_status {
    SDR_OK = 0,
    SDR_EINVAL = 1,     /* bad argument (the reference's std::invalid_argument) */
    SDR_ENOMEM = 2,
    SDR_EHIP = 3,       /* a HIP runtime call failed; see sdr_last_error() */
    SDR_EALIGN = 4,     /* pointer not aligned to its sample type */
    SDR_ECAPACITY = 5,  /* output buffer too small */
    SDR_ENODEV = 6,     /* no gfx950 device */
    SDR_ERANGE = 7      /* requested samples are outside the resident window */
};

/* output of the FIR-decimate chain
 *
 * Accuracy of the float32 outputs against the float64 reference arithmetic (BASELINE north_star:
 * "<= 1e-5 of full scale"), as the GPU tests hold it (tests/test_gpu_parity.py check_linear / check_fm,
 * tools/fuzz_parity.py):
 *   IQ, ENV, REAL  |err| <= e = 1e-5 * FS, FS = max(1, max|y|) (Q11 full scale is 1.0).  No exemption.
 *   FM             full scale is scale*pi (0.5 for scale 1/(2*pi)); outputs compare modulo one turn.
 *                  The output is the ANGLE of y[m+1]*conj(y[m]), so the linear bound e on y turns into
 *                  scale*(asin(e/|y[m]|) + asin(e/|y[m+1]|)) on the angle (~ scale*(e/|y[m]| + e/|y[m+1]|);
 *                  a vector shorter than e can point anywhere).  On the constant-envelope FM test signals
 *                  the tests demand 1e-5 of full scale wherever both |y| exceed 1e-3 of the block's rms,
 *                  and this bound everywhere else (noise inputs: this bound everywhere).  Two remarks, both
 *                  "the angle of a (nearly) zero vector", neither an exemption from the bound:
 *                  (1) start-up transient: the first ceil((T-1)/D)+1 outputs of a stream come from partly
 *                      filled windows (zero pre-history, bladerf_apt_demod.cpp:299-305) and are small, so
 *                      the bound is wide there; nothing else is special about them;
 *                  (2) where y[m] or y[m+1] is EXACTLY zero in the reference (zero pre-history, a run of
 *                      >= T zero samples in mid-stream) its output is 0 or +-scale*pi by the signs of its
 *                      zeros (bladerf_apt_demod.cpp:419-421): reproduced bit for bit by every kernel.  The
 *                      int8 matrix kernel (pipeline 5), whose fixed-point taps lose anything below 2^-31 of
 *                      the largest tap, recomputes with float32 taps every output of its own that came out
 *                      as an exact zero over a window that is not all zero, at a stream's start and in
 *                      mid-stream alike (round 2 documented the mid-stream case as a deviation; it is gone:
 *                      tests/test_gpu_parity.py::test_matrix_kernel_zero_runs). */
enum sdr_mode {
    SDR_MODE_IQ = 0,    /* complex y[m] (float2)           -- polyphase_decimate alone        */
    SDR_MODE_FM = 1,    /* scale*arg(y[m+1] conj(y[m]))    -- + polar_discriminator           */
    SDR_MODE_ENV = 2,   /* |y[m]|                          -- + std::abs envelope             */
    SDR_MODE_REAL = 3   /* Re y[m] (float): real-input filters, e.g. the FM audio post filters
                           af::fir(af_lpf_de, x5), af::fir(af_lpf_a, x6) + x6(seq_audio)
                           (bladerf/rx_fm_demod/src/bladerf_fm_demod.cpp:301-307)           */
};

/* sample type of the input stream */
enum sdr_input {
    SDR_IN_I16_IQ = 0,  /* interleaved int16 I,Q; the 1/2048 Q11 scale is applied            */
    SDR_IN_CF32 = 1,    /* interleaved float32 re,im (already scaled)                         */
    SDR_IN_F32_REAL = 2,/* float32 real (e.g. the discriminator output fed to APT stage 2)    */
    SDR_IN_I16_QI = 3,  /* interleaved int16 Q,I (the "Q16 I16" wire order the B205 example
                           records, b205_mini/rx_example/src/b205m_rx_main.cpp:54,210); Q11 scale   */
    SDR_IN_I8_IQ = 4    /* interleaved int8 I,Q (HackRF transfer buffers,
                           hackrf/rx_example/src/hackrf_rx_main.cpp:28,54; written as (int8_t) pairs
                           in hackrf/tx_example/src/hackrf_tx_main.cpp:150-151); scale 1/128.  Without
                           mixer these run on the int8 matrix kernel (pipeline 5) like int16 samples */
};

/* how many decimated samples y[m] = (h * x)[D*m] a block of n_in input samples yields */
enum sdr_count_rule {
    SDR_COUNT_FLOOR = 0,  /* m < floor(n_in/D): polyphase_decimate, bladerf_apt_demod.cpp:282 (APT chain) */
    SDR_COUNT_AF_SEQ = 1  /* D*m <= n_in-1, i.e. floor((n_in-1)/D)+1 samples: the ArrayFire loops'
                             x3(dec_seq), dec_seq = af::seq(0, num_samples, dec_rate) (inclusive end;
                             bladerf_fm_demod.cpp:212,227,294,307, bladerf_am_demod.cpp:213,297).  One
                             sample more than FLOOR unless D divides n_in -- then the seq's last index
                             IS n_in, one past the block (a latent out-of-bounds read in the reference,
                             SURVEY 8 a4): that index is never read here and the count stays n_in/D */
};

enum sdr_window_id {
    SDR_WIN_RECTANGULAR = 0, SDR_WIN_HANN = 1, SDR_WIN_HAMMING = 2, SDR_WIN_BLACKMAN = 3,
    SDR_WIN_NUTTALL = 4, SDR_WIN_BLACKMAN_HARRIS = 5
};

typedef struct sdr_fir_plan sdr_fir_plan;
typedef struct sdr_fft_plan sdr_fft_plan;

/* ---- library ------------------------------------------------------------------------------ */

/* Select `device` (HIP ordinal) for the calling thread and check that it is gfx950.
 * No reference counterpart (the reference is CPU-only). */
int sdr_init(int device);
/* "libsdrhip product build (gfx950)", or a text that starts with "libsdrhip MEASUREMENT BUILD" when
 * a translation unit was compiled with -DSDR_MEASUREMENT_BUILD (kernel parts compiled out to time the
 * rest: wrong results by construction; sdr_init() then prints this text to stderr).  The build of
 * sdr_amd/capi.py / __graft_entry__.build() never sets it. */
const char *sdr_build_info(void);
const char *sdr_strerror(int status);
/* text of the last failure on the calling thread (HIP error string, argument name...) */
const char *sdr_last_error(void);
/* device memory helpers so that non-HIP hosts (ctypes, cgo...) can own device buffers */
int sdr_dev_alloc(void **ptr, size_t bytes);
int sdr_dev_free(void *ptr);
int sdr_dev_upload(void *dst_dev, const void *src_host, size_t bytes);
int sdr_dev_download(void *dst_host, const void *src_dev, size_t bytes);
int sdr_dev_sync(void);
/* Number of device / pinned-host allocations the library has made in this process, its own (plan tables,
 * staging, scratch) and sdr_dev_alloc's.  A diagnostic with no reference counterpart: a steady-state
 * caller -- the same filter block after block through the C++ shims or the hand-off -- sees it stand still. */
unsigned long long sdr_alloc_count(void);

/* ---- a11: filter design (host code, no kernel) ---------------------------------------------
 * Replaces DSP::create_fir_filter<T>(n_taps, fc, &DSP::xxx_window) and DSP::xxx_window(N) of the
 * un-vendored davemers0160/Common repo; call sites bladerf_fm_demod.cpp:215,229,232,
 * bladerf_am_demod.cpp:217, bladerf_apt_demod.cpp:571,574.  PARITY UNPINNED (source absent):
 * symmetric window w[n] over n/(N-1); h[n] = w[n]*sin(pi*fc*t)/(pi*t), t = n-(N-1)/2, h = fc*w
 * at t = 0; no gain normalisation. */
int sdr_window(int window_id, size_t n, double *w);
int sdr_create_fir_filter(size_t n_taps, double fc, int window_id, double *h);

/* ---- a2/a3/a4/a5/a6/a7: FIR + decimate (+ demod) plan -------------------------------------- */
typedef struct sdr_fir_desc {
    int input;          /* enum sdr_input */
    int mode;           /* enum sdr_mode */
    int decim;          /* D >= 1 */
    int n_taps;         /* 1 <= T <= 16384; NO "T % D == 0" restriction.  A tile of
                           T + 63*D (*2 for odd D) samples has to fit the 160 KB of LDS: float
                           input or the mixer (8 B per sample) at 16384 taps stops near D = 30 for
                           odd D, SDR_EINVAL beyond */
    const double *taps; /* host, T doubles (rounded to float32 on upload; pipeline 5: to 31-bit fixed
                           point).  Sums are accumulated in float32: <= 4e-7 of full scale at the
                           reference's filter lengths (64..511 taps), growing like sqrt(T) to about
                           1e-5 at 16384 taps */
    double scale;       /* FM: discriminator scale, e.g. 1/(2*pi) (bladerf_apt_demod.cpp:560) */
    double mix_fr;      /* != 0: multiply x[n] by exp(+j*2*pi*mix_fr*n) before the FIR
                           (frequency_shift, bladerf_apt_demod.cpp:476-495; sign as
                           create_freq_rotation, sdr_functions.h:63-74).  n is the index in the
                           whole stream, so time-block shards stay phase-continuous. */
    int outputs_per_lane; /* 0 = choose automatically; else 1, 2, 4 or 8 (tuning knob) */
    int threads;          /* 0 = choose automatically; else 64, 128 or 256 */
    int pipeline;         /* kernel choice: 0 = automatic; 1 = raw int16 tiles in LDS filled by
                             LDS-DMA, one tile per workgroup (int16 input without mixer only);
                             2 = float tiles converted once, persistent workgroups that prefetch
                             the next tile into registers (any input);
                             3 = int16 converted once while staging into two float planes in LDS,
                             one tile per workgroup (int16 input without mixer only);
                             4 = the same float planes, window walk as a Toeplitz GEMM on the
                             matrix cores in float32 (long filters; int16 without mixer only);
                             5 = Toeplitz GEMM on the int8 matrix pipe: samples as their two
                             bytes (SDR_IN_I8_IQ: the sample bytes themselves, four byte products per
                             step instead of seven), taps as four base-256 digits, exact int32
                             accumulation (windows T + 15*D up to 768 samples; at least 2 taps, 6 for
                             int8 samples; int16 or int8 input without mixer only; leave
                             outputs_per_lane and threads 0; 512 lanes).  An output is
                             one exact integer sum over its window: its bits do not depend on tiling,
                             block hand-off or sharding.  Chosen automatically where a measured cost model
                             (tiles vs taps per sample) puts it ahead -- everywhere but small decimation
                             factors with few taps.  Taps
                             smaller than 2^-31 of the largest one are dropped by the fixed point;
                             discriminator outputs that such taps alone decide (a y that is exactly
                             zero in fixed point over a window that is not) are recomputed with
                             float32 taps, at a stream's start and after zero runs in mid-stream.
                             6 = K-split float tiles for SHORT real-input streams (SDR_IN_F32_REAL, any
                             mode but FM; 4 outputs per lane, 256 threads): the four waves of a workgroup
                             walk a quarter of the filter each and add their partial sums in a fixed
                             order -- a quarter of the latency of pipeline 2 where a launch has too few
                             lane windows to fill the chip (e.g. APT stage 2, bladerf_apt_demod.cpp:744-754,
                             whose input is a tenth of the capture block).  Chosen automatically from
                             size_hint.
                             7 = float tiles filled by LDS-DMA, one tile per workgroup (SDR_IN_CF32 or
                             SDR_IN_F32_REAL; no per-channel mixers): float samples need no conversion, so the
                             DMA'd image is the tile the walk reads -- the af::fir call shape of the FM receiver
                             (cf_samples block, bladerf_fm_demod.cpp:288-294).  A mixer (mix_fr) is folded into
                             complex taps h[i]*exp(-j*2*pi*fr*i) plus one phasor per output (phase of the
                             output's own sample, reduced in double), so mixer outputs are bit-identical under
                             shards and blocks here.  The automatic choice for complex float input (1.4-2.2x
                             pipeline 2) and for real float input at lane strides up to 40 samples; without
                             mixer the same float32 taps and summation order as pipeline 2. */
    int count_rule;       /* enum sdr_count_rule; 0 = the polyphase_decimate rule */
    long long size_hint;  /* 0 = unknown; else the input samples a launch will typically see.  Geometry is
                             chosen for long streams by default; with a hint that says "short" (fewer lane
                             windows than two waves per SIMD) real-input plans take the K-split kernel
                             (pipeline 6), and int16 / int8 plans stay on the vector-unit tile kernels where the
                             persistent matrix kernel's ~13 us launch latency would outweigh its rate
                             (blocks below ~2^26 samples at 64 taps / 10, ~2^22 at 511 / 10).  The choice is made once, at plan creation: a plan's outputs do
                             not depend on how a stream is cut into launches. */
    int n_mix_channels;   /* > 0: the channelizer (BASELINE config 3 as SURVEY 8 a7 words it) -- one NCO per
                             channel in front of the filter: channel c of a batched launch is multiplied by
                             exp(+j*2*pi*mix_fr_channels[c]*n), n = the sample's index in its stream
                             (create_freq_rotation per channel, sdr_functions.h:63-74; frequency_shift,
                             bladerf_apt_demod.cpp:476-495; the FM receiver's offset rotator
                             exp(-2j*pi*(f_offset/fs)*n), bladerf_fm_demod.cpp:237-241, is the negative
                             frequency).  An alternative to mix_fr (leave that 0); complex or integer input.
                             Launches through sdr_fir_execute_channels_dev; the single-stream entry points
                             (execute, range, host, stream hand-off) use channel 0's mixer. */
    const double *mix_fr_channels; /* host, n_mix_channels doubles (cycles per sample; copied) */
} sdr_fir_desc;

/* Replaces the set-up half of polyphase_decimate (bladerf_apt_demod.cpp:249-279: argument
 * checks + polyphase table) and af::array(lpf) (bladerf_fm_demod.cpp:217). */
int sdr_fir_plan_create(const sdr_fir_desc *desc, sdr_fir_plan **plan);
int sdr_fir_plan_destroy(sdr_fir_plan *plan);
/* number of outputs for n_in input samples: floor(n_in/D) (bladerf_apt_demod.cpp:282) or, for plans
 * with SDR_COUNT_AF_SEQ, floor((n_in-1)/D)+1; minus one in FM mode (:409; x4(1:end) * conjg(x4(0:end-1)),
 * bladerf_fm_demod.cpp:297). */
size_t sdr_fir_out_count(const sdr_fir_plan *plan, size_t n_in);
/* what the plan chose: outputs per lane, threads per workgroup, LDS bytes per workgroup */
int sdr_fir_plan_info(const sdr_fir_plan *plan, int *outputs_per_lane, int *threads,
                      int *lds_bytes);
/* which kernel the plan runs: 1 = raw-tile, 2 = streaming, 3 = planar, 4 = MFMA float32,
 * 5 = MFMA int8, 6 = K-split (see `pipeline`) */
int sdr_fir_plan_kernel(const sdr_fir_plan *plan);

/* One block from zero pre-history, host buffers.  Replaces, fused in one pass,
 *   the int16 -> float scale loop      bladerf_apt_demod.cpp:728-732 / bladerf_sdr.h:288-291
 *   polyphase_decimate(x, D, h)        bladerf_apt_demod.cpp:249-313
 *     (== af::fir + x3(dec_seq)        bladerf_fm_demod.cpp:291-294, bladerf_am_demod.cpp:294-297)
 *   polar_discriminator(x4, scale)     bladerf_apt_demod.cpp:405-425   (SDR_MODE_FM)
 *   std::abs / af::abs envelope        bladerf_apt_demod.cpp:750-754, bladerf_am_demod.cpp:311
 * out: float (FM, ENV) or float2 (IQ); out_capacity in output elements. */
int sdr_fir_execute_host(sdr_fir_plan *plan, const void *in, size_t n_in, void *out,
                         size_t out_capacity, size_t *n_out);

/* Same, device buffers, asynchronous on `stream`. */
int sdr_fir_execute_dev(sdr_fir_plan *plan, const void *d_in, size_t n_in, void *d_out,
                        size_t out_capacity, size_t *n_out, void *stream);

/* Time-block sharding: the stream has n_total samples; d_in points at the sample
 * with stream index in_first and n_resident samples are readable from it.  Computes the
 * whole-stream outputs j in [out_begin, out_end) into d_out[0 .. out_end-out_begin).  The
 * caller keeps the FIR history halo resident: samples D*out_begin-(T-1) .. D*(out_end-1)
 * (one more output, i.e. up to D*out_end, in FM mode) clipped to [0, n_total); otherwise
 * SDR_ERANGE.  No reference counterpart: the reference filters every block from zero state
 * (bladerf_fm_demod.cpp:284-291). */
int sdr_fir_execute_range_dev(sdr_fir_plan *plan, const void *d_in, uint64_t in_first,
                              uint64_t n_resident, uint64_t n_total, uint64_t out_begin,
                              uint64_t out_end, void *d_out, void *stream);

/* Batched independent channels (BASELINE config 3): channel c reads n_in samples at
 * d_in + c*in_stride samples and writes sdr_fir_out_count() outputs at d_out + c*out_stride
 * output elements; every channel starts from zero pre-history.  One launch. */
int sdr_fir_execute_batched_dev(sdr_fir_plan *plan, const void *d_in, size_t n_in,
                                size_t in_stride, size_t n_channels, void *d_out,
                                size_t out_stride, void *stream);

/* The channelizer launch: channels chan_first .. chan_first + n_channels - 1 of a plan created with
 * per-channel mixers (sdr_fir_desc.n_mix_channels), one launch.  Channel k reads n_in samples at
 * d_in + k*in_stride samples -- in_stride 0: every channel reads the SAME wideband buffer, the
 * channelizer proper -- is mixed with its own NCO (mix_fr_channels[chan_first + k]), filtered,
 * decimated and demodulated, and writes sdr_fir_out_count() outputs at d_out + k*out_stride elements.
 * chan_first lets a rank that owns a block of the channels run it against the same plan.
 * Replaces, per channel, create_freq_rotation / frequency_shift (sdr_functions.h:63-74,
 * bladerf_apt_demod.cpp:476-495) + polyphase_decimate (:249-313) + polar_discriminator (:405-425). */
int sdr_fir_execute_channels_dev(sdr_fir_plan *plan, const void *d_in, size_t n_in, size_t in_stride,
                                 size_t chan_first, size_t n_channels, void *d_out, size_t out_stride,
                                 void *stream);

/* ---- a10: windowed FFT -> power in dB ------------------------------------------------------
 * Replaces bladerf_main_v1.cpp:183-196: complex<float>(I,Q)*(1/2048) -> af::fftInPlace(x, 1/N)
 * -> 20*log10(shift(abs(X), N/2)); plus the window multiply and hop of BASELINE config 4.
 * Frame f covers samples [f*hop, f*hop+nfft); out[f*nfft + i] = 10*log10(|X_f[(i+nfft/2) %
 * nfft]|^2) with X = (1/nfft) * DFT(w .* x/2048).  No log floor (exact zeros give -inf, as in
 * the reference).  nfft: a power of two, 64 .. 1048576 (up to 4096 the frame is transformed in
 * registers; longer frames, like the reference's one 131072-sample block, go through multi-pass
 * kernels with plan-owned scratch).  window may be NULL (rectangular). */
int sdr_fft_plan_create(int nfft, int hop, const double *window, sdr_fft_plan **plan);
int sdr_fft_plan_destroy(sdr_fft_plan *plan);
/* Kernel choice for 4096-point frames at hop 2048 on long streams (tuning / A-B measurement knob; every
 * variant computes the same transform): 0 = automatic, 1 = one frame per 256-lane workgroup with
 * barrier-separated LDS exchanges (fft_db_kernel), 2 = two frames per lane (fft4096_pair_kernel),
 * 3 = one frame per wavefront, barrier-free, four waves per CU (fft4096_wave_kernel<false>),
 * 4 = the same with eight waves per CU (fft4096_wave_kernel<true>; the automatic choice).  Other
 * shapes ignore it. */
int sdr_fft_plan_set_variant(sdr_fft_plan *plan, int variant);
size_t sdr_fft_frame_count(const sdr_fft_plan *plan, size_t n_in);
int sdr_fft_execute_i16_host(sdr_fft_plan *plan, const int16_t *iq, size_t n_in, float *out_db,
                             size_t frame_capacity, size_t *n_frames);
int sdr_fft_execute_i16_dev(sdr_fft_plan *plan, const int16_t *d_iq, size_t n_in,
                            float *d_out_db, size_t frame_capacity, size_t *n_frames,
                            void *stream);

/* Allocation-free variant of sdr_fft_execute_i16_dev.  It has NO counterpart in the reference: it
 * exists for callers that cannot allocate device memory at the call -- code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  Same arguments, same results bit for bit,
 * and no hipMalloc / hipFree / allocating library routine on any path: frames longer than 4096
 * points, whose multi-pass transform needs two scratch arrays (sdr_fft_execute_i16_dev allocates and
 * grows them inside the plan on demand), take them from a file-scope device array of the library
 * instead -- 2 MiB per GPU, held until the process ends, resolved once per device by sdr_init() or
 * by the first call on that device.
 *   Limit: nfft <= 131072 (the reference's own capture block, bladerf_main_v1.cpp:139); larger
 *   plans return SDR_EINVAL with a text that says so.
 *   Every call on one GPU shares that GPU's array (its contents are overwritten before they are
 *   read, so nothing needs clearing): calls that overlap in time on one GPU -- different streams,
 *   different plans -- race.  Serialise them, or use sdr_fft_execute_i16_dev. */
int sdr_fft_execute_i16_noalloc_dev(sdr_fft_plan *plan, const int16_t *d_iq, size_t n_in,
                                    float *d_out_db, size_t frame_capacity, size_t *n_frames,
                                    void *stream);

/* ---- a6 (waterfall row): |x|/2048 straight from int16 --------------------------------------
 * Replaces the std::transform at bladerf_waterfall_main.cpp:272. */
int sdr_magnitude_i16_dev(const int16_t *d_iq, size_t n, float *d_mag, void *stream);

/* ---- f1: FM audio PCM stage --------------------------------------------------------------------
 * Replaces bladerf_fm_demod.cpp:310-313:  x7 = x7 * (1.0 / max(abs(x7)));  x7 = ((x7+1)*30).as(u8).
 * d_x: n floats (the decimated audio, e.g. a SDR_MODE_REAL plan's output); d_max: one float of
 * device scratch (receives max|x|); d_pcm: n bytes.  An all-zero block gives 30 everywhere (the
 * reference divides by zero there).  The float -> u8 conversion truncates like as(u8). */
int sdr_audio_u8_dev(const float *d_x, size_t n, float *d_max, uint8_t *d_pcm, void *stream);

/* ---- f3: APT line synchronisation and image assembly (host code, fed by the GPU chain) -----------
 * Replaces the tail of the consumer loop, bladerf_apt_demod.cpp:756-793, with
 * get_correlation_peaks (:181-213) and add_line_to_image (:216-246): min/max normalisation of the
 * pending envelope samples to 0..255, sliding correlation with the 39-sample sync pulse (:563),
 * one image row of `width` pixels per detected line, scrolling `rows` x `width` 8-bit image
 * (reference: 2080 x 700, :565,:580), consumed samples dropped up to 500 before the last peak.
 * push: x9 = n envelope samples (the SDR_MODE_ENV output of the second stage), *lines_added =
 * rows appended by this call.  get: copies the image (rows*width bytes, newest line last). */
typedef struct sdr_apt_image sdr_apt_image;
int sdr_apt_image_create(int width, int rows, sdr_apt_image **image);
int sdr_apt_image_destroy(sdr_apt_image *image);
int sdr_apt_image_push(sdr_apt_image *image, const float *x9, size_t n, int *lines_added);
int sdr_apt_image_get(const sdr_apt_image *image, uint8_t *dst, uint64_t *total_lines,
                      size_t *pending_samples);

/* ---- a2 alone: int16 -> complex float, x 1/2048 ---------------------------------------------
 * Replaces the scale loop of bladerf_sdr.h:288-291,334-337 for callers that want the float
 * stream itself. */
int sdr_unpack_i16_dev(const int16_t *d_iq, size_t n, float *d_cf32, void *stream);

/* ---- a7 alone: frequency shift / rotation vector --------------------------------------------
 * frequency_shift (bladerf_apt_demod.cpp:476-495) on a real (is_complex = 0) or complex float
 * stream: out[n] = in[n]*exp(+j*2*pi*fr*(n0+n)); with d_in == NULL it writes the rotation
 * itself, i.e. create_freq_rotation (sdr_functions.h:63-74). */
int sdr_frequency_shift_dev(const float *d_in, int is_complex, size_t n, uint64_t n0, double fr,
                            float *d_out_cf32, void *stream);

/* ---- a8 support: gather --------------------------------------------------------------------
 * out[i] = in[idx[i]] for elements of elem_dwords 32-bit words.  decimate_vec
 * (bladerf_apt_demod.cpp:374-401; cv_decimate, opencv_complex_functions.h:22-61) is this gather
 * with the reference's index sequence floor(0), floor(rate), floor(rate+rate), ... which the
 * C++ shim builds on the host exactly as the reference accumulates it.  An index >= n_in writes a
 * zero element (never a read outside d_in). */
int sdr_gather_dev(const void *d_in, int elem_dwords, size_t n_in, const uint64_t *d_idx,
                   size_t n_idx, void *d_out, void *stream);

/* ---- a12: block hand-off with persistent filter history ------------------------------------
 * Replaces SDR_BASE::start / wait_for_samples / stop (sdr_base.h:42-55), the capture thread's
 * buffer hand-off (bladerf_sdr.h:270-302) and apt_demod's iq_data_queue double buffer
 * (bladerf_apt_demod.cpp:64-71,152-178).  The producer pushes blocks of the plan's input type
 * (host memory); each is copied through pinned staging to the device behind the FIR history of
 * the previous block and filtered asynchronously on the stream's own HIP stream, so block
 * boundaries leave no transient (the reference restarts every block from zero state).  pop()
 * waits for the oldest block and returns the outputs that became computable with it; over all
 * blocks the popped outputs concatenate to exactly the one-shot result on the concatenated
 * input.  Two blocks may be in flight; a third push returns SDR_ECAPACITY.  Single producer /
 * single consumer thread, like the reference. */
typedef struct sdr_fir_stream sdr_fir_stream;
int sdr_fir_stream_create(sdr_fir_plan *plan, size_t max_block_samples, sdr_fir_stream **stream);
int sdr_fir_stream_destroy(sdr_fir_stream *stream);
int sdr_fir_stream_push(sdr_fir_stream *stream, const void *block_host, size_t n_samples);
int sdr_fir_stream_pop(sdr_fir_stream *stream, void *out_host, size_t capacity, size_t *n_out);
/* Zero-copy variant of push: *buffer = the pinned host buffer (max_block_samples long) of the slot
 * the next push will use -- the counterpart of handing iq_data_queue[write_index] to
 * bladerf_sync_rx (bladerf_apt_demod.cpp:166).  Fill it, then push that same pointer. */
int sdr_fir_stream_next_buffer(sdr_fir_stream *stream, void **buffer);
/* File replay source.  Replaces temp_get_data (bladerf/apt_demod/src/bladerf_apt_demod.cpp:92-148):
 * the raw sample file at `path` (samples of the plan's input kind, e.g. a .sc16 recording) is cut
 * to a multiple of two blocks (:113-117) and pushed block by block, `passes` times over (the
 * reference cycles until stopped, :139-145).  `sink` is called once per block, in order, with that
 * block's outputs (elements of the plan's output type); a non-zero return stops the replay
 * (SDR_ERANGE).  The stream is left empty.  *samples_replayed (optional) = samples pushed. */
typedef int (*sdr_block_sink)(void *user, const void *outputs, size_t n_outputs);
int sdr_fir_stream_replay_file(sdr_fir_stream *stream, const char *path, size_t block_samples,
                               int passes, sdr_block_sink sink, void *user,
                               uint64_t *samples_replayed);
/* input kind, mode, decimation and tap count of a plan */
int sdr_fir_plan_shape(const sdr_fir_plan *plan, int *input, int *mode, int *decim, int *n_taps);

/* ---- synthetic input (test/bench support; no reference counterpart) --------------------------
 * Writes n int16 IQ samples of an FM-modulated tone, stream indices n0 .. n0+n-1:
 * phi = (dev/ftone)*sin(2*pi*ftone*t/fs), I = round(amp*cos phi), Q = round(amp*sin phi), plus
 * uniform noise of +-noise_lsb.  Same law as the FM generators of the reference's TX examples
 * (amplitude ~2000, bladerf/tx_example/src/bladerf_tx_main_v1.cpp:120-138) and SURVEY 8d. */
int sdr_synth_fm_i16_dev(int16_t *d_iq, size_t n, uint64_t n0, double fs, double ftone,
                         double dev, double amp, int noise_lsb, unsigned seed, void *stream);

/* ---- timing helper for roofline measurement -------------------------------------------------
 * Enqueues `iters` back-to-back launches of the plan's kernel bracketed by HIP events on
 * `stream` and returns the average milliseconds per launch (synchronises). */
int sdr_fir_time_dev(sdr_fir_plan *plan, const void *d_in, size_t n_in, size_t in_stride,
                     size_t n_channels, void *d_out, size_t out_stride, int iters, void *stream,
                     float *ms_per_launch);
int sdr_fft_time_dev(sdr_fft_plan *plan, const int16_t *d_iq, size_t n_in, float *d_out_db,
                     int iters, void *stream, float *ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* SDR_HIP_H_ */
