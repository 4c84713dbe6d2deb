/*
 * sdr_oracle.c -- CPU restatement of the davemers0160/SDR receive-side DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/, the smoke check in
 * __graft_entry__.py and bench.py's cpu_baseline leg may load it.  The product path
 * (sdr_amd/csrc) never links or calls anything in here.
 *
 * Every function restates, in plain C (double precision unless noted), one function of the
 * reference and cites the file:line it follows (paths relative to the reference root).
 * Complex vectors are interleaved (re, im) doubles, layout-compatible with
 * std::vector<std::complex<double>>.
 *
 * Pinning: the restatement is checked bit-for-bit (f64) against the reference's own functions
 * compiled from /root/reference (oracle/_ref, see oracle/build_ref.sh) by
 * tests/golden/make_golden.py, whose outputs are committed under tests/golden/ and re-checked
 * by tests/test_oracle_golden.py.  Two pieces have NO reference source in the tree and are
 * "parity unpinned": the windowed-sinc designer (orc_create_fir_filter, orc_window -- the
 * reference calls DSP::create_fir_filter from the un-vendored davemers0160/Common repo) and the
 * FFT itself (ArrayFire's af::fftInPlace); for those the oracle states the documented
 * mathematical definition and the tests say so.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define ORC_OK 0
#define ORC_EINVAL 1

/* ---------------------------------------------------------------------------------------------
 * a1/a2: int16 IQ unpack + Q11 scale.
 * bladerf/apt_demod/src/bladerf_apt_demod.cpp:557,728-732
 *   cf_samples[j] = cf_scale * complex<double>(re, im), cf_scale = (1/2048, 0)
 * bladerf/common/include/bladerf_sdr.h:274,288-291 (same in float).
 * A complex multiply by (s, 0) is (re*s - im*0, re*0 + im*s); for finite inputs that equals
 * (re*s, im*s) including the sign of zero (int16 zeros convert to +0).
 * ------------------------------------------------------------------------------------------- */
void orc_unpack_scale_i16(const int16_t *iq, size_t n, double *out)
{
    const double s = 1.0 / 2048.0;
    for (size_t j = 0; j < n; ++j) {
        out[2 * j + 0] = (double)iq[2 * j + 0] * s;
        out[2 * j + 1] = (double)iq[2 * j + 1] * s;
    }
}

/* SURVEY 8(f2) wire formats.  Q-first int16 ("Q16 I16", b205_mini/rx_example/src/b205m_rx_main.cpp:54;
 * written by write_qi_data, :210): the pair is (Q, I), same Q11 scale.  The reader/writer
 * themselves live in the un-vendored Common repository (iq_utils.h): layout only, parity unpinned. */
void orc_unpack_scale_i16_qi(const int16_t *qi, size_t n, double *out)
{
    const double s = 1.0 / 2048.0;
    for (size_t j = 0; j < n; ++j) {
        out[2 * j + 0] = (double)qi[2 * j + 1] * s;
        out[2 * j + 1] = (double)qi[2 * j + 0] * s;
    }
}

/* HackRF transfer buffers: interleaved int8 I, Q (hackrf/rx_example/src/hackrf_rx_main.cpp:28,54;
 * the TX example fills them with (int8_t)real, (int8_t)imag, hackrf_tx_main.cpp:150-151).  The
 * reference never scales them; full scale 128 -> 1/128 is this build's documented choice. */
void orc_unpack_scale_i8(const int8_t *iq, size_t n, double *out)
{
    const double s = 1.0 / 128.0;
    for (size_t j = 0; j < 2 * n; ++j) out[j] = (double)iq[j] * s;
}

/* float flavour of the same loop (bladerf_sdr.h:288-291): exact, power-of-two scale. */
void orc_unpack_scale_i16_f32(const int16_t *iq, size_t n, float *out)
{
    const float s = 1.0f / 2048.0f;
    for (size_t j = 0; j < 2 * n; ++j) out[j] = (float)iq[j] * s;
}

/* ---------------------------------------------------------------------------------------------
 * a3: polyphase_decimate  -- bladerf/apt_demod/src/bladerf_apt_demod.cpp:249-313
 *   y[m] = sum_{k=0}^{D-1} sum_{p=0}^{T/D-1} h[p*D+k] * x[D*(m-p)-k], out-of-range terms skipped
 *   (zero pre-history, :299-305); n_out = floor(N/D) (:282); T % D != 0 or D <= 0 is an error
 *   (:254-259).  The summation order (phase outer, tap-in-phase inner) is kept so that the f64
 *   result is bit-identical to the reference.
 * x: n complex, h: T real, y: floor(n/D) complex.
 * ------------------------------------------------------------------------------------------- */
int orc_polyphase_decimate(const double *x, size_t n, int D, const double *h, int T, double *y)
{
    if (D <= 0) return ORC_EINVAL;
    if (T % D != 0) return ORC_EINVAL;
    const int P = T / D;
    const int64_t n_out = (int64_t)(n / (size_t)D);
    for (int64_t m = 0; m < n_out; ++m) {
        double sr = 0.0, si = 0.0;
        for (int k = 0; k < D; ++k) {
            for (int p = 0; p < P; ++p) {
                const int64_t index = (int64_t)D * (m - p) - k;
                if (index >= 0 && index < (int64_t)n) {
                    const double e = h[p * D + k];
                    sr += e * x[2 * index + 0];
                    si += e * x[2 * index + 1];
                }
            }
        }
        y[2 * m + 0] = sr;
        y[2 * m + 1] = si;
    }
    return ORC_OK;
}

/* Same arithmetic without the T % D restriction: the taps are zero-padded to the next multiple of
 * D (adding h=0 terms leaves every partial sum unchanged), which is how SURVEY 0.6 drives the
 * reference for 64- and 511-tap filters.  `pre` gives readable history x[-pre..-1] (pre = 0 is
 * the reference's zero pre-history); used for time-block sharding checks. */
int orc_fir_decimate(const double *x, size_t n, size_t pre, int D, const double *h, int T,
                     double *y)
{
    if (D <= 0 || T <= 0) return ORC_EINVAL;
    const int P = (T + D - 1) / D;
    const int64_t n_out = (int64_t)(n / (size_t)D);
    for (int64_t m = 0; m < n_out; ++m) {
        double sr = 0.0, si = 0.0;
        for (int k = 0; k < D; ++k) {
            for (int p = 0; p < P; ++p) {
                const int t = p * D + k;
                const int64_t index = (int64_t)D * (m - p) - k;
                if (t < T && index >= -(int64_t)pre && index < (int64_t)n) {
                    sr += h[t] * x[2 * index + 0];
                    si += h[t] * x[2 * index + 1];
                }
            }
        }
        y[2 * m + 0] = sr;
        y[2 * m + 1] = si;
    }
    return ORC_OK;
}

/* a4: af::fir + stride pick -- bladerf/rx_fm_demod/src/bladerf_fm_demod.cpp:212,291,294 and
 * bladerf/rx_am_demod/src/bladerf_am_demod.cpp:213,294,297.  af::fir is the causal length-N
 * convolution y[n] = sum h[k] x[n-k] from zero state; dec_seq = af::seq(0, N, D) picks n = 0, D,
 * 2D, ... <= N (inclusive end).  Index N is out of bounds when N % D == 0 (latent OOB in the
 * reference); this restatement returns the in-range picks only: count = floor((N-1)/D) + 1.
 * Direct tap-order summation (k ascending).  ArrayFire is absent: "parity unpinned" beyond the
 * identity with a3 at the kept indices. */
size_t orc_fir_pick(const double *x, size_t n, int D, const double *h, int T, double *y)
{
    if (D <= 0 || n == 0) return 0;
    const size_t n_out = (n - 1) / (size_t)D + 1;
    for (size_t m = 0; m < n_out; ++m) {
        double sr = 0.0, si = 0.0;
        for (int k = 0; k < T; ++k) {
            const int64_t index = (int64_t)m * D - k;
            if (index >= 0) {
                sr += h[k] * x[2 * index + 0];
                si += h[k] * x[2 * index + 1];
            }
        }
        y[2 * m + 0] = sr;
        y[2 * m + 1] = si;
    }
    return n_out;
}

/* ---------------------------------------------------------------------------------------------
 * a5: polar_discriminator -- bladerf/apt_demod/src/bladerf_apt_demod.cpp:405-425
 *   tmp = v[m+1] * conj(v[m]);  d[m] = scale * atan2(tmp.imag, tmp.real);  length n-1.
 * (AF twin: bladerf_fm_demod.cpp:297-298; cv twin: opencv_complex_functions.h:65-92.)
 * The complex product is written out as libstdc++ evaluates it for finite operands,
 * (a+bi)(c+ei) = (a*c - b*e) + (a*e + b*c)i with e = -d, so that signed zeros (a zero first
 * output when h[0] == 0) reach atan2 exactly as they do in the reference.
 * ------------------------------------------------------------------------------------------- */
void orc_polar_discriminator(const double *v, size_t n, double scale, double *d)
{
    if (n < 2) return;
    for (size_t m = 0; m + 1 < n; ++m) {
        const double a = v[2 * (m + 1) + 0], b = v[2 * (m + 1) + 1];
        const double c = v[2 * m + 0], e = -v[2 * m + 1];
        const double re = a * c - b * e;
        const double im = a * e + b * c;
        d[m] = scale * atan2(im, re);
    }
}

/* ---------------------------------------------------------------------------------------------
 * a6: envelope -- bladerf/apt_demod/src/bladerf_apt_demod.cpp:750-754 (std::abs(complex<double>)),
 * bladerf/rx_am_demod/src/bladerf_am_demod.cpp:311 (af::abs),
 * bladerf/common/include/opencv_complex_functions.h:142-158 (cv_cmplx_abs).
 * std::abs(std::complex<double>) is cabs(), i.e. hypot(re, im).
 * ------------------------------------------------------------------------------------------- */
void orc_envelope(const double *v, size_t n, double *e)
{
    for (size_t j = 0; j < n; ++j) e[j] = hypot(v[2 * j + 0], v[2 * j + 1]);
}

/* Waterfall row: bladerf/rx_waterfall/src/bladerf_waterfall_main.cpp:147,272
 *   |complex<double>(re*scale, im*scale)| straight from the int16 pair. */
void orc_magnitude_i16(const int16_t *iq, size_t n, double *mag)
{
    const double s = 1.0 / 2048.0;
    for (size_t j = 0; j < n; ++j) mag[j] = hypot(iq[2 * j] * s, iq[2 * j + 1] * s);
}

/* ---------------------------------------------------------------------------------------------
 * a7: frequency_shift -- bladerf/apt_demod/src/bladerf_apt_demod.cpp:476-495
 *   frc = (0, 2*pi*fr);  out[n] = data[n] * exp(frc * (double)n)
 * std::exp(complex) is cexp(); frc*index has real part 0*index = 0.  Real input (T = double,
 * the APT call at :744) multiplies both components by the scalar; complex input does a full
 * complex product.
 * ------------------------------------------------------------------------------------------- */
void orc_frequency_shift_real(const double *data, size_t n, double fr, double *out)
{
    const double w = 2.0 * M_PI * fr;
    double index = 0;
    for (size_t j = 0; j < n; ++j) {
        const double complex r = cexp(CMPLX(0.0 * index, w * index));
        out[2 * j + 0] = data[j] * creal(r);
        out[2 * j + 1] = data[j] * cimag(r);
        index += 1;
    }
}

void orc_frequency_shift_cplx(const double *data, size_t n, double fr, double *out)
{
    const double w = 2.0 * M_PI * fr;
    double index = 0;
    for (size_t j = 0; j < n; ++j) {
        const double complex r = cexp(CMPLX(0.0 * index, w * index));
        const double a = data[2 * j + 0], b = data[2 * j + 1];
        const double c = creal(r), e = cimag(r);
        out[2 * j + 0] = a * c - b * e;
        out[2 * j + 1] = a * e + b * c;
        index += 1;
    }
}

/* create_freq_rotation -- bladerf/common/include/sdr_functions.h:63-74
 *   res[idx] = exp(1i * 2.0 * M_PI * fr * (double)idx): the products associate left to right,
 *   ((((0,1)*2.0)*M_PI)*fr)*idx, giving imaginary part ((2.0*M_PI)*fr)*idx. */
void orc_create_freq_rotation(uint64_t N, double fr, double *out)
{
    for (uint64_t idx = 0; idx < N; ++idx) {
        const double re = ((0.0 * 2.0) * M_PI) * fr * (double)idx;
        const double im = ((1.0 * 2.0) * M_PI) * fr * (double)idx;
        const double complex r = cexp(CMPLX(re, im));
        out[2 * idx + 0] = creal(r);
        out[2 * idx + 1] = cimag(r);
    }
}

/* ---------------------------------------------------------------------------------------------
 * a8: decimate_vec -- bladerf/apt_demod/src/bladerf_apt_demod.cpp:374-401
 *   num = ceil(N / rate); res[i] = v[floor(index)], index += rate (accumulated in double).
 * Returns num; idx_out (optional) receives the picked indices.  Elements are `elem` doubles wide
 * (1 = real, 2 = complex).  Like the reference, an index that lands past the end is NOT
 * guarded there; here it is clamped to N-1 and counted in *n_oob so tests can see it.
 * ------------------------------------------------------------------------------------------- */
size_t orc_decimate_vec(const double *v, size_t n, int elem, double rate, double *res,
                        int64_t *idx_out, size_t *n_oob)
{
    const size_t num = (size_t)ceil((double)n / rate);
    double index = 0.0;
    size_t oob = 0;
    for (size_t i = 0; i < num; ++i) {
        int64_t k = (int64_t)floor(index);
        if (k >= (int64_t)n) { k = (int64_t)n - 1; ++oob; }
        if (idx_out) idx_out[i] = k;
        if (res) memcpy(res + (size_t)elem * i, v + (size_t)elem * k, sizeof(double) * elem);
        index += rate;
    }
    if (n_oob) *n_oob = oob;
    return num;
}

/* ---------------------------------------------------------------------------------------------
 * a9: filter_vec -- bladerf/apt_demod/src/bladerf_apt_demod.cpp:428-455
 *   centred "same" FIR: h2 = T>>1; for each idx: jmn = idx>=h2 ? 0 : h2-idx;
 *   jmx = idx < N-h2 ? T-1 : N-idx  (the tail bound truncates early -- kept as is);
 *   kdx = max(0, idx-h2); res[idx] = sum_{j=jmn}^{jmx} v[kdx++] * h[T-1-j].
 * Real input, real taps.  The reference indexes v[kdx] and h[T-1-j] unguarded; for the tail
 * rows with jmx > T-1 (N-idx > T-1 cannot happen there) both stay in range when N >= T.
 * Requires n >= T (returns ORC_EINVAL otherwise, where the reference would read out of range).
 * ------------------------------------------------------------------------------------------- */
int orc_filter_vec(const double *v, size_t n, const double *h, int T, double *res)
{
    if (T <= 0 || n < (size_t)T) return ORC_EINVAL;
    const uint64_t h2 = (uint64_t)T >> 1;
    for (uint64_t idx = 0; idx < n; ++idx) {
        const uint64_t jmn = (idx >= h2) ? 0 : (h2 - idx);
        const uint64_t jmx = (idx < n - h2) ? (uint64_t)T - 1 : n - idx;
        double acc = 0.0;
        uint64_t kdx = (idx > h2) ? idx - h2 : 0;
        for (uint64_t j = jmn; j <= jmx; ++j, ++kdx) acc += v[kdx] * h[T - j - 1];
        res[idx] = acc;
    }
    return ORC_OK;
}

/* ---------------------------------------------------------------------------------------------
 * a11: windows + windowed-sinc low-pass designer.  PARITY UNPINNED.
 * The reference calls DSP::create_fir_filter<T>(N, fc, &DSP::hann_window) etc. from the external,
 * un-vendored repo davemers0160/Common (dsp/dsp_windows.h, dsp/dsp_filtering.h); only the call
 * sites are in the tree (bladerf_fm_demod.cpp:215,229,232; bladerf_am_demod.cpp:217;
 * bladerf_apt_demod.cpp:571,574).  The definition used here (and by the product's host code):
 *   symmetric windows of length N evaluated at n/(N-1), n = 0..N-1;
 *   h[n] = w[n] * sinc_lp(n - (N-1)/2),  sinc_lp(t) = sin(pi*fc*t)/(pi*t),  sinc_lp(0) = fc
 *   (fc = cutoff / (fs/2) would make this the classic fir1; the call sites pass cutoff/fs-like
 *   ratios, which we take as normalised to Nyquist = 1); no DC-gain renormalisation.
 * window ids: 0 rectangular, 1 hann, 2 hamming, 3 blackman (0.42/0.5/0.08), 4 nuttall
 * (0.355768/0.487396/0.144232/0.012604), 5 blackman-harris (0.35875/0.48829/0.14128/0.01168).
 * ------------------------------------------------------------------------------------------- */
int orc_window(int id, size_t N, double *w)
{
    if (N == 0) return ORC_EINVAL;
    if (N == 1) { w[0] = 1.0; return ORC_OK; }
    const double den = (double)(N - 1);
    for (size_t n = 0; n < N; ++n) {
        const double t = (double)n / den;
        switch (id) {
        case 0: w[n] = 1.0; break;
        case 1: w[n] = 0.5 - 0.5 * cos(2.0 * M_PI * t); break;
        case 2: w[n] = 0.54 - 0.46 * cos(2.0 * M_PI * t); break;
        case 3: w[n] = 0.42 - 0.5 * cos(2.0 * M_PI * t) + 0.08 * cos(4.0 * M_PI * t); break;
        case 4:
            w[n] = 0.355768 - 0.487396 * cos(2.0 * M_PI * t) + 0.144232 * cos(4.0 * M_PI * t) -
                   0.012604 * cos(6.0 * M_PI * t);
            break;
        case 5:
            w[n] = 0.35875 - 0.48829 * cos(2.0 * M_PI * t) + 0.14128 * cos(4.0 * M_PI * t) -
                   0.01168 * cos(6.0 * M_PI * t);
            break;
        default: return ORC_EINVAL;
        }
    }
    return ORC_OK;
}

int orc_create_fir_filter(size_t N, double fc, int window_id, double *h)
{
    if (N == 0) return ORC_EINVAL;
    double *w = (double *)malloc(sizeof(double) * N);
    if (!w) return ORC_EINVAL;
    if (orc_window(window_id, N, w) != ORC_OK) { free(w); return ORC_EINVAL; }
    const double c = ((double)N - 1.0) / 2.0;
    for (size_t n = 0; n < N; ++n) {
        const double t = (double)n - c;
        const double s = (t == 0.0) ? fc : sin(M_PI * fc * t) / (M_PI * t);
        h[n] = w[n] * s;
    }
    free(w);
    return ORC_OK;
}

/* ---------------------------------------------------------------------------------------------
 * a10: FFT -> dB.  bladerf/rx_example/src/bladerf_main_v1.cpp:139,183-196
 *   c = complex<float>(I, Q) * (1/2048);  fftInPlace(c, 1/N)  [forward DFT e^{-j2pi kn/N} * 1/N];
 *   out = 20*log10(shift(abs(X), N/2)).   No window, no overlap, no log floor in the reference.
 * BASELINE config 4 adds a window (multiplied into the samples before the DFT), frames of `nfft`
 * every `hop` samples.  out[f*nfft + i] = 20*log10(|X_f[(i + nfft/2) mod nfft]|), computed as
 * 10*log10(re^2+im^2).  Radix-2 iterative DIT in double; nfft must be a power of two.
 * The DFT itself is ArrayFire in the reference: "parity unpinned" (mathematical definition).
 * window may be NULL (rectangular).  Returns the number of frames.
 * ------------------------------------------------------------------------------------------- */
static void fft_pow2(double *re, double *im, size_t n)
{
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t half = len >> 1;
        for (size_t k = 0; k < half; ++k) {
            const double ang = -2.0 * M_PI * (double)k / (double)len;
            const double wr = cos(ang), wi = sin(ang);
            for (size_t s = k; s < n; s += len) {
                const size_t u = s, v = s + half;
                const double xr = re[v] * wr - im[v] * wi;
                const double xi = re[v] * wi + im[v] * wr;
                re[v] = re[u] - xr; im[v] = im[u] - xi;
                re[u] += xr;        im[u] += xi;
            }
        }
    }
}

size_t orc_fft_power_db_i16(const int16_t *iq, size_t n, size_t nfft, size_t hop,
                            const double *window, double *out_db, double *out_lin)
{
    if (nfft == 0 || (nfft & (nfft - 1)) || hop == 0 || n < nfft) return 0;
    const size_t frames = (n - nfft) / hop + 1;
    double *re = (double *)malloc(sizeof(double) * nfft * 2);
    double *im = re + nfft;
    const double s = 1.0 / 2048.0, inv = 1.0 / (double)nfft;
    for (size_t f = 0; f < frames; ++f) {
        const int16_t *p = iq + 2 * f * hop;
        for (size_t j = 0; j < nfft; ++j) {
            const double w = window ? window[j] : 1.0;
            re[j] = (double)p[2 * j + 0] * s * w;
            im[j] = (double)p[2 * j + 1] * s * w;
        }
        fft_pow2(re, im, nfft);
        for (size_t i = 0; i < nfft; ++i) {
            const size_t k = (i + nfft / 2) & (nfft - 1);
            const double xr = re[k] * inv, xi = im[k] * inv;
            const double pw = xr * xr + xi * xi;
            if (out_lin) out_lin[f * nfft + i] = pw;
            if (out_db) out_db[f * nfft + i] = 10.0 * log10(pw);
        }
    }
    free(re);
    return frames;
}

/* ---------------------------------------------------------------------------------------------
 * Chains (the consumer loops), expressed with the functions above.
 * FM / envelope chain on int16 IQ: bladerf_apt_demod.cpp:728-742 (scale, polyphase_decimate,
 * polar_discriminator) and :750-754 (abs).  mode: 0 = complex y (2 doubles per output),
 * 1 = discriminator (n_out-1 reals), 2 = envelope (n_out reals).  Returns the output count.
 * ------------------------------------------------------------------------------------------- */
size_t orc_chain_i16(const int16_t *iq, size_t n, int D, const double *h, int T, int mode,
                     double scale, double *out)
{
    if (D <= 0 || T <= 0) return 0;
    const size_t ny = n / (size_t)D;
    double *x = (double *)malloc(sizeof(double) * 2 * (n ? n : 1));
    double *y = (double *)malloc(sizeof(double) * 2 * (ny ? ny : 1));
    size_t n_ret = 0;
    orc_unpack_scale_i16(iq, n, x);
    orc_fir_decimate(x, n, 0, D, h, T, y);
    if (mode == 0) { memcpy(out, y, sizeof(double) * 2 * ny); n_ret = ny; }
    else if (mode == 1) { orc_polar_discriminator(y, ny, scale, out); n_ret = ny ? ny - 1 : 0; }
    else { orc_envelope(y, ny, out); n_ret = ny; }
    free(x); free(y);
    return n_ret;
}

/* APT two-stage chain: bladerf_apt_demod.cpp:728-754
 *   x4 = polyphase_decimate(cf, D1, h1); x6 = polar_discriminator(x4, 1/(2pi));
 *   x7 = frequency_shift(x6, fr); x8 = polyphase_decimate(x7, D2, h2); x9 = |x8|.
 * Returns len(x9) = floor((floor(n/D1)-1)/D2).  x6_out (optional) receives the stage-1 output. */
size_t orc_apt_chain_i16(const int16_t *iq, size_t n, int D1, const double *h1, int T1,
                         double disc_scale, double fr, int D2, const double *h2, int T2,
                         double *x9, double *x6_out)
{
    const size_t n4 = n / (size_t)D1;
    if (n4 < 2) return 0;
    const size_t n6 = n4 - 1;
    double *x6 = (double *)malloc(sizeof(double) * n6);
    orc_chain_i16(iq, n, D1, h1, T1, 1, disc_scale, x6);
    if (x6_out) memcpy(x6_out, x6, sizeof(double) * n6);
    double *x7 = (double *)malloc(sizeof(double) * 2 * n6);
    orc_frequency_shift_real(x6, n6, fr, x7);
    const size_t n8 = n6 / (size_t)D2;
    double *x8 = (double *)malloc(sizeof(double) * 2 * (n8 ? n8 : 1));
    orc_fir_decimate(x7, n6, 0, D2, h2, T2, x8);
    orc_envelope(x8, n8, x9);
    free(x6); free(x7); free(x8);
    return n8;
}

/* Synthetic FM test signal: phase accumulated in double,
 * I = round(A cos phi), Q = round(A sin phi), phi' = 2pi*(dev/fs)*cos(2pi*(ftone/fs)*n).
 * Not a reference function -- the deterministic input generator shared by tests and bench. */
void orc_synth_fm_i16(int16_t *iq, size_t n, uint64_t n0, double fs, double ftone, double dev,
                      double amp)
{
    /* closed-form phase so that any block [n0, n0+n) can be generated independently:
     * phi(n) = (dev/ftone) * sin(2pi ftone n / fs) */
    const double beta = dev / ftone;
    for (size_t j = 0; j < n; ++j) {
        const double t = (double)(n0 + j);
        const double ph = beta * sin(2.0 * M_PI * ftone * t / fs);
        iq[2 * j + 0] = (int16_t)lrint(amp * cos(ph));
        iq[2 * j + 1] = (int16_t)lrint(amp * sin(ph));
    }
}


/* ---------------------------------------------------------------------------------------------
 * f1: FM audio post-chain -- bladerf/rx_fm_demod/src/bladerf_fm_demod.cpp:301-313
 *   x6 = af::fir(af_lpf_de, x5);  x6 = af::fir(af_lpf_a, x6);  x7 = x6(seq_audio);
 *   x7 = x7 * (1.0 / max(abs(x7)));  x7 = ((x7 + 1) * 30).as(u8);
 * af::fir is the causal, length-preserving FIR from zero state.  ArrayFire is not in
 * the tree: PARITY UNPINNED beyond these documented semantics.  Divergences restated from the
 * build: kept indices are m*D for m < floor(n/D) (the reference's af::seq end is inclusive and can
 * index one past the data); an all-zero block gives 30 (the reference divides by zero).
 * x5: n floats as doubles; audio (optional): floor(n/D) normalised values before quantisation;
 * pcm: floor(n/D) bytes.  Returns floor(n/D). */
size_t orc_fm_audio_chain_rule(const double *x5, size_t n, const double *h_de, int T_de,
                               const double *h_a, int T_a, int D, int af_seq, double *audio, uint8_t *pcm);
size_t orc_fm_audio_chain(const double *x5, size_t n, const double *h_de, int T_de,
                          const double *h_a, int T_a, int D, double *audio, uint8_t *pcm)
{
    return orc_fm_audio_chain_rule(x5, n, h_de, T_de, h_a, T_a, D, 0, audio, pcm);
}

/* af_seq != 0: the kept indices are those of x6(seq_audio), seq_audio = af::seq(0, dec_seq.dims(0),
 * dec_audio) (bladerf_fm_demod.cpp:227,307), that lie inside x6: 0, D, 2D, ... <= n-1, i.e.
 * floor((n-1)/D)+1 samples (dec_seq.dims(0) is n+1 or n+2, so the seq reaches at least n-1; its
 * indices >= n would be out of bounds in the reference and are not picked). */
size_t orc_fm_audio_chain_rule(const double *x5, size_t n, const double *h_de, int T_de,
                               const double *h_a, int T_a, int D, int af_seq, double *audio, uint8_t *pcm)
{
    if (D <= 0 || T_de <= 0 || T_a <= 0) return 0;
    double *a = (double *)calloc(n ? n : 1, sizeof(double));
    double *b = (double *)calloc(n ? n : 1, sizeof(double));
    for (size_t i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int k = 0; k < T_de && (size_t)k <= i; ++k) acc += h_de[k] * x5[i - k];
        a[i] = acc;
    }
    for (size_t i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int k = 0; k < T_a && (size_t)k <= i; ++k) acc += h_a[k] * a[i - k];
        b[i] = acc;
    }
    const size_t m = af_seq ? (n ? (n - 1) / (size_t)D + 1 : 0) : n / (size_t)D;
    double mx = 0.0;
    for (size_t j = 0; j < m; ++j) { const double v = fabs(b[j * D]); if (v > mx) mx = v; }
    const double r = mx > 0.0 ? 1.0 / mx : 0.0;
    for (size_t j = 0; j < m; ++j) {
        const double v = b[j * D] * r;
        if (audio) audio[j] = v;
        pcm[j] = (uint8_t)(int)((v + 1.0) * 30.0);
    }
    free(a); free(b);
    return m;
}

/* ---------------------------------------------------------------------------------------------
 * f3: APT line sync + image assembly -- bladerf/apt_demod/src/bladerf_apt_demod.cpp:756-793,
 * get_correlation_peaks :181-213, add_line_to_image :216-246, sync pulse :563.
 * OpenCV is not in the tree: PARITY UNPINNED; restated semantics: scalar*(A - s) evaluates as
 * A*alpha + beta in double, convertTo = cvRound (half to even) + saturate, Mat::dot sums in double.
 * Divergences restated from the build: fewer than 39 pending samples or a constant block do
 * nothing (the reference's unsigned loop bound wraps / it divides by zero); the drop count is
 * max(last_peak - 500, 0) with a real clamp (the reference's unsigned max cannot clamp).
 *
 * One block: x10 (capacity cap10, *n10 valid) gets x9 appended, img (rows x width) gets one row
 * per detected line, the consumed head of x10 is dropped.  peaks_out (optional, up to max_peaks
 * entries) receives the peak positions of this block.  Returns the number of rows added, or -1
 * when x10 would overflow. */
static const int16_t orc_sync_pulse[39] = {-128, -128, -128, -128, 127, 127, -128, -128, 127, 127, -128,
    -128, 127, 127, -128, -128, 127, 127, -128, -128, 127, 127, -128, -128, 127, 127, -128, -128, 127,
    127, -128, -128, -128, -128, -128, -128, -128, -128, -128};

int orc_apt_sync_block(double *x10, size_t *n10, size_t cap10, const double *x9, size_t n9,
                       uint8_t *img, int rows, int width, long long *peaks_out, int max_peaks,
                       int *n_peaks)
{
    if (n_peaks) *n_peaks = 0;
    if (*n10 + n9 > cap10) return -1;
    memcpy(x10 + *n10, x9, n9 * sizeof(double));
    *n10 += n9;
    const size_t N = *n10;
    if (N < 39) return 0;
    double lo = x10[0], hi = x10[0];
    for (size_t i = 1; i < N; ++i) { if (x10[i] < lo) lo = x10[i]; if (x10[i] > hi) hi = x10[i]; }
    const double delta = hi - lo;
    if (!(delta > 0.0)) return 0;
    const double alpha = 255.0 / delta, beta = -lo * alpha;
    int16_t *x12 = (int16_t *)malloc(N * sizeof(int16_t));
    uint8_t *x11 = (uint8_t *)malloc(N);
    for (size_t i = 0; i < N; ++i) {
        const double v = x10[i] * alpha + beta;
        double s = nearbyint(v - 128.0);
        if (s < -32768.0) s = -32768.0;
        if (s > 32767.0) s = 32767.0;
        x12[i] = (int16_t)s;
        double u = nearbyint(v);
        if (u < 0.0) u = 0.0;
        if (u > 255.0) u = 255.0;
        x11[i] = (uint8_t)u;
    }
    /* peaks: list starts as {(0, 0)} */
    size_t pcap = N / 500 + 4, np = 1;
    long long *ppos = (long long *)calloc(pcap, sizeof(long long));
    double *pval = (double *)calloc(pcap, sizeof(double));
    size_t index = 0;
    while (index <= N - 39) {
        double dot = 0.0;
        for (int k = 0; k < 39; ++k) dot += (double)x12[index + k] * (double)orc_sync_pulse[k];
        const double corr = dot / 39.0;
        if ((long long)index - ppos[np - 1] > 2000) {
            ppos[np] = (long long)index; pval[np] = corr; ++np;
            index += 500;
        } else if (corr > pval[np - 1]) {
            ppos[np - 1] = (long long)index; pval[np - 1] = corr;
        }
        ++index;
    }
    int added = 0;
    for (size_t j = 0; j + 1 < np; ++j) {
        if (ppos[j] + width >= (long long)N) continue;
        if (rows > 1) memmove(img, img + width, (size_t)width * (size_t)(rows - 1));
        memcpy(img + (size_t)width * (size_t)(rows - 1), x11 + ppos[j], (size_t)width);
        ++added;
    }
    if (n_peaks) *n_peaks = (int)np;
    if (peaks_out) for (size_t j = 0; j < np && (int)j < max_peaks; ++j) peaks_out[j] = ppos[j];
    long long drop = ppos[np - 1] - 500;
    if (drop < 0) drop = 0;
    memmove(x10, x10 + drop, (N - (size_t)drop) * sizeof(double));
    *n10 = N - (size_t)drop;
    free(x12); free(x11); free(ppos); free(pval);
    return added;
}
