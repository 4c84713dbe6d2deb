// sdr_hip_shims.hpp -- the reference's C++ function signatures on top of the libsdrhip C ABI.
//
// A reference main() (e.g. bladerf/apt_demod/src/bladerf_apt_demod.cpp) keeps calling
//     polyphase_decimate(x, D, h), polar_discriminator(x4, scale), frequency_shift(x6, fr),
//     decimate_vec(v, rate), filter_vec(v, h), create_freq_rotation(N, fr),
//     DSP::create_fir_filter<T>(n, fc, &DSP::hann_window)
// with std::vector arguments; these header-only shims forward to sdr_hip.h.  Status codes become
// the exceptions the reference throws (std::invalid_argument, bladerf_apt_demod.cpp:254-259) or
// std::runtime_error.  std::complex<float> / std::complex<int16_t> are layout compatible with
// float[2] / int16_t[2], so nothing is repacked at the boundary.
//
// Precision: the reference functions compute in double; the GPU path computes in float32 and the
// shims widen the result (BASELINE tolerance: 1e-5 of full scale).  The fused entry points at the
// bottom (sdrhip::fm_demod_i16 ...) are what a maintainer would switch the consumer loop to: one
// kernel instead of scale loop + polyphase_decimate + polar_discriminator.
#ifndef SDR_HIP_SHIMS_HPP_
#define SDR_HIP_SHIMS_HPP_

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "sdr_hip.h"

namespace sdrhip {

inline void check(int status)
{
    if (status == SDR_OK) return;
    const std::string msg = std::string(sdr_strerror(status)) + ": " + sdr_last_error();
    if (status == SDR_EINVAL) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}

// RAII device buffer: freed on every path, exceptions included
class DevMem {
public:
    explicit DevMem(size_t bytes) { check(sdr_dev_alloc(&p_, bytes)); }
    ~DevMem() { sdr_dev_free(p_); }
    DevMem(const DevMem&) = delete;
    DevMem& operator=(const DevMem&) = delete;
    void* get() const { return p_; }
    template <class T> T* as() const { return static_cast<T*>(p_); }
private:
    void* p_ = nullptr;
};

// RAII plan
class FirPlan {
public:
    FirPlan(const std::vector<double>& taps, int decim, int mode, int input, double scale = 1.0,
            double mix_fr = 0.0, int count_rule = SDR_COUNT_FLOOR, long long size_hint = 0)
    {
        sdr_fir_desc d{};
        d.input = input; d.mode = mode; d.decim = decim; d.n_taps = (int)taps.size();
        d.taps = taps.data(); d.scale = scale; d.mix_fr = mix_fr; d.count_rule = count_rule;
        d.size_hint = size_hint;
        check(sdr_fir_plan_create(&d, &p_));
    }
    ~FirPlan() { sdr_fir_plan_destroy(p_); }
    FirPlan(const FirPlan&) = delete;
    FirPlan& operator=(const FirPlan&) = delete;
    sdr_fir_plan* get() const { return p_; }
    size_t out_count(size_t n) const { return sdr_fir_out_count(p_, n); }
    // host buffers in, host buffers out
    template <class In, class Out>
    std::vector<Out> run(const In* in, size_t n) const
    {
        std::vector<Out> out(std::max<size_t>(out_count(n), 1));
        size_t n_out = 0;
        check(sdr_fir_execute_host(p_, in, n, out.data(), out_count(n), &n_out));
        out.resize(n_out);
        return out;
    }
private:
    sdr_fir_plan* p_ = nullptr;
};

// ---- fused consumer-loop entry points (what the reference's while-loop body becomes) ---------
// bladerf_apt_demod.cpp:728-742 in one call: int16 IQ -> scale -> FIR/D -> discriminator
inline std::vector<float> fm_demod_i16(const std::vector<std::complex<int16_t>>& iq, int D,
                                       const std::vector<double>& h, double scale)
{
    FirPlan plan(h, D, SDR_MODE_FM, SDR_IN_I16_IQ, scale, 0.0, SDR_COUNT_FLOOR, (long long)iq.size());   // the block's size: kernel choice
    return plan.run<std::complex<int16_t>, float>(iq.data(), iq.size());
}
// bladerf_am_demod.cpp:291-311 / bladerf_apt_demod.cpp:747-754: FIR/D -> |.|
inline std::vector<float> envelope_i16(const std::vector<std::complex<int16_t>>& iq, int D,
                                       const std::vector<double>& h)
{
    FirPlan plan(h, D, SDR_MODE_ENV, SDR_IN_I16_IQ, 1.0, 0.0, SDR_COUNT_FLOOR, (long long)iq.size());
    return plan.run<std::complex<int16_t>, float>(iq.data(), iq.size());
}
// bladerf_apt_demod.cpp:744-754: frequency_shift -> polyphase_decimate -> abs on a real stream
inline std::vector<float> shift_decimate_envelope(const std::vector<float>& x, double fr, int D,
                                                  const std::vector<double>& h)
{
    FirPlan plan(h, D, SDR_MODE_ENV, SDR_IN_F32_REAL, 1.0, fr, SDR_COUNT_FLOOR, (long long)x.size());   // short blocks: K-split kernel
    return plan.run<float, float>(x.data(), x.size());
}

// bladerf_fm_demod.cpp:288-298 in one launch, on the float block the reference's source hands over:
//   x3 = af::fir(af_lpf, x2);  x4 = x3(dec_seq);  x5 = atan2(imag, real)(x4(1:end) * conjg(x4(0:end-1))) * phasor_scale
// dec_seq = af::seq(0, num_samples, dec_rate) has an inclusive end, hence SDR_COUNT_AF_SEQ (sdr_hip.h).
inline std::vector<float> af_fir_pick_discriminate(const std::vector<std::complex<float>>& x2,
                                                   const std::vector<float>& lpf, int dec_rate, double phasor_scale)
{
    FirPlan plan(std::vector<double>(lpf.begin(), lpf.end()), dec_rate, SDR_MODE_FM, SDR_IN_CF32, phasor_scale, 0.0,
                 SDR_COUNT_AF_SEQ);
    return plan.run<std::complex<float>, float>(x2.data(), x2.size());
}
// the same from the raw int16 block (the scale loop of bladerf_sdr.h:288-291 fused in as well)
inline std::vector<float> af_fir_pick_discriminate_i16(const std::vector<std::complex<int16_t>>& iq,
                                                       const std::vector<float>& lpf, int dec_rate, double phasor_scale)
{
    FirPlan plan(std::vector<double>(lpf.begin(), lpf.end()), dec_rate, SDR_MODE_FM, SDR_IN_I16_IQ, phasor_scale, 0.0,
                 SDR_COUNT_AF_SEQ, (long long)iq.size());
    return plan.run<std::complex<int16_t>, float>(iq.data(), iq.size());
}
// bladerf_am_demod.cpp:294-311: af::fir, x3(dec_seq), af::abs
inline std::vector<float> af_fir_pick_abs(const std::vector<std::complex<float>>& x2, const std::vector<float>& lpf,
                                          int dec_rate)
{
    FirPlan plan(std::vector<double>(lpf.begin(), lpf.end()), dec_rate, SDR_MODE_ENV, SDR_IN_CF32, 1.0, 0.0, SDR_COUNT_AF_SEQ);
    return plan.run<std::complex<float>, float>(x2.data(), x2.size());
}

// bladerf_fm_demod.cpp:301-313: de-emphasis FIR, audio low-pass FIR, pick every dec_audio-th sample,
// normalise to +-1 and convert to unsigned 8-bit PCM.  The two full-rate filters are folded into
// one (their convolution), so only the kept samples are computed -- one real-input FIR/decimate
// launch plus the PCM stage.  count_rule: SDR_COUNT_AF_SEQ picks like x6(seq_audio) (:227,307; the
// seq's indices past the end of x6 are not read), SDR_COUNT_FLOOR keeps floor(n / dec_audio) samples.
inline std::vector<uint8_t> fm_audio_u8(const std::vector<float>& x5, const std::vector<float>& lpf_de,
                                        const std::vector<float>& lpf_a, int dec_audio,
                                        int count_rule = SDR_COUNT_FLOOR)
{
    std::vector<double> hc(lpf_de.size() + lpf_a.size() - 1, 0.0);
    for (size_t i = 0; i < lpf_de.size(); ++i)
        for (size_t j = 0; j < lpf_a.size(); ++j) hc[i + j] += (double)lpf_de[i] * (double)lpf_a[j];
    FirPlan plan(hc, dec_audio, SDR_MODE_REAL, SDR_IN_F32_REAL, 1.0, 0.0, count_rule, (long long)x5.size());
    const size_t m = plan.out_count(x5.size());
    std::vector<uint8_t> pcm(m);
    if (m == 0) return pcm;
    DevMem d_in(x5.size() * sizeof(float)), d_a((m + 1) * sizeof(float)), d_pcm(m);   // d_a: audio + the max scratch
    check(sdr_dev_upload(d_in.get(), x5.data(), x5.size() * sizeof(float)));
    size_t n_out = 0;
    check(sdr_fir_execute_dev(plan.get(), d_in.get(), x5.size(), d_a.get(), m, &n_out, nullptr));
    check(sdr_audio_u8_dev(d_a.as<float>(), m, d_a.as<float>() + m, d_pcm.as<uint8_t>(), nullptr));
    check(sdr_dev_sync());
    check(sdr_dev_download(pcm.data(), d_pcm.get(), m));
    return pcm;
}

}  // namespace sdrhip

// =============================================================================================
// The reference signatures (global namespace, as in the reference's translation units)
// =============================================================================================

// bladerf_apt_demod.cpp:249 -- same argument checks and messages (:254-259)
inline std::vector<std::complex<double>>
polyphase_decimate(const std::vector<std::complex<double>>& x, int32_t decimation_factor,
                   const std::vector<double>& h)
{
    if (decimation_factor <= 0) throw std::invalid_argument("Decimation factor M must be positive.");
    if ((int32_t)h.size() % decimation_factor != 0)
        throw std::invalid_argument("Filter length must be a multiple of M.");
    std::vector<std::complex<float>> xf(x.size());
    for (size_t i = 0; i < x.size(); ++i) xf[i] = std::complex<float>((float)x[i].real(), (float)x[i].imag());
    sdrhip::FirPlan plan(h, decimation_factor, SDR_MODE_IQ, SDR_IN_CF32);
    const std::vector<std::complex<float>> yf = plan.run<std::complex<float>, std::complex<float>>(xf.data(), xf.size());
    std::vector<std::complex<double>> y(yf.size());
    for (size_t i = 0; i < yf.size(); ++i) y[i] = std::complex<double>(yf[i].real(), yf[i].imag());
    return y;
}

// bladerf_apt_demod.cpp:405-406.  On its own the discriminator is a 1-tap FIR (h = 1, D = 1)
// followed by the fused epilogue.
template <typename T>
std::vector<T> polar_discriminator(std::vector<std::complex<T>>& v1, double scale)
{
    std::vector<std::complex<float>> xf(v1.size());
    for (size_t i = 0; i < v1.size(); ++i) xf[i] = std::complex<float>((float)v1[i].real(), (float)v1[i].imag());
    sdrhip::FirPlan plan(std::vector<double>{1.0}, 1, SDR_MODE_FM, SDR_IN_CF32, scale);
    const std::vector<float> d = plan.run<std::complex<float>, float>(xf.data(), xf.size());
    return std::vector<T>(d.begin(), d.end());
}

// bladerf_apt_demod.cpp:476-477 (real or complex data)
namespace sdrhip {
inline std::vector<std::complex<double>> frequency_shift_impl(const float* in, int is_complex, size_t n, double fr)
{
    std::vector<std::complex<double>> res(n);
    if (n == 0) return res;
    const size_t bin = n * (is_complex ? 8 : 4);
    DevMem d_in(bin), d_out(n * 8);
    check(sdr_dev_upload(d_in.get(), in, bin));
    check(sdr_frequency_shift_dev(d_in.as<float>(), is_complex, n, 0, fr, d_out.as<float>(), nullptr));
    std::vector<std::complex<float>> out(n);
    check(sdr_dev_download(out.data(), d_out.get(), n * 8));
    for (size_t i = 0; i < n; ++i) res[i] = std::complex<double>(out[i].real(), out[i].imag());
    return res;
}
}  // namespace sdrhip

template <typename T>
inline std::vector<std::complex<double>> frequency_shift(const std::vector<T>& data, double fr)
{
    std::vector<float> f(data.begin(), data.end());
    return sdrhip::frequency_shift_impl(f.data(), 0, f.size(), fr);
}
template <typename T>
inline std::vector<std::complex<double>> frequency_shift(const std::vector<std::complex<T>>& data, double fr)
{
    std::vector<std::complex<float>> f(data.size());
    for (size_t i = 0; i < data.size(); ++i) f[i] = std::complex<float>((float)data[i].real(), (float)data[i].imag());
    return sdrhip::frequency_shift_impl((const float*)f.data(), 1, f.size(), fr);
}

// bladerf/common/include/sdr_functions.h:63
inline std::vector<std::complex<double>> create_freq_rotation(uint64_t N, double fr)
{
    std::vector<std::complex<double>> res(N);
    if (N == 0) return res;
    sdrhip::DevMem d_out(N * 8);
    sdrhip::check(sdr_frequency_shift_dev(nullptr, 1, N, 0, fr, d_out.as<float>(), nullptr));
    std::vector<std::complex<float>> out(N);
    sdrhip::check(sdr_dev_download(out.data(), d_out.get(), N * 8));
    for (size_t i = 0; i < N; ++i) res[i] = std::complex<double>(out[i].real(), out[i].imag());
    return res;
}

// bladerf_apt_demod.cpp:374-375: fractional-rate pick.  The index sequence is the reference's
// (floor of a double accumulated by repeated addition), built on the host; the gather runs on
// the device.  The reference does not guard an index that lands on N (it reads one past the
// end); here such an index raises std::out_of_range.
template <typename T>
std::vector<T> decimate_vec(std::vector<T>& v1, double rate)
{
    static_assert(sizeof(T) % 4 == 0, "element size must be a multiple of 4 bytes");
    const uint64_t num = (uint64_t)std::ceil(v1.size() / rate);
    std::vector<uint64_t> idx(num);
    double index = 0.0;
    for (uint64_t i = 0; i < num; ++i) {
        idx[i] = (uint64_t)std::floor(index);
        if (idx[i] >= v1.size()) throw std::out_of_range("decimate_vec: index past the end");
        index += rate;
    }
    std::vector<T> res(num);
    if (num == 0) return res;
    sdrhip::DevMem d_in(v1.size() * sizeof(T)), d_idx(num * 8), d_out(num * sizeof(T));
    sdrhip::check(sdr_dev_upload(d_in.get(), v1.data(), v1.size() * sizeof(T)));
    sdrhip::check(sdr_dev_upload(d_idx.get(), idx.data(), num * 8));
    sdrhip::check(sdr_gather_dev(d_in.get(), (int)(sizeof(T) / 4), v1.size(), d_idx.as<uint64_t>(), num, d_out.get(), nullptr));
    sdrhip::check(sdr_dev_download(res.data(), d_out.get(), num * sizeof(T)));
    return res;
}

// bladerf_apt_demod.cpp:428-429: centred "same" FIR on a real vector (real taps).
// res[idx] is the causal FIR output at n = idx + (T-1-h2), h2 = T>>1, from zero pre-history:
// computed on the device.  The last h2 outputs follow the reference's early-truncated tail
// bound (jmx = N - idx, :444) and are evaluated on the host from their (few) terms.
template <typename T, typename U>
std::vector<T> filter_vec(std::vector<T>& v1, std::vector<U>& h)
{
    const uint64_t N = v1.size(), Th = h.size(), h2 = Th >> 1;
    if (Th == 0 || N < Th) throw std::invalid_argument("filter_vec: needs v1.size() >= h.size() > 0");
    std::vector<T> res(N, T(0));
    // device: y[n] = sum h[i] v[n-i] for n < N + Th (v zero-extended on the right)
    std::vector<float> vf(N + Th, 0.0f);
    for (uint64_t i = 0; i < N; ++i) vf[i] = (float)v1[i];
    std::vector<double> hd(h.begin(), h.end());
    sdrhip::FirPlan plan(hd, 1, SDR_MODE_IQ, SDR_IN_F32_REAL);
    const std::vector<std::complex<float>> y = plan.run<float, std::complex<float>>(vf.data(), vf.size());
    const uint64_t shift = Th - 1 - h2;
    for (uint64_t idx = 0; idx + h2 < N; ++idx) res[idx] = (T)y[idx + shift].real();
    for (uint64_t idx = N - h2; idx < N; ++idx) {          // the reference's tail rows
        const uint64_t jmn = (idx >= h2) ? 0 : (h2 - idx), jmx = N - idx;
        uint64_t kdx = (idx > h2) ? idx - h2 : 0;
        T acc = T(0);
        for (uint64_t j = jmn; j <= jmx; ++j, ++kdx) acc += v1[kdx] * h[Th - j - 1];
        res[idx] = acc;
    }
    return res;
}

// DSP::create_fir_filter / DSP::*_window (davemers0160/Common dsp/dsp_windows.h; call sites
// bladerf_fm_demod.cpp:215, bladerf_apt_demod.cpp:571).  PARITY UNPINNED, see sdr_hip.h.
namespace DSP {
inline std::vector<double> rectangular_window(int64_t N) { std::vector<double> w(N); sdrhip::check(sdr_window(SDR_WIN_RECTANGULAR, N, w.data())); return w; }
inline std::vector<double> hann_window(int64_t N) { std::vector<double> w(N); sdrhip::check(sdr_window(SDR_WIN_HANN, N, w.data())); return w; }
inline std::vector<double> hamming_window(int64_t N) { std::vector<double> w(N); sdrhip::check(sdr_window(SDR_WIN_HAMMING, N, w.data())); return w; }
inline std::vector<double> blackman_window(int64_t N) { std::vector<double> w(N); sdrhip::check(sdr_window(SDR_WIN_BLACKMAN, N, w.data())); return w; }
inline std::vector<double> nuttall_window(int64_t N) { std::vector<double> w(N); sdrhip::check(sdr_window(SDR_WIN_NUTTALL, N, w.data())); return w; }
inline std::vector<double> blackman_harris_window(int64_t N) { std::vector<double> w(N); sdrhip::check(sdr_window(SDR_WIN_BLACKMAN_HARRIS, N, w.data())); return w; }

template <typename T>
std::vector<T> create_fir_filter(uint64_t n_taps, double fc, std::vector<double> (*window)(int64_t))
{
    const std::vector<double> w = window((int64_t)n_taps);
    std::vector<T> h(n_taps);
    const double c = ((double)n_taps - 1.0) / 2.0;
    for (uint64_t n = 0; n < n_taps; ++n) {
        const double t = (double)n - c;
        const double s = (t == 0.0) ? fc : std::sin(M_PI * fc * t) / (M_PI * t);
        h[n] = (T)(w[n] * s);
    }
    return h;
}
}  // namespace DSP

#endif  // SDR_HIP_SHIMS_HPP_
