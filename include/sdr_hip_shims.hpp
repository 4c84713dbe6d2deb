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

// ---- per-thread plan cache and device scratch ---------------------------------------------------
// The reference's functions carry no state: every call of polyphase_decimate / af::fir hands over the taps
// again (bladerf_apt_demod.cpp:740,747; bladerf_fm_demod.cpp:291).  Building a plan per call -- tap upload,
// geometry choice, device allocations -- costs more than filtering one 131072-sample block, so the shims
// keep the plans they have built: a small per-thread LRU keyed by everything a plan depends on (the taps'
// values, decimation, mode, input kind, scale, mixer, count rule, the block's size class).  A second call
// with the same filter allocates nothing: the plan, its tap tables and its grow-only staging buffers
// (sdr_fir_execute_host) are reused.  Per thread, because a plan's host staging serves one call at a time;
// the reference calls its DSP from the one consumer thread.
struct PlanKey {
    std::vector<double> taps;
    int decim, mode, input, count_rule, size_class;
    double scale, mix_fr;
    bool operator==(const PlanKey& o) const
    {
        return decim == o.decim && mode == o.mode && input == o.input && count_rule == o.count_rule &&
               size_class == o.size_class && scale == o.scale && mix_fr == o.mix_fr && taps == o.taps;
    }
};

class PlanCache {
public:
    static constexpr size_t kCapacity = 16;
    ~PlanCache() { for (auto& e : entries_) sdr_fir_plan_destroy(e.plan); }
    // size_hint: the samples a call hands over (0 = unknown); plans for block sizes within one power of two share an entry
    sdr_fir_plan* get(const std::vector<double>& taps, int decim, int mode, int input, double scale, double mix_fr,
                      int count_rule, long long size_hint)
    {
        int cls = 0;
        for (long long v = size_hint; v > 1; v >>= 1) ++cls;
        PlanKey key{taps, decim, mode, input, count_rule, size_hint > 0 ? cls : -1, scale, mix_fr};
        for (size_t i = 0; i < entries_.size(); ++i)
            if (entries_[i].key == key) {
                if (i) std::rotate(entries_.begin(), entries_.begin() + i, entries_.begin() + i + 1);   // most recent first
                return entries_.front().plan;
            }
        sdr_fir_desc d{};
        d.input = input; d.mode = mode; d.decim = decim; d.n_taps = (int)taps.size();
        d.taps = taps.data(); d.scale = scale; d.mix_fr = mix_fr; d.count_rule = count_rule;
        d.size_hint = size_hint > 0 ? (1LL << cls) : 0;
        sdr_fir_plan* plan = nullptr;
        check(sdr_fir_plan_create(&d, &plan));
        if (entries_.size() == kCapacity) { sdr_fir_plan_destroy(entries_.back().plan); entries_.pop_back(); }
        entries_.insert(entries_.begin(), Entry{std::move(key), plan});
        return plan;
    }
private:
    struct Entry { PlanKey key; sdr_fir_plan* plan; };
    std::vector<Entry> entries_;
};
inline PlanCache& plan_cache() { static thread_local PlanCache c; return c; }

// host buffers in, host buffers out, through a cached plan
template <class In, class Out>
std::vector<Out> run_cached(const std::vector<double>& taps, int decim, int mode, int input, double scale, double mix_fr,
                            int count_rule, const In* in, size_t n)
{
    sdr_fir_plan* plan = plan_cache().get(taps, decim, mode, input, scale, mix_fr, count_rule, (long long)n);
    const size_t cnt = sdr_fir_out_count(plan, n);
    std::vector<Out> out(std::max<size_t>(cnt, 1));
    size_t n_out = 0;
    check(sdr_fir_execute_host(plan, in, n, out.data(), cnt, &n_out));
    out.resize(n_out);
    return out;
}

// grow-only device scratch of the element-wise shims (frequency_shift, decimate_vec, the audio stage): a few
// buffers per thread, allocated on first use and kept
class Scratch {
public:
    ~Scratch() { for (void* q : p_) if (q) sdr_dev_free(q); }
    void* get(int slot, size_t bytes)
    {
        if (bytes > cap_[slot]) {
            if (p_[slot]) { sdr_dev_free(p_[slot]); p_[slot] = nullptr; cap_[slot] = 0; }
            const size_t want = bytes + bytes / 4;
            check(sdr_dev_alloc(&p_[slot], want));
            cap_[slot] = want;
        }
        return p_[slot];
    }
private:
    void* p_[3] = {nullptr, nullptr, nullptr};
    size_t cap_[3] = {0, 0, 0};
};
inline Scratch& scratch() { static thread_local Scratch s; return s; }

// ---- fused consumer-loop entry points (what the reference's while-loop body becomes) ---------
// bladerf_apt_demod.cpp:728-742 in one call: int16 IQ -> scale -> FIR/D -> discriminator
inline std::vector<float> fm_demod_i16(const std::vector<std::complex<int16_t>>& iq, int D,
                                       const std::vector<double>& h, double scale)
{
    // (the block's size picks the kernel: short blocks stay off the persistent matrix kernel)
    return run_cached<std::complex<int16_t>, float>(h, D, SDR_MODE_FM, SDR_IN_I16_IQ, scale, 0.0, SDR_COUNT_FLOOR, iq.data(), iq.size());
}
// bladerf_am_demod.cpp:291-311 / bladerf_apt_demod.cpp:747-754: FIR/D -> |.|
inline std::vector<float> envelope_i16(const std::vector<std::complex<int16_t>>& iq, int D,
                                       const std::vector<double>& h)
{
    return run_cached<std::complex<int16_t>, float>(h, D, SDR_MODE_ENV, SDR_IN_I16_IQ, 1.0, 0.0, SDR_COUNT_FLOOR, iq.data(), iq.size());
}
// bladerf_apt_demod.cpp:744-754: frequency_shift -> polyphase_decimate -> abs on a real stream
inline std::vector<float> shift_decimate_envelope(const std::vector<float>& x, double fr, int D,
                                                  const std::vector<double>& h)
{
    // (short blocks: the K-split kernel)
    return run_cached<float, float>(h, D, SDR_MODE_ENV, SDR_IN_F32_REAL, 1.0, fr, SDR_COUNT_FLOOR, x.data(), x.size());
}

// bladerf_fm_demod.cpp:288-298 in one launch, on the float block the reference's source hands over:
//   x3 = af::fir(af_lpf, x2);  x4 = x3(dec_seq);  x5 = atan2(imag, real)(x4(1:end) * conjg(x4(0:end-1))) * phasor_scale
// dec_seq = af::seq(0, num_samples, dec_rate) has an inclusive end, hence SDR_COUNT_AF_SEQ (sdr_hip.h).
inline std::vector<float> af_fir_pick_discriminate(const std::vector<std::complex<float>>& x2,
                                                   const std::vector<float>& lpf, int dec_rate, double phasor_scale)
{
    return run_cached<std::complex<float>, float>(std::vector<double>(lpf.begin(), lpf.end()), dec_rate, SDR_MODE_FM, SDR_IN_CF32,
                                                  phasor_scale, 0.0, SDR_COUNT_AF_SEQ, x2.data(), x2.size());
}
// the same from the raw int16 block (the scale loop of bladerf_sdr.h:288-291 fused in as well)
inline std::vector<float> af_fir_pick_discriminate_i16(const std::vector<std::complex<int16_t>>& iq,
                                                       const std::vector<float>& lpf, int dec_rate, double phasor_scale)
{
    return run_cached<std::complex<int16_t>, float>(std::vector<double>(lpf.begin(), lpf.end()), dec_rate, SDR_MODE_FM, SDR_IN_I16_IQ,
                                                    phasor_scale, 0.0, SDR_COUNT_AF_SEQ, iq.data(), iq.size());
}
// bladerf_am_demod.cpp:294-311: af::fir, x3(dec_seq), af::abs
inline std::vector<float> af_fir_pick_abs(const std::vector<std::complex<float>>& x2, const std::vector<float>& lpf,
                                          int dec_rate)
{
    return run_cached<std::complex<float>, float>(std::vector<double>(lpf.begin(), lpf.end()), dec_rate, SDR_MODE_ENV, SDR_IN_CF32,
                                                  1.0, 0.0, SDR_COUNT_AF_SEQ, x2.data(), x2.size());
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
    sdr_fir_plan* plan = plan_cache().get(hc, dec_audio, SDR_MODE_REAL, SDR_IN_F32_REAL, 1.0, 0.0, count_rule, (long long)x5.size());
    const size_t m = sdr_fir_out_count(plan, x5.size());
    std::vector<uint8_t> pcm(m);
    if (m == 0) return pcm;
    void* d_in = scratch().get(0, x5.size() * sizeof(float));
    float* d_a = static_cast<float*>(scratch().get(1, (m + 1) * sizeof(float)));      // audio + the max scratch
    uint8_t* d_pcm = static_cast<uint8_t*>(scratch().get(2, m));
    check(sdr_dev_upload(d_in, x5.data(), x5.size() * sizeof(float)));
    size_t n_out = 0;
    check(sdr_fir_execute_dev(plan, d_in, x5.size(), d_a, m, &n_out, nullptr));
    check(sdr_audio_u8_dev(d_a, m, d_a + m, d_pcm, nullptr));
    check(sdr_dev_sync());
    check(sdr_dev_download(pcm.data(), d_pcm, m));
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
    const std::vector<std::complex<float>> yf = sdrhip::run_cached<std::complex<float>, std::complex<float>>(
        h, decimation_factor, SDR_MODE_IQ, SDR_IN_CF32, 1.0, 0.0, SDR_COUNT_FLOOR, xf.data(), xf.size());
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
    const std::vector<float> d = sdrhip::run_cached<std::complex<float>, float>(std::vector<double>{1.0}, 1, SDR_MODE_FM, SDR_IN_CF32, scale,
                                                                                0.0, SDR_COUNT_FLOOR, xf.data(), xf.size());
    return std::vector<T>(d.begin(), d.end());
}

// bladerf_apt_demod.cpp:476-477 (real or complex data)
namespace sdrhip {
inline std::vector<std::complex<double>> frequency_shift_impl(const float* in, int is_complex, size_t n, double fr)
{
    std::vector<std::complex<double>> res(n);
    if (n == 0) return res;
    const size_t bin = n * (is_complex ? 8 : 4);
    float* d_in = static_cast<float*>(scratch().get(0, bin));
    float* d_out = static_cast<float*>(scratch().get(1, n * 8));
    check(sdr_dev_upload(d_in, in, bin));
    check(sdr_frequency_shift_dev(d_in, is_complex, n, 0, fr, d_out, nullptr));
    std::vector<std::complex<float>> out(n);
    check(sdr_dev_download(out.data(), d_out, n * 8));
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
    float* d_out = static_cast<float*>(sdrhip::scratch().get(1, N * 8));
    sdrhip::check(sdr_frequency_shift_dev(nullptr, 1, N, 0, fr, d_out, nullptr));
    std::vector<std::complex<float>> out(N);
    sdrhip::check(sdr_dev_download(out.data(), d_out, N * 8));
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
    void* d_in = sdrhip::scratch().get(0, v1.size() * sizeof(T));
    uint64_t* d_idx = static_cast<uint64_t*>(sdrhip::scratch().get(1, num * 8));
    void* d_out = sdrhip::scratch().get(2, num * sizeof(T));
    sdrhip::check(sdr_dev_upload(d_in, v1.data(), v1.size() * sizeof(T)));
    sdrhip::check(sdr_dev_upload(d_idx, idx.data(), num * 8));
    sdrhip::check(sdr_gather_dev(d_in, (int)(sizeof(T) / 4), v1.size(), d_idx, num, d_out, nullptr));
    sdrhip::check(sdr_dev_download(res.data(), d_out, num * sizeof(T)));
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
    const std::vector<std::complex<float>> y = sdrhip::run_cached<float, std::complex<float>>(hd, 1, SDR_MODE_IQ, SDR_IN_F32_REAL, 1.0, 0.0,
                                                                                              SDR_COUNT_FLOOR, vf.data(), vf.size());
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
    // the designer lives in the library (sdr_create_fir_filter, one definition): the window functor only
    // names the window -- the reference passes &DSP::hann_window etc.
    int id = -1;
    if (window == &rectangular_window) id = SDR_WIN_RECTANGULAR;
    else if (window == &hann_window) id = SDR_WIN_HANN;
    else if (window == &hamming_window) id = SDR_WIN_HAMMING;
    else if (window == &blackman_window) id = SDR_WIN_BLACKMAN;
    else if (window == &nuttall_window) id = SDR_WIN_NUTTALL;
    else if (window == &blackman_harris_window) id = SDR_WIN_BLACKMAN_HARRIS;
    std::vector<double> hd(n_taps);
    if (id >= 0) {
        sdrhip::check(sdr_create_fir_filter(n_taps, fc, id, hd.data()));
    } else {
        // a window of the caller's own: the library's prototype (rectangular window) times the caller's window
        sdrhip::check(sdr_create_fir_filter(n_taps, fc, SDR_WIN_RECTANGULAR, hd.data()));
        const std::vector<double> w = window((int64_t)n_taps);
        for (uint64_t n = 0; n < n_taps; ++n) hd[n] *= w[n];
    }
    return std::vector<T>(hd.begin(), hd.end());
}
}  // namespace DSP

#endif  // SDR_HIP_SHIMS_HPP_
