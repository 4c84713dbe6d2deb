// apt.hip -- APT line synchronisation and image assembly.  Host code: the input is
// the 4160 Hz envelope the GPU chain delivers (a few thousand samples per block) and the peak
// search is sequentially dependent, so this stage stays on the CPU, fed by the HIP kernels.
//
// Replaces, per processed block, bladerf/apt_demod/src/bladerf_apt_demod.cpp:756-793
//   x10 += x9;  x11 = (255/delta) * (x10 - min);  x12 = int16(x11 - 128);
//   peaks = get_correlation_peaks(x12, sync_pulse)        (:181-213)
//   for every peak but the last: image row = uint8(x11)[peak, peak + 2080)   (:775-787)
//   add_line_to_image: scroll up one row, new row at the bottom               (:216-246)
//   x10.erase(first max(last_peak - 500, 0) samples)                          (:789-793)
// The reference does this with OpenCV (not in the tree: PARITY UNPINNED); the OpenCV semantics
// restated here: a matrix expression scalar*(A - s) is evaluated as A*alpha + beta in double,
// convertTo rounds half to even (cvRound) and saturates, Mat::dot of two int16 rows sums in double.

#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

#include "sdr_internal.h"

namespace {

// the reference's sync pulse, bladerf_apt_demod.cpp:563
const int16_t kSyncPulse[39] = {-128, -128, -128, -128, 127, 127, -128, -128, 127, 127, -128, -128, 127,
                                127,  -128, -128, 127,  127, -128, -128, 127, 127, -128, -128, 127, 127,
                                -128, -128, 127,  127,  -128, -128, -128, -128, -128, -128, -128, -128, -128};
constexpr long long kMinDistance = 2000;     // :185
constexpr long long kSkip = 500;             // :202 and :791

inline long long round_half_even(double v) { return (long long)std::nearbyint(v); }   // cvRound

}  // namespace

struct sdr_apt_image {
    int width = 0, rows = 0;
    std::vector<double> x10;                 // envelope samples not yet consumed
    std::vector<uint8_t> img;                // rows x width, newest line last
    unsigned long long lines = 0;
};

extern "C" int sdr_apt_image_create(int width, int rows, sdr_apt_image **out)
{
    if (!out || width <= 0 || rows <= 0) { sdr::set_error("bad argument"); return SDR_EINVAL; }
    sdr_apt_image *a = new sdr_apt_image();
    a->width = width;
    a->rows = rows;
    a->img.assign((size_t)width * rows, 0);          // cv::Mat::zeros(700, 2080, CV_8UC1), :580
    *out = a;
    return SDR_OK;
}

extern "C" int sdr_apt_image_destroy(sdr_apt_image *a)
{
    delete a;
    return SDR_OK;
}

extern "C" int sdr_apt_image_push(sdr_apt_image *a, const float *x9, size_t n, int *lines_added)
{
    if (!a || (!x9 && n)) { sdr::set_error("bad argument"); return SDR_EINVAL; }
    if (lines_added) *lines_added = 0;
    a->x10.insert(a->x10.end(), x9, x9 + n);                                   // :756
    const size_t N = a->x10.size();
    const size_t S = sizeof(kSyncPulse) / sizeof(kSyncPulse[0]);
    if (N < S) return SDR_OK;             // the reference's unsigned loop bound would wrap here
    double x_min = a->x10[0], x_max = a->x10[0];                               // :758-761
    for (size_t i = 1; i < N; ++i) {
        if (a->x10[i] < x_min) x_min = a->x10[i];
        if (a->x10[i] > x_max) x_max = a->x10[i];
    }
    const double delta = x_max - x_min;
    if (!(delta > 0.0)) {
        // constant backlog (squelched or zero input): the reference divides by zero here.  No line can
        // be found in it, so keep what a later line could still start in -- the last width + 500
        // samples, the same margin the drop rule below leaves -- and let the rest go: the state stays
        // bounded however long the input stays constant
        const size_t keep = (size_t)a->width + (size_t)kSkip;
        if (N > keep) a->x10.erase(a->x10.begin(), a->x10.begin() + (N - keep));
        return SDR_OK;
    }
    const double alpha = 255.0 / delta, beta = -x_min * alpha;                  // :766
    std::vector<int16_t> x12(N);
    std::vector<uint8_t> x11u(N);
    for (size_t i = 0; i < N; ++i) {
        const double v = a->x10[i] * alpha + beta;
        long long s = round_half_even(v - 128.0);                               // :768
        x12[i] = (int16_t)(s < -32768 ? -32768 : (s > 32767 ? 32767 : s));
        long long u = round_half_even(v);                                       // :773
        x11u[i] = (uint8_t)(u < 0 ? 0 : (u > 255 ? 255 : u));
    }
    // get_correlation_peaks, :181-213
    std::vector<std::pair<long long, double>> peaks(1, std::make_pair(0LL, 0.0));
    for (unsigned long long index = 0; index <= N - S; ++index) {
        double dot = 0.0;
        for (size_t k = 0; k < S; ++k) dot += (double)x12[index + k] * (double)kSyncPulse[k];
        const double corr = dot / (double)S;
        if ((long long)index - peaks.back().first > kMinDistance) {
            peaks.push_back(std::make_pair((long long)index, corr));
            index += kSkip;
        } else if (corr > peaks.back().second) {
            peaks.back() = std::make_pair((long long)index, corr);
        }
    }
    int added = 0;
    for (size_t j = 0; j + 1 < peaks.size(); ++j) {                            // :775-787
        if (peaks[j].first + a->width >= (long long)N) continue;
        if (a->rows > 1) memmove(a->img.data(), a->img.data() + a->width, (size_t)a->width * (a->rows - 1));
        memcpy(a->img.data() + (size_t)a->width * (a->rows - 1), x11u.data() + peaks[j].first, (size_t)a->width);
        ++added;
    }
    a->lines += (unsigned long long)added;
    long long drop = peaks.back().first - kSkip;                               // :789-793
    if (drop < 0) drop = 0;               // the reference's unsigned max() cannot clamp; this does
    a->x10.erase(a->x10.begin(), a->x10.begin() + drop);
    if (lines_added) *lines_added = added;
    return SDR_OK;
}

extern "C" int sdr_apt_image_get(const sdr_apt_image *a, uint8_t *dst, uint64_t *total_lines,
                                 size_t *pending_samples)
{
    if (!a) { sdr::set_error("bad argument"); return SDR_EINVAL; }
    if (dst) memcpy(dst, a->img.data(), a->img.size());
    if (total_lines) *total_lines = a->lines;
    if (pending_samples) *pending_samples = a->x10.size();
    return SDR_OK;
}
