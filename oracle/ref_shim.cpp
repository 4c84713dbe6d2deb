// ref_shim.cpp -- extern "C" wrapper around the reference's own functions (TEST INFRASTRUCTURE).
//
// This file contains no reference code.  oracle/build_ref.sh extracts the function bodies from
// /root/reference into a temporary "ref_extract.inc" and this translation unit includes it, so
// that oracle/_ref/libsdrref.so is the reference's arithmetic, compiled as its CMake would
// (-O2, C++17).  The wrappers below only convert pointers <-> std::vector and call the
// functions in the order the reference's main loop does
// (bladerf/apt_demod/src/bladerf_apt_demod.cpp:728-754).
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <complex>
#include <stdexcept>
#include <algorithm>
#include <iostream>
#include <chrono>

using namespace std;                      // the reference relies on this leaking from Common
using namespace std::complex_literals;    // sdr_functions.h:70 uses 1i

#include "ref_extract.inc"

typedef std::vector<std::complex<double>> cvec;

extern "C" {

// returns 0 ok, 1 if the reference threw std::invalid_argument
int ref_polyphase_decimate(const double* x, size_t n, int D, const double* h, int T, double* y)
{
    cvec xv(n);
    std::memcpy(xv.data(), x, sizeof(double) * 2 * n);
    std::vector<double> hv(h, h + T);
    try {
        cvec out = polyphase_decimate(xv, D, hv);
        std::memcpy(y, out.data(), sizeof(double) * 2 * out.size());
    } catch (const std::invalid_argument&) {
        return 1;
    }
    return 0;
}

void ref_polar_discriminator(const double* v, size_t n, double scale, double* d)
{
    cvec vv(n);
    std::memcpy(vv.data(), v, sizeof(double) * 2 * n);
    std::vector<double> out = polar_discriminator(vv, scale);
    std::memcpy(d, out.data(), sizeof(double) * out.size());
}

void ref_frequency_shift_real(const double* data, size_t n, double fr, double* out)
{
    std::vector<double> dv(data, data + n);
    cvec r = frequency_shift(dv, fr);
    std::memcpy(out, r.data(), sizeof(double) * 2 * n);
}

void ref_frequency_shift_cplx(const double* data, size_t n, double fr, double* out)
{
    cvec dv(n);
    std::memcpy(dv.data(), data, sizeof(double) * 2 * n);
    cvec r = frequency_shift(dv, fr);
    std::memcpy(out, r.data(), sizeof(double) * 2 * n);
}

void ref_create_freq_rotation(uint64_t N, double fr, double* out)
{
    cvec r = create_freq_rotation(N, fr);
    std::memcpy(out, r.data(), sizeof(double) * 2 * N);
}

size_t ref_decimate_vec_real(const double* v, size_t n, double rate, double* res, size_t cap)
{
    std::vector<double> vv;
    vv.reserve(n + 8);          // the reference can index one past the end for some rates;
    vv.assign(v, v + n);        // spare capacity keeps that stray read inside the allocation
    std::vector<double> out = decimate_vec(vv, rate);
    const size_t m = std::min(out.size(), cap);
    std::memcpy(res, out.data(), sizeof(double) * m);
    return out.size();
}

void ref_filter_vec_real(const double* v, size_t n, const double* h, int T, double* res)
{
    std::vector<double> vv(v, v + n), hv(h, h + T);
    std::vector<double> out = filter_vec(vv, hv);
    std::memcpy(res, out.data(), sizeof(double) * n);
}

// The consumer loop of bladerf_apt_demod.cpp:728-742 / :750-754 on one block:
// scale -> polyphase_decimate -> (mode 1) polar_discriminator | (mode 2) abs | (mode 0) complex.
// Taps must already satisfy T % D == 0 (callers zero-pad).  Returns the output count.
size_t ref_chain_i16(const int16_t* iq, size_t n, int D, const double* h, int T, int mode,
                     double scale, double* out)
{
    std::complex<double> cf_scale(1.0 / 2048.0, 0.0);
    cvec cf(n);
    for (size_t j = 0; j < n; ++j)
        cf[j] = cf_scale * std::complex<double>((double)iq[2 * j], (double)iq[2 * j + 1]);
    std::vector<double> hv(h, h + T);
    cvec x4 = polyphase_decimate(cf, D, hv);
    if (mode == 0) {
        std::memcpy(out, x4.data(), sizeof(double) * 2 * x4.size());
        return x4.size();
    }
    if (mode == 1) {
        if (x4.size() < 2) return 0;
        std::vector<double> x6 = polar_discriminator(x4, scale);
        std::memcpy(out, x6.data(), sizeof(double) * x6.size());
        return x6.size();
    }
    for (size_t j = 0; j < x4.size(); ++j) out[j] = std::abs(x4[j]);
    return x4.size();
}

// Full APT block: bladerf_apt_demod.cpp:728-754.
size_t ref_apt_chain_i16(const int16_t* iq, size_t n, int D1, const double* h1, int T1,
                         double disc_scale, double fr, int D2, const double* h2, int T2,
                         double* x9_out, double* x6_out)
{
    std::complex<double> cf_scale(1.0 / 2048.0, 0.0);
    cvec cf(n);
    for (size_t j = 0; j < n; ++j)
        cf[j] = cf_scale * std::complex<double>((double)iq[2 * j], (double)iq[2 * j + 1]);
    std::vector<double> lpf_fm(h1, h1 + T1), lpf_am(h2, h2 + T2);
    cvec x4 = polyphase_decimate(cf, D1, lpf_fm);
    if (x4.size() < 2) return 0;
    std::vector<double> x6 = polar_discriminator(x4, disc_scale);
    if (x6_out) std::memcpy(x6_out, x6.data(), sizeof(double) * x6.size());
    cvec x7 = frequency_shift(x6, fr);
    cvec x8 = polyphase_decimate(x7, D2, lpf_am);
    for (size_t j = 0; j < x8.size(); ++j) x9_out[j] = std::abs(x8[j]);
    return x8.size();
}

// Wall-clock of the FM chain for bench.py's cpu_baseline leg ("reference" kind): seconds for
// `reps` passes over the block, single thread, as the reference runs it.
double ref_time_chain_i16(const int16_t* iq, size_t n, int D, const double* h, int T, int mode,
                          double scale, int reps, double* sink)
{
    std::vector<double> out(n / (size_t)D + 8);
    auto t0 = std::chrono::high_resolution_clock::now();
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        size_t m = ref_chain_i16(iq, n, D, h, T, mode, scale, out.data());
        acc += out[m / 2];
    }
    auto t1 = std::chrono::high_resolution_clock::now();
    if (sink) *sink = acc;
    return std::chrono::duration<double>(t1 - t0).count();
}

}   // extern "C"
