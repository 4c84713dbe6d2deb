// block_latency.cpp -- what one capture block costs through the drop-in boundary, PCIe included.
//
// The reference's receivers process one block per loop iteration: 131072 samples in the FM receiver
// (bladerf_sdr.h:79, bladerf_fm_demod.cpp:281-313: 101 taps, decimate by 5) and sample_rate * 2 s =
// 1 248 000 samples in the APT receiver (bladerf_apt_demod.cpp:531,709,728-754: 200 taps / 10 + discriminator,
// +2400/62400 shift, 195 taps / 15, envelope).  This program times, per block and end to end (host vector
// in, host vector out):
//   shim          the Level-1 shims of sdr_hip_shims.hpp (per-thread plan cache: a steady-state call builds
//                 and allocates nothing)
//   plan_per_call the same call with a fresh plan every time (what the shims did before they cached plans)
//   stream        sdr_fir_stream_push / _pop (pinned staging, history carried across blocks), copying push
//   stream_zc     the same with the producer filling the slot's pinned buffer (sdr_fir_stream_next_buffer)
// and prints one JSON object per line.  tools/bench_blocks.py adds the reference's CPU time for the same blocks.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>

#include "sdr_hip_shims.hpp"

using clk = std::chrono::steady_clock;

static std::vector<std::complex<int16_t>> synth(size_t n, double fs, double ftone, double dev)
{
    std::vector<std::complex<int16_t>> iq(n);
    for (size_t i = 0; i < n; ++i) {
        const double t = (double)i / fs;
        const double ph = (dev / ftone) * std::sin(2.0 * M_PI * ftone * t);
        iq[i] = std::complex<int16_t>((int16_t)std::lrint(2000.0 * std::cos(ph)), (int16_t)std::lrint(2000.0 * std::sin(ph)));
    }
    return iq;
}

struct Stat { double first_us, median_us, min_us; unsigned long long allocs_steady; };

static Stat time_calls(const std::function<void()>& fn, int reps)
{
    Stat s{};
    auto t0 = clk::now();
    fn();
    s.first_us = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
    for (int i = 0; i < 5; ++i) fn();
    const unsigned long long a0 = sdr_alloc_count();
    std::vector<double> us(reps);
    for (int i = 0; i < reps; ++i) {
        t0 = clk::now();
        fn();
        us[i] = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
    }
    s.allocs_steady = sdr_alloc_count() - a0;
    std::sort(us.begin(), us.end());
    s.median_us = us[reps / 2];
    s.min_us = us[0];
    return s;
}

static void emit(const char* chain, const char* path, size_t n, const Stat& s)
{
    std::printf("{\"chain\": \"%s\", \"path\": \"%s\", \"block_samples\": %zu, \"first_call_us\": %.1f, \"median_us\": %.1f, "
                "\"min_us\": %.1f, \"Msamples_per_s\": %.1f, \"allocations_in_steady_state\": %llu}\n",
                chain, path, n, s.first_us, s.median_us, s.min_us, (double)n / s.median_us, s.allocs_steady);
    std::fflush(stdout);
}

static void stream_paths(const char* chain, const std::vector<std::complex<int16_t>>& iq, const std::vector<double>& h, int D,
                         double scale, int reps)
{
    const size_t n = iq.size();
    sdrhip::FirPlan plan(h, D, SDR_MODE_FM, SDR_IN_I16_IQ, scale, 0.0, SDR_COUNT_FLOOR, (long long)n);
    sdr_fir_stream* st = nullptr;
    sdrhip::check(sdr_fir_stream_create(plan.get(), n, &st));
    std::vector<float> out(n / D + 16);
    size_t n_out = 0;
    emit(chain, "stream", n, time_calls([&] {
        sdrhip::check(sdr_fir_stream_push(st, iq.data(), n));
        sdrhip::check(sdr_fir_stream_pop(st, out.data(), out.size(), &n_out));
    }, reps));
    emit(chain, "stream_zc", n, time_calls([&] {
        void* buf = nullptr;
        sdrhip::check(sdr_fir_stream_next_buffer(st, &buf));       // (a live source writes here; the buffer keeps the last block)
        sdrhip::check(sdr_fir_stream_push(st, buf, n));
        sdrhip::check(sdr_fir_stream_pop(st, out.data(), out.size(), &n_out));
    }, reps));
    // two blocks in flight, as the reference's capture thread and consumer overlap: time per block at steady state
    sdrhip::check(sdr_fir_stream_push(st, iq.data(), n));
    emit(chain, "stream_pipelined", n, time_calls([&] {
        sdrhip::check(sdr_fir_stream_push(st, iq.data(), n));
        sdrhip::check(sdr_fir_stream_pop(st, out.data(), out.size(), &n_out));
    }, reps));
    sdrhip::check(sdr_fir_stream_pop(st, out.data(), out.size(), &n_out));
    sdr_fir_stream_destroy(st);
}

int main()
{
    try {
        sdrhip::check(sdr_init(0));
        const int reps = 60;
        // ---- FM receiver block (bladerf_fm_demod.cpp:124-141, 206-233: test_case 1)
        {
            const size_t n = 131072;
            const auto iq = synth(n, 1.0e6, 1000.0, 60000.0);
            const std::vector<float> lpf = DSP::create_fir_filter<float>(101, (200000 / 2.0) / 1.0e6, &DSP::hann_window);
            const std::vector<double> lpfd(lpf.begin(), lpf.end());
            const double scale = 1.0 / (2.0 * M_PI);
            std::vector<float> x5;
            emit("fm 131072 x (101 taps / 5 + discriminator)", "shim", n,
                 time_calls([&] { x5 = sdrhip::af_fir_pick_discriminate_i16(iq, lpf, 5, scale); }, reps));
            emit("fm 131072 x (101 taps / 5 + discriminator)", "plan_per_call", n, time_calls([&] {
                sdrhip::FirPlan plan(lpfd, 5, SDR_MODE_FM, SDR_IN_I16_IQ, scale, 0.0, SDR_COUNT_AF_SEQ, (long long)n);
                x5 = plan.run<std::complex<int16_t>, float>(iq.data(), n);
            }, reps));
            // the reference's own hand-over: the float block its source delivers (bladerf_fm_demod.cpp:288)
            std::vector<std::complex<float>> cf(n);
            for (size_t i = 0; i < n; ++i) cf[i] = std::complex<float>(iq[i].real(), iq[i].imag()) * (1.0f / 2048.0f);
            emit("fm 131072 x (101 taps / 5 + discriminator), cf32 block", "shim", n,
                 time_calls([&] { x5 = sdrhip::af_fir_pick_discriminate(cf, lpf, 5, scale); }, reps));
            stream_paths("fm 131072 x (101 taps / 5 + discriminator)", iq, lpfd, 5, scale, reps);
        }
        // ---- APT receiver block (bladerf_apt_demod.cpp:521-546, 728-754)
        {
            const size_t n = 1248000;
            const auto iq = synth(n, 624000.0, 2400.0, 17000.0);
            const std::vector<double> lpf_fm = DSP::create_fir_filter<double>(200, 20800.0 / 62400.0 / 5.0, &DSP::hann_window);
            const std::vector<double> lpf_am = DSP::create_fir_filter<double>(195, 2400.0 / 62400.0, &DSP::hann_window);
            const double scale = 1.0 / (2.0 * M_PI), fr = 2400.0 / 62400.0;
            std::vector<float> x6, x9;
            emit("apt 1248000 x (200 taps / 10 + discriminator -> shift -> 195 taps / 15 -> abs)", "shim", n, time_calls([&] {
                x6 = sdrhip::fm_demod_i16(iq, 10, lpf_fm, scale);
                x9 = sdrhip::shift_decimate_envelope(x6, fr, 15, lpf_am);
            }, reps));
            emit("apt 1248000 x (200 taps / 10 + discriminator -> shift -> 195 taps / 15 -> abs)", "plan_per_call", n, time_calls([&] {
                sdrhip::FirPlan p1(lpf_fm, 10, SDR_MODE_FM, SDR_IN_I16_IQ, scale, 0.0, SDR_COUNT_FLOOR, (long long)n);
                x6 = p1.run<std::complex<int16_t>, float>(iq.data(), n);
                sdrhip::FirPlan p2(lpf_am, 15, SDR_MODE_ENV, SDR_IN_F32_REAL, 1.0, fr, SDR_COUNT_FLOOR, (long long)x6.size());
                x9 = p2.run<float, float>(x6.data(), x6.size());
            }, reps));
            stream_paths("apt stage 1 only: 1248000 x (200 taps / 10 + discriminator)", iq, lpf_fm, 10, scale, reps);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "block_latency: %s\n", e.what());
        return 1;
    }
    return 0;
}
