// apt_chain_example.cpp -- the reference's APT consumer loop, recompiled against libsdrhip.
//
// The body of main() below is the block-processing part of
// bladerf/apt_demod/src/bladerf_apt_demod.cpp:728-754 with the reference's own function names;
// only the #include changed (sdr_hip_shims.hpp instead of the in-file definitions and the
// Common DSP headers), and the capture thread is replaced by a file replay like the reference's
// own temp_get_data() (:92-148).  A second pass runs the fused entry points, which is what a
// maintainer would switch the loop to.
//
//   apt_chain_example <in.sc16> <out_prefix>
// writes <out_prefix>.x6.f64 / .x9.f64 (shim path, double like the reference) and
// <out_prefix>.x6f.f32 / .x9f.f32 (fused path), plus .dec.f64 / .filt.f64 from decimate_vec and
// filter_vec on x6.
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>

#include "sdr_hip_shims.hpp"

template <class T>
static void dump(const std::string& path, const std::vector<T>& v)
{
    std::ofstream f(path, std::ios::binary);
    f.write((const char*)v.data(), (std::streamsize)(v.size() * sizeof(T)));
}

int main(int argc, char** argv)
{
    if (argc < 3) { std::cerr << "usage: " << argv[0] << " <in.sc16> <out_prefix>" << std::endl; return 2; }
    try {
        sdrhip::check(sdr_init(0));

        // interleaved int16 I,Q as written by rx_record (bladerf_rec_main_v1.cpp:161-171)
        std::ifstream in(argv[1], std::ios::binary | std::ios::ate);
        if (!in) { std::cerr << "cannot open " << argv[1] << std::endl; return 2; }
        const size_t bytes = (size_t)in.tellg();
        in.seekg(0);
        std::vector<std::complex<int16_t>> samples(bytes / 4);
        in.read((char*)samples.data(), (std::streamsize)(samples.size() * 4));
        const uint64_t block_size = samples.size();

        // bladerf_apt_demod.cpp:534-574
        uint64_t fm_taps = 200, audio_taps = 195;
        int64_t rf_decimation_factor = 10, audio_decimation_factor = 15;
        double desired_rf_sample_rate = 62400.0;
        int64_t fc_fm = 20800, fc_am = 2400, am_offset = 2400;
        std::complex<double> cf_scale(1.0 / 2048.0, 0.0);
        double phasor_scale = 1.0 / (2.0 * M_PI);
        std::vector<double> lpf_fm = DSP::create_fir_filter<double>(fm_taps, fc_fm / desired_rf_sample_rate, &DSP::hann_window);
        std::vector<double> lpf_am = DSP::create_fir_filter<double>(audio_taps, (fc_am) / (double)desired_rf_sample_rate, &DSP::hann_window);

        auto t0 = std::chrono::high_resolution_clock::now();
        // ---- bladerf_apt_demod.cpp:728-754, unchanged ---------------------------------------
        std::vector<std::complex<double>> cf_samples(block_size);
        for (uint64_t jdx = 0; jdx < block_size; ++jdx)
            cf_samples[jdx] = cf_scale * std::complex<double>(std::real(samples[jdx]), std::imag(samples[jdx]));
        std::vector<std::complex<double>> x4 = polyphase_decimate(cf_samples, rf_decimation_factor, lpf_fm);
        std::vector<double> x6 = polar_discriminator(x4, phasor_scale);
        std::vector<std::complex<double>> x7 = frequency_shift(x6, (double)am_offset / (double)desired_rf_sample_rate);
        std::vector<std::complex<double>> x8 = polyphase_decimate(x7, audio_decimation_factor, lpf_am);
        std::vector<double> x9(x8.size());
        for (uint64_t jdx = 0; jdx < x8.size(); ++jdx) x9[jdx] = std::abs(x8[jdx]);
        // -------------------------------------------------------------------------------------
        auto t1 = std::chrono::high_resolution_clock::now();

        // the fused form of the same loop body: two launches, no intermediate host vectors
        std::vector<float> x6f = sdrhip::fm_demod_i16(samples, (int)rf_decimation_factor, lpf_fm, phasor_scale);
        std::vector<float> x9f = sdrhip::shift_decimate_envelope(x6f, (double)am_offset / desired_rf_sample_rate,
                                                                 (int)audio_decimation_factor, lpf_am);
        auto t2 = std::chrono::high_resolution_clock::now();

        // the two helpers the reference defines but does not call in its live chain
        std::vector<double> dec = decimate_vec(x6, 2.5);
        std::vector<double> h7 = {0.1, -0.2, 0.3, 0.4, 0.3, -0.2, 0.1};
        std::vector<double> filt = filter_vec(x6, h7);

        // the FM receiver's audio stage (bladerf_fm_demod.cpp:301-313) on the discriminator output:
        // 64-tap rectangular "de-emphasis" design, 101-tap Hann audio low-pass, pick, normalise, u8
        std::vector<float> lpf_de = DSP::create_fir_filter<float>(64, 1.0 / (62400.0 * 75e-6), &DSP::rectangular_window);
        std::vector<float> lpf_a = DSP::create_fir_filter<float>(101, (10000.0 / 2.0) / 62400.0, &DSP::hann_window);
        std::vector<uint8_t> pcm = sdrhip::fm_audio_u8(x6f, lpf_de, lpf_a, 6);
        dump(std::string(argv[2]) + ".pcm.u8", pcm);
        dump(std::string(argv[2]) + ".lpf_de.f32", lpf_de);
        dump(std::string(argv[2]) + ".lpf_a.f32", lpf_a);

        // argument checks behave like the reference (bladerf_apt_demod.cpp:254-259)
        bool threw = false;
        try { std::vector<double> bad(64, 1.0); polyphase_decimate(cf_samples, 10, bad); }
        catch (const std::invalid_argument&) { threw = true; }
        if (!threw) { std::cerr << "polyphase_decimate accepted taps % D != 0" << std::endl; return 1; }

        const std::string pfx = argv[2];
        dump(pfx + ".x6.f64", x6);
        dump(pfx + ".x9.f64", x9);
        dump(pfx + ".x6f.f32", x6f);
        dump(pfx + ".x9f.f32", x9f);
        dump(pfx + ".dec.f64", dec);
        dump(pfx + ".filt.f64", filt);
        std::cout << "block " << block_size << " samples: shim path "
                  << std::chrono::duration<double, std::milli>(t1 - t0).count() << " ms, fused path "
                  << std::chrono::duration<double, std::milli>(t2 - t1).count() << " ms, x9 " << x9.size()
                  << " samples" << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
