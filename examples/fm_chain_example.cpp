// fm_chain_example.cpp -- the reference's FM receiver loop, recompiled against libsdrhip.
//
// Set-up and loop follow bladerf/rx_fm_demod/src/bladerf_fm_demod.cpp (test_case 1, :133-141):
//   :62-79    SDR_BASE::build()            -> a sdrhip::REPLAY_SDR over a recording instead of a BLADERF_SDR
//   :206-233  dec_rate, filters, phasor_scale, dec_audio
//   :279-284  sdr->start(cf_samples); while (...) { sdr->wait_for_samples(); ...
//   :288-298  af::fir, x3(dec_seq), polar discriminator      -> sdrhip::af_fir_pick_discriminate (one launch)
//   :301-313  de-emphasis FIR, audio FIR, x6(seq_audio), normalise, u8 -> sdrhip::fm_audio_u8 (AF seq count)
//   :335      sdr->stop()
// ArrayFire's arithmetic is replaced by the HIP kernels; the WinMM playback (:255-262,:315-321) by a file.
//
//   fm_chain_example <in.sc16> <out_prefix> [block_samples]
// writes <out_prefix>.x5.f32 (all blocks' discriminator outputs), .pcm.u8 (all blocks' audio),
// .x5raw.f32 (the same blocks through the fused int16 entry point), .single.f32 (start_single's block),
// .lpf.f32 / .lpf_de.f32 / .lpf_a.f32 (the filters used) and prints the block count.
#include <cstdio>
#include <fstream>
#include <iostream>

#include "sdr_hip_source.hpp"

static std::string g_recording;
static uint64_t g_block = 4096 * 32;

// bladerf_fm_demod.cpp:62-79
std::unique_ptr<SDR_BASE> SDR_BASE::build()
{
    std::unique_ptr<sdrhip::REPLAY_SDR> dev(new sdrhip::REPLAY_SDR(
        std::unique_ptr<sdrhip::BlockSource>(new sdrhip::FileSource(g_recording, 1)), g_block));
    dev->set_rx_frequency(104000000);
    dev->set_rx_samplerate(1000000);
    dev->set_rx_gain(66, 1);
    dev->set_rx_bandwidth(1000000);
    return std::unique_ptr<SDR_BASE>(dev.release());
}

template <class T>
static void dump(const std::string& path, const std::vector<T>& v)
{
    std::ofstream f(path, std::ios::binary);
    f.write((const char*)v.data(), (std::streamsize)(v.size() * sizeof(T)));
}

int main(int argc, char** argv)
{
    if (argc < 3) { std::cerr << "usage: " << argv[0] << " <in.sc16> <out_prefix> [block_samples]" << std::endl; return 2; }
    g_recording = argv[1];
    const std::string pfx = argv[2];
    if (argc > 3) g_block = std::stoull(argv[3]);
    try {
        sdrhip::check(sdr_init(0));

        // bladerf_fm_demod.cpp:133-141 (test_case 1) and :206-233
        uint64_t fs = 1000000, rx_freq = 103700000;
        int64_t channel_bw = 200000, audio_freq = 44100;
        uint64_t n_taps = 101;

        std::unique_ptr<SDR_BASE> sdr = SDR_BASE::build();
        sdr->set_rx_frequency(rx_freq);
        sdr->set_rx_samplerate(fs);
        sdr->set_rx_gain(66, 1);
        sdr->set_rx_bandwidth((uint32_t)fs);
        const uint64_t num_samples = sdr->num_samples;

        int64_t dec_rate = (int64_t)(fs / (float)channel_bw);
        float fs_d = fs / (float)dec_rate;
        std::vector<float> lpf = DSP::create_fir_filter<float>(n_taps, (channel_bw / 2.0) / (float)fs, &DSP::hann_window);
        int64_t dec_audio = (int64_t)(fs_d / (float)audio_freq);
        float phasor_scale = 1 / ((2 * M_PI) / (fs_d / channel_bw));
        std::vector<float> lpf_de = DSP::create_fir_filter<float>(64, 1.0f / (float)(fs_d * 75e-6), &DSP::rectangular_window);
        std::vector<float> lpf_a = DSP::create_fir_filter<float>(n_taps, (audio_freq / 2.0) / (float)fs_d, &DSP::hann_window);

        std::vector<float> all_x5, all_x5raw;
        std::vector<uint8_t> all_pcm;

        // the one-shot form first (commented alternative at :283): one block, converted, synchronously
        std::vector<std::complex<float>> single;
        sdr->start_single(single, (uint32_t)num_samples);
        dump(pfx + ".single.f32", single);

        std::vector<std::complex<float>> cf_samples(num_samples);
        sdr->start(cf_samples);                                                   // :279
        sdrhip::REPLAY_SDR* replay = dynamic_cast<sdrhip::REPLAY_SDR*>(sdr.get());
        uint64_t blocks = 0;
        while (1) {                                                               // :281
            try {
                sdr->wait_for_samples();                                          // :284
            } catch (const sdrhip::REPLAY_SDR::end_of_stream&) {
                break;                                                            // a recording ends; a radio does not
            }
            if (cf_samples.empty()) break;
            // :288-298
            std::vector<float> x5 = sdrhip::af_fir_pick_discriminate(cf_samples, lpf, (int)dec_rate, phasor_scale);
            // :301-313
            std::vector<uint8_t> x7 = sdrhip::fm_audio_u8(x5, lpf_de, lpf_a, (int)dec_audio, SDR_COUNT_AF_SEQ);
            // what a maintainer would switch to: the raw int16 block through one fused launch
            std::vector<float> x5raw = sdrhip::af_fir_pick_discriminate_i16(replay->raw_block(), lpf, (int)dec_rate, phasor_scale);
            all_x5.insert(all_x5.end(), x5.begin(), x5.end());
            all_x5raw.insert(all_x5raw.end(), x5raw.begin(), x5raw.end());
            all_pcm.insert(all_pcm.end(), x7.begin(), x7.end());
            ++blocks;
        }
        sdr->stop();                                                              // :335

        dump(pfx + ".x5.f32", all_x5);
        dump(pfx + ".x5raw.f32", all_x5raw);
        dump(pfx + ".pcm.u8", all_pcm);
        dump(pfx + ".lpf.f32", lpf);
        dump(pfx + ".lpf_de.f32", lpf_de);
        dump(pfx + ".lpf_a.f32", lpf_a);
        std::cout << "blocks " << blocks << " block_samples " << num_samples << " dec_rate " << dec_rate << " dec_audio "
                  << dec_audio << " phasor_scale " << phasor_scale << " x5 " << all_x5.size() << " pcm " << all_pcm.size()
                  << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
