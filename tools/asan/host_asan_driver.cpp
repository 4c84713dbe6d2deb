// Host-side AddressSanitizer stress of libsdrhip: the library's translation units are compiled with
// -Xarch_host -fsanitize=address and linked into this one executable (no preload needed), which
// creates, runs and destroys plans over a grid of shapes, streams blocks and transforms frames.
// It checks no numerics (the parity tests do); it is here for the host code around the kernels:
// table construction, geometry search, staging buffers, stream bookkeeping.
//   build + run: tools/asan/run_host_asan.sh   (on a GPU box)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <unistd.h>

#include "sdr_hip.h"

static unsigned rng_state = 12345u;
static unsigned rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    if (sdr_init(0) != SDR_OK) { std::printf("no device: %s\n", sdr_last_error()); return 2; }
    const int Ts[] = {1, 2, 7, 64, 65, 101, 199, 300, 511, 640, 2047, 16384};
    const int Ds[] = {1, 2, 3, 5, 8, 10, 15, 16, 20, 25, 33, 41};
    long plans = 0, refused = 0, runs = 0;
    for (int T : Ts)
        for (int D : Ds)
            for (int input = 0; input <= 4; ++input)
                for (int mode = 0; mode <= 3; ++mode)
                    for (int pipe = 0; pipe <= 7; ++pipe) {
                        if (pipe != 0 && (rnd() & 3)) continue;          // thin out the explicit choices
                        std::vector<double> h(T);
                        for (int i = 0; i < T; ++i) h[i] = std::sin(0.37 * i + 0.1) / (1.0 + 0.01 * i);
                        sdr_fir_desc d{};
                        d.input = input; d.mode = mode; d.decim = D; d.n_taps = T; d.taps = h.data();
                        d.scale = 0.159; d.mix_fr = (rnd() & 1) ? 0.0 : 0.0371; d.pipeline = pipe;
                        sdr_fir_plan *p = nullptr;
                        if (sdr_fir_plan_create(&d, &p) != SDR_OK) { ++refused; continue; }
                        ++plans;
                        const size_t ns[] = {0, 1, 3, (size_t)D, (size_t)(T + D), 5000, (size_t)(rnd() % 60000)};
                        const size_t esz = input == SDR_IN_CF32 ? 8 : (input == SDR_IN_I8_IQ ? 2 : 4);
                        for (size_t n : ns) {
                            std::vector<unsigned char> in(n * esz + 16);
                            for (auto &b : in) b = (unsigned char)(rnd() & (input == SDR_IN_CF32 || input == SDR_IN_F32_REAL ? 0x3f : 0xff));
                            const size_t cnt = sdr_fir_out_count(p, n);
                            std::vector<float> out(2 * cnt + 4);
                            size_t n_out = 0;
                            const int rc = sdr_fir_execute_host(p, in.data(), n, out.data(), cnt, &n_out);
                            if (rc != SDR_OK && !(pipe == 5 && n < 8))  /* pinned pipeline 5 refuses streams under one 16-byte load */ { std::printf("execute failed T=%d D=%d in=%d mode=%d pipe=%d n=%zu: %s\n", T, D, input, mode, pipe, n, sdr_last_error()); return 1; }
                            ++runs;
                        }
                        if ((rnd() & 7) == 0) {                          // block hand-off on this plan
                            sdr_fir_stream *st = nullptr;
                            const size_t mb = 1 + rnd() % 5000;
                            if (sdr_fir_stream_create(p, mb, &st) == SDR_OK) {
                                std::vector<unsigned char> blk(mb * esz, 1);
                                std::vector<float> out(2 * (mb / D + 4200));
                                int in_flight = 0;
                                for (int b = 0; b < 9; ++b) {
                                    const size_t n = 1 + rnd() % mb;
                                    if (sdr_fir_stream_push(st, blk.data(), n) != SDR_OK) { std::printf("push failed: %s\n", sdr_last_error()); return 1; }
                                    if (++in_flight == 2) { size_t no; if (sdr_fir_stream_pop(st, out.data(), out.size() / 2, &no) != SDR_OK) { std::printf("pop failed: %s\n", sdr_last_error()); return 1; } --in_flight; }
                                }
                                while (in_flight--) { size_t no; sdr_fir_stream_pop(st, out.data(), out.size() / 2, &no); }
                                sdr_fir_stream_destroy(st);
                            }
                        }
                        sdr_fir_plan_destroy(p);
                    }
    const int Ns[] = {64, 128, 256, 512, 1024, 2048, 4096, 8192, 131072};
    for (int N : Ns)
        for (int rep = 0; rep < 4; ++rep) {
            const int hop = rep == 0 ? N / 2 : 1 + (int)(rnd() % (2 * N));
            std::vector<double> w(N, 0.5);
            sdr_fft_plan *fp = nullptr;
            if (sdr_fft_plan_create(N, hop, rep & 1 ? w.data() : nullptr, &fp) != SDR_OK) { std::printf("fft plan failed: %s\n", sdr_last_error()); return 1; }
            const size_t n = (size_t)N + (size_t)hop * (rnd() % 40) + rnd() % hop;
            std::vector<int16_t> iq(2 * n);
            for (auto &v : iq) v = (int16_t)(rnd() & 0xfff) - 2048;
            const size_t fr = sdr_fft_frame_count(fp, n);
            std::vector<float> out(fr * N + 1);
            size_t nf = 0;
            if (sdr_fft_execute_i16_host(fp, iq.data(), n, out.data(), fr, &nf) != SDR_OK) { std::printf("fft failed: %s\n", sdr_last_error()); return 1; }
            sdr_fft_plan_destroy(fp);
            ++runs;
        }
    // APT image: host code
    sdr_apt_image *img = nullptr;
    if (sdr_apt_image_create(2080, 40, &img) == SDR_OK) {
        std::vector<float> x(50000);
        for (size_t i = 0; i < x.size(); ++i) x[i] = (float)((i % 2080) < 39 ? ((i >> 1) & 1) : 0.3 + 0.001 * (i % 97));
        int lines = 0;
        for (int k = 0; k < 5; ++k) sdr_apt_image_push(img, x.data(), x.size() - 1000 * k, &lines);
        std::vector<uint8_t> px(2080 * 40);
        sdr_apt_image_get(img, px.data(), nullptr, nullptr);
        sdr_apt_image_destroy(img);
    }
    std::printf("host asan driver: %ld plans, %ld refused shapes, %ld runs, no findings\n", plans, refused, runs);
    // leave without the runtime's exit-time teardown: the ROCm ASan runtime trips a CHECK of its own
    // there (sanitizer_allocator_device.h, "dev_runtime_unloaded") that has nothing to do with this code
    _exit(0);
}
