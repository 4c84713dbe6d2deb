// sdr_hip_source.hpp -- the reference's sample-source interface (SDR_BASE) over libsdrhip, fed by a
// file or a synthetic signal instead of a radio.
//
// Replaces, for consumer loops written against it (bladerf/rx_fm_demod/src/bladerf_fm_demod.cpp:279-284,
// bladerf/rx_am_demod/src/bladerf_am_demod.cpp:282-287):
//   class SDR_BASE                                   bladerf/common/include/sdr_base.h:11-59
//     start(cf_samples) / start_single(cf_samples, n) / stop() / wait_for_samples()      :42-55
//   BLADERF_SDR::get_continuous_samples / start / start_single / stop
//                                                    bladerf/common/include/bladerf_sdr.h:270-346
// SDR_BASE below has the reference's public members and virtuals, so a reference main() that holds a
// std::unique_ptr<SDR_BASE> compiles against it unchanged; sdrhip::REPLAY_SDR is the implementation
// to return from SDR_BASE::build() when no bladeRF is attached (the reference's own stand-in for that
// is the file replay temp_get_data, bladerf/apt_demod/src/bladerf_apt_demod.cpp:92-148).
//
// What the producer thread does per block is the reference's (bladerf_sdr.h:279-296): take
// num_samples int16 IQ samples, convert them to complex<float> * (1/2048), publish them in the
// caller's cf_samples under sdr_mtx, set `processed`, notify.  Here the conversion runs on the GPU
// (sdr_unpack_i16_dev, bit-exact) and two things differ on purpose:
//   * a replay is not real time, so no block is dropped: the producer publishes block k+1 only after
//     the consumer has come back to wait_for_samples() (the reference overwrites cf_samples whenever
//     the radio delivers, racing with a slow consumer);
//   * the raw int16 block is kept as well (raw_block()): the fused kernels take it directly, so a
//     consumer that switches to sdrhip::fm_demod_i16 never materialises the float stream.
#ifndef SDR_HIP_SOURCE_HPP_
#define SDR_HIP_SOURCE_HPP_

#include <atomic>
#include <exception>
#include <cmath>
#include <complex>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "sdr_hip_shims.hpp"

// bladerf/common/include/sdr_base.h:11-59 -- same members, same virtuals
class SDR_BASE
{
public:
    bool collect_samples = false;
    uint64_t num_samples = 0;

    std::mutex sdr_mtx;
    std::condition_variable sdr_cv;
    bool processed = false;

    static std::unique_ptr<SDR_BASE> build();       // defined by the application (bladerf_fm_demod.cpp:62-79)
    virtual ~SDR_BASE() {}

    virtual uint64_t get_rx_samplerate() const = 0;
    virtual uint64_t get_tx_samplerate() const = 0;
    virtual uint64_t get_rx_frequency() const = 0;
    virtual uint64_t get_tx_frequency() const = 0;
    virtual void set_rx_samplerate(uint64_t fs) = 0;
    virtual void set_rx_gain(int32_t gain, uint32_t mode) = 0;
    virtual void set_rx_bandwidth(uint32_t bw) = 0;
    virtual void set_rx_frequency(uint64_t f) = 0;
    virtual void set_tx_frequency(uint64_t f) = 0;

    virtual void start(std::vector<std::complex<float>>& cf_samples) = 0;
    virtual void start_single(std::vector<std::complex<float>>& cf_samples, uint32_t num_samples) = 0;
    virtual void stop() = 0;

    virtual void wait_for_samples()                  // sdr_base.h:50-55
    {
        std::unique_lock<std::mutex> lck(sdr_mtx);
        sdr_cv.wait(lck, [this] { return processed; });
        processed = false;
    }
};

namespace sdrhip {

// Where the int16 IQ blocks come from: what bladerf_sync_rx (bladerf_sdr.h:281) is to the reference.
class BlockSource {
public:
    virtual ~BlockSource() {}
    // fill dst[0 .. n); false when the source has no further full block
    virtual bool next(std::complex<int16_t>* dst, size_t n) = 0;
};

// A recording of interleaved int16 I,Q (.sc16 / .bin as rx_record writes it,
// bladerf/rx_record/src/bladerf_rec_main_v1.cpp:161-171), replayed block by block; `passes` times over,
// starting again at the top like temp_get_data (bladerf_apt_demod.cpp:139-145).
class FileSource : public BlockSource {
public:
    explicit FileSource(const std::string& path, int passes = 1) : passes_(passes)
    {
        f_ = std::fopen(path.c_str(), "rb");
        if (!f_) throw std::runtime_error("FileSource: cannot open " + path);
    }
    ~FileSource() override { if (f_) std::fclose(f_); }
    bool next(std::complex<int16_t>* dst, size_t n) override
    {
        for (;;) {
            if (std::fread(dst, sizeof(std::complex<int16_t>), n, f_) == n) return true;
            if (++pass_ >= passes_) return false;            // the tail shorter than a block is dropped
            std::rewind(f_);
        }
    }
private:
    std::FILE* f_ = nullptr;
    int passes_, pass_ = 0;
};

// FM-modulated tone, the law of sdr_synth_fm_i16_dev and of the reference's TX generators
// (bladerf/tx_example/src/bladerf_tx_main_v1.cpp:120-138): phi = (dev/ftone) sin(2 pi ftone t / fs).
class SynthFmSource : public BlockSource {
public:
    SynthFmSource(double fs, double ftone, double dev, double amp, uint64_t total_samples)
        : fs_(fs), ftone_(ftone), dev_(dev), amp_(amp), left_(total_samples) {}
    bool next(std::complex<int16_t>* dst, size_t n) override
    {
        if (left_ < n) return false;
        const double beta = dev_ / ftone_, w = ftone_ / fs_;
        for (size_t i = 0; i < n; ++i, ++t_) {
            const double cyc = w * (double)t_;
            const double ph = beta * std::sin(2.0 * M_PI * (cyc - std::rint(cyc)));
            dst[i] = std::complex<int16_t>((int16_t)std::rint(amp_ * std::cos(ph)), (int16_t)std::rint(amp_ * std::sin(ph)));
        }
        left_ -= n;
        return true;
    }
private:
    double fs_, ftone_, dev_, amp_;
    uint64_t left_, t_ = 0;
};

// The SDR_BASE implementation without a radio.
class REPLAY_SDR : public SDR_BASE {
public:
    explicit REPLAY_SDR(std::unique_ptr<BlockSource> src, uint64_t block_samples = 4096 * 32, int device = 0)
        : src_(std::move(src)), device_(device)
    {
        num_samples = block_samples;                 // bladerf_sdr.h:79
    }
    ~REPLAY_SDR() override { if (thread_.joinable()) stop(); }

    // device settings are recorded, nothing to tune (bladerf_sdr.h:97-267 talk to the radio)
    uint64_t get_rx_samplerate() const override { return fs_; }
    uint64_t get_tx_samplerate() const override { return 0; }
    uint64_t get_rx_frequency() const override { return rx_freq_; }
    uint64_t get_tx_frequency() const override { return 0; }
    void set_rx_samplerate(uint64_t fs) override { fs_ = fs; }
    void set_rx_gain(int32_t, uint32_t) override {}
    void set_rx_bandwidth(uint32_t) override {}
    void set_rx_frequency(uint64_t f) override { rx_freq_ = f; }
    void set_tx_frequency(uint64_t) override {}

    // bladerf_sdr.h:305-319
    void start(std::vector<std::complex<float>>& cf_samples) override
    {
        stop();                                      // a second start() without stop(): end the running producer first
        cf_out_ = &cf_samples;
        error_ = nullptr;
        collect_samples = true;
        processed = false;
        finished_ = false;
        consumed_ = true;
        cf_samples.clear();
        cf_samples.resize(num_samples);
        thread_ = std::thread(&REPLAY_SDR::get_continuous_samples, this, std::ref(cf_samples));
    }

    // bladerf_sdr.h:322-338: one block, converted, synchronously
    void start_single(std::vector<std::complex<float>>& cf_samples, uint32_t n) override
    {
        check(sdr_init(device_));
        cf_samples.clear();
        cf_samples.resize(n);
        raw_.assign(n, std::complex<int16_t>(0, 0));
        if (!src_->next(raw_.data(), n)) { finished_ = true; return; }
        convert(raw_.data(), n, cf_samples.data());
    }

    // bladerf_sdr.h:341-346
    void stop() override
    {
        {
            std::lock_guard<std::mutex> lg(sdr_mtx);
            collect_samples = false;
            consumed_ = true;
        }
        sdr_cv.notify_all();
        if (thread_.joinable()) thread_.join();
    }

    // sdr_base.h:50-55, plus: coming back here tells the producer that the previous block has been
    // consumed.  A radio never ends; a recording does, and the producer can fail.  The reference's loops
    // are `while (run) { wait_for_samples(); process(cf_samples); }` and know neither case, so both are
    // made visible through what such a loop already has: a producer error is rethrown here (the mains
    // wrap the loop in try / catch (std::exception), bladerf_apt_demod.cpp:857), and at the end of the
    // source cf_samples comes back EMPTY (not the previous block once more) -- and, unless
    // set_throw_at_end(false), sdrhip::end_of_stream (a std::runtime_error) is thrown.
    struct end_of_stream : std::runtime_error { end_of_stream() : std::runtime_error("replay source exhausted") {} };
    void set_throw_at_end(bool on) { throw_at_end_ = on; }
    void wait_for_samples() override
    {
        std::unique_lock<std::mutex> lck(sdr_mtx);
        consumed_ = true;
        sdr_cv.notify_all();
        sdr_cv.wait(lck, [this] { return processed || finished_; });
        const bool fresh = processed;
        processed = false;
        if (fresh) return;
        if (cf_out_) cf_out_->clear();
        if (error_) std::rethrow_exception(error_);
        if (throw_at_end_) throw end_of_stream();
    }

    bool finished() const { return finished_; }
    // the int16 block behind the current cf_samples (valid until the next wait_for_samples())
    const std::vector<std::complex<int16_t>>& raw_block() const { return raw_; }
    uint64_t blocks_published() const { return blocks_; }

private:
    void convert(const std::complex<int16_t>* in, size_t n, std::complex<float>* out)
    {
        // bladerf_sdr.h:288-291 on the GPU: cf = complex<float>(re, im) * (1/2048)
        // (the calling thread's grow-only device scratch: no allocation per block)
        int16_t* d_i = static_cast<int16_t*>(scratch().get(0, n * 4));
        float* d_f = static_cast<float*>(scratch().get(1, n * 8));
        check(sdr_dev_upload(d_i, in, n * 4));
        check(sdr_unpack_i16_dev(d_i, n, d_f, nullptr));
        check(sdr_dev_download(out, d_f, n * 8));
    }

    // bladerf_sdr.h:270-302
    void get_continuous_samples(std::vector<std::complex<float>>& cf_samples)
    {
        try {
            check(sdr_init(device_));                // the device is per thread
            std::vector<std::complex<int16_t>> samples(num_samples);
            std::vector<std::complex<float>> cs(num_samples);
            for (;;) {
                {
                    std::lock_guard<std::mutex> lg(sdr_mtx);
                    if (!collect_samples) break;
                }
                const bool got = src_->next(samples.data(), num_samples);
                if (got) convert(samples.data(), num_samples, cs.data());      // overlaps the consumer's work on block k
                std::unique_lock<std::mutex> lck(sdr_mtx);
                sdr_cv.wait(lck, [this] { return consumed_ || !collect_samples; });
                if (!collect_samples) break;
                if (!got) { finished_ = true; lck.unlock(); sdr_cv.notify_all(); break; }
                cf_samples.swap(cs);                 // publish (the reference converts in place under the lock)
                raw_.swap(samples);
                samples.resize(num_samples);
                cs.resize(num_samples);
                consumed_ = false;
                processed = true;
                ++blocks_;
                lck.unlock();
                sdr_cv.notify_all();
            }
        } catch (const std::exception& e) {
            std::fprintf(stderr, "error in collection thread: %s\n", e.what());    // bladerf_sdr.h:298-301
            std::lock_guard<std::mutex> lg(sdr_mtx);
            error_ = std::current_exception();       // handed to the consumer by wait_for_samples()
            finished_ = true;
            sdr_cv.notify_all();
        }
    }

    std::unique_ptr<BlockSource> src_;
    int device_;
    uint64_t fs_ = 0, rx_freq_ = 0;
    std::thread thread_;
    std::vector<std::complex<int16_t>> raw_;
    bool consumed_ = true;
    bool throw_at_end_ = true;
    std::vector<std::complex<float>>* cf_out_ = nullptr;
    std::exception_ptr error_;
    std::atomic<bool> finished_{false};
    uint64_t blocks_ = 0;
};

}  // namespace sdrhip

#endif  // SDR_HIP_SOURCE_HPP_
