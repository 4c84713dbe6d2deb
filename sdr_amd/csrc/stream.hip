// stream.hip -- block hand-off with persistent filter history.
//
// Replaces the capture -> consumer hand-off of the reference:
//   SDR_BASE::start / wait_for_samples / stop     bladerf/common/include/sdr_base.h:42-55
//   BLADERF_SDR::get_continuous_samples           bladerf/common/include/bladerf_sdr.h:270-302
//   iq_data_queue[2] + read_index/write_index     bladerf/apt_demod/src/bladerf_apt_demod.cpp:64-71,152-178
// There the consumer filters every block from zero state (a T-1 sample transient per block).
// Here the producer pushes blocks; each block goes pinned-host -> device asynchronously behind
// the tail of the previous block (the FIR history halo), one range launch computes exactly the
// outputs that became computable, and pop() hands them back in order.  Two slots alternate so
// that the producer can fill slot k+1 while slot k is still in flight.

#include <cstdio>
#include <cstring>
#include <vector>

#include "sdr_internal.h"

struct sdr_fir_plan;
extern "C" int sdr_fir_plan_shape(const sdr_fir_plan *p, int *input, int *mode, int *decim, int *n_taps);

namespace {
constexpr int kSlots = 2;
size_t in_bytes(int input) { return sdr::in_elem_bytes(input); }
size_t out_bytes(int mode) { return mode == SDR_MODE_IQ ? 8 : 4; }
}

struct sdr_fir_stream {
    sdr_fir_plan *plan = nullptr;
    int input = 0, mode = 0, D = 1, T = 1;
    size_t max_block = 0, halo = 0;          // samples
    hipStream_t st = nullptr;
    // per slot: pinned input staging, device window [halo | block], device + pinned output
    void *h_in[kSlots] = {}, *d_win[kSlots] = {}, *d_out[kSlots] = {}, *h_out[kSlots] = {};
    hipEvent_t done[kSlots] = {};
    size_t out_n[kSlots] = {};
    bool busy[kSlots] = {};
    int head = 0, tail = 0;                  // next slot to push into / to pop from
    unsigned long long n_pushed = 0;         // samples so far
    unsigned long long n_emitted = 0;        // outputs launched so far
    size_t prev_valid = 0;                   // samples of history held in the previous slot's window
    int prev_slot = -1;
    size_t prev_len = 0;                     // halo_used + block length of the previous window
};

extern "C" int sdr_fir_stream_create(sdr_fir_plan *plan, size_t max_block_samples, sdr_fir_stream **out)
{
    if (!plan || !out || max_block_samples == 0) { sdr::set_error("bad argument"); return SDR_EINVAL; }
    *out = nullptr;
    int rc = sdr::ensure_device();
    if (rc != SDR_OK) return rc;
    sdr_fir_stream *s = new sdr_fir_stream();
    s->plan = plan;
    rc = sdr_fir_plan_shape(plan, &s->input, &s->mode, &s->D, &s->T);
    if (rc != SDR_OK) { delete s; return rc; }
    s->max_block = max_block_samples;
    // history a block needs from its predecessor: T-1 samples for the FIR plus D so that the first
    // new discriminator output has its predecessor y, plus D of slack for the y that was not yet
    // computable last time
    s->halo = (size_t)(s->T - 1 + 2 * s->D + 4);
    const size_t win = s->halo + s->max_block;
    const size_t max_out = win / (size_t)s->D + 2;
    hipError_t e = hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking);
    for (int k = 0; k < kSlots && e == hipSuccess; ++k) {
        e = sdr::host_malloc(&s->h_in[k], s->max_block * in_bytes(s->input), hipHostMallocDefault);
        if (e == hipSuccess) e = sdr::dev_malloc(&s->d_win[k], win * in_bytes(s->input));
        if (e == hipSuccess) e = sdr::dev_malloc(&s->d_out[k], max_out * out_bytes(s->mode));
        if (e == hipSuccess) e = sdr::host_malloc(&s->h_out[k], max_out * out_bytes(s->mode), hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->done[k], hipEventDisableTiming);
    }
    if (e != hipSuccess) { sdr_fir_stream_destroy(s); return sdr::hip_fail(e, "stream allocation"); }
    *out = s;
    return SDR_OK;
}

extern "C" int sdr_fir_stream_destroy(sdr_fir_stream *s)
{
    if (!s) return SDR_OK;
    if (s->st) (void)hipStreamSynchronize(s->st);
    for (int k = 0; k < kSlots; ++k) {
        if (s->h_in[k]) (void)hipHostFree(s->h_in[k]);
        if (s->d_win[k]) (void)hipFree(s->d_win[k]);
        if (s->d_out[k]) (void)hipFree(s->d_out[k]);
        if (s->h_out[k]) (void)hipHostFree(s->h_out[k]);
        if (s->done[k]) (void)hipEventDestroy(s->done[k]);
    }
    if (s->st) (void)hipStreamDestroy(s->st);
    delete s;
    return SDR_OK;
}

// The pinned staging buffer of the slot the next push will use, max_block samples long: a capture
// thread can let the radio write straight into it (what bladerf_sync_rx does with
// iq_data_queue[write_index], bladerf_apt_demod.cpp:166) and push that pointer, which skips the
// host-side copy.  SDR_ECAPACITY while that slot is still in flight.
extern "C" int sdr_fir_stream_next_buffer(sdr_fir_stream *s, void **buffer)
{
    if (!s || !buffer) { sdr::set_error("bad argument"); return SDR_EINVAL; }
    *buffer = nullptr;
    if (s->busy[s->head]) { sdr::set_error("both slots in flight: pop before taking a buffer"); return SDR_ECAPACITY; }
    *buffer = s->h_in[s->head];
    return SDR_OK;
}

// Enqueue one block.  Returns SDR_ECAPACITY when both slots are still waiting to be popped.
extern "C" int sdr_fir_stream_push(sdr_fir_stream *s, const void *block, size_t n)
{
    if (!s || (!block && n)) { sdr::set_error("bad argument"); return SDR_EINVAL; }
    if (n == 0) return SDR_OK;
    if (n > s->max_block) { sdr::set_error("block of %zu samples exceeds max_block %zu", n, s->max_block); return SDR_EINVAL; }
    const int k = s->head;
    if (s->busy[k]) { sdr::set_error("both slots in flight: pop before pushing again"); return SDR_ECAPACITY; }
    const size_t ib = in_bytes(s->input);
    // history: the last `h` samples of the previous window, device to device
    size_t h = 0;
    if (s->prev_slot >= 0) {
        h = s->prev_len < s->halo ? s->prev_len : s->halo;
        SDR_HIP(hipMemcpyAsync(s->d_win[k], (const unsigned char *)s->d_win[s->prev_slot] + (s->prev_len - h) * ib,
                               h * ib, hipMemcpyDeviceToDevice, s->st));
    }
    if (block != s->h_in[k]) memcpy(s->h_in[k], block, n * ib);      // not already in the slot's buffer
    SDR_HIP(hipMemcpyAsync((unsigned char *)s->d_win[k] + h * ib, s->h_in[k], n * ib, hipMemcpyHostToDevice, s->st));
    const unsigned long long n_total = s->n_pushed + n;
    const unsigned long long first = s->n_pushed - h;            // stream index of d_win[k][0]
    const unsigned long long cnt = sdr_fir_out_count(s->plan, (size_t)n_total);
    s->out_n[k] = (size_t)(cnt - s->n_emitted);
    if (s->out_n[k] > 0) {
        int rc = sdr_fir_execute_range_dev(s->plan, s->d_win[k], first, h + n, n_total, s->n_emitted, cnt,
                                           s->d_out[k], s->st);
        if (rc != SDR_OK) return rc;
        SDR_HIP(hipMemcpyAsync(s->h_out[k], s->d_out[k], s->out_n[k] * out_bytes(s->mode), hipMemcpyDeviceToHost, s->st));
    }
    SDR_HIP(hipEventRecord(s->done[k], s->st));
    s->busy[k] = true;
    s->n_emitted = cnt;
    s->n_pushed = n_total;
    s->prev_slot = k;
    s->prev_len = h + n;
    s->head = (k + 1) % kSlots;
    return SDR_OK;
}

// Wait for the oldest pushed block and copy its outputs (the reference's wait_for_samples()).
extern "C" int sdr_fir_stream_pop(sdr_fir_stream *s, void *out, size_t capacity, size_t *n_out)
{
    if (!s || !n_out) { sdr::set_error("bad argument"); return SDR_EINVAL; }
    const int k = s->tail;
    if (!s->busy[k]) { *n_out = 0; sdr::set_error("nothing in flight"); return SDR_EINVAL; }
    if (s->out_n[k] && !out) { sdr::set_error("null output"); return SDR_EINVAL; }
    if (s->out_n[k] > capacity) { sdr::set_error("output capacity %zu < %zu", capacity, s->out_n[k]); return SDR_ECAPACITY; }
    SDR_HIP(hipEventSynchronize(s->done[k]));
    if (s->out_n[k]) memcpy(out, s->h_out[k], s->out_n[k] * out_bytes(s->mode));
    *n_out = s->out_n[k];
    s->busy[k] = false;
    s->tail = (k + 1) % kSlots;
    return SDR_OK;
}

// File replay: the reference's stand-in for a live capture,
//   temp_get_data                    bladerf/apt_demod/src/bladerf_apt_demod.cpp:92-148
// reads a raw sample file (.sc16: interleaved int16 I,Q as rx_record writes it,
// bladerf_rec_main_v1.cpp:169), drops the tail so that the length is a multiple of two capture
// blocks (:113-117), and hands the blocks to the consumer one after the other, starting over at
// the end of the file (:139-145).  Here: the same blocks go through push/pop; `passes` bounds the
// cycling (the reference cycles until it is stopped); the sink sees every block's outputs in
// order.  Filter history persists across blocks and across the wrap-around, as everywhere in
// this hand-off (the reference filters each block from zero state).
extern "C" int sdr_fir_stream_replay_file(sdr_fir_stream *s, const char *path, size_t block_samples,
                                          int passes, sdr_block_sink sink, void *user,
                                          uint64_t *samples_replayed)
{
    if (!s || !path || !sink || block_samples == 0 || passes <= 0) { sdr::set_error("bad argument"); return SDR_EINVAL; }
    if (block_samples > s->max_block) { sdr::set_error("block of %zu samples exceeds max_block %zu", block_samples, s->max_block); return SDR_EINVAL; }
    if (samples_replayed) *samples_replayed = 0;
    FILE *f = fopen(path, "rb");
    if (!f) { sdr::set_error("cannot open %s", path); return SDR_EINVAL; }
    const size_t ib = in_bytes(s->input), ob = out_bytes(s->mode);
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); sdr::set_error("cannot seek %s", path); return SDR_EINVAL; }
    const long long fbytes = ftell(f);
    unsigned long long n_file = fbytes > 0 ? (unsigned long long)fbytes / ib : 0;
    n_file -= n_file % (2ULL * block_samples);                  // :113-117
    if (n_file == 0) { fclose(f); sdr::set_error("%s holds less than two blocks of %zu samples", path, block_samples); return SDR_EINVAL; }
    std::vector<unsigned char> blk(block_samples * ib);
    const size_t cap = (s->halo + s->max_block) / (size_t)s->D + 2;
    std::vector<unsigned char> outbuf(cap * ob);
    int rc = SDR_OK;
    unsigned long long done = 0;
    int in_flight = 0;
    auto drain_one = [&]() -> int {
        size_t n_out = 0;
        int r = sdr_fir_stream_pop(s, outbuf.data(), cap, &n_out);
        if (r != SDR_OK) return r;
        --in_flight;
        return sink(user, outbuf.data(), n_out) == 0 ? SDR_OK : SDR_ERANGE;
    };
    for (int pass = 0; pass < passes && rc == SDR_OK; ++pass) {
        if (fseek(f, 0, SEEK_SET) != 0) { sdr::set_error("cannot rewind %s", path); rc = SDR_EINVAL; break; }
        for (unsigned long long at = 0; at < n_file && rc == SDR_OK; at += block_samples) {
            if (fread(blk.data(), ib, block_samples, f) != block_samples) { sdr::set_error("short read from %s", path); rc = SDR_EINVAL; break; }
            if (in_flight == kSlots) rc = drain_one();          // keep one block in flight behind the read
            if (rc != SDR_OK) break;
            rc = sdr_fir_stream_push(s, blk.data(), block_samples);
            if (rc == SDR_OK) { ++in_flight; done += block_samples; }
        }
    }
    while (in_flight > 0) {                                     // always leave the stream empty
        const int r = drain_one();
        if (rc == SDR_OK) rc = r;
        if (r != SDR_OK && r != SDR_ERANGE) break;
    }
    fclose(f);
    if (rc == SDR_ERANGE) sdr::set_error("sink stopped the replay");
    if (samples_replayed) *samples_replayed = done;
    return rc;
}
