// fir_plan.hip -- host side of the FIR + decimate chain: plan creation (kernel choice from measured cost models,
// tile geometry, tap / digit / phasor tables), launch set-up and the C ABI (include/sdr_hip.h, rows a2-a7).
// Replaces the set-up half of polyphase_decimate (bladerf/apt_demod/src/bladerf_apt_demod.cpp:249-279) and
// af::array(lpf) (bladerf/rx_fm_demod/src/bladerf_fm_demod.cpp:217); the kernels are in fir_tiles.hip and
// fir_mfma_i8.hip.

#include "fir_common.h"

using namespace sdr::fir;

namespace {

int roundup(int v, int m) { return (v + m - 1) / m * m; }

struct Geometry { int Tback, Lw4, U, ngran, look, q, pad, s_gran, a_gran, n_inst, lds_bytes; };

// Tile geometry for (T, D, R, NT) with spg samples per LDS granule.  Returns false when the lane
// stride is not a whole number of granules.
// entries of the coarse mixer phasor table for a tile of s_gran 2-sample granules (even count)
int ph_hi_entries(int s_gran) { return ((2 * s_gran) / 64 + 2 + 1) & ~1; }

bool make_geometry(int T, int D, int R, int NT, int spg, bool stream, Geometry *o, bool planar = false, bool mixer = false,
                   int step_mult = 2)
{
    if (((long long)D * R) % spg) return false;
    o->Tback = roundup(T - 1, 4);
    o->Lw4 = roundup(o->Tback + D * (R - 1) + 1, 4);
    o->U = (spg * R >= 16) ? 1 : 16 / (spg * R);
    const int steps = roundup((o->Lw4 / spg + o->U - 1) / o->U, step_mult);
    o->ngran = steps * o->U;                  // granules walked (padded with zero taps)
    o->look = o->U;                           // one look-ahead step is loaded, never used
    o->q = D * R / spg;
    o->pad = (o->q % 2 == 0) ? 1 : 0;
    const long long s_need = (long long)D * R * (NT - 1) + (long long)(o->ngran + o->look) * spg;
    const long long s_tile = s_need;
    if (s_tile > (1LL << 22)) return false;
    o->s_gran = (int)((s_tile + spg - 1) / spg);
    const int a = o->s_gran + (o->pad ? o->s_gran / o->q + 1 : 0);
    // the persistent streaming kernel stages whole rounds of NT granules; the one-tile kernels
    // mask the last round, so their image is exact (64 taps / 10: 22.8 KB, a seventh workgroup
    // per CU fits in the 160 KB)
    o->a_gran = stream ? roundup(a, NT) : a;
    o->n_inst = (o->a_gran + NT - 1) / NT;
    o->lds_bytes = o->a_gran * (planar ? 32 : 16) + (NT + 2) * 8;
    if (mixer) o->lds_bytes += (ph_hi_entries(o->s_gran) + 64) * 8;
    return true;
}

size_t in_elem_bytes(int input) { return sdr::in_elem_bytes(input); }
size_t out_elem_bytes(int mode) { return mode == SDR_MODE_IQ ? 8 : 4; }

// Enqueue one launch.  `in` points at stream sample index in_first of channel 0.
int launch(sdr_fir_plan *p, const void *in, long long in_first, long long n_resident,
           long long n_total, long long out_begin, long long out_end, long long in_stride,
           size_t n_ch, void *out, long long out_stride, hipStream_t st, long long chan_first = 0)
{
    if (out_end <= out_begin || n_ch == 0) return SDR_OK;
    if (n_ch > 65535) { sdr::set_error("n_channels > 65535"); return SDR_EINVAL; }
    if (int rc = sdr::same_device(p->device)) return rc;
    const size_t ia = in_elem_bytes(p->input);
    if (((uintptr_t)in) % ia) { sdr::set_error("input pointer not sample aligned"); return SDR_EALIGN; }
    if (((uintptr_t)out) % out_elem_bytes(p->mode)) { sdr::set_error("output pointer misaligned"); return SDR_EALIGN; }
    FirGeom g;
    g.in_stride = in_stride;
    g.out_stride = out_stride;
    g.lo_valid = in_first > 0 ? in_first : 0;
    g.hi_valid = (in_first + n_resident < n_total) ? in_first + n_resident : n_total;
    if (p->mfma_i8 && g.hi_valid - g.lo_valid < (p->input == SDR_IN_I8_IQ ? 8 : 4)) {
        // its staging loads are 16 bytes wide (4 int16 or 8 int8 samples) and clamped into the valid range
        if (p->tiny) return launch(p->tiny, in, in_first, n_resident, n_total, out_begin, out_end, in_stride, n_ch, out, out_stride, st);
        sdr::set_error("pipeline 5 needs at least %d resident samples", p->input == SDR_IN_I8_IQ ? 8 : 4);
        return SDR_EINVAL;
    }
    g.tile0 = out_begin & ~3LL;
    g.out_begin = out_begin;
    g.out_end = out_end;
    g.store_from = out_begin;
    if (p->mfma_i8 && p->d_ftaps && p->mode == SDR_MODE_FM && out_begin * p->D < p->T - 1) {
        // the start-up outputs are computed with float32 taps (i8_startup_fm_output), not by the tiles
        const long long split = (p->T - 1 + p->D - 1) / p->D + 1;
        g.store_from = split < out_end ? split : out_end;
    }
    g.mix_fr = p->mix_fr;
    g.D = p->D; g.T = p->T; g.Tback = p->Tback; g.ngran = p->ngran; g.q = p->q; g.pad = p->pad;
    g.magic = (unsigned)((0x100000000ULL + (unsigned)p->q) / (unsigned)(p->q + 1));
    g.magic_q = (unsigned)((0x100000000ULL + (unsigned)p->q - 1) / (unsigned)p->q);
    g.s_gran = p->s_gran; g.a_gran = p->a_gran; g.n_inst = p->n_inst; g.mode = p->mode;
    g.swap_iq = (p->input == SDR_IN_I16_QI) ? 1 : 0;
    g.ph_off = p->ph_off; g.n_ph_hi = p->n_ph_hi;
    g.zfix = (p->mfma_i8 && p->mode == SDR_MODE_FM && p->zfix) ? 1 : 0;
    g.chan_fr = nullptr; g.chan_tab = nullptr; g.chan_first = 0; g.chan_tab_stride = 0; g.chan_bcast = p->chan_bcast ? 1 : 0;
    if (p->chan_bcast) chan_first = 0;
    if (p->n_chan_mix > 0) {
        if (!p->chan_bcast && (chan_first < 0 || chan_first + (long long)n_ch > p->n_chan_mix)) {
            sdr::set_error("channels [%lld, %lld) but the plan has %d channel mixers", chan_first, chan_first + (long long)n_ch, p->n_chan_mix);
            return SDR_EINVAL;
        }
        g.chan_fr = p->d_chan_fr; g.chan_tab = p->cpx ? p->d_chan_e : p->d_chan_tab; g.chan_first = (int)chan_first; g.chan_tab_stride = p->chan_tab_stride;
    }
    g.scale = (float)p->scale;
    // with the discriminator a tile's last lane only supplies its neighbour's successor y
    const long long per_tile = (long long)(p->mode == SDR_MODE_FM ? p->NT - 1 : p->NT) * p->R;
    g.per_tile = (int)per_tile;
    const long long tiles = (out_end - g.tile0 + per_tile - 1) / per_tile;
    if (tiles * (long long)n_ch > 0x7fffffffLL) { sdr::set_error("too many tiles"); return SDR_EINVAL; }
    g.tiles_per_ch = (int)tiles;
    g.tiles_per_xcd = (int)((tiles + 7) / 8);
    g.total_tiles = (int)(tiles * (long long)n_ch);
    // virtual pointer to stream sample 0 of channel 0 (only dereferenced inside the valid window)
    const void *base = (const unsigned char *)in - (size_t)in_first * in_elem_bytes(p->input);
    hipError_t e;
    const char *what = "fir kernel launch";
    if (p->mfma_i8) e = launch_mfma_i8(p, st, base, out, g, &what);
    else e = launch_tile_kernels(p, st, base, out, g, (unsigned)tiles, (unsigned)n_ch, &what);
    if (e != hipSuccess) return sdr::hip_fail(e, what);
    p->lds_opt_in.store(true, std::memory_order_relaxed);
    return SDR_OK;
}

// number of final outputs for a stream of n samples
long long out_count(const sdr_fir_plan *p, long long n)
{
    // y[m] = filtered sample D*m: m < floor(n/D) as polyphase_decimate (bladerf_apt_demod.cpp:282), or
    // every index 0, D, 2D, ... <= n-1 as x3(af::seq(0, n, D)) picks them (bladerf_fm_demod.cpp:212,294;
    // the seq's inclusive end n itself is past the block and never read here)
    const long long ny = (p->count_rule == SDR_COUNT_AF_SEQ) ? (n > 0 ? (n - 1) / p->D + 1 : 0) : n / p->D;
    if (p->mode == SDR_MODE_FM) return ny > 0 ? ny - 1 : 0;
    return ny;
}

}  // namespace

extern "C" int sdr_fir_plan_create(const sdr_fir_desc *d, sdr_fir_plan **out)
{
    if (!d || !out) { sdr::set_error("null argument"); return SDR_EINVAL; }
    *out = nullptr;
    // the reference throws std::invalid_argument for D <= 0 (bladerf_apt_demod.cpp:254-255)
    if (d->decim <= 0) { sdr::set_error("Decimation factor must be positive."); return SDR_EINVAL; }
    if (d->n_taps <= 0 || !d->taps) { sdr::set_error("n_taps must be positive"); return SDR_EINVAL; }
    if (d->n_taps > 16384) { sdr::set_error("n_taps > 16384 unsupported"); return SDR_EINVAL; }
    if (d->input < SDR_IN_I16_IQ || d->input > SDR_IN_I8_IQ) { sdr::set_error("bad input kind"); return SDR_EINVAL; }
    if (d->mode < SDR_MODE_IQ || d->mode > SDR_MODE_REAL) { sdr::set_error("bad mode"); return SDR_EINVAL; }
    if (d->pipeline < 0 || d->pipeline > 7) { sdr::set_error("bad pipeline"); return SDR_EINVAL; }
    if (d->count_rule != SDR_COUNT_FLOOR && d->count_rule != SDR_COUNT_AF_SEQ) { sdr::set_error("bad count_rule"); return SDR_EINVAL; }
    int rc = sdr::ensure_device();
    if (rc != SDR_OK) return rc;

    // One mixer on integer input (frequency_shift in front of the filter): when the window fits
    // the complex-tap matrix kernel (256 samples), the plan is built as a channelizer whose one table entry
    // every channel uses -- 0.31 against 0.55 ms per 2^28 samples at 64 taps / 10 on the float-tile kernel,
    // and exact integer sums, so shards and blocks reproduce the one-piece result bit for bit.
    if (d && out && d->mix_fr != 0.0 && d->n_mix_channels == 0 && (d->pipeline == 0 || d->pipeline == 5) && !d->outputs_per_lane &&
        !d->threads && d->taps && d->decim > 0 && d->n_taps >= 6 && d->n_taps <= 256 &&
        (d->input == SDR_IN_I16_IQ || d->input == SDR_IN_I16_QI || d->input == SDR_IN_I8_IQ) &&
        (d->n_taps - 1 + 3) / 4 * 4 + 15 * d->decim + 1 <= 256 && (d->size_hint == 0 || d->size_hint >= (1LL << 22))) {
        sdr_fir_desc dd = *d;
        const double fr1 = d->mix_fr;
        dd.mix_fr = 0.0; dd.n_mix_channels = 1; dd.mix_fr_channels = &fr1; dd.pipeline = 5;
        sdr_fir_plan *q = nullptr;
        if (sdr_fir_plan_create(&dd, &q) == SDR_OK && q) {
            q->chan_bcast = true;
            q->mix_fr = d->mix_fr;
            if (q->tiny) { q->tiny->chan_bcast = true; q->tiny->mix_fr = d->mix_fr; }
            *out = q;
            return SDR_OK;
        }
        if (d->pipeline == 5) return SDR_EINVAL;       // (the explicit choice does not fit: the error text is set)
    }
    const bool chan_mix = d->n_mix_channels > 0;
    if (d->n_mix_channels < 0 || (chan_mix && !d->mix_fr_channels)) { sdr::set_error("n_mix_channels without mix_fr_channels"); return SDR_EINVAL; }
    if (chan_mix && d->mix_fr != 0.0) { sdr::set_error("mix_fr and mix_fr_channels are alternatives"); return SDR_EINVAL; }
    if (chan_mix && d->input == SDR_IN_F32_REAL) { sdr::set_error("per-channel mixers need complex or integer input"); return SDR_EINVAL; }
    if (chan_mix && d->n_mix_channels > 65535) { sdr::set_error("n_mix_channels > 65535"); return SDR_EINVAL; }

    sdr_fir_plan *p = new sdr_fir_plan();
    p->input = d->input; p->mode = d->mode; p->D = d->decim; p->T = d->n_taps;
    p->scale = d->scale; p->mix_fr = d->mix_fr; p->count_rule = d->count_rule;
    (void)hipGetDevice(&p->device);
    const bool raw_possible = ((d->input == SDR_IN_I16_IQ || d->input == SDR_IN_I16_QI) && d->mix_fr == 0.0 && !chan_mix);
    // the int8 matrix kernel also takes int8 sample pairs (one byte plane per component)
    const bool int_input = d->input == SDR_IN_I16_IQ || d->input == SDR_IN_I16_QI || d->input == SDR_IN_I8_IQ;
    // (channelizer plans: the per-channel mixer folded into complex taps, windows of up to 256 samples)
    const bool i8_possible = int_input && d->mix_fr == 0.0;
    if ((d->pipeline == 1 || d->pipeline == 3 || d->pipeline == 4) && !raw_possible) { delete p; sdr::set_error("pipelines 1, 3 and 4 need int16 input without mixer"); return SDR_EINVAL; }
    if (d->pipeline == 5 && !i8_possible) { delete p; sdr::set_error("pipeline 5 needs int16 or int8 input without mixer"); return SDR_EINVAL; }
    const bool ksplit_possible = d->input == SDR_IN_F32_REAL && d->mode != SDR_MODE_FM;
    // float input without mixer: the DMA'd image is the float tile (fir_raw_kernel<R, IN_CF32 | IN_REAL>)
    // (a mixer goes along: it lives in complex taps and one phasor per output)
    const bool rawf_possible = (d->input == SDR_IN_CF32 || d->input == SDR_IN_F32_REAL) && !chan_mix;
    if (d->pipeline == 7 && !rawf_possible) { delete p; sdr::set_error("pipeline 7 needs complex or real float input (one mixer at most, no per-channel mixers)"); return SDR_EINVAL; }
    if (d->pipeline == 6 && !ksplit_possible) { delete p; sdr::set_error("pipeline 6 needs real float input and a mode other than FM"); return SDR_EINVAL; }
    // The planar kernel (float planes, convert once) is an explicit choice only: beyond the int8
    // matrix kernel's window it is 8-12 % ahead of the raw tile at D = 8 and 10 but 1.4-1.8x behind
    // at D = 16 and 20 (T = 640..2047, tools/sweep_shapes.py), so the automatic choice there is the raw tile.
    const bool planar = (d->pipeline == 3);

    // ---- choose kernel, outputs per lane (R) and workgroup size (NT) ------------------------
    // Raw-tile / planar kernels (int16, no mixer), from geometry sweeps on MI355X
    // (tools/tune_fir.py; D = 2..20, T = 32..511): two outputs per lane halve the conversions per
    // output and stay at an odd lane stride for most D; four when D is small (a lane should own
    // >= 16 input samples); one or eight are never the best (one: a conversion pair per MAC pair;
    // eight: 2-3x slower, tap SGPR sets and accumulators crowd the wave).  The largest workgroup
    // whose tile leaves two workgroups per CU wins within an R.
    // The streaming kernel (float inputs, mixer) keeps the instruction-count model below.
    const int Rs[4] = {1, 2, 4, 8};
    const int NTs[3] = {256, 128, 64};
    Geometry best{};
    int bestR = 0, bestNT = 0;
    bool best_raw = false;
    double best_cost = 0;
    if (d->pipeline == 4) {
        // Toeplitz GEMM geometry: 256 lanes, 512 outputs per tile, groups of 16 outputs
        if (d->outputs_per_lane && d->outputs_per_lane != 2) { delete p; sdr::set_error("pipeline 4 has 2 outputs per lane"); return SDR_EINVAL; }
        if (d->threads && d->threads != 256) { delete p; sdr::set_error("pipeline 4 has 256 threads"); return SDR_EINVAL; }
        Geometry gm{};
        gm.Tback = roundup(p->T - 1, 4);
        const int W = roundup(gm.Tback + 15 * p->D + 1, 16);
        const int nq = roundup(W / 16, kMfmaAhead);
        gm.ngran = nq;                                      // k-chunks of 16
        const long long s_a = (long long)p->D * 16 * 31 + 16LL * (nq + kMfmaAhead);   // last group + look-ahead chunks
        const long long s_tile = s_a;
        gm.s_gran = (int)((s_tile + 3) / 4);
        gm.a_gran = gm.s_gran;
        gm.n_inst = (gm.a_gran + 255) / 256;
        gm.q = 1; gm.pad = 0;
        gm.lds_bytes = gm.a_gran * 32 + 513 * 8 + 8;
        if (gm.lds_bytes > 160 * 1024) { sdr::set_error("pipeline 4: tile does not fit in LDS (decim=%d taps=%d)", p->D, p->T); delete p; return SDR_EINVAL; }
        best = gm; bestR = 2; bestNT = 256; best_raw = true;
        p->mfma = true;
    }
    // Automatic choice between the vector-unit kernels and the int8 matrix kernel, from a sweep of
    // fifteen shapes on MI355X (T = 64..700, D = 4..25), in ms per 2^28 samples:
    //   vector unit:  max(0.256 [HBM], 0.17 + 0.0125 * T/D)
    //   matrix pipe:  max(HBM at 5.4 TB/s, tiles per CU * time per tile); a tile of 1024 outputs takes
    //                 0.7 + 0.112 * (k steps of the window) + 0.085 * D microseconds (matrix waves: MFMAs
    //                 per step; helper waves: samples per tile), a tile of 512 outputs (when the LDS or
    //                 the helper lanes' registers do not hold the larger one) 0.55 of that
    // The matrix kernel's time follows the number of tiles, not the filter length; it loses where
    // tiles are many and windows short (D <= 4 with few taps).
    bool auto_i8 = false;
    const bool s8 = d->input == SDR_IN_I8_IQ;
    if (d->pipeline == 0 && i8_possible && !d->outputs_per_lane && !d->threads && d->n_taps >= (s8 ? 6 : 2)) {
        // int8 sample pairs (tools/sweep_int8_input.py, eighteen shapes): the float-tile kernel is the
        // alternative, max(0.30, 0.25 + 0.02 * T/D, 0.2 + 0.85 / D); a matrix tile takes
        // 0.78 + 0.06 * steps + 0.045 * D microseconds (four byte products per step instead of seven,
        // half the bytes per sample)
        const double tpd = (double)d->n_taps / (double)d->decim;
        const double t_valu = s8 ? std::fmax(std::fmax(0.30, 0.25 + 0.02 * tpd), 0.2 + 0.85 / d->decim)
                                 : std::fmax(0.256, 0.17 + 0.0125 * tpd);
        const int Wn = roundup(d->n_taps - 1, 4) + 15 * d->decim + 1;
        const int nks = (Wn + 63) / 64, steps = (nks + 3) / 4;
        const long long s_big = 16LL * d->decim * 63 + 256LL * steps;
        const bool big = s_big <= 8 * 256 * 8 &&
                         2 * 4 * (roundup((int)s_big, 256) + 240) + 2 * 8 * kI8PartSlots * 16 + (steps > 2 ? 12 * 1024 : 0) + 64 <= 160 * 1024;
        const double tile_us = (s8 ? 0.78 + 0.06 * nks + 0.045 * d->decim : 0.7 + 0.112 * nks + 0.085 * d->decim) * (big ? 1.0 : 0.55);
        const double tiles_per_cu = (big ? 1024.0 : 2048.0) / (double)d->decim;
        const double t_hbm = ((s8 ? 2.0 : 4.0) + 4.0 / d->decim) * 268435456.0 / 5.4e12 * 1e3;
        const double t_i8 = std::fmax(t_hbm, tiles_per_cu * tile_us * 1e-3);
        auto_i8 = t_i8 < t_valu || chan_mix;     // channelizer: 2.7x the float-tile kernel with per-channel tables
        if (d->size_hint > 0 && !chan_mix) {
            // a launch of the persistent matrix kernel costs ~13 us before its first output (tap digits
            // into registers, three pipeline stages), one of the tile kernels ~4 us: short launches
            // (below 2^26 samples at the headline shape, 2^22 at 511 / 10) stay on the vector unit
            const double f = (double)d->size_hint / 268435456.0;
            auto_i8 = 0.013 + t_i8 * f < 0.004 + t_valu * f;
        }
    }
    if (d->pipeline == 5 && d->n_taps < (s8 ? 6 : 2)) {
        // no history halo: a tile at the stream's end could hold fewer than the four (int8 input: eight)
        // valid samples its 16-byte staging loads need, and a one-tap "filter" has nothing for the matrix
        // pipe anyway
        delete p;
        sdr::set_error("pipeline 5 needs at least %d taps", s8 ? 6 : 2);
        return SDR_EINVAL;
    }
    if (d->pipeline == 5 || auto_i8) {
        bool fits = true;
        // byte-split Toeplitz GEMM: 1024 or 512 outputs per tile, k steps of 64 dealt to 4 matrix waves
        if (d->outputs_per_lane || d->threads) { delete p; sdr::set_error("pipeline 5 chooses its own geometry (512 lanes): leave outputs_per_lane and threads 0"); return SDR_EINVAL; }
        Geometry gm{};
        gm.Tback = roundup(p->T - 1, 4);
        // complex taps (the channelizer), even D: the row form of fir_mfma_i8_kernel -- groups of eight outputs, steps
        // of 32 window offsets (x two components), up to eight of them; odd D: the column form, four steps of 64
        const bool rows = chan_mix && p->D % 2 == 0;
        const int W = gm.Tback + (rows ? 7 : 15) * p->D + 1;
        const int nks = rows ? (W + 31) / 32 : (W + 63) / 64;
        if (chan_mix && nks > (rows ? 8 : 4)) {
            // complex taps double the digit registers: 256 samples at most
            if (!auto_i8) { sdr::set_error("pipeline 5 with per-channel mixers: window of %d samples exceeds 256 (decim=%d taps=%d)", W, p->D, p->T); delete p; return SDR_EINVAL; }
            fits = false;
        }
        if (nks > kI8MaxSteps) {
            if (!auto_i8) { sdr::set_error("pipeline 5: window of %d samples exceeds %d (decim=%d taps=%d)", W, 64 * kI8MaxSteps, p->D, p->T); delete p; return SDR_EINVAL; }
            fits = false;
        }
        gm.ngran = nks;                                     // k steps of 64
        const int ns = (nks + 3) / 4;                       // per matrix wave
        // tile of 1024 outputs (8 column blocks, 4 outputs per finishing lane) when the LDS holds it,
        // else 512 outputs
        int ncb = 0;
        for (int cand = 8; cand >= 4 && !ncb; cand -= 4) {
            const long long s_tile = rows ? (long long)p->D * 8 * (16 * cand - 1) + 32LL * (nks <= 4 ? 4 : 8)
                                          : (long long)p->D * 16 * (8 * cand - 1) + 64LL * 4 * ns;   // last group's base + every wave's steps
            // byte planes, 1 B per sample; the offset between the I and Q planes is chosen so that the
            // 16-lane groups of the matrix waves' ds_read_b128 (MI355X_MICROARCH.md, LDS) meet no bank twice
            static const int grp[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                           {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                           {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                           {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
            const int base = roundup((int)s_tile, 256);
            int best_skew = 0, best_cost = 1 << 30;
            for (int skew = 0; skew < 256; skew += 16) {
                int cost = 0;
                for (int gi = 0; gi < 4; ++gi) {
                    int per_slot[16] = {0}, worst = 0;
                    for (int li = 0; li < 16; ++li) {
                        const int lane = grp[gi][li], jj = lane & 15, kq = lane >> 4;
                        const int addr = rows ? (kq & 1) * (base + skew) + 8 * p->D * jj + 16 * (kq >> 1)
                                              : (jj & 1) * (base + skew) + 16 * p->D * (jj >> 1) + 16 * kq;
                        const int slot = (addr / 16) & 15;
                        if (++per_slot[slot] > worst) worst = per_slot[slot];
                    }
                    cost += worst;
                }
                if (cost < best_cost) { best_cost = cost; best_skew = skew; }
            }
            const int plane = base + best_skew;
            // two tile buffers + two tiles of outputs + the lowest tap digits of a 12-step window
            const int lds = 2 * 4 * plane + 2 * cand * kI8PartSlots * 16 + (!rows && ns > 2 ? kI8MaxSteps * 1024 : 0) + 64;   // + the tile queue
            // the helper lanes hold the staged tile in registers, at most 8 octets of samples each
            const bool wide = (s_tile / 4 + 1) / 2 > 256 * 8;
            if ((lds > 160 * 1024 || wide) && cand > 4) continue;
            ncb = cand;
            p->plane_bytes = plane;
            gm.lds_bytes = wide ? 161 * 1024 : lds;         // (reported as "does not fit" below)
            gm.s_gran = (int)(s_tile / 4);                  // quads of samples staged
        }
        gm.a_gran = gm.s_gran;
        gm.n_inst = (gm.a_gran + 255) / 256;
        gm.q = 1; gm.pad = 0;
        if (gm.lds_bytes > 160 * 1024) {
            if (!auto_i8) { sdr::set_error("pipeline 5: tile does not fit in LDS (decim=%d taps=%d)", p->D, p->T); delete p; return SDR_EINVAL; }
            fits = false;
        }
        if (fits && chan_mix && ncb != 8) {
            if (!auto_i8) { sdr::set_error("pipeline 5 with per-channel mixers needs tiles of 1024 outputs (decim=%d taps=%d)", p->D, p->T); delete p; return SDR_EINVAL; }
            fits = false;
        }
        if (fits) {
            best = gm; bestR = ncb / 2; bestNT = 256; best_raw = true;   // 256 finishing lanes, two or four outputs each
            p->mfma_i8 = true;
            p->cpx = chan_mix ? (rows ? 2 : 1) : 0;
        }
    }
    // Short real-input streams (the plan's size hint): when the streaming kernel would put fewer than two
    // waves on a SIMD its launch lasts as long as one lane's walk over the whole filter; the K-split
    // kernel cuts that chain in four (APT stage 2 at cfg5's size: 32 -> 12 us)
    bool want_ksplit = d->pipeline == 6;
    if (d->pipeline == 0 && ksplit_possible && d->size_hint > 0 && !d->outputs_per_lane && !d->threads && d->n_taps >= 64) {
        hipDeviceProp_t prop;
        int ncu = 256;
        if (hipGetDeviceProperties(&prop, p->device) == hipSuccess) ncu = prop.multiProcessorCount;
        const long long chains = d->size_hint / d->decim / 256;            // waves of 64 lanes x 4 outputs
        want_ksplit = chains < 8LL * ncu;
    }
    if (want_ksplit) {
        if ((d->outputs_per_lane && d->outputs_per_lane != 4) || (d->threads && d->threads != 256)) { delete p; sdr::set_error("pipeline 6 has 4 outputs per lane and 256 threads"); return SDR_EINVAL; }
        Geometry gm;
        bool ok = make_geometry(p->T, p->D, 4, 64, 4, false, &gm, false, false, 8);
        if (ok) {
            gm.lds_bytes = gm.a_gran * 16 > 4 * 64 * 4 * 8 ? gm.a_gran * 16 : 4 * 64 * 4 * 8;    // tile, then the partial sums in its place
            ok = gm.lds_bytes <= 160 * 1024;
        }
        if (ok) { best = gm; bestR = 4; bestNT = 64; best_raw = false; p->ksplit = true; }
        else if (d->pipeline == 6) { delete p; sdr::set_error("pipeline 6: tile does not fit in LDS (decim=%d taps=%d)", p->D, p->T); return SDR_EINVAL; }
    }
    if (!bestR && raw_possible && d->pipeline != 2) {
        const int pref_small[4] = {4, 2, 1, 8}, pref_large[4] = {2, 4, 1, 8};
        const int *pref = (p->D * 2 >= 16) ? pref_large : pref_small;
        for (int ri = 0; ri < 4 && !bestR; ++ri) {
            const int R = pref[ri];
            if (d->outputs_per_lane && d->outputs_per_lane != R) continue;
            for (int ni = 0; ni < 3; ++ni) {
                const int NT = NTs[ni];
                if (d->threads && d->threads != NT) continue;
                Geometry gm;
                if (!make_geometry(p->T, p->D, R, NT, 4, false, &gm, planar)) break;
                const bool forced = d->threads != 0;
                const int budget = (forced || ni == 2) ? 160 * 1024 : 56 * 1024;
                if (gm.lds_bytes > budget) continue;
                best = gm; bestR = R; bestNT = NT; best_raw = true;
                break;
            }
        }
    }
    // Float input without mixer: the DMA-filled tile kernel (fir_raw_kernel<R, IN_CF32 | IN_REAL>), explicitly
    // (pipeline 7) or automatically.  From tools/sweep_float_dma.py (twelve shapes, T = 32..511, D = 2..20, all nine
    // R x workgroup geometries, 2^27 samples): complex float runs 1.4-2.2x the persistent float-tile kernel on every
    // shape (64 taps / 10: 3.9 -> 5.6 TB/s) with the SMALLEST number of outputs per lane that gives a lane ten
    // samples of its own -- nothing is converted, so sharing a loaded sample between outputs saves LDS reads only,
    // and wide lanes cost tile bytes; real float gains 12-77 % with four outputs per lane on long windows (the walk
    // is bound by LDS reads there) and the smallest lane of sixteen samples on short ones, and loses 10-15 % where
    // the lane stride exceeds ten granules (D = 15: 195 and 511 taps), which stay on the float-tile kernel.
    if (!bestR && rawf_possible && (d->pipeline == 7 || (d->pipeline == 0 && !d->outputs_per_lane && !d->threads))) {
        const bool cf = d->input == SDR_IN_CF32;
        const int spg = cf ? 2 : 4;
        const bool pinned = d->pipeline == 7;
        int R = 0;
        if (d->outputs_per_lane) R = d->outputs_per_lane;
        else {
            const int own = cf ? 10 : 16;                      // samples a lane should own
            const bool long_real = !cf && p->T >= 16 * p->D;
            for (int ri = (long_real ? 2 : 0); ri < 3 && !R; ++ri)
                if ((p->D * Rs[ri]) % spg == 0 && (p->D * Rs[ri] >= own || ri == 2)) R = Rs[ri];
            for (int ri = 2; ri >= 0 && !R; --ri)
                if ((p->D * Rs[ri]) % spg == 0) R = Rs[ri];
        }
        const bool wanted = pinned || (R && (cf || p->D * R <= 40));
        for (int ni = 0; ni < 3 && R && wanted && !bestR; ++ni) {
            const int NT = NTs[ni];
            if (d->threads && d->threads != NT) continue;
            Geometry gm;
            if (!make_geometry(p->T, p->D, R, NT, spg, false, &gm)) break;
            // automatic choice: two workgroups per CU at least (tiles up to 80 KB); pinned: anything the LDS holds
            if (gm.lds_bytes > ((pinned && (d->threads || ni == 2)) ? 160 * 1024 : 80 * 1024)) continue;
            best = gm; bestR = R; bestNT = NT; best_raw = true; p->rawf = true;
        }
        if (!bestR && pinned) { delete p; sdr::set_error("pipeline 7: no tile geometry for decim=%d taps=%d (outputs_per_lane=%d threads=%d)", d->decim, d->n_taps, d->outputs_per_lane, d->threads); return SDR_EINVAL; }
    }
    // second pass: shapes that fit no preferred tile (large decimation factors, very long filters)
    // take any tile the LDS holds, and more granules per lane than the kernel prefetches
    for (int pass = 0; pass < 2 && !bestR; ++pass) {
        if (d->pipeline == 1 || d->pipeline == 3) continue;
        const bool raw = false;
        const int spg = (d->input == SDR_IN_F32_REAL) ? 4 : 2;     // real input: four samples per LDS granule
        for (int ri = 0; ri < 4; ++ri) {
            const int R = Rs[ri];
            if (d->outputs_per_lane && d->outputs_per_lane != R) continue;
            for (int ni = 0; ni < 3; ++ni) {
                const int NT = NTs[ni];
                if (d->threads && d->threads != NT) continue;
                Geometry gm;
                if (!make_geometry(p->T, p->D, R, NT, spg, !raw, &gm, false, (d->mix_fr != 0.0 || chan_mix) && spg == 2)) break;
                if (gm.n_inst > kSlots && pass == 0) continue;
                // channelizer: a lane carries at most two table entries across the walk (fir_stream_kernel)
                if (chan_mix && ph_hi_entries(gm.s_gran) + 64 > 2 * NT) continue;
                const bool forced = d->threads != 0;
                // automatic choice: float tiles <= 80 KB (two workgroups per CU; three below 53 KB)
                const int budget = (forced || ni == 2 || pass == 1) ? 160 * 1024 : 80 * 1024;
                if (gm.lds_bytes > budget) continue;
                // SIMD cycles per output: one packed FMA per (sample, output), one ds_read_b128
                // per granule
                const double walked = (double)gm.ngran * spg;                  // samples per lane
                double cost = walked * 4.0 + walked / spg * 8.0 / R;         // a 16-byte LDS read per wave: 8 cycles of a pipe four SIMDs share
                if (gm.pad) cost *= 1.1;                                       // padded image: extra index work
                // smaller tiles re-read more halo; real tiles (half the bytes per sample) run best
                // from 64-lane workgroups all the same (nine of twelve swept shapes, up to 25 %)
                if (spg == 4 ? ni != 2 : ni > 0) cost *= 1.05;
                if (R == 8) cost *= 2.0;
                // Tiles with more than 20 granules per lane go to the 32-slot instantiation, whose
                // prefetch set (128 VGPRs for complex float input) spills: measured 1.6-2.7x slower
                // than the best 12- or 20-slot geometry of the same shape (complex float input,
                // T = 64..511, D = 3..20; tools/sweep_float_tiles.py), so it is a last resort.
                // Two workgroups per CU instead of three cost little on long filters.
                if (gm.n_inst > kSlotsReal) cost *= 2.0;
                if (gm.lds_bytes > 53 * 1024) cost *= 1.08;
                if (gm.n_inst > kSlots) cost *= 1.0 + 0.05 * (gm.n_inst - kSlots);   // unprefetched rest of the tile
                if (!bestR || cost < best_cost) { best = gm; bestR = R; bestNT = NT; best_raw = raw; best_cost = cost; }
                if (spg != 4) break;          // complex tiles: the largest workgroup that fits; real tiles: compare all
            }
        }
    }
    if (!bestR) {
        delete p;
        sdr::set_error("no kernel geometry for decim=%d taps=%d (outputs_per_lane=%d threads=%d pipeline=%d)",
                       d->decim, d->n_taps, d->outputs_per_lane, d->threads, d->pipeline);
        return SDR_EINVAL;
    }
    p->R = bestR; p->NT = bestNT; p->raw = best_raw; p->planar = best_raw && planar && !p->mfma && !p->mfma_i8; p->spg = ((best_raw && !p->rawf) || p->input == SDR_IN_F32_REAL) ? 4 : 2;
    p->Tback = best.Tback; p->Lw4 = best.Lw4; p->ngran = best.ngran; p->q = best.q; p->pad = best.pad;
    p->s_gran = best.s_gran; p->a_gran = best.a_gran; p->n_inst = best.n_inst; p->lds_bytes = best.lds_bytes;
    {
        hipDeviceProp_t prop;
        int ncu = 256;
        if (hipGetDeviceProperties(&prop, p->device) == hipSuccess) ncu = prop.multiProcessorCount;
        int per_cu = (160 * 1024) / p->lds_bytes;
        const int wave_cap = 32 / (p->NT / 64);
        if (per_cu > wave_cap) per_cu = wave_cap;
        if (per_cu > 8) per_cu = 8;
        if (per_cu < 1) per_cu = 1;
        if (p->mfma_i8) per_cu = 1;                           // 512 lanes with 256 registers each, one workgroup per CU
        p->stream_blocks = ncu * per_cu;
    }

    // ---- tap tables ---------------------------------------------------------------------------
    // taps rounded to float32; for integer input the exact power-of-two scale (2^-11 int16, 2^-7 int8) is folded in
    const float fold = (p->input == SDR_IN_I16_IQ || p->input == SDR_IN_I16_QI) ? sdr::kQ11
                       : (p->input == SDR_IN_I8_IQ ? sdr::kQ7 : 1.0f);
    std::vector<float> hf(p->T);
    for (int i = 0; i < p->T; ++i) hf[i] = (float)d->taps[i] * fold;
    // gtab[j][r] = h[D*r + Tback - j]  (tap that window offset j contributes to output r);
    // rows past the window (padding + look-ahead step) stay zero
    const size_t rows = (size_t)(best.ngran + best.look) * p->spg;
    // complex tap table (two floats per tap, the mixer folded in): real input on the float kernels, and complex float
    // input with a mixer on the DMA tile
    const bool real_tile = ((!p->raw || p->rawf) && p->input == SDR_IN_F32_REAL) || (p->rawf && p->input == SDR_IN_CF32 && p->mix_fr != 0.0);
    std::vector<float> gt(rows * p->R * (real_tile ? 2 : 1), 0.0f);
    for (size_t j = 0; j < rows; ++j)
        for (int r = 0; r < p->R; ++r) {
            const long long i = (long long)p->D * r + p->Tback - (long long)j;
            if (i < 0 || i >= p->T) continue;
            if (!real_tile) { gt[j * p->R + r] = hf[i]; continue; }
            // complex taps h[i] * exp(-j*2*pi*fr*i), phase reduced in double
            const double ph = p->mix_fr * (double)i, f = ph - std::rint(ph);
            gt[2 * (j * p->R + r)] = (float)(d->taps[i] * std::cos(2.0 * M_PI * f));
            gt[2 * (j * p->R + r) + 1] = (float)(-d->taps[i] * std::sin(2.0 * M_PI * f));
        }
    // mixer phasor tables (see fir_stream_kernel), phases reduced in double: HI[j] = the phasor of offset
    // 64*j, LO[i] of offset i
    auto phasor_tables = [](double fr, int n_hi, std::vector<float> &dst) {
        auto phasor = [&](long long i) {
            const double ph = fr * (double)i;
            const double f = ph - std::rint(ph);
            dst.push_back((float)std::cos(2.0 * M_PI * f));
            dst.push_back((float)std::sin(2.0 * M_PI * f));
        };
        for (int j = 0; j < n_hi; ++j) phasor(64LL * j);
        for (int i = 0; i < 64; ++i) phasor(i);
    };
    if (!p->raw && p->mix_fr != 0.0 && !real_tile) {
        while (gt.size() % 4) gt.push_back(0.0f);                 // 16-byte aligned
        p->ph_off = (int)gt.size();
        p->n_ph_hi = ph_hi_entries(p->s_gran);
        phasor_tables(p->mix_fr, p->n_ph_hi, gt);
    }
    std::vector<float> chan_tab;
    if (chan_mix) p->n_chan_mix = d->n_mix_channels;
    if (chan_mix && !p->cpx) {
        // the channelizer: one pair of tables per channel (frequency_shift / create_freq_rotation per
        // channel, bladerf_apt_demod.cpp:476-495, sdr_functions.h:63-74), ~1 KB each
        p->n_ph_hi = ph_hi_entries(p->s_gran);
        p->chan_tab_stride = p->n_ph_hi + 64;
        p->n_chan_mix = d->n_mix_channels;
        chan_tab.reserve((size_t)p->n_chan_mix * p->chan_tab_stride * 2);
        for (int c = 0; c < p->n_chan_mix; ++c) phasor_tables(d->mix_fr_channels[c], p->n_ph_hi, chan_tab);
    }
    std::vector<float4> at;
    if (p->mfma) {
        // atab[q][lane]: the four taps lane (i = lane % 16, kq = lane / 16) multiplies with window
        // samples 16q + 4kq + 0..3 of its group: H[i][k] = h[D*i + Tb - k]; zero look-ahead chunks at the end
        at.assign((size_t)(best.ngran + kMfmaAhead) * 64, make_float4(0.f, 0.f, 0.f, 0.f));
        for (int q = 0; q < best.ngran; ++q)
            for (int lane = 0; lane < 64; ++lane) {
                float v[4];
                for (int c = 0; c < 4; ++c) {
                    const long long t = (long long)p->D * (lane & 15) + p->Tback - (16LL * q + 4 * (lane >> 4) + c);
                    v[c] = (t >= 0 && t < p->T) ? hf[t] : 0.0f;
                }
                at[(size_t)q * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
            }
    }
    std::vector<uint4> at8, bt8;
    std::vector<float> chan_e;
    if (p->mfma_i8) {
        // taps as 31-bit fixed point q = round(h * 2^F), F from the largest tap, in four balanced
        // base-256 digits; atab8[wave][s][digit][lane] = sixteen digits of H[i][64*ks + 16*kq + c],
        // ks = wave + 4*s, H[i][k] = q[D*i + Tb - k]
        double hmax = 0.0;
        for (int i = 0; i < p->T; ++i) hmax = std::fmax(hmax, std::fabs(d->taps[i]));
        int e2 = 0;
        if (hmax > 0.0) (void)std::frexp(hmax, &e2);           // hmax = m * 2^e2, 0.5 <= m < 1
        const int F = 30 - e2;                                 // |q| <= 2^30
        std::vector<long long> q(p->T);
        for (int i = 0; i < p->T; ++i) q[i] = std::llround(std::ldexp(d->taps[i], F));
        // A tap that is not zero but vanishes here -- it rounds to zero (below 2^-31 of the largest: the
        // numerical zeros of a windowed sinc where fc*t is an integer), or it has a lowest digit only
        // (below 2^-24) and meets a sample of 0..255, whose one byte product is the one left out -- is the
        // only way an output can be an exact 0 in this arithmetic and a tiny vector in the reference's
        // (short of an exact cancellation over the whole window).  Only plans with such a tap pay for
        // i8_zero_fix's look at their zeros.  (SDRHIP_ZFIX_ALWAYS=1 turns the look on for every plan:
        // A/B measurements of its cost; it cannot change a result.)
        static const bool zfix_always = std::getenv("SDRHIP_ZFIX_ALWAYS") != nullptr;
        p->zfix = zfix_always;
        for (int i = 0; i < p->T; ++i) p->zfix |= (std::llabs(q[i]) < 128 && d->taps[i] != 0.0);
        // Q11 samples, the kernel's sums start at weight 2^8 (int16) | Q7 samples, sums from weight 2^0 (int8)
        p->yscale = (float)std::ldexp(1.0, p->input == SDR_IN_I8_IQ ? -(F + 7) : -(F + 11) + 8);
        // sixteen-digit groups of one tap set: dst[(ks * ks_stride + digit) * 64 + lane], row sums added to rowsum[digit][row]
        auto digits_of = [&](const std::vector<long long> &qq, int n_steps, size_t ks_stride, uint4 *dst, int *rowsum) {
            for (int ks = 0; ks < n_steps; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    unsigned char dg[4][16];
                    for (int c = 0; c < 16; ++c) {
                        const long long k = 64LL * ks + 16 * (lane >> 4) + c;
                        const long long t = (long long)p->D * (lane & 15) + p->Tback - k;
                        long long r = (t >= 0 && t < p->T) ? qq[t] : 0;
                        for (int di = 0; di < 4; ++di) {
                            const long long dgt = ((r + 128) & 255) - 128;       // balanced digit in [-128, 127]
                            dg[di][c] = (unsigned char)(dgt & 255);
                            rowsum[(size_t)di * 16 + (lane & 15)] += (int)dgt;
                            r = (r - dgt) / 256;
                        }
                    }
                    for (int di = 0; di < 4; ++di) {
                        uint32_t v[4];
                        for (int u = 0; u < 4; ++u)
                            v[u] = dg[di][4 * u] | ((uint32_t)dg[di][4 * u + 1] << 8) | ((uint32_t)dg[di][4 * u + 2] << 16) | ((uint32_t)dg[di][4 * u + 3] << 24);
                        dst[((size_t)ks * ks_stride + di) * 64 + lane] = make_uint4(v[0], v[1], v[2], v[3]);
                    }
                }
        };
        // bias of lane (j, kq): rows 4*kq + 0..3; digit d meets the low bytes in accumulator d-1 (weight 0,
        // digit 0, is not computed): 128 * rowsum, combined like the kernel's accumulators
        auto bias_of = [](const int *rs, int lane) {
            float bf[4];
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (lane >> 4) + r;
                const float lo = std::fmaf((float)(128 * rs[32 + row]), 256.0f, (float)(128 * rs[16 + row]));
                const float hi = std::fmaf(0.0f, 256.0f, (float)(128 * rs[48 + row]));
                bf[r] = std::fmaf(hi, 65536.0f, lo);
            }
            uint4 u;
            std::memcpy(&u, bf, 16);
            return u;
        };
        if (p->cpx == 2) {
            // the channelizer in row form: per channel [step][digit][lane], sixteen digits of
            // A[(o, c)][(k, c')], row i = lane % 16 = (output o = i / 2, component c = i % 2), k = 32 ks + 16 (kq / 2) + b,
            // c' = kq % 2 (kq = lane / 16), t = D o + Tb - k:  (re, I) Gr[t]  (re, Q) -Gi[t]  (im, I) Gi[t]  (im, Q) Gr[t]
            // with g[i] = h[i] exp(-j 2 pi fr i) (phase reduced in double); the bias from the rows' digit sums as ever
            const int NSC = p->ngran <= 4 ? 4 : 8;             // the kernel instantiation's steps (launch_mfma_i8)
            at8.assign((size_t)p->n_chan_mix * NSC * 4 * 64, make_uint4(0u, 0u, 0u, 0u));
            bt8.assign((size_t)p->n_chan_mix * 64, make_uint4(0u, 0u, 0u, 0u));
            chan_e.resize((size_t)p->n_chan_mix * 2);
            std::vector<long long> qr(p->T), qi(p->T);
            p->zfix = true;      // the discriminator's shortcut (a constant turn of the product) needs every exact zero looked at
            for (int c = 0; c < p->n_chan_mix; ++c) {
                const double fr = d->mix_fr_channels[c];
                for (int i = 0; i < p->T; ++i) {
                    const double ph = fr * (double)i, f = ph - std::rint(ph);
                    qr[i] = std::llround(std::ldexp(d->taps[i] * std::cos(2.0 * M_PI * f), F));
                    qi[i] = std::llround(std::ldexp(-d->taps[i] * std::sin(2.0 * M_PI * f), F));
                }
                std::vector<int> rs(64, 0);                        // [digit][row]
                uint4 *dst = at8.data() + (size_t)c * NSC * 4 * 64;
                for (int ks = 0; ks < NSC; ++ks)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int row = lane & 15, kq = lane >> 4, o = row >> 1, co = row & 1, ci = kq & 1;
                        unsigned char dg[4][16];
                        for (int b = 0; b < 16; ++b) {
                            const long long k = 32LL * ks + 16 * (kq >> 1) + b;
                            const long long t = (long long)p->D * o + p->Tback - k;
                            long long r = 0;
                            if (t >= 0 && t < p->T) r = co == ci ? qr[t] : (co ? qi[t] : -qi[t]);
                            for (int di = 0; di < 4; ++di) {
                                const long long dgt = ((r + 128) & 255) - 128;       // balanced digit in [-128, 127]
                                dg[di][b] = (unsigned char)(dgt & 255);
                                rs[(size_t)di * 16 + row] += (int)dgt;
                                r = (r - dgt) / 256;
                            }
                        }
                        for (int di = 0; di < 4; ++di) {
                            uint32_t v[4];
                            for (int u = 0; u < 4; ++u)
                                v[u] = dg[di][4 * u] | ((uint32_t)dg[di][4 * u + 1] << 8) | ((uint32_t)dg[di][4 * u + 2] << 16) | ((uint32_t)dg[di][4 * u + 3] << 24);
                            dst[((size_t)ks * 4 + di) * 64 + lane] = make_uint4(v[0], v[1], v[2], v[3]);
                        }
                    }
                for (int lane = 0; lane < 64; ++lane) bt8[(size_t)c * 64 + lane] = bias_of(rs.data(), lane);
                const double phd = fr * (double)p->D, fd = phd - std::rint(phd);
                chan_e[2 * (size_t)c] = (float)std::cos(2.0 * M_PI * fd);
                chan_e[2 * (size_t)c + 1] = (float)std::sin(2.0 * M_PI * fd);
            }
        } else if (!p->cpx) {
            std::vector<int> rowsum((size_t)4 * 16, 0);            // [digit][row]
            at8.assign((size_t)kI8MaxSteps * 4 * 64, make_uint4(0u, 0u, 0u, 0u));
            digits_of(q, kI8MaxSteps, 4, at8.data(), rowsum.data());
            bt8.assign((size_t)64, make_uint4(0u, 0u, 0u, 0u));
            for (int lane = 0; lane < 64; ++lane) bt8[lane] = bias_of(rowsum.data(), lane);
        } else {
            // the channelizer: per channel the digits of Re g and Im g, g[i] = h[i] * exp(-j*2*pi*fr*i) (phase
            // reduced in double), laid out [channel][step][part][digit][lane]; the bias of an even lane (real
            // output = Gr*Xi - Gi*Xq) comes from the row sums' difference, of an odd lane from their sum
            constexpr int NSC = 4;
            at8.assign((size_t)p->n_chan_mix * NSC * 2 * 4 * 64, make_uint4(0u, 0u, 0u, 0u));
            bt8.assign((size_t)p->n_chan_mix * 64, make_uint4(0u, 0u, 0u, 0u));
            chan_e.resize((size_t)p->n_chan_mix * 2);
            std::vector<long long> qr(p->T), qi(p->T);
            p->zfix = true;      // the discriminator's shortcut (a constant turn of the product) needs every exact zero looked at
            for (int c = 0; c < p->n_chan_mix; ++c) {
                const double fr = d->mix_fr_channels[c];
                for (int i = 0; i < p->T; ++i) {
                    const double ph = fr * (double)i, f = ph - std::rint(ph);
                    qr[i] = std::llround(std::ldexp(d->taps[i] * std::cos(2.0 * M_PI * f), F));
                    qi[i] = std::llround(std::ldexp(-d->taps[i] * std::sin(2.0 * M_PI * f), F));
                }
                std::vector<int> rs_r(64, 0), rs_i(64, 0), rs(64, 0);
                uint4 *dst = at8.data() + (size_t)c * NSC * 2 * 4 * 64;
                digits_of(qr, NSC, 8, dst, rs_r.data());
                digits_of(qi, NSC, 8, dst + 4 * 64, rs_i.data());
                for (int lane = 0; lane < 64; ++lane) {
                    for (int k = 0; k < 64; ++k) rs[k] = (lane & 1) ? rs_r[k] + rs_i[k] : rs_r[k] - rs_i[k];
                    bt8[(size_t)c * 64 + lane] = bias_of(rs.data(), lane);
                }
                const double phd = fr * (double)p->D, fd = phd - std::rint(phd);
                chan_e[2 * (size_t)c] = (float)std::cos(2.0 * M_PI * fd);
                chan_e[2 * (size_t)c + 1] = (float)std::sin(2.0 * M_PI * fd);
            }
        }
    }
    hipError_t e = sdr::dev_malloc((void **)&p->d_gtab, gt.size() * sizeof(float));
    if (e == hipSuccess && p->mfma_i8) e = sdr::dev_malloc((void **)&p->d_atab8, at8.size() * sizeof(uint4));
    if (e == hipSuccess && p->mfma_i8) e = hipMemcpy(p->d_atab8, at8.data(), at8.size() * sizeof(uint4), hipMemcpyHostToDevice);
    if (e == hipSuccess && p->mfma_i8) e = sdr::dev_malloc((void **)&p->d_btab8, bt8.size() * sizeof(uint4));
    if (e == hipSuccess && p->mfma_i8) e = hipMemcpy(p->d_btab8, bt8.data(), bt8.size() * sizeof(uint4), hipMemcpyHostToDevice);
    if (e == hipSuccess && p->mfma_i8 && p->mode == SDR_MODE_FM) e = sdr::dev_malloc((void **)&p->d_ftaps, hf.size() * sizeof(float));
    if (e == hipSuccess && p->d_ftaps) e = hipMemcpy(p->d_ftaps, hf.data(), hf.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && p->mfma) e = sdr::dev_malloc((void **)&p->d_atab, at.size() * sizeof(float4));
    if (e == hipSuccess && p->mfma) e = hipMemcpy(p->d_atab, at.data(), at.size() * sizeof(float4), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_gtab, gt.data(), gt.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && chan_mix && !p->cpx) e = sdr::dev_malloc((void **)&p->d_chan_tab, chan_tab.size() * sizeof(float));
    if (e == hipSuccess && chan_mix && !p->cpx) e = hipMemcpy(p->d_chan_tab, chan_tab.data(), chan_tab.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && p->cpx) e = sdr::dev_malloc((void **)&p->d_chan_e, chan_e.size() * sizeof(float));
    if (e == hipSuccess && p->cpx) e = hipMemcpy(p->d_chan_e, chan_e.data(), chan_e.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && chan_mix) e = sdr::dev_malloc((void **)&p->d_chan_fr, (size_t)p->n_chan_mix * sizeof(double));
    if (e == hipSuccess && chan_mix) e = hipMemcpy(p->d_chan_fr, d->mix_fr_channels, (size_t)p->n_chan_mix * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        sdr_fir_plan_destroy(p);
        return sdr::hip_fail(e, "plan upload");
    }
    if (p->mfma_i8 && (d->pipeline == 0 || p->cpx)) {
        sdr_fir_desc d2 = *d;
        d2.pipeline = (d->input == SDR_IN_I8_IQ || chan_mix) ? 2 : 1;        // float tiles | raw tile
        if (sdr_fir_plan_create(&d2, &p->tiny) != SDR_OK) p->tiny = nullptr;
    }
    *out = p;
    return SDR_OK;
}

extern "C" int sdr_fir_plan_destroy(sdr_fir_plan *p)
{
    if (!p) return SDR_OK;
    if (p->d_gtab) (void)hipFree(p->d_gtab);
    if (p->d_atab) (void)hipFree(p->d_atab);
    if (p->tiny) (void)sdr_fir_plan_destroy(p->tiny);
    if (p->d_atab8) (void)hipFree(p->d_atab8);
    if (p->d_btab8) (void)hipFree(p->d_btab8);
    if (p->d_i8q) (void)hipFree(p->d_i8q);
    if (p->h_q_done) (void)hipHostFree(p->h_q_done);
    if (p->d_ftaps) (void)hipFree(p->d_ftaps);
    if (p->d_chan_tab) (void)hipFree(p->d_chan_tab);
    if (p->d_chan_fr) (void)hipFree(p->d_chan_fr);
    if (p->d_chan_e) (void)hipFree(p->d_chan_e);
    if (p->d_in) (void)hipFree(p->d_in);
    if (p->d_out) (void)hipFree(p->d_out);
    delete p;
    return SDR_OK;
}

extern "C" size_t sdr_fir_out_count(const sdr_fir_plan *p, size_t n_in)
{
    return p ? (size_t)out_count(p, (long long)n_in) : 0;
}

extern "C" int sdr_fir_plan_info(const sdr_fir_plan *p, int *R, int *NT, int *lds)
{
    if (!p) return SDR_EINVAL;
    if (R) *R = p->R;
    if (NT) *NT = p->mfma_i8 ? 512 : p->NT;       // (pipeline 5: NT is its 256 finishing lanes; the workgroup has 512)
    if (lds) *lds = p->lds_bytes;
    return SDR_OK;
}

extern "C" int sdr_fir_plan_shape(const sdr_fir_plan *p, int *input, int *mode, int *decim, int *n_taps)
{
    if (!p) return SDR_EINVAL;
    if (input) *input = p->input;
    if (mode) *mode = p->mode;
    if (decim) *decim = p->D;
    if (n_taps) *n_taps = p->T;
    return SDR_OK;
}

extern "C" int sdr_fir_plan_kernel(const sdr_fir_plan *p)
{
    return p ? (p->ksplit ? 6 : (p->mfma_i8 ? 5 : (p->mfma ? 4 : (p->planar ? 3 : (p->rawf ? 7 : (p->raw ? 1 : 2)))))) : 0;
}

extern "C" int sdr_fir_execute_dev(sdr_fir_plan *p, const void *d_in, size_t n_in, void *d_out,
                                   size_t out_capacity, size_t *n_out, void *stream)
{
    if (!p || (!d_in && n_in)) { sdr::set_error("null argument"); return SDR_EINVAL; }
    const long long cnt = out_count(p, (long long)n_in);
    if (n_out) *n_out = (size_t)cnt;
    if ((size_t)cnt > out_capacity) { sdr::set_error("output capacity %zu < %lld", out_capacity, cnt); return SDR_ECAPACITY; }
    if (cnt == 0) return SDR_OK;
    if (!d_out) { sdr::set_error("null output"); return SDR_EINVAL; }
    return launch(p, d_in, 0, (long long)n_in, (long long)n_in, 0, cnt, 0, 1, d_out, 0,
                  (hipStream_t)stream);
}

extern "C" int sdr_fir_execute_range_dev(sdr_fir_plan *p, const void *d_in, uint64_t in_first,
                                         uint64_t n_resident, uint64_t n_total,
                                         uint64_t out_begin, uint64_t out_end, void *d_out,
                                         void *stream)
{
    if (!p || !d_in || !d_out) { sdr::set_error("null argument"); return SDR_EINVAL; }
    const long long cnt = out_count(p, (long long)n_total);
    if (out_begin > out_end || (long long)out_end > cnt) { sdr::set_error("output range outside [0, %lld]", cnt); return SDR_EINVAL; }
    if (out_begin == out_end) return SDR_OK;
    // samples the range needs: history halo on the left, one more output on the right for FM
    long long need_lo = (long long)out_begin * p->D - (p->T - 1);
    if (need_lo < 0) need_lo = 0;
    long long need_hi = ((long long)out_end - 1 + (p->mode == SDR_MODE_FM ? 1 : 0)) * p->D + 1;
    if (need_hi > (long long)n_total) need_hi = (long long)n_total;
    if (need_lo < (long long)in_first || need_hi > (long long)(in_first + n_resident)) {
        sdr::set_error("range needs samples [%lld, %lld) but [%llu, %llu) are resident", need_lo,
                       need_hi, (unsigned long long)in_first, (unsigned long long)(in_first + n_resident));
        return SDR_ERANGE;
    }
    return launch(p, d_in, (long long)in_first, (long long)n_resident, (long long)n_total,
                  (long long)out_begin, (long long)out_end, 0, 1, d_out, 0, (hipStream_t)stream);
}

extern "C" int sdr_fir_execute_batched_dev(sdr_fir_plan *p, const void *d_in, size_t n_in,
                                           size_t in_stride, size_t n_channels, void *d_out,
                                           size_t out_stride, void *stream)
{
    if (!p || !d_in || !d_out) { sdr::set_error("null argument"); return SDR_EINVAL; }
    const long long cnt = out_count(p, (long long)n_in);
    if (n_channels > 1 && (in_stride < n_in || out_stride < (size_t)cnt)) { sdr::set_error("channel stride smaller than channel length"); return SDR_EINVAL; }
    return launch(p, d_in, 0, (long long)n_in, (long long)n_in, 0, cnt, (long long)in_stride,
                  n_channels, d_out, (long long)out_stride, (hipStream_t)stream);
}

extern "C" int sdr_fir_execute_channels_dev(sdr_fir_plan *p, const void *d_in, size_t n_in, size_t in_stride,
                                            size_t chan_first, size_t n_channels, void *d_out, size_t out_stride,
                                            void *stream)
{
    if (!p || !d_in || !d_out) { sdr::set_error("null argument"); return SDR_EINVAL; }
    if (p->n_chan_mix <= 0 || p->chan_bcast) { sdr::set_error("the plan has no per-channel mixers (sdr_fir_desc.n_mix_channels)"); return SDR_EINVAL; }
    const long long cnt = out_count(p, (long long)n_in);
    // in_stride 0: every channel reads the same wideband buffer
    if (n_channels > 1 && ((in_stride != 0 && in_stride < n_in) || out_stride < (size_t)cnt)) { sdr::set_error("channel stride smaller than channel length"); return SDR_EINVAL; }
    return launch(p, d_in, 0, (long long)n_in, (long long)n_in, 0, cnt, (long long)in_stride,
                  n_channels, d_out, (long long)out_stride, (hipStream_t)stream, (long long)chan_first);
}

extern "C" int sdr_fir_execute_host(sdr_fir_plan *p, const void *in, size_t n_in, void *out,
                                    size_t out_capacity, size_t *n_out)
{
    if (!p || (!in && n_in)) { sdr::set_error("null argument"); return SDR_EINVAL; }
    const long long cnt = out_count(p, (long long)n_in);
    if (n_out) *n_out = (size_t)cnt;
    if ((size_t)cnt > out_capacity) { sdr::set_error("output capacity %zu < %lld", out_capacity, cnt); return SDR_ECAPACITY; }
    if (cnt == 0) return SDR_OK;
    if (!out) { sdr::set_error("null output"); return SDR_EINVAL; }
    if (int rc = sdr::same_device(p->device)) return rc;
    const size_t bin = n_in * in_elem_bytes(p->input), bout = (size_t)cnt * out_elem_bytes(p->mode);
    if (bin > p->cap_in) {
        if (p->d_in) (void)hipFree(p->d_in);
        p->d_in = nullptr; p->cap_in = 0;
        SDR_HIP(sdr::dev_malloc(&p->d_in, bin));
        p->cap_in = bin;
    }
    if (bout > p->cap_out) {
        if (p->d_out) (void)hipFree(p->d_out);
        p->d_out = nullptr; p->cap_out = 0;
        SDR_HIP(sdr::dev_malloc(&p->d_out, bout));
        p->cap_out = bout;
    }
    SDR_HIP(hipMemcpyAsync(p->d_in, in, bin, hipMemcpyHostToDevice, 0));
    int rc = launch(p, p->d_in, 0, (long long)n_in, (long long)n_in, 0, cnt, 0, 1, p->d_out, 0, 0);
    if (rc != SDR_OK) return rc;
    SDR_HIP(hipMemcpyAsync(out, p->d_out, bout, hipMemcpyDeviceToHost, 0));
    SDR_HIP(hipStreamSynchronize(0));
    return SDR_OK;
}

extern "C" int sdr_fir_time_dev(sdr_fir_plan *p, const void *d_in, size_t n_in, size_t in_stride,
                                size_t n_channels, void *d_out, size_t out_stride, int iters,
                                void *stream, float *ms_per_launch)
{
    if (!p || !d_in || !d_out || iters <= 0 || !ms_per_launch) { sdr::set_error("bad argument"); return SDR_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const long long cnt = out_count(p, (long long)n_in);
    int rc = SDR_OK;
    float ms = 0.f;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, st);
    for (int i = 0; e == hipSuccess && i < iters && rc == SDR_OK; ++i)
        rc = launch(p, d_in, 0, (long long)n_in, (long long)n_in, 0, cnt, (long long)in_stride,
                    n_channels ? n_channels : 1, d_out, (long long)out_stride, st);
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) return sdr::hip_fail(e, "timing events");
    *ms_per_launch = ms / (float)iters;
    return rc;
}
