// rg_enqueue.hip -- batch set-up and kernel launches for one enqueue (host side of
// analyze_track_internal's set-up, src/replaygain.rs:848-878, for a whole batch at once).
//
// Stream order of one batch:
//   H2D launch descriptors -> memset(histograms, peaks)
//   variant 2 groups (one per sample rate x sample format x channel count):
//        rg_tm_main_kernel -> rg_tm_fix_kernel
//   variant 1 list (all tracks when forced, otherwise only tracks of the unstable 88.2 kHz row)
//   per-track percentile/result kernel
//   [album] merge kernel
#include <stdlib.h>
#include <string.h>

#include <math.h>

#include <algorithm>

#include "rg_ctx.h"

extern "C" {
hipError_t rg_launch_k1_halo(const RgTrackDev *, uint32_t, uint32_t, const RgCoefDev *, uint32_t *,
                             unsigned long long *, unsigned long long *, hipStream_t);
hipError_t rg_launch_track_results(const uint32_t *, const unsigned long long *, const RgTrackDev *,
                                   const unsigned long long *, rg_track_result *, uint32_t, hipStream_t);
hipError_t rg_launch_album_merge(const uint32_t *, const unsigned long long *, uint32_t, uint32_t *, double *,
                                 hipStream_t);
hipError_t rg_launch_tm_main(int fmt, int nch, const RgTmCoef *, const RgTmGeom *, const RgTmTrack *, uint32_t, uint32_t,
                             double *, uint32_t, double *, uint32_t, uint32_t *, uint32_t *, uint64_t, hipStream_t);
hipError_t rg_launch_tm_fix(int nch, const RgTmGeom *, const RgTmFixTables *, const RgTmTrack *, uint32_t, uint32_t,
                            const double *, uint32_t, const double *, uint32_t, uint32_t *, uint32_t *, uint32_t *,
                            unsigned long long *, uint32_t *, rg_track_result *, hipStream_t);
}

namespace {

const uint32_t kMinSegment = 96;              // shorter segments need more than RG_TM_MAX_ROUNDS doubling rounds

struct TmGroup {
    int rate_idx, fmt, nch;
    std::vector<uint32_t> ids;
    uint64_t frames = 0;
};

int validate(rg_ctx *c, const rg_track_desc *tracks, size_t n, size_t pcm_bytes) {
    for (size_t t = 0; t < n; ++t) {
        const rg_track_desc &d = tracks[t];
        if (rg_rate_index(d.sample_rate) < 0)
            return rg_set_err(c, RG_ERR_UNSUPPORTED_RATE,
                              "Unsupported sample rate: %u Hz. Supported rates: 96000, 88200, 64000, 48000, 44100, "
                              "32000, 24000, 22050, 16000, 12000, 11025, 8000",
                              d.sample_rate);
        if (d.channels == 0) return rg_set_err(c, RG_ERR_INVALID_ARG, "track %zu: channels == 0", t);
        if (d.format > RG_FMT_S32_PLANAR) return rg_set_err(c, RG_ERR_INVALID_ARG, "track %zu: unknown format %u", t, d.format);
        const size_t bps = rg_bytes_per_sample(d.format);
        if (d.offset_bytes % bps) return rg_set_err(c, RG_ERR_INVALID_ARG, "track %zu: offset not sample-aligned", t);
        const uint64_t need = d.offset_bytes + (uint64_t)d.channels * d.frames * bps;
        if (need > pcm_bytes)
            return rg_set_err(c, RG_ERR_INVALID_ARG, "track %zu: extends past the PCM arena (%llu > %zu)", t,
                              (unsigned long long)need, pcm_bytes);
        const uint32_t W = rg_window_samples(d.sample_rate);
        if ((d.frames + W - 1) / W > 0x7FFFFFFFull) return rg_set_err(c, RG_ERR_INVALID_ARG, "track %zu: too long", t);
    }
    return RG_OK;
}

// descriptor every kernel family shares (also what the result kernel reads sample_rate / file_type from)
void fill_common(rg_ctx *c, const rg_track_desc &d, const unsigned char *d_base, uint32_t track_index, RgTrackDev &o) {
    const int ri = rg_rate_index(d.sample_rate);
    const size_t bps = rg_bytes_per_sample(d.format);
    memset(&o, 0, sizeof o);
    o.ch0 = d_base + d.offset_bytes;
    o.ch1 = d.channels >= 2 ? d_base + d.offset_bytes + d.frames * bps : nullptr;
    o.frames = d.frames;
    o.window = rg_window_samples(d.sample_rate);
    o.n_windows = (uint32_t)((d.frames + o.window - 1) / o.window);
    o.coef_idx = (uint32_t)ri;
    o.format = d.format;
    o.sample_rate = d.sample_rate;
    o.file_type = RG_FILE_MP3;
    o.track_index = track_index;
    (void)c;
}

// variant 1 work split for a list of tracks already filled by fill_common
int finish_k1_list(rg_ctx *c, RgTrackDev *list, size_t n, uint32_t *total_items) {
    uint64_t total_windows = 0;
    for (size_t t = 0; t < n; ++t) total_windows += list[t].n_windows;
    uint32_t seg_windows = 1;  // enough work items to fill the chip, long enough to amortise the halo
    if (total_windows > (1u << 18)) {
        uint64_t s = total_windows >> 17;
        seg_windows = (uint32_t)(s > 16 ? 16 : s);
    }
    uint64_t items = 0;
    for (size_t t = 0; t < n; ++t) {
        RgTrackDev &o = list[t];
        // a wave (64 items) never mixes sample rates: the kernel keeps the filter constants in scalar registers
        if (t > 0 && list[t - 1].coef_idx != o.coef_idx) items = (items + 63) & ~(uint64_t)63;
        if (c->design[o.coef_idx].stable) {
            o.seg_windows = seg_windows;
            o.halo = c->design[o.coef_idx].halo_frames;
        } else {  // 88.2 kHz row: the recursion diverges, only the sequential order is defined
            o.seg_windows = o.n_windows ? o.n_windows : 1;
            o.halo = 0xFFFFFFFFu;
        }
        o.n_segments = (o.n_windows + o.seg_windows - 1) / o.seg_windows;
        o.item_base = (uint32_t)items;
        items += o.n_segments;
        if (items > 0x7FFFFFFFull) return rg_set_err(c, RG_ERR_INVALID_ARG, "batch too large");
    }
    *total_items = (uint32_t)items;
    return RG_OK;
}

// ---- variant 2 tables --------------------------------------------------------------------------
int get_tm_tables(rg_ctx *c, int rate_idx, uint32_t L, RgTmDeviceTables **out, uint32_t m = 1) {
    const uint32_t key = ((uint32_t)rate_idx << 24) | (m << 16) | L;
    auto it = c->tm_tables.find(key);
    if (it != c->tm_tables.end()) {
        *out = it->second;
        return it->second->design.ok ? RG_OK : RG_ERR_INVALID_ARG;
    }
    RgTmDeviceTables *tb = new RgTmDeviceTables();
    rg_tm_design(RG_RATE_TABLE[rate_idx], L, &tb->design, m);
    c->tm_tables[key] = tb;
    *out = tb;
    const RgTmDesign &D = tb->design;
    if (!D.ok) return RG_ERR_INVALID_ARG;
    const size_t nT = D.T.size(), nG = D.Gp.size(), nY = D.PhiY.size(), nB = D.PhiB.size();
    const size_t nL = (size_t)D.H10 * 12 + (size_t)(D.L - D.H10) * 2;
    const size_t nST = D.ST.size();  // servo: [L][12] prefix sums of T
    const size_t total = nT + nG + nY + nB + 24 + 12 + 100 + 12 + 8 + nL + 2 + nST;
    std::vector<double> blob(total, 0.0);
    size_t o = 0;
    const size_t oT = o; memcpy(&blob[o], D.T.data(), nT * 8); o += nT;
    const size_t oG = o; memcpy(&blob[o], D.Gp.data(), nG * 8); o += nG;
    const size_t oY = o; if (nY) memcpy(&blob[o], D.PhiY.data(), nY * 8); o += nY;
    const size_t oB = o; if (nB) memcpy(&blob[o], D.PhiB.data(), nB * 8); o += nB;
    const size_t oX = o; memcpy(&blob[o], &D.Xs[0][0], 24 * 8); o += 24;
    const size_t oS = o; memcpy(&blob[o], D.sigma0, 12 * 8); o += 12;
    memcpy(&blob[o], D.Wf, 100 * 8); o += 100;  // [last Gram | PhiY | PhiB | Xs | sigma0 | Wf | ST of a full segment] is one image for the fix-up kernel
    if (nST) memcpy(&blob[o], &D.ST[(size_t)(D.L - 1) * 12], 12 * 8);
    o += 12;
    o = (o + 1) & ~(size_t)1;  // 16-byte alignment of the LDS image
    const size_t oL = o;
    for (uint32_t n = 0; n < D.H10; ++n)
        for (int j = 0; j < 12; ++j) blob[o++] = D.T[(size_t)n * 12 + j];
    for (uint32_t n = D.H10; n < D.L; ++n) { blob[o++] = D.T[(size_t)n * 12 + 10]; blob[o++] = D.T[(size_t)n * 12 + 11]; }
    o = (o + 1) & ~(size_t)1;
    const size_t oST = o;
    if (nST) memcpy(&blob[o], D.ST.data(), nST * 8);
    o += nST;
    RG_HIP(c, hipMalloc((void **)&tb->d_blob, total * 8));
    RG_HIP(c, hipMemcpy(tb->d_blob, blob.data(), total * 8, hipMemcpyHostToDevice));
    RgTmGeom &g = tb->geom;
    g.L = D.L;
    g.W = D.W;
    g.k = D.W / D.L;
    g.H10 = D.H10;
    g.rounds = D.rounds;
    g.rounds_fast = D.rounds_fast;
    g.warm = 1u << D.rounds;
    g.fix_windows = (RG_TM_BLOCK - g.warm) / g.k;
    g.block = rg_tm_choose_block(D.L, D.H10, m);
    g.m = m;
    g.whiten = D.whiten ? 1u : 0u;
    g.servo = D.servo ? 1u : 0u;
    // servo: the affine term of a window of n frames is aff_lin * (v2_end - v2_start) + aff_n * n (rg_tm.h)
    g.aff_lin = D.servo ? 2.0 * D.dinf / D.beta : 0.0;
    g.aff_n = D.servo ? D.dinf * D.dinf : 0.0;
    g.aff_sig = D.servo ? 2.0 * D.dinf : 0.0;
    for (int j = 0; j < 10; ++j) g.tau10[j] = D.tau10[j];
    if (m > 1 && rg_tm_lds_bytes(D.L, D.H10, g.block, m) > RG_TM_LDS_BYTES) {  // multi-window segments run on the LDS path only
        tb->design.ok = false;
        return RG_ERR_INVALID_ARG;
    }
    g.T = tb->d_blob + oT;
    g.Tlds = tb->d_blob + oL;
    tb->fix.Gp = tb->d_blob + oG;
    tb->fix.PhiY = tb->d_blob + oY;
    tb->fix.PhiB = tb->d_blob + oB;
    tb->fix.X = tb->d_blob + oX;
    tb->fix.sigma0 = tb->d_blob + oS;
    tb->fix.ST = nST ? tb->d_blob + oST : nullptr;
    if (g.fix_windows == 0) {
        tb->design.ok = false;
        return RG_ERR_INVALID_ARG;
    }
    return RG_OK;
}

// Segment length for one group.  Every divisor L of the window W (>= kMinSegment) is a legal
// segment; the choice trades arithmetic (all 12 transient moments are live for the first H10 frames
// of a segment, 2 afterwards, so short segments cost up to 40 instead of 30 FP64 ops per sample)
// against how evenly the resulting waves fill 256 CUs x 4 SIMDs x 4 resident waves.
//   cost(L)  = L*30 + min(L, H10)*10 + fixed          [VALU slots per lane; fixed covers the fix-up kernel too]
//   waves(L) = sum over tracks and channels of ceil(nseg / 256) * 4
//   time(L)  ~ ceil(waves / 1024) * cost   when everything is resident at once (waves <= 3072),
//              (waves / 1024 + 1) * cost   otherwise (many rounds, one extra for the ragged tail)
// Consecutive batches overlap across the context's pipeline slots, so `waves` counts n_slots batches.
int choose_tm_tables(rg_ctx *c, const TmGroup &g, const rg_track_desc *tracks, RgTmDeviceTables **out) {
    const uint32_t W = rg_window_samples(RG_RATE_TABLE[g.rate_idx].sample_rate);
    if (c->tune_tm_segment && W % c->tune_tm_segment == 0) {
        const uint32_t m = c->tune_tm_segment == W && c->tune_tm_windows > 1 ? c->tune_tm_windows : 1;
        if (get_tm_tables(c, g.rate_idx, c->tune_tm_segment, out, m) == RG_OK) return RG_OK;
    }
    // The choice depends on the group's track lengths alone (and on the mode): remember it, so that a caller that enqueues
    // the same batch shape again and again does not pay for the candidate sweep (about 270 candidates x tracks) every time.
    uint64_t sig = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { sig = (sig ^ v) * 1099511628211ull; };
    mix((uint64_t)g.rate_idx); mix((uint64_t)g.fmt); mix((uint64_t)g.nch); mix(c->one_shot ? 1 : 0); mix((uint64_t)c->n_slots);
    mix(c->tune_tm_windows); mix(c->tune_tm_target_lanes);
    for (uint32_t id : g.ids) mix(tracks[id].frames);
    {
        auto it = c->tm_choice.find(sig);
        if (it != c->tm_choice.end() && get_tm_tables(c, g.rate_idx, it->second.first, out, it->second.second) == RG_OK) return RG_OK;
    }
    // candidates: (L, 1) for every divisor L of the window, and (W, m) -- a lane runs m whole windows and pays for the
    // transient moments in the first one only (rg_tm.h)
    struct Cand { uint32_t L, m; };
    std::vector<Cand> cand;
    for (uint32_t d = 1; d <= W; ++d)
        if (W % d == 0 && (d >= kMinSegment || d == W)) cand.push_back(Cand{d, 1});
    if (c->tune_tm_windows != 1) {
        // every m is legal (a track's last segment simply holds fewer windows); which one wins is mostly a matter of how
        // well ceil(windows / m) lanes fill whole blocks and whole rounds of the chip.  Up to round 3 m stopped at 16, and a
        // 1000-track batch was 4.58 rounds of indivisible 768-lane blocks quantised to 5; with m free, 1000 three-minute
        // tracks are ONE round (m = 37: 196 000 lanes = 255.2 blocks on 256 CUs).  No lane needs more windows than the
        // longest track has.
        uint64_t longest = 1;
        for (uint32_t id : g.ids) longest = std::max<uint64_t>(longest, (tracks[id].frames + W - 1) / W);
        const uint32_t m_cap = c->tune_tm_windows ? c->tune_tm_windows : 255;  // the table key holds eight bits of m
        const uint32_t m_max = (uint32_t)std::min<uint64_t>(m_cap, longest);
        for (uint32_t m = 2; m <= m_max; ++m) cand.push_back(Cand{W, m});
    }
    auto lanes_of = [&](const Cand &q) {
        uint64_t lanes = 0;
        const uint64_t stride = (uint64_t)q.L * q.m;
        for (uint32_t id : g.ids) lanes += (tracks[id].frames + stride - 1) / stride;
        return lanes;
    };
    if (c->tune_tm_target_lanes) {  // explicit lane target: the longest segment reaching it, else the shortest
        for (size_t i = cand.size(); i-- > 0;)
            if (lanes_of(cand[i]) >= c->tune_tm_target_lanes && get_tm_tables(c, g.rate_idx, cand[i].L, out, cand[i].m) == RG_OK) return RG_OK;
        for (size_t i = 0; i < cand.size(); ++i)
            if (get_tm_tables(c, g.rate_idx, cand[i].L, out, cand[i].m) == RG_OK) return RG_OK;
        return rg_set_err(c, RG_ERR_INVALID_ARG, "no admissible segment length for %u Hz",
                          RG_RATE_TABLE[g.rate_idx].sample_rate);
    }
    // H10 of this rate from the one-window design (the decay length does not depend on L)
    RgTmDeviceTables *full = nullptr;
    uint32_t H10 = W;
    if (get_tm_tables(c, g.rate_idx, W, &full) == RG_OK) H10 = full->design.H10;
    // Cost model.  Per lane: 28 VALU slots per frame (27 FMA + convert), + 2 per frame of the first window (slow pair of
    // moments), + 10 for its first H10 frames (fast block), + a fixed part (prologue, record, the fix-up kernel's share:
    // one 208-byte record per channel read back plus ~400 FMAs; measured 2.4 ms per 10.8 M segments beside 24 ms of main)
    //   waves(L, m) = ceil(sum over tracks and channels of nseg / block) * block / 64
    //   time        ~ ceil(waves / 1024) * cost   when everything is resident at once,
    //                 (waves / 1024 + 1) * cost   otherwise (many rounds, one extra for the ragged tail)
    // Consecutive batches overlap across the context's pipeline slots, so `waves` counts the batches in flight.
    double best = 1e300;
    size_t best_i = 0;
    std::vector<double> score(cand.size(), 1e300);
    for (size_t i = 0; i < cand.size(); ++i) {
        const uint32_t L = cand[i].L, m = cand[i].m;
        const uint32_t Hl = H10 >= L ? L : H10;
        const uint32_t block = rg_tm_choose_block(L, Hl, m);
        const uint64_t stride = (uint64_t)L * m;
        double lanes = 0;
        for (uint32_t id : g.ids) lanes += (double)((tracks[id].frames + stride - 1) / stride) * g.nch;
        double waves = ceil(lanes / block) * (block / 64);  // blocks run through track boundaries: no per-track padding
        // batches in flight = streams when the caller pipelines (rg_enqueue_* / rg_collect); a synchronous entry point has one:
        // its launch's last, ragged round of blocks is not filled by a neighbour, so the windows per lane are chosen to make
        // the rounds come out even (256 three-minute tracks: m = 8 is 300 blocks = two rounds on 256 CUs, m = 10 is 240 = one)
        if (!c->one_shot) waves *= c->n_slots < RG_SLOT_STREAMS ? c->n_slots : RG_SLOT_STREAMS;
        const double cost = (double)stride * 28.0 + (double)L * 2.0 + (double)std::min(L, H10) * 10.0 + 1500.0 + 1500.0 + 60.0 * (m - 1);
        // residency: the LDS image of the response tables + one 4 KiB tile per wave bound the blocks per CU
        const double lds = (double)rg_tm_lds_bytes(L, Hl, block, m);
        if (m > 1 && lds > (double)RG_TM_LDS_BYTES) continue;
        // waves per SIMD that can be resident: three narrow blocks, or one wide block, per CU
        const double blocks_cu = block == RG_TM_BLOCK ? std::max(1.0, std::min(3.0, floor((double)RG_TM_LDS_BYTES / lds))) : (double)block / 256.0;
        const double cap = 1024.0 * blocks_cu;
        // pipelined: the ragged last round of a batch is filled by the batches behind it; what is left is the tail of the
        // whole pipeline, one round shared by the batches in flight (charged in full to every batch until round 3, which
        // made long lanes look a round of their own length worse than they are)
        const double in_flight = c->n_slots < RG_SLOT_STREAMS ? c->n_slots : RG_SLOT_STREAMS;
        double rounds = waves <= cap ? ceil(waves / 1024.0) : waves / 1024.0 + 1.0 / in_flight;
        if (c->one_shot) {
            // one batch in flight: nothing fills its last round, and a block is indivisible -- the busiest CU runs
            // ceil(blocks / 256) of them, each block / 256 waves per SIMD deep (tools/oneshot_sweep.py: 256 three-minute tracks
            // take 5.1 ms at m = 5 or 10, 7.9 ms at m = 8, in step with this count)
            const double blocks = ceil(lanes / block);
            rounds = ceil(blocks / 256.0) * ((double)block / 256.0);
        }
        // FP64 issue efficiency by waves per SIMD (tools/ubench/frame.hip: 188 / 160 / 147 cycles per frame)
        const double wps = std::min(blocks_cu, std::max(1.0, c->one_shot ? rounds : waves / 1024.0));
        double eff = wps >= 3.0 ? 1.0 : (wps >= 2.0 ? 1.09 + (3.0 - wps) * 0.0 : 1.28 - (wps - 1.0) * 0.19);
        // Where the response tables, not the batch, keep a CU at one or two waves per SIMD (96 kHz: H10 = 1212 frames, 135 KB of
        // tables at L = 2400) the kernel loses less than the frame-only microbenchmark: fitted on the 96 kHz sweep of round 5
        // (L = 2400 / 1600 / 1200 at one wave per SIMD against L = 960 / 800 / 600 / 480 at three), 1.10 at one wave.
        if (blocks_cu < 3.0 && (c->one_shot ? rounds : waves / 1024.0) >= 3.0) eff = blocks_cu >= 2.0 ? 1.04 : 1.10;
        // tables that do not fit a CU's LDS (160 KiB) send the whole launch down the generic path
        // rounds * cost covers every lane of the batch whatever the segment stride is, so candidates compare on it directly
        score[i] = rounds * cost * eff * (lds > (double)RG_TM_LDS_BYTES ? 8.0 : 1.0);
        if (c->trace_tm >= 2)
                fprintf(stderr, "[tm]   L %u m %u: H10 %u block %u lds %.0f blocks/CU %.0f waves %.0f rounds %.2f wps %.2f eff %.2f cost/frame %.2f score %.4g\n", L, m, H10,
                        block, lds, blocks_cu, waves, rounds, wps, eff, cost / (double)stride, score[i]);
        if (score[i] < best) { best = score[i]; best_i = i; }
    }
    // try the candidates from the best score on (tiny L can need too many scan rounds)
    std::vector<size_t> order(cand.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return score[a] < score[b]; });
    (void)best_i;
    for (size_t i : order)
        if (get_tm_tables(c, g.rate_idx, cand[i].L, out, cand[i].m) == RG_OK) {
            if (c->trace_tm)
                fprintf(stderr, "[tm] %zu track(s) at %u Hz, %s: L = %u, m = %u (score %.3g; runner-up L = %u, m = %u: %.3g)\n", g.ids.size(),
                        RG_RATE_TABLE[g.rate_idx].sample_rate, c->one_shot ? "one batch in flight" : "pipelined", cand[i].L, cand[i].m, score[i],
                        cand[order.size() > 1 ? order[1] : i].L, cand[order.size() > 1 ? order[1] : i].m, score[order.size() > 1 ? order[1] : i]);
            if (c->tm_choice.size() > 4096) c->tm_choice.clear();
            c->tm_choice[sig] = std::make_pair(cand[i].L, cand[i].m);
            return RG_OK;
        }
    return rg_set_err(c, RG_ERR_INVALID_ARG, "no admissible segment length for %u Hz",
                      RG_RATE_TABLE[g.rate_idx].sample_rate);
}

int timing_begin(rg_ctx *c, hipEvent_t *e1) {
    *e1 = nullptr;
    if (!c->timing) return RG_OK;
    RgSlot &S = c->slot();
    if (S.ev_used == S.ev_pool.size()) {
        hipEvent_t a, b;
        RG_HIP(c, hipEventCreate(&a));
        RG_HIP(c, hipEventCreate(&b));
        S.ev_pool.emplace_back(a, b);
    }
    hipEvent_t e0 = S.ev_pool[S.ev_used].first;
    *e1 = S.ev_pool[S.ev_used].second;
    S.ev_used += 1;
    RG_HIP(c, hipEventRecord(e0, S.stream));
    if (!c->timing_first) c->timing_first = e0;
    return RG_OK;
}

}  // namespace

int rg_validate_batch(rg_ctx *c, const rg_track_desc *tracks, size_t n, size_t pcm_bytes) { return validate(c, tracks, n, pcm_bytes); }

void rg_tm_tables_release(rg_ctx *c) {
    for (auto &kv : c->tm_tables) {
        if (kv.second->d_blob) (void)hipFree(kv.second->d_blob);
        delete kv.second;
    }
    c->tm_tables.clear();
}

int rg_enqueue_impl(rg_ctx *c, const rg_track_desc *tracks, size_t n, const void *d_pcm_base, size_t pcm_bytes,
                    int album) {
    if (!c) return RG_ERR_INVALID_ARG;
    if (n && (!tracks || !d_pcm_base)) return rg_set_err(c, RG_ERR_INVALID_ARG, "null tracks / pcm_base");
    if (n > 0x7FFFFFFFull) return rg_set_err(c, RG_ERR_INVALID_ARG, "too many tracks");
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    // A design is cached per (rate, windows per lane, segment length) -- up to 4 MB of device memory each -- and a long-lived
    // context that sees many batch shapes would keep them all: past kTmTablesMax the cache is emptied at this one safe point
    // (no launch group of this enqueue holds a table yet; the streams are drained because earlier batches may still read
    // theirs) and what is needed is designed again.
    constexpr size_t kTmTablesMax = 48;
    if (c->tm_tables.size() > kTmTablesMax) {
        for (int k = 0; k < RG_MAX_SLOTS; ++k)
            if (c->slots[k].stream) RG_HIP(c, hipStreamSynchronize(c->slots[k].stream));
        rg_tm_tables_release(c);
        c->tm_choice.clear();
    }
    rc = validate(c, tracks, n, pcm_bytes);
    if (rc != RG_OK) return rc;
    // next pipeline slot: its stream orders this batch behind the batch that used the slot before
    c->cur = (c->cur + 1) % c->n_slots;
    RgSlot &S = c->slot();
    hipStream_t s = c->enqueue_stream ? c->enqueue_stream : S.stream;
    if (c->user_dirty) {  // inputs produced on the caller's stream (or by rg_synth_fill_device) must be complete first
        if (c->user_attached) RG_HIP(c, hipEventRecord(c->user_ev, c->user_stream));
        for (int k = 0; k < RG_MAX_SLOTS; ++k) RG_HIP(c, hipStreamWaitEvent(c->slots[k].stream, c->user_ev, 0));
        c->user_dirty = false;
    }
    if (c->enqueue_wait_ev) RG_HIP(c, hipStreamWaitEvent(s, c->enqueue_wait_ev, 0));  // the batch's PCM is still being produced (rg_ctx.h)
    if (S.album_pending) {  // the previous album tail that used this slot's buffers ran on another stream
        RG_HIP(c, hipStreamWaitEvent(s, S.album_done, 0));
        S.album_pending = false;
    }

    // the pinned staging buffers may still feed the previous batch's H2D copies
    if (S.staging_pending) {
        RG_HIP(c, hipEventSynchronize(S.staging_done));
        S.staging_pending = false;
    }

    c->h_tracks.resize(n ? n : 1);
    c->h_k1_tracks.resize(n ? n : 1);
    c->h_tm_tracks.resize(n ? n : 1);
    // histograms, peaks and per-track arrival counters share one allocation: [hist n*12000 | peak n*2 | done n]
    // words, cleared together (by the first main kernel of the batch, or by one memset)
    const size_t acc_words = n * (size_t)(RG_HISTOGRAM_SIZE + 3);
    RG_HIP(c, S.d_hist.reserve(acc_words));
    S.peak_ptr = reinterpret_cast<unsigned long long *>(S.d_hist.p + n * (size_t)RG_HISTOGRAM_SIZE);
    uint32_t *const done_ptr = S.d_hist.p + n * (size_t)(RG_HISTOGRAM_SIZE + 2);
    RG_HIP(c, S.d_results.reserve(n));
    RG_HIP(c, S.h_results.reserve(n));
    if (n > S.d_nonfinite.cap) {  // self-cleaning flags: zero once, when the buffer is (re)allocated
        RG_HIP(c, S.d_nonfinite.reserve(n));
        RG_HIP(c, hipMemsetAsync(S.d_nonfinite.p, 0, S.d_nonfinite.cap * sizeof(uint32_t), s));
    }
    if (n > S.d_imprecise.cap) {
        RG_HIP(c, S.d_imprecise.reserve(n));
        RG_HIP(c, hipMemsetAsync(S.d_imprecise.p, 0, S.d_imprecise.cap * sizeof(uint32_t), s));
    }

    const unsigned char *base = (const unsigned char *)d_pcm_base;
    const bool use_tm = c->kernel_variant != 1;

    // ---- split the batch --------------------------------------------------------------------------
    std::vector<TmGroup> groups;
    size_t n_k1 = 0;
    for (size_t t = 0; t < n; ++t) {
        fill_common(c, tracks[t], base, (uint32_t)t, c->h_tracks[t]);
        const int ri = rg_rate_index(tracks[t].sample_rate);
        // Every stable rate runs on variant 2.  (Until round 3 auto mode kept 64 and 96 kHz on variant 1: there the Yule-Walker
        // poles crowd z = 1, unit DF2T states reach the output with gains of 71 and 478, and with the state carried in DF2T
        // coordinates the moments cancelled to ~1e-8 of their terms -- 14 of 4000 random tracks had a displaced window and the
        // self-check missed 83 of 1200 pathological ones.  The fix-up kernel now carries each block in coordinates in which
        // its Gram matrix is the identity (rg_design.cpp: whitening): 0 of 2227 random 64 / 96 kHz tracks differ, 1 of 2400
        // pathological ones does and is flagged, so the exact repeat covers it like at every other rate.)
        const bool tm_rate = !(c->force_exact.size() == n && c->force_exact[t]);
        if (use_tm && tm_rate && c->design[ri].stable) {
            const int nch = tracks[t].channels >= 2 ? 2 : 1;
            TmGroup *g = nullptr;
            for (auto &q : groups)
                if (q.rate_idx == ri && q.fmt == (int)tracks[t].format && q.nch == nch) { g = &q; break; }
            if (!g) {
                groups.push_back(TmGroup{ri, (int)tracks[t].format, nch, {}, 0});
                g = &groups.back();
            }
            g->ids.push_back((uint32_t)t);
            g->frames += tracks[t].frames;
        } else {
            c->h_k1_tracks[n_k1++] = c->h_tracks[t];
        }
    }
    uint32_t k1_items = 0;
    rc = finish_k1_list(c, c->h_k1_tracks.data(), n_k1, &k1_items);
    if (rc != RG_OK) return rc;

    // ---- variant 2 launch lists -----------------------------------------------------------------------
    struct GroupLaunch {
        RgTmDeviceTables *tb;
        RgTmCoef K;
        size_t list_off, list_n;
        uint32_t main_grid, fix_grid, total_recs, total_windows;
        int fmt, nch;
    };
    std::vector<GroupLaunch> launches;
    size_t tm_off = 0;
    size_t max_rec_doubles = 0, max_win_doubles = 0;
    for (const TmGroup &g : groups) {
        GroupLaunch gl{};
        rc = choose_tm_tables(c, g, tracks, &gl.tb);
        if (rc != RG_OK) return rc;
        const RgTmGeom &geo = gl.tb->geom;
        const rg_rate_coeffs &rcf = RG_RATE_TABLE[g.rate_idx];
        // the power-of-two input scale of the sample format is folded into the feed-forward taps
        // (exact): F32 x 32768 (src/replaygain.rs:969), S16 x 1 (:990), S32 x 32768/2^31 (:1005)
        const double scale = g.fmt == RG_FMT_F32_PLANAR ? 32768.0 : (g.fmt == RG_FMT_S16_PLANAR ? 1.0 : 32768.0 / 2147483648.0);
        const RgTmDesign &des = gl.tb->design;
        // servo form (rg_tm.h): butter b0 is folded into the Yule taps (one rounding per tap, then the exact scale)
        for (int i = 0; i < 11; ++i) {
            gl.K.b[i] = (des.servo ? (double)((long double)des.g * (long double)rcf.yule_b[i]) : rcf.yule_b[i]) * scale;
            gl.K.a[i] = rcf.yule_a[i];
        }
        for (int i = 0; i < 3; ++i) { gl.K.bb[i] = rcf.butter_b[i]; gl.K.ba[i] = rcf.butter_a[i]; }
        gl.K.c0 = 1e-10;
        gl.K.alpha = des.alpha;
        gl.K.beta = des.beta;
        gl.K.aff_lin = geo.aff_lin;
        gl.K.aff_n = geo.aff_n;
        gl.fmt = g.fmt;
        gl.nch = g.nch;
        gl.list_off = tm_off;
        gl.list_n = g.ids.size();
        uint64_t recs = 0, mb = 0, fb = 0, wins = 0;
        const uint64_t seg_stride = (uint64_t)geo.L * geo.m;
        const uint32_t NB = geo.fix_windows * geo.k;
        const uint32_t geom_block = geo.block;
        for (uint32_t id : g.ids) {
            const RgTrackDev &cd = c->h_tracks[id];
            RgTmTrack &o = c->h_tm_tracks[tm_off++];
            o.ch0 = cd.ch0;
            o.ch1 = cd.ch1;
            o.frames = cd.frames;
            o.nseg = (uint32_t)((cd.frames + seg_stride - 1) / seg_stride);
            o.n_windows = cd.n_windows;
            o.rec_base = (uint32_t)recs;
            o.lane_base = (uint32_t)mb;
            o.fix_block_base = (uint32_t)fb;
            o.track_index = id;
            o.fix_blocks = (o.nseg + NB - 1) / NB;
            o.sample_rate = cd.sample_rate;
            o.file_type = cd.file_type;
            o.win_base = (uint32_t)wins;
            wins += cd.n_windows;
            if (o.fix_blocks == 0) {  // empty track: no block will finish it, the result kernel does
                c->h_k1_tracks[n_k1] = cd;
                c->h_k1_tracks[n_k1].n_segments = 0;
                c->h_k1_tracks[n_k1].item_base = k1_items;
                ++n_k1;
            }
            recs += o.nseg;
            mb += (uint64_t)o.nseg * g.nch;  // lanes, not blocks: the main kernel's blocks run through track boundaries
            fb += (o.nseg + NB - 1) / NB;
            if (recs > 0x7FFFFFFFull || mb > 0x7FFFFFFFull || fb > 0x7FFFFFFFull || wins > 0x7FFFFFFFull)
                return rg_set_err(c, RG_ERR_INVALID_ARG, "batch too large");
        }
        gl.total_recs = (uint32_t)recs;
        gl.total_windows = (uint32_t)wins;
        if (geo.m > 1) max_win_doubles = std::max(max_win_doubles, (size_t)wins * g.nch);
        gl.main_grid = (uint32_t)((mb + geom_block - 1) / geom_block);
        gl.fix_grid = (uint32_t)fb;
        max_rec_doubles = std::max(max_rec_doubles, (size_t)recs * RG_TM_REC * g.nch);
        launches.push_back(gl);
    }
    RG_HIP(c, S.d_tm_rec.reserve(max_rec_doubles ? max_rec_doubles : 1));
    if (max_win_doubles) RG_HIP(c, S.d_tm_win.reserve(max_win_doubles));

    S.n_enqueued = n;
    S.enq_album = album;
    S.enq_base = d_pcm_base;
    S.enq_bytes = pcm_bytes;
    S.album_ready = false;
    const RgTrackDev *d_tracks = nullptr, *d_k1_tracks = nullptr;
    const RgTmTrack *d_tm_tracks = nullptr;
    if (n) {
        // ---- one descriptor blob, one H2D copy (skipped when the blob is what the device already holds) ----
        const size_t o_all = 0;
        const size_t o_k1 = (o_all + n * sizeof(RgTrackDev) + 15) & ~(size_t)15;
        const size_t o_tm = (o_k1 + n_k1 * sizeof(RgTrackDev) + 15) & ~(size_t)15;
        const size_t blob_bytes = o_tm + tm_off * sizeof(RgTmTrack);
        const size_t old_desc_cap = S.d_desc.cap;  // (a reallocation may hand the freed address out again: compare capacities)
        RG_HIP(c, S.d_desc.reserve(blob_bytes));
        RG_HIP(c, S.h_desc.reserve(blob_bytes));
        if (S.d_desc.cap != old_desc_cap) S.desc_shadow.clear();
        std::vector<unsigned char> blob(blob_bytes, 0);
        memcpy(&blob[o_all], c->h_tracks.data(), n * sizeof(RgTrackDev));
        if (n_k1) memcpy(&blob[o_k1], c->h_k1_tracks.data(), n_k1 * sizeof(RgTrackDev));
        if (tm_off) memcpy(&blob[o_tm], c->h_tm_tracks.data(), tm_off * sizeof(RgTmTrack));
        if (blob != S.desc_shadow) {
            memcpy(S.h_desc.p, blob.data(), blob_bytes);
            RG_HIP(c, hipMemcpyAsync(S.d_desc.p, S.h_desc.p, blob_bytes, hipMemcpyHostToDevice, s));
            RG_HIP(c, hipEventRecord(S.staging_done, s));
            S.staging_pending = true;
            S.desc_shadow.swap(blob);
        }
        d_tracks = reinterpret_cast<const RgTrackDev *>(S.d_desc.p + o_all);
        (void)d_tracks;
        d_k1_tracks = reinterpret_cast<const RgTrackDev *>(S.d_desc.p + o_k1);
        d_tm_tracks = reinterpret_cast<const RgTmTrack *>(S.d_desc.p + o_tm);
        bool cleared = false;
        bool any_main = false;
        for (const GroupLaunch &gl : launches) any_main = any_main || gl.main_grid != 0;
        if (!any_main) {
            RG_HIP(c, hipMemsetAsync(S.d_hist.p, 0, acc_words * sizeof(uint32_t), s));
            cleared = true;
        }

        for (const GroupLaunch &gl : launches) {
            hipEvent_t e1;
            rc = timing_begin(c, &e1);
            if (rc != RG_OK) return rc;
            RG_HIP(c, rg_launch_tm_main(gl.fmt, gl.nch, &gl.K, &gl.tb->geom, d_tm_tracks + gl.list_off,
                                        (uint32_t)gl.list_n, gl.main_grid, S.d_tm_rec.p, gl.total_recs, S.d_tm_win.p,
                                        gl.total_windows, S.d_nonfinite.p, cleared ? nullptr : S.d_hist.p, (uint64_t)acc_words, s));
            if (gl.main_grid != 0) cleared = true;
            if (e1) RG_HIP(c, hipEventRecord(e1, s));
            RG_HIP(c, rg_launch_tm_fix(gl.nch, &gl.tb->geom, &gl.tb->fix, d_tm_tracks + gl.list_off,
                                       (uint32_t)gl.list_n, gl.fix_grid, S.d_tm_rec.p, gl.total_recs, S.d_tm_win.p,
                                       gl.total_windows, S.d_nonfinite.p, S.d_imprecise.p, S.d_hist.p, S.peak_ptr, done_ptr, S.d_results.p, s));
        }
        if (n_k1) {
            RG_HIP(c, S.d_k1_bad.reserve(n));
            RG_HIP(c, hipMemsetAsync(S.d_k1_bad.p, 0xFF, n * sizeof(unsigned long long), s));  // ~0 = all samples finite
            hipEvent_t e1;
            rc = timing_begin(c, &e1);
            if (rc != RG_OK) return rc;
            RG_HIP(c, rg_launch_k1_halo(d_k1_tracks, (uint32_t)n_k1, k1_items, c->d_coefs.p, S.d_hist.p,
                                        S.peak_ptr, S.d_k1_bad.p, s));
            if (e1) RG_HIP(c, hipEventRecord(e1, s));
        }
        // tracks the fix-up kernel did not finish: variant 1's, and empty ones
        RG_HIP(c, rg_launch_track_results(S.d_hist.p, S.peak_ptr, d_k1_tracks, S.d_k1_bad.p, S.d_results.p, (uint32_t)n_k1, s));
    }
    if (album) {
        RG_HIP(c, rg_launch_album_merge(S.d_hist.p, S.peak_ptr, (uint32_t)n, S.d_album_hist.p,
                                        S.d_album_peak.p, s));
        S.album_ready = true;
        if (c->user_attached) {  // the caller's collective (on its own stream) follows the merge
            RG_HIP(c, hipEventRecord(S.batch_done, s));
            RG_HIP(c, hipStreamWaitEvent(c->user_stream, S.batch_done, 0));
        }
    }
    return RG_OK;
}
