// rg_mp3dev_host.hip -- host orchestration of the split MP3 decoder's device half (rg_mp3dev.hip).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "rg_ctx.h"
#include "rg_mp3dev.h"
#include "rg_mp3dev_host.h"
#include "rg_mp3_frame.h"

extern "C" {
hipError_t rg_launch_mp3_huffman(const RgMp3DevTables *, const RgMp3DevHuff *, const RgMp3DevTrack *, uint32_t, const RgMp3HuffRec *,
                                 const uint8_t *, rg_mp3_unit *, int16_t *, uint64_t, const uint32_t *, const uint32_t *, hipStream_t);
hipError_t rg_launch_mp3_sort(const RgMp3DevTrack *, uint32_t, const RgMp3HuffRec *, uint64_t, uint32_t *, uint32_t *, hipStream_t);
hipError_t rg_launch_mp3_backhalf(const RgMp3DevTables *, const RgMp3DevTrack *, uint32_t, uint32_t, const rg_mp3_unit *,
                                  const int16_t *, uint32_t, hipStream_t);
hipError_t rg_launch_mp3_frames(RgMp3DevTrack *, uint32_t, uint32_t, const uint8_t *, uint32_t *, RgMp3HuffRec *, uint32_t *, hipStream_t);
}

extern "C" int rg_cpu_has_fma(void);  // rg_mp3gain.cpp

namespace {
// units per chunk: 4.6 KB of IMDCT halves + 1.2 KB of input each (256 K units = 1.5 GB); a longer single track gets a
// chunk of its own size
const uint64_t kChunkUnits = 1ull << 18;
}  // namespace

int rg_mp3_rate_row(uint32_t sample_rate) {
    static const uint32_t rates[9] = {44100, 48000, 32000, 22050, 24000, 16000, 11025, 12000, 8000};
    for (int r = 0; r < 9; ++r)
        if (rates[r] == sample_rate) return r;
    return -1;
}

// the lane sort's buffers of staging set `set` for a chunk of `units` units
static int ensure_sort(rg_ctx *c, int set, uint64_t units, hipStream_t s) {
    if (!c->d_mp3_sortw_set[set].p) {
        RG_HIP(c, c->d_mp3_sortw_set[set].reserve(RG_MP3_SORT_WORDS));
        RG_HIP(c, hipMemsetAsync(c->d_mp3_sortw_set[set].p, 0, RG_MP3_SORT_WORDS * sizeof(uint32_t), s));
        RG_HIP(c, hipStreamSynchronize(s));  // the sort may run on another stream than `s`
    }
    RG_HIP(c, c->d_mp3_perm_set[set].reserve(units ? units : 1));
    return RG_OK;
}

static int ensure_tables(rg_ctx *c) {
    if (!c->mp3_tab_ready && !rg_cpu_has_fma()) return rg_set_err(c, RG_ERR_DEVICE, "this build of the MP3 decoder needs a host CPU with FMA3");
    if (!c->mp3_tab_ready) {
        RgMp3DevTables *tab = new RgMp3DevTables();
        rg_mp3_fill_device_tables(tab);
        hipError_t e = c->d_mp3_tab.reserve(sizeof(RgMp3DevTables));
        if (e == hipSuccess) e = hipMemcpy(c->d_mp3_tab.p, tab, sizeof(RgMp3DevTables), hipMemcpyHostToDevice);
        delete tab;
        RG_HIP(c, e);
        RgMp3DevHuff *hf = new RgMp3DevHuff();
        rg_mp3_fill_device_huff(hf);
        if (hf->n_entries == 0xFFFFFFFFu || hf->n_entries + 2 > RG_MP3_HUFF_LDS_ENTRIES) {  // (the sentinel: an entry the 16-bit image cannot hold)
            const uint32_t ne = hf->n_entries;
            delete hf;
            return rg_set_err(c, RG_ERR_DEVICE, "Huffman tables (%u entries) do not fit the kernel's LDS image", ne);
        }
        e = c->d_mp3_huff.reserve(sizeof(RgMp3DevHuff));
        if (e == hipSuccess) e = hipMemcpy(c->d_mp3_huff.p, hf, sizeof(RgMp3DevHuff), hipMemcpyHostToDevice);
        delete hf;
        RG_HIP(c, e);
        c->mp3_tab_ready = true;
    }
    return RG_OK;
}

int rg_mp3dev_decode(rg_ctx *c, const RgMp3SplitItem *items, size_t n, hipStream_t s) {
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    rc = ensure_tables(c);
    if (rc != RG_OK) return rc;
    const RgMp3DevHuff *d_huff = reinterpret_cast<const RgMp3DevHuff *>(c->d_mp3_huff.p);
    const RgMp3DevTables *d_tab = reinterpret_cast<const RgMp3DevTables *>(c->d_mp3_tab.p);
    for (size_t first = 0; first < n;) {
        uint64_t units = 0;
        size_t last = first;
        while (last < n && (last == first || units + items[last].n_units <= kChunkUnits)) units += items[last++].n_units;
        std::vector<RgMp3DevTrack> tr(last - first);
        uint64_t ub = 0, mainb = 0;
        uint32_t gb = 0, hb = 0;
        bool any_recs = false;
        for (size_t i = first; i < last; ++i) {
            const RgMp3SplitItem &it = items[i];
            RgMp3DevTrack &t = tr[i - first];
            memset(&t, 0, sizeof t);
            t.unit_base = ub;
            t.granule_base = gb;
            t.channels = it.channels;
            t.n_granules = (uint32_t)(it.channels ? it.n_units / it.channels : 0);
            t.rate_row = it.rate_row;
            t.lsf = it.lsf;
            t.ch0 = it.d_ch0;
            t.ch1 = it.d_ch1;
            // the back half addresses a lane's PCM as the first plane + 32 bits (rg_mp3dev.hip, wave 3)
            if (t.ch1 && (uint64_t)(t.ch1 - t.ch0) >= (1ull << 30))
                return rg_set_err(c, RG_ERR_INVALID_ARG, "a track of 2^30 frames or more per channel cannot take the device decoder (rg_set_tuning(ctx, 6, 0) selects the host decoder)");
            t.run_base = hb;
            hb += (t.n_granules + RG_MP3_RUN - 1) / RG_MP3_RUN;
            t.main_base = mainb;
            ub += it.n_units;
            gb += t.n_granules;
            if (it.recs) { any_recs = true; mainb += (it.main_len + 15) & ~(uint64_t)15; }
        }
        if (units) {
            RG_HIP(c, c->d_mp3_is.reserve(units * (RG_MP3_ROW_BYTES / 2) + 64));
            RG_HIP(c, c->d_mp3_units.reserve(units * sizeof(rg_mp3_unit)));
            RG_HIP(c, c->d_mp3_tracks.reserve(tr.size() * sizeof(RgMp3DevTrack)));
            if (any_recs) {
                RG_HIP(c, c->d_mp3_recs.reserve(units * sizeof(RgMp3HuffRec)));
                RG_HIP(c, c->d_mp3_main.reserve(mainb + RG_MP3_READ_AHEAD_BYTES));
                RG_HIP(c, hipMemsetAsync(c->d_mp3_main.p + mainb, 0, RG_MP3_READ_AHEAD_BYTES, s));  // the bit reader runs ahead (rg_mp3dev.h)
            }
            for (size_t i = first; i < last; ++i) {
                const RgMp3SplitItem &it = items[i];
                if (!it.n_units) continue;
                const uint64_t off = tr[i - first].unit_base;
                if (it.recs) {  // frame index + main data: the device decodes scalefactors and Huffman itself
                    RG_HIP(c, hipMemcpyAsync(c->d_mp3_recs.p + off * sizeof(RgMp3HuffRec), it.recs, it.n_units * sizeof(RgMp3HuffRec),
                                             hipMemcpyHostToDevice, s));
                    RG_HIP(c, hipMemcpyAsync(c->d_mp3_main.p + tr[i - first].main_base, it.main, it.main_len, hipMemcpyHostToDevice, s));
                    continue;
                }
                RG_HIP(c, hipMemcpyAsync(c->d_mp3_is.p + off * 576, it.is, it.n_units * 576 * sizeof(int16_t), hipMemcpyHostToDevice, s));
                RG_HIP(c, hipMemcpyAsync(c->d_mp3_units.p + off * sizeof(rg_mp3_unit), it.units, it.n_units * sizeof(rg_mp3_unit),
                                         hipMemcpyHostToDevice, s));
            }
            RG_HIP(c, hipMemcpyAsync(c->d_mp3_tracks.p, tr.data(), tr.size() * sizeof(RgMp3DevTrack), hipMemcpyHostToDevice, s));
            const RgMp3DevTrack *d_tr = reinterpret_cast<const RgMp3DevTrack *>(c->d_mp3_tracks.p);
            if (any_recs) {
                // a chunk is all of one kind (the file layer never mixes them); the Huffman stage fills d_mp3_is / d_mp3_units
                rc = ensure_sort(c, 0, ub, s);
                if (rc != RG_OK) return rc;
                RG_HIP(c, rg_launch_mp3_sort(d_tr, (uint32_t)tr.size(), reinterpret_cast<const RgMp3HuffRec *>(c->d_mp3_recs.p), ub,
                                             c->d_mp3_sortw_set[0].p, c->d_mp3_perm_set[0].p, s));
                RG_HIP(c, rg_launch_mp3_huffman(d_tab, d_huff, d_tr, (uint32_t)tr.size(), reinterpret_cast<const RgMp3HuffRec *>(c->d_mp3_recs.p),
                                                c->d_mp3_main.p, reinterpret_cast<rg_mp3_unit *>(c->d_mp3_units.p), c->d_mp3_is.p, ub,
                                                c->d_mp3_perm_set[0].p, c->d_mp3_sortw_set[0].p, s));
            }
            RG_HIP(c, rg_launch_mp3_backhalf(d_tab, d_tr, (uint32_t)tr.size(), hb, reinterpret_cast<const rg_mp3_unit *>(c->d_mp3_units.p),
                                             c->d_mp3_is.p, any_recs ? 1u : 0u, s));
            // the chunk buffers (and `tr`) are reused by the next chunk
            RG_HIP(c, hipStreamSynchronize(s));
        }
        first = last;
    }
    return RG_OK;
}

// ---- tuning key 6 = 3 ------------------------------------------------------------------------------------------------
size_t rg_mp3dev_track_bytes(size_t n_items) { return n_items * sizeof(RgMp3DevTrack); }

int rg_mp3dev_reserve_results(rg_ctx *c, size_t n, hipStream_t s) {
    RG_HIP(c, c->d_mp3_results.reserve(n ? n : 1));
    RG_HIP(c, c->h_mp3_results.reserve(n ? n : 1));
    // a stream without a single frame never reaches the frame parser: its count stays zero
    RG_HIP(c, hipMemsetAsync(c->d_mp3_results.p, 0, (n ? n : 1) * sizeof(uint32_t), s));
    // the frame parser writes the counts from the copy stream, which is not ordered behind `s`: the zeros are in place first
    RG_HIP(c, hipStreamSynchronize(s));
    return RG_OK;
}

int rg_mp3dev_fetch_results(rg_ctx *c, size_t n, hipStream_t s) {
    if (n) RG_HIP(c, hipMemcpyAsync(c->h_mp3_results.p, c->d_mp3_results.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    return RG_OK;
}

const uint32_t *rg_mp3dev_results(rg_ctx *c) { return c->h_mp3_results.p; }

int rg_mp3dev_enqueue_chunk(rg_ctx *c, int set, uint8_t *staging, size_t bytes, size_t tracks_off, hipEvent_t staged,
                            const RgMp3StreamItem *items, size_t n, hipStream_t s, uint32_t *counts_out, size_t n_counts, hipEvent_t counts_ev) {
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    rc = ensure_tables(c);
    if (rc != RG_OK) return rc;
    // The copies (and the frame parser and lane sort behind them) run on the SECOND pipeline stream, the decode on the first (the
    // callers pass it): the context's four pipeline streams are the runtime's four hardware queues, a fifth stream would share
    // one of them -- with whichever stream happens to sit there (rg_files.hip).  Nothing else is in flight on the pipeline
    // streams while files are loaded; what the caller enqueues afterwards on this stream is ordered behind the copies.
    c->mp3_copy_stream = c->slots[1].stream;
    if (s == c->mp3_copy_stream) return rg_set_err(c, RG_ERR_STATE, "the MP3 decode may not run on the copies' stream");
    for (int k = 0; k < 2; ++k)
        if (!c->mp3_set_free[k]) RG_HIP(c, hipEventCreateWithFlags(&c->mp3_set_free[k], hipEventDisableTiming));
    // the descriptors are written into the staging block itself: check that they fit BEFORE writing them
    const size_t total = tracks_off + n * sizeof(RgMp3DevTrack);
    if (total > bytes || (tracks_off & 7) != 0) return rg_set_err(c, RG_ERR_INVALID_ARG, "MP3 staging block: descriptors do not fit");
    // laid out for the upper bound "every walked frame decodes"
    RgMp3DevTrack *tr = reinterpret_cast<RgMp3DevTrack *>(staging + tracks_off);
    uint64_t ub = 0;
    uint32_t gb = 0, tb = 0, hb = 0;
    for (size_t i = 0; i < n; ++i) {
        const RgMp3StreamItem &it = items[i];
        RgMp3DevTrack &t = tr[i];
        memset(&t, 0, sizeof t);
        const uint32_t granules = it.n_frames * (it.lsf ? 1u : 2u);
        t.tile_base = tb;
        t.tiles_base = it.tiles_off;
        tb += (it.n_frames + RG_MP3_FRAME_TILE - 1) / RG_MP3_FRAME_TILE;
        t.unit_base = ub;
        t.granule_base = gb;
        t.n_granules = granules;
        t.channels = it.channels;
        t.rate_row = it.rate_row;
        t.lsf = it.lsf;
        t.ch0 = it.d_ch0;
        t.ch1 = it.channels == 2 ? it.d_ch0 + (size_t)granules * 576 : nullptr;
        if (t.ch1 && (uint64_t)granules * 576 >= (1ull << 30))  // (the planes lie 32 bits apart at most: rg_mp3dev.hip, wave 3)
            return rg_set_err(c, RG_ERR_INVALID_ARG, "a track of 2^30 frames or more per channel cannot take the device decoder (rg_set_tuning(ctx, 6, 0) selects the host decoder)");
        t.main_base = it.main_off;
        t.run_base = hb;
        t.n_frames = it.n_frames;
        t.slots_base = it.slots_off;
        t.result_index = it.result_index;
        ub += (uint64_t)granules * it.channels;
        gb += granules;
        hb += (granules + RG_MP3_RUN - 1) / RG_MP3_RUN;
    }
    // grow-only buffers; growing one frees the old allocation, which waits for the kernels still using it
    RG_HIP(c, c->d_mp3_stage[set].reserve(bytes + RG_MP3_READ_AHEAD_BYTES));  // the main data is the block's first part: the reader runs ahead into the rest or past it
    if (ub) {
        RG_HIP(c, c->d_mp3_is.reserve(ub * (RG_MP3_ROW_BYTES / 2) + 64));
        RG_HIP(c, c->d_mp3_units.reserve(ub * sizeof(rg_mp3_unit)));
        RG_HIP(c, c->d_mp3_recs_set[set].reserve(ub * sizeof(RgMp3HuffRec)));
        RG_HIP(c, c->d_mp3_tiles_set[set].reserve((size_t)tb * 2));
        rc = ensure_sort(c, set, ub, s);
        if (rc != RG_OK) return rc;
    }
    hipStream_t cs = c->mp3_copy_stream;
    hipEvent_t *ev = c->mp3_bench_ev;  // measurement hook (rg_mp3_decode_bench): the kernels' boundaries on their own stream
    const uint8_t *d_chunk = c->d_mp3_stage[set].p;
    RgMp3DevTrack *d_tr = reinterpret_cast<RgMp3DevTrack *>(c->d_mp3_stage[set].p + tracks_off);
    RgMp3HuffRec *d_recs = reinterpret_cast<RgMp3HuffRec *>(c->d_mp3_recs_set[set].p);
    if (c->mp3_set_used[set]) RG_HIP(c, hipStreamWaitEvent(cs, c->mp3_set_free[set], 0));  // the set's previous chunk has been decoded
    RG_HIP(c, hipMemcpyAsync(c->d_mp3_stage[set].p, staging, bytes, hipMemcpyHostToDevice, cs));
    // The frame parser -- three small launches, a block per 256 frames: latency, not work -- follows its block's copy on the copy
    // stream, i.e. it runs beside the Huffman / back-half kernels of the chunk before instead of between them and this chunk's.
    // (Under the measurement hook it stays in line, so that the three stages' times remain what the events around them say.)
    // The lane sort of the Huffman stage (three more small launches) reads the records the parser has just written.
    if (ub && !ev) {
        RG_HIP(c, rg_launch_mp3_frames(d_tr, (uint32_t)n, tb, d_chunk, c->d_mp3_tiles_set[set].p, d_recs, c->d_mp3_results.p, cs));
        RG_HIP(c, rg_launch_mp3_sort(d_tr, (uint32_t)n, d_recs, ub, c->d_mp3_sortw_set[set].p, c->d_mp3_perm_set[set].p, cs));
    }
    if (counts_out && !ev) {  // (the counts of every stream parsed so far: the frame parser runs on this stream, chunk after chunk)
        RG_HIP(c, hipMemcpyAsync(counts_out, c->d_mp3_results.p, n_counts * sizeof(uint32_t), hipMemcpyDeviceToHost, cs));
        RG_HIP(c, hipEventRecord(counts_ev, cs));
    }
    RG_HIP(c, hipEventRecord(staged, cs));
    RG_HIP(c, hipStreamWaitEvent(s, staged, 0));
    if (ub) {
        const RgMp3DevHuff *d_huff = reinterpret_cast<const RgMp3DevHuff *>(c->d_mp3_huff.p);
        const RgMp3DevTables *d_tab = reinterpret_cast<const RgMp3DevTables *>(c->d_mp3_tab.p);
        if (ev) {
            RG_HIP(c, hipEventRecord(ev[0], s));
            RG_HIP(c, rg_launch_mp3_frames(d_tr, (uint32_t)n, tb, d_chunk, c->d_mp3_tiles_set[set].p, d_recs, c->d_mp3_results.p, s));
            RG_HIP(c, rg_launch_mp3_sort(d_tr, (uint32_t)n, d_recs, ub, c->d_mp3_sortw_set[set].p, c->d_mp3_perm_set[set].p, s));
            RG_HIP(c, hipEventRecord(ev[1], s));
        }
        RG_HIP(c, rg_launch_mp3_huffman(d_tab, d_huff, d_tr, (uint32_t)n, d_recs, d_chunk, reinterpret_cast<rg_mp3_unit *>(c->d_mp3_units.p),
                                        c->d_mp3_is.p, ub, c->d_mp3_perm_set[set].p, c->d_mp3_sortw_set[set].p, s));
        if (ev) RG_HIP(c, hipEventRecord(ev[2], s));
        RG_HIP(c, rg_launch_mp3_backhalf(d_tab, d_tr, (uint32_t)n, hb, reinterpret_cast<const rg_mp3_unit *>(c->d_mp3_units.p), c->d_mp3_is.p,
                                         1u, s));
        if (ev) RG_HIP(c, hipEventRecord(ev[3], s));
    }
    RG_HIP(c, hipEventRecord(c->mp3_set_free[set], s));
    c->mp3_set_used[set] = true;
    return RG_OK;
}
