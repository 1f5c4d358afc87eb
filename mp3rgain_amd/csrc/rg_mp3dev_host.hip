// rg_mp3dev_host.hip -- host orchestration of the split MP3 decoder's device half (rg_mp3dev.hip).
#include <string.h>

#include <vector>

#include "rg_ctx.h"
#include "rg_mp3dev.h"
#include "rg_mp3dev_host.h"

extern "C" {
hipError_t rg_launch_mp3_huffman(const RgMp3DevTables *, const RgMp3DevHuff *, const RgMp3DevTrack *, uint32_t, const RgMp3HuffRec *,
                                 const uint8_t *, rg_mp3_unit *, int16_t *, uint32_t, hipStream_t);
hipError_t rg_launch_mp3_hybrid(const RgMp3DevTables *, const RgMp3DevTrack *, uint32_t, uint32_t, const rg_mp3_unit *,
                                const int16_t *, float *, hipStream_t);
hipError_t rg_launch_mp3_synth(const RgMp3DevTables *, const RgMp3DevTrack *, uint32_t, uint32_t, const float *, hipStream_t);
}

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

int rg_mp3dev_decode(rg_ctx *c, const RgMp3SplitItem *items, size_t n, hipStream_t s) {
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    if (!c->mp3_tab_ready) {
        RgMp3DevTables *tab = new RgMp3DevTables();
        rg_mp3_fill_device_tables(tab);
        hipError_t e = c->d_mp3_tab.reserve(sizeof(RgMp3DevTables));
        if (e == hipSuccess) e = hipMemcpy(c->d_mp3_tab.p, tab, sizeof(RgMp3DevTables), hipMemcpyHostToDevice);
        delete tab;
        RG_HIP(c, e);
        RgMp3DevHuff *hf = new RgMp3DevHuff();
        rg_mp3_fill_device_huff(hf);
        e = c->d_mp3_huff.reserve(sizeof(RgMp3DevHuff));
        if (e == hipSuccess) e = hipMemcpy(c->d_mp3_huff.p, hf, sizeof(RgMp3DevHuff), hipMemcpyHostToDevice);
        delete hf;
        RG_HIP(c, e);
        c->mp3_tab_ready = true;
    }
    const RgMp3DevHuff *d_huff = reinterpret_cast<const RgMp3DevHuff *>(c->d_mp3_huff.p);
    const RgMp3DevTables *d_tab = reinterpret_cast<const RgMp3DevTables *>(c->d_mp3_tab.p);
    for (size_t first = 0; first < n;) {
        uint64_t units = 0;
        size_t last = first;
        while (last < n && (last == first || units + items[last].n_units <= kChunkUnits)) units += items[last++].n_units;
        std::vector<RgMp3DevTrack> tr(last - first);
        uint64_t ub = 0, mainb = 0;
        uint32_t gb = 0, fcb = 0, sb = 0;
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
            t.fc_base = fcb;
            t.synth_base = sb;
            sb += ((t.n_granules + RG_MP3_SYNTH_RUN - 1) / RG_MP3_SYNTH_RUN) * t.channels;  // runs of granules x channels
            t.main_base = mainb;
            ub += it.n_units;
            gb += t.n_granules;
            fcb += (uint32_t)(it.n_units / (it.lsf ? 1u : 2u));  // one Huffman-stage thread per (frame, channel)
            if (it.recs) { any_recs = true; mainb += (it.main_len + 15) & ~(uint64_t)15; }
        }
        if (units) {
            RG_HIP(c, c->d_mp3_is.reserve(units * 576));
            RG_HIP(c, c->d_mp3_units.reserve(units * sizeof(rg_mp3_unit)));
            RG_HIP(c, c->d_mp3_hyb.reserve(units * 2 * 576));
            RG_HIP(c, c->d_mp3_tracks.reserve(tr.size() * sizeof(RgMp3DevTrack)));
            if (any_recs) {
                RG_HIP(c, c->d_mp3_recs.reserve(units * sizeof(RgMp3HuffRec)));
                RG_HIP(c, c->d_mp3_main.reserve(mainb + 64));
                RG_HIP(c, hipMemsetAsync(c->d_mp3_main.p + mainb, 0, 64, s));  // the bit reader looks a few bytes ahead
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
                RG_HIP(c, rg_launch_mp3_huffman(d_tab, d_huff, d_tr, (uint32_t)tr.size(), reinterpret_cast<const RgMp3HuffRec *>(c->d_mp3_recs.p),
                                                c->d_mp3_main.p, reinterpret_cast<rg_mp3_unit *>(c->d_mp3_units.p), c->d_mp3_is.p, fcb, s));
            }
            RG_HIP(c, rg_launch_mp3_hybrid(d_tab, d_tr, (uint32_t)tr.size(), gb, reinterpret_cast<const rg_mp3_unit *>(c->d_mp3_units.p),
                                           c->d_mp3_is.p, c->d_mp3_hyb.p, s));
            RG_HIP(c, rg_launch_mp3_synth(d_tab, d_tr, (uint32_t)tr.size(), sb, c->d_mp3_hyb.p, s));
            // the chunk buffers (and `tr`) are reused by the next chunk
            RG_HIP(c, hipStreamSynchronize(s));
        }
        first = last;
    }
    return RG_OK;
}
