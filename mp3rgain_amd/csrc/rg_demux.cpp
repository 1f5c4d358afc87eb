// rg_demux.cpp -- include/mp3rgain_amd_demux.h: ISO base media sample tables and ADTS, in front of the analysis path.
// ISO/IEC 14496-12 (box structure 4.2, mdhd 8.4.2, hdlr 8.4.3, stsd 8.5.2, stsc 8.7.4, stsz / stz2 8.7.3, stco / co64 8.7.5),
// 14496-1 7.2.6 (ES_Descriptor, DecoderConfigDescriptor), 14496-3 1.6.2.1 (AudioSpecificConfig), 1.A.2 (ADTS).
// Every length is checked against the enclosing box before it is used; a damaged table ends the walk, it never reads past it.
#include "../../include/mp3rgain_amd_demux.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

namespace {
thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint64_t be64(const uint8_t *p) { return ((uint64_t)be32(p) << 32) | be32(p + 4); }
inline uint32_t be16(const uint8_t *p) { return ((uint32_t)p[0] << 8) | p[1]; }

struct Box {
    uint64_t pos, size, hdr;  // absolute position, total size, header bytes
    uint32_t type;
    uint64_t body() const { return pos + hdr; }
    uint64_t end() const { return pos + size; }
};
constexpr uint32_t fcc(const char (&s)[5]) { return ((uint32_t)(uint8_t)s[0] << 24) | ((uint32_t)(uint8_t)s[1] << 16) | ((uint32_t)(uint8_t)s[2] << 8) | (uint8_t)s[3]; }

// the box starting at `pos` inside [pos, limit); false when there is none / it does not fit
bool read_box(const uint8_t *d, uint64_t pos, uint64_t limit, Box *b) {
    if (pos + 8 > limit) return false;
    uint64_t size = be32(d + pos);
    uint64_t hdr = 8;
    if (size == 1) {
        if (pos + 16 > limit) return false;
        size = be64(d + pos + 8);
        hdr = 16;
    } else if (size == 0) {
        size = limit - pos;  // to the end of the enclosing box
    }
    if (size < hdr || size > limit - pos) return false;
    b->pos = pos;
    b->size = size;
    b->hdr = hdr;
    b->type = be32(d + pos + 4);
    return true;
}
// first child of `type` in [start, limit)
bool find_child(const uint8_t *d, uint64_t start, uint64_t limit, uint32_t type, Box *out) {
    uint64_t pos = start;
    Box b;
    while (read_box(d, pos, limit, &b)) {
        if (b.type == type) { *out = b; return true; }
        pos = b.end();
    }
    return false;
}

struct Track {
    rg_mp4_audio_track info{};
    Box stsz{}, stsc{}, stco{};
    bool has_stsz = false, has_stz2 = false, has_stsc = false, has_stco = false, co64 = false;
};

const uint32_t kAscRates[13] = {96000, 88200, 64000, 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000, 7350};

struct BitR {
    const uint8_t *p;
    size_t n, bit = 0;
    uint32_t get(int k) {
        uint32_t v = 0;
        for (int i = 0; i < k; ++i, ++bit) v = (v << 1) | (bit / 8 < n ? (p[bit / 8] >> (7 - bit % 8)) & 1u : 0u);
        return v;
    }
};

// MPEG-4 descriptor header: tag, then a length in up to four 7-bit groups
bool descriptor(const uint8_t *d, uint64_t &pos, uint64_t limit, uint8_t *tag, uint64_t *len) {
    if (pos + 2 > limit) return false;
    *tag = d[pos++];
    uint64_t v = 0;
    for (int i = 0; i < 4; ++i) {
        if (pos >= limit) return false;
        const uint8_t c = d[pos++];
        v = (v << 7) | (c & 0x7F);
        if (!(c & 0x80)) break;
    }
    if (v > limit - pos) return false;
    *len = v;
    return true;
}

void parse_esds(const uint8_t *d, const Box &esds, rg_mp4_audio_track *t) {
    uint64_t pos = esds.body() + 4, limit = esds.end();  // version + flags
    uint8_t tag;
    uint64_t len;
    if (pos > limit || !descriptor(d, pos, limit, &tag, &len) || tag != 0x03) return;
    uint64_t es_end = pos + len;
    if (pos + 3 > es_end) return;
    const uint8_t flags = d[pos + 2];
    pos += 3;
    if (flags & 0x80) pos += 2;                                   // dependsOn_ES_ID
    if (flags & 0x40) { if (pos >= es_end) return; pos += 1 + d[pos]; }  // URL
    if (flags & 0x20) pos += 2;                                   // OCR_ES_Id
    if (pos > es_end || !descriptor(d, pos, es_end, &tag, &len) || tag != 0x04 || len < 13) return;
    const uint64_t dc_end = pos + len;
    t->object_type = d[pos];
    pos += 13;
    if (pos < dc_end && descriptor(d, pos, dc_end, &tag, &len) && tag == 0x05) {
        t->asc_len = (uint32_t)(len < sizeof t->asc ? len : sizeof t->asc);
        memcpy(t->asc, d + pos, t->asc_len);
        BitR br{d + pos, (size_t)len};
        uint32_t aot = br.get(5);
        if (aot == 31) aot = 32 + br.get(6);
        const uint32_t fi = br.get(4);
        const uint32_t rate = fi == 15 ? br.get(24) : (fi < 13 ? kAscRates[fi] : 0);
        const uint32_t cc = br.get(4);
        t->audio_object_type = aot;
        if (rate) t->sample_rate = rate;
        if (cc >= 1 && cc <= 7) t->channels = cc == 7 ? 8 : cc;
    }
}

// the trak's audio description, or false when it is not an audio track this library's reference would decode
bool parse_trak(const uint8_t *d, const Box &trak, Track *t) {
    Box tkhd, mdia, mdhd, hdlr, minf, stbl, stsd;
    if (find_child(d, trak.body(), trak.end(), fcc("tkhd"), &tkhd) && tkhd.size >= tkhd.hdr + 4) {
        const uint8_t ver = d[tkhd.body()];
        const uint64_t off = tkhd.body() + 4 + (ver == 1 ? 16 : 8);
        if (off + 4 <= tkhd.end()) t->info.track_id = be32(d + off);
    }
    if (!find_child(d, trak.body(), trak.end(), fcc("mdia"), &mdia)) return false;
    if (!find_child(d, mdia.body(), mdia.end(), fcc("hdlr"), &hdlr) || hdlr.size < hdlr.hdr + 12) return false;
    if (be32(d + hdlr.body() + 8) != fcc("soun")) return false;
    if (find_child(d, mdia.body(), mdia.end(), fcc("mdhd"), &mdhd) && mdhd.size >= mdhd.hdr + 4) {
        const uint8_t ver = d[mdhd.body()];
        if (ver == 1 && mdhd.size >= mdhd.hdr + 32) {
            t->info.timescale = be32(d + mdhd.body() + 20);
            t->info.duration = be64(d + mdhd.body() + 24);
        } else if (ver == 0 && mdhd.size >= mdhd.hdr + 20) {
            t->info.timescale = be32(d + mdhd.body() + 12);
            t->info.duration = be32(d + mdhd.body() + 16);
        }
    }
    if (!find_child(d, mdia.body(), mdia.end(), fcc("minf"), &minf)) return false;
    if (!find_child(d, minf.body(), minf.end(), fcc("stbl"), &stbl)) return false;
    if (!find_child(d, stbl.body(), stbl.end(), fcc("stsd"), &stsd) || stsd.size < stsd.hdr + 8) return false;
    if (be32(d + stsd.body() + 4) < 1) return false;
    Box entry;
    if (!read_box(d, stsd.body() + 8, stsd.end(), &entry) || entry.size < entry.hdr + 28) return false;
    const uint64_t e = entry.body();
    const uint32_t version = be16(d + e + 8);
    t->info.channels = be16(d + e + 16);
    t->info.sample_rate = be32(d + e + 24) >> 16;
    uint64_t children = e + 28;
    if (version == 1) children += 16;
    else if (version == 2) {  // QuickTime sound description v2
        if (entry.size >= entry.hdr + 64) {
            double sr;
            uint64_t bits = be64(d + e + 32);
            memcpy(&sr, &bits, 8);
            if (sr > 0 && sr < 1e7) t->info.sample_rate = (uint32_t)sr;
            t->info.channels = be32(d + e + 40);
        }
        children += 36;
    }
    if (entry.type == fcc(".mp3")) {
        t->info.codec = RG_CODEC_MP3;
    } else if (entry.type == fcc("mp4a")) {
        Box esds;
        if (children <= entry.end() && find_child(d, children, entry.end(), fcc("esds"), &esds)) parse_esds(d, esds, &t->info);
        const uint32_t oti = t->info.object_type;
        if (oti == 0x40 || (oti >= 0x66 && oti <= 0x68)) t->info.codec = RG_CODEC_AAC;
        else if (oti == 0x69 || oti == 0x6B) t->info.codec = RG_CODEC_MP3;
        else return false;  // another codec in an mp4a entry (or no esds): not one the reference's build decodes
    } else {
        return false;       // alac, ac-3, Opus ...: CODEC_TYPE_NULL in the reference's build (Cargo.toml:24)
    }
    t->has_stsz = find_child(d, stbl.body(), stbl.end(), fcc("stsz"), &t->stsz);
    if (!t->has_stsz) t->has_stz2 = find_child(d, stbl.body(), stbl.end(), fcc("stz2"), &t->stsz);
    t->has_stsc = find_child(d, stbl.body(), stbl.end(), fcc("stsc"), &t->stsc);
    t->has_stco = find_child(d, stbl.body(), stbl.end(), fcc("stco"), &t->stco);
    if (!t->has_stco) { t->has_stco = find_child(d, stbl.body(), stbl.end(), fcc("co64"), &t->stco); t->co64 = t->has_stco; }
    if ((t->has_stsz || t->has_stz2) && t->stsz.size >= t->stsz.hdr + 12) t->info.n_samples = be32(d + t->stsz.body() + 8);
    return true;
}

int audio_tracks(const uint8_t *d, size_t len, std::vector<Track> *out) {
    Box moov;
    if (!find_child(d, 0, len, fcc("moov"), &moov)) return fail(RG_DEMUX_ERR_FORMAT, "no moov box");
    uint64_t pos = moov.body();
    Box b;
    while (read_box(d, pos, moov.end(), &b)) {
        if (b.type == fcc("trak")) {
            Track t;
            if (parse_trak(d, b, &t)) out->push_back(t);
        }
        pos = b.end();
    }
    return RG_DEMUX_OK;
}
}  // namespace

extern "C" const char *rg_demux_last_error(void) { return g_err.c_str(); }

extern "C" int rg_mp4_audio_tracks(const void *data, size_t len, rg_mp4_audio_track *out, size_t cap, size_t *n_audio) {
    if (!data || !n_audio || (cap && !out)) return fail(RG_DEMUX_ERR_ARG, "null argument");
    std::vector<Track> tr;
    const int rc = audio_tracks(static_cast<const uint8_t *>(data), len, &tr);
    if (rc != RG_DEMUX_OK) return rc;
    *n_audio = tr.size();
    for (size_t i = 0; i < tr.size() && i < cap; ++i) out[i] = tr[i].info;
    return RG_DEMUX_OK;
}

extern "C" int rg_mp4_access_units(const void *data, size_t len, size_t audio_index, uint64_t *offsets, uint32_t *sizes, size_t cap, size_t *n) {
    if (!data || !n) return fail(RG_DEMUX_ERR_ARG, "null argument");
    const uint8_t *d = static_cast<const uint8_t *>(data);
    std::vector<Track> tr;
    int rc = audio_tracks(d, len, &tr);
    if (rc != RG_DEMUX_OK) return rc;
    if (audio_index >= tr.size()) return fail(RG_DEMUX_ERR_RANGE, "audio track %zu of %zu", audio_index, tr.size());
    const Track &t = tr[audio_index];
    *n = 0;
    if (!(t.has_stsz || t.has_stz2) || !t.has_stsc || !t.has_stco) return fail(RG_DEMUX_ERR_FORMAT, "sample table incomplete (stsz / stsc / stco)");
    // sample sizes
    const uint64_t zb = t.stsz.body();
    if (t.stsz.size < t.stsz.hdr + 12) return fail(RG_DEMUX_ERR_FORMAT, "stsz too short");
    uint32_t fixed = 0, field = 32;
    const uint32_t count = be32(d + zb + 8);
    if (t.has_stsz) fixed = be32(d + zb + 4);
    else field = d[zb + 7];
    if (t.has_stz2 && field != 4 && field != 8 && field != 16) return fail(RG_DEMUX_ERR_FORMAT, "stz2 field size %u", field);
    const uint64_t ztab = zb + 12, zend = t.stsz.end();
    auto size_of = [&](uint64_t i, uint32_t *sz) -> bool {
        if (fixed) { *sz = fixed; return true; }
        if (field == 32) { if (ztab + 4 * (i + 1) > zend) return false; *sz = be32(d + ztab + 4 * i); return true; }
        if (field == 16) { if (ztab + 2 * (i + 1) > zend) return false; *sz = be16(d + ztab + 2 * i); return true; }
        if (field == 8) { if (ztab + i + 1 > zend) return false; *sz = d[ztab + i]; return true; }
        if (ztab + i / 2 + 1 > zend) return false;
        *sz = (i & 1) ? (d[ztab + i / 2] & 15) : (d[ztab + i / 2] >> 4);
        return true;
    };
    // chunk runs and offsets
    if (t.stsc.size < t.stsc.hdr + 8 || t.stco.size < t.stco.hdr + 8) return fail(RG_DEMUX_ERR_FORMAT, "stsc / stco too short");
    const uint64_t cb = t.stsc.body(), ob = t.stco.body();
    const uint64_t runs = be32(d + cb + 4), chunks = be32(d + ob + 4);
    if (cb + 8 + runs * 12 > t.stsc.end()) return fail(RG_DEMUX_ERR_FORMAT, "stsc entries run past the box");
    if (ob + 8 + chunks * (t.co64 ? 8u : 4u) > t.stco.end()) return fail(RG_DEMUX_ERR_FORMAT, "chunk offsets run past the box");
    uint64_t sample = 0, run = 0;
    uint64_t total_bytes = 0;  // the samples of one track do not overlap: together they cannot exceed the file
    for (uint64_t c = 1; c <= chunks && sample < count; ++c) {
        while (run + 1 < runs && be32(d + cb + 8 + (run + 1) * 12) <= c) ++run;
        if (runs == 0 || be32(d + cb + 8 + run * 12) > c) continue;  // a chunk before the first run: no samples
        const uint32_t per = be32(d + cb + 8 + run * 12 + 4);
        uint64_t off = t.co64 ? be64(d + ob + 8 + (c - 1) * 8) : be32(d + ob + 8 + (c - 1) * 4);
        for (uint32_t k = 0; k < per && sample < count; ++k, ++sample) {
            uint32_t sz = 0;
            if (!size_of(sample, &sz)) return fail(RG_DEMUX_ERR_FORMAT, "sample sizes run past the box");
            if (off > len || sz > len - off) return RG_DEMUX_OK;  // truncated file: the reader's UnexpectedEof ends the track
            total_bytes += sz;
            // (a crafted table -- fixed size 1, count 2^32 - 1, every chunk at offset 0 -- would otherwise list four billion
            // samples of a 400 KB file and have the caller allocate for them)
            if (total_bytes > len) return fail(RG_DEMUX_ERR_FORMAT, "sample table describes more bytes than the file holds");
            if (*n < cap) {
                if (offsets) offsets[*n] = off;
                if (sizes) sizes[*n] = sz;
            }
            ++*n;
            off += sz;
        }
    }
    return RG_DEMUX_OK;
}

// ---- ADTS ------------------------------------------------------------------------------------------------------------
namespace {
struct AdtsHdr { uint32_t id, profile, fi, cc, frame_len, blocks, hdr_len; };
bool adts_header(const uint8_t *p, size_t avail, AdtsHdr *h) {
    if (avail < 7) return false;
    if (p[0] != 0xFF || (p[1] & 0xF6) != 0xF0) return false;  // syncword 12 bits, layer == 0
    h->id = (p[1] >> 3) & 1;
    const bool crc = !(p[1] & 1);
    h->profile = (p[2] >> 6) & 3;
    h->fi = (p[2] >> 2) & 15;
    h->cc = ((p[2] & 1) << 2) | (p[3] >> 6);
    h->frame_len = ((uint32_t)(p[3] & 3) << 11) | ((uint32_t)p[4] << 3) | (p[5] >> 5);
    h->blocks = (p[6] & 3) + 1;
    h->hdr_len = crc ? 9 : 7;
    return h->fi < 13 && h->frame_len >= h->hdr_len;
}
template <typename F>
int adts_walk(const uint8_t *d, size_t len, rg_adts_info *info, F &&on_frame) {
    size_t pos = 0;
    if (len >= 10 && memcmp(d, "ID3", 3) == 0) {
        const size_t sz = ((size_t)(d[6] & 0x7F) << 21) | ((size_t)(d[7] & 0x7F) << 14) | ((size_t)(d[8] & 0x7F) << 7) | (d[9] & 0x7F);
        pos = 10 + sz + ((d[5] & 0x10) ? 10 : 0);
    }
    bool first = true, synced = false;
    AdtsHdr h, nx;
    while (pos + 7 <= len) {
        // In sync (the frame before ended right here) a valid header is a frame.  Out of sync -- the start of the stream,
        // or after junk -- it also takes the next frame's header where this one's length says (or the end of the stream):
        // twelve set bits alone occur in any data.
        bool ok = adts_header(d + pos, len - pos, &h) && pos + h.frame_len <= len;
        if (ok && !synced)
            ok = pos + h.frame_len + 7 > len || (adts_header(d + pos + h.frame_len, len - pos - h.frame_len, &nx) && nx.fi == h.fi);
        if (ok) {
            if (first) {
                info->sample_rate = kAscRates[h.fi];
                info->channels = h.cc == 7 ? 8 : h.cc;
                info->profile = h.profile + 1;
                info->mpeg_version = h.id;
                info->first_frame_offset = pos;
                first = false;
            }
            ++info->frames;
            info->raw_blocks += h.blocks;
            on_frame(pos + h.hdr_len, h.frame_len - h.hdr_len);
            pos += h.frame_len;
            synced = true;
        } else {
            ++pos;
            synced = false;
            if (!first) ++info->junk_bytes;
        }
    }
    return first ? fail(RG_DEMUX_ERR_FORMAT, "no ADTS frame found") : RG_DEMUX_OK;
}
}  // namespace

extern "C" int rg_adts_scan(const void *data, size_t len, rg_adts_info *out) {
    if (!data || !out) return fail(RG_DEMUX_ERR_ARG, "null argument");
    memset(out, 0, sizeof *out);
    return adts_walk(static_cast<const uint8_t *>(data), len, out, [](uint64_t, uint32_t) {});
}

extern "C" int rg_adts_access_units(const void *data, size_t len, uint64_t *offsets, uint32_t *sizes, size_t cap, size_t *n) {
    if (!data || !n) return fail(RG_DEMUX_ERR_ARG, "null argument");
    rg_adts_info info;
    memset(&info, 0, sizeof info);
    *n = 0;
    return adts_walk(static_cast<const uint8_t *>(data), len, &info, [&](uint64_t off, uint32_t sz) {
        if (*n < cap) {
            if (offsets) offsets[*n] = off;
            if (sizes) sizes[*n] = sz;
        }
        ++*n;
    });
}
