// rg_mp3dec.cpp -- MPEG-1 / MPEG-2 / MPEG-2.5 Layer III decoder (host), see include/mp3rgain_amd_dec.h.
//
// the reference hands this job to symphonia 0.5.5 (Cargo.lock:230-311), whose
// source is not in the reference tree; its call sites are src/replaygain.rs:807-822 (probe), :861-863 (decoder),
// :881-904 (packet loop).  This is an independent implementation of the standard's decoding process
// (ISO/IEC 11172-3 clause 2.4.3.4 and Annex B; 13818-3 clause 2.4.3 for the low-sampling-frequency extension),
// written as a chain of per-granule stages so that the data-parallel back half can later run on the GPU:
//
//   A  bitstream (serial per frame): header, side information, bit reservoir, scalefactors, Huffman  -> is[576]
//   B  requantisation                                                                                -> xr[576]
//   C  joint stereo (mid/side, intensity)                                       needs both channels of a granule
//   D  short-block reordering, alias reduction, IMDCT + windowing + overlap-add, frequency inversion
//   E  polyphase synthesis filterbank (matrixing + 512-tap window)                                  -> 576 PCM
//
// Packet semantics follow the reference's loop: one Layer III frame = one packet = 1152 / 576 PCM frames; a frame
// whose main data starts before what the reservoir holds, or whose side information is invalid, is dropped whole
// (decoder.decode -> DecodeError -> continue, :896-899) but still feeds the reservoir; a Xing / Info / VBRI header
// frame is not decoded (the reference's own frame walker skips it too, src/lib.rs:388-408); nothing is trimmed
// (FormatOptions::default(): gapless off).  All arithmetic is f32 like the reference's decoder; tables are
// computed in double at start-up.  Tabulated constants: rg_mp3_tables.h (generated, see its header).
#include "../../include/mp3rgain_amd_dec.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "rg_mp3_tables.h"
#include "rg_mp3dev.h"
#include "rg_mp3_math.h"
#include "rg_mp3_frame.h"

extern "C" int rg_cpu_has_fma(void);  // rg_mp3gain.cpp (built without -mfma)

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

// this file is compiled with -mfma (Makefile): on a CPU without FMA3 it must not be entered
#define RG_NEED_FMA() \
    do { if (!rg_cpu_has_fma()) return fail(RG_MP3DEC_ERR_ARG, "this build of the MP3 decoder needs a CPU with FMA3"); } while (0)

// ---------------------------------------------------------------------------------------------------------------
// frame header (the same field tables as src/lib.rs:152-252; ISO 11172-3 2.4.1.3)
// ---------------------------------------------------------------------------------------------------------------
const uint16_t kBitrateV1L3[16] = {0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 0};
const uint16_t kBitrateV2L3[16] = {0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160, 0};
const uint32_t kRateV1[3] = {44100, 48000, 32000};

struct Header {
    int version;      // 1, 2, 25
    bool lsf;         // MPEG-2 / 2.5: one granule per frame
    bool crc;
    int bitrate_kbps;
    uint32_t rate;
    int rate_row;     // row of kMp3BandLong / kMp3BandShort
    int padding;
    int mode;         // 0 stereo, 1 joint stereo, 2 dual channel, 3 mono
    int mode_ext;     // joint stereo: bit 0 intensity, bit 1 mid/side
    int channels;
    int frame_bytes;  // whole frame including the header
    int side_bytes;
    int samples;      // PCM frames per channel: 1152 or 576
};

bool parse_header(const uint8_t *p, Header *h) {
    if (p[0] != 0xFF || (p[1] & 0xE0) != 0xE0) return false;
    const int ver = (p[1] >> 3) & 3, layer = (p[1] >> 1) & 3;
    if (ver == 1 || layer != 1) return false;  // reserved version; only Layer III (layer bits 01)
    const int br = p[2] >> 4, sr = (p[2] >> 2) & 3;
    if (br == 0 || br == 15 || sr == 3) return false;  // free format is not supported, like the reference's decoder
    h->version = ver == 3 ? 1 : (ver == 2 ? 2 : 25);
    h->lsf = ver != 3;
    h->crc = (p[1] & 1) == 0;
    h->bitrate_kbps = h->lsf ? kBitrateV2L3[br] : kBitrateV1L3[br];
    h->rate = kRateV1[sr] >> (ver == 3 ? 0 : (ver == 2 ? 1 : 2));
    h->rate_row = sr + (ver == 3 ? 0 : (ver == 2 ? 3 : 6));
    h->padding = (p[2] >> 1) & 1;
    h->mode = p[3] >> 6;
    h->mode_ext = (p[3] >> 4) & 3;
    h->channels = h->mode == 3 ? 1 : 2;
    h->samples = h->lsf ? 576 : 1152;
    h->frame_bytes = (h->lsf ? 72 : 144) * h->bitrate_kbps * 1000 / (int)h->rate + h->padding;
    h->side_bytes = h->lsf ? (h->channels == 1 ? 9 : 17) : (h->channels == 1 ? 17 : 32);
    return h->frame_bytes >= 4 + (h->crc ? 2 : 0) + h->side_bytes;
}

// two headers belong to the same stream when version, layer and sampling rate agree
bool same_stream(const uint8_t *a, const uint8_t *b) { return a[1] == b[1] ? ((a[2] ^ b[2]) & 0x0C) == 0 : ((a[1] ^ b[1]) & 0xFE) == 0 && ((a[2] ^ b[2]) & 0x0C) == 0; }

size_t id3v2_size(const uint8_t *d, size_t len) {
    if (len < 10 || memcmp(d, "ID3", 3) != 0 || d[3] == 0xFF || d[4] == 0xFF) return 0;
    if ((d[6] | d[7] | d[8] | d[9]) & 0x80) return 0;
    size_t n = ((size_t)d[6] << 21) | ((size_t)d[7] << 14) | ((size_t)d[8] << 7) | d[9];
    n += 10 + ((d[5] & 0x10) ? 10 : 0);  // footer
    return n <= len ? n : len;
}

// Xing / Info (after the side information) or VBRI (at offset 36): a header frame, not audio
bool is_info_frame(const uint8_t *f, const Header &h) {
    const int off = 4 + h.side_bytes;  // the tag sits where main data would start (no CRC in such frames, in practice)
    if (off + 4 <= h.frame_bytes && (memcmp(f + off, "Xing", 4) == 0 || memcmp(f + off, "Info", 4) == 0)) return true;
    if (h.crc && off + 6 <= h.frame_bytes && (memcmp(f + off + 2, "Xing", 4) == 0 || memcmp(f + off + 2, "Info", 4) == 0)) return true;
    return 36 + 4 <= h.frame_bytes && memcmp(f + 36, "VBRI", 4) == 0;
}

// first position >= pos with a plausible frame: a valid header whose successor (if the data reaches that far) is a
// header of the same stream
size_t find_sync(const uint8_t *d, size_t len, size_t pos, Header *h, bool need_confirm) {
    for (; pos + 4 <= len; ++pos) {
        if (d[pos] != 0xFF || (d[pos + 1] & 0xE0) != 0xE0) continue;
        if (!parse_header(d + pos, h)) continue;
        if (!need_confirm) return pos;
        const size_t nx = pos + (size_t)h->frame_bytes;
        if (nx + 4 > len) return pos;  // last frame of the data (possibly truncated)
        Header h2;
        if (parse_header(d + nx, &h2) && same_stream(d + pos, d + nx)) return pos;
    }
    return len;
}

// ---------------------------------------------------------------------------------------------------------------
// bit reader over a byte buffer (caller guarantees 64 readable bytes past the end)
// ---------------------------------------------------------------------------------------------------------------
struct Bits {
    const uint8_t *p;
    size_t pos, end;  // in bits
    inline uint32_t peek(int n) const {  // n <= 25
        const uint8_t *q = p + (pos >> 3);
        const uint32_t w = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
        return (w << (pos & 7)) >> (32 - n);
    }
    inline uint32_t get(int n) {
        if (n == 0) return 0;
        const uint32_t v = peek(n);
        pos += n;
        return v;
    }
    inline uint32_t get1() {
        const uint32_t v = (p[pos >> 3] >> (7 - (pos & 7))) & 1u;
        ++pos;
        return v;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// tables computed at start-up
// ---------------------------------------------------------------------------------------------------------------
struct HuffLut {
    int primary_bits = 0;
    std::vector<uint32_t> e;  // leaf: len | xy << 8;  link: 0x80000000 | sub_bits | offset << 8
};

struct Tables {
    HuffLut huff[32];
    uint8_t quadA[64];            // 6 peeked bits -> len << 4 | vwxy
    float pow43[8208];
    uint16_t sfb_long[9][23];
    uint16_t sfb_short[9][14];
    float is_ratio_l[7], is_ratio_r[7];   // MPEG-1 intensity stereo, is_pos 0..6
    float cs[8], ca[8];                   // alias reduction butterflies
    float win[4][36];                     // IMDCT windows per block type (2 = short, 12 taps used)
    float imdct36[36][18];
    float imdct12[12][6];
    float sec[32];                        // secants of the 32-point DCT behind the synthesis matrixing (rg_mp3_math.h: Lee's form)
    float dct16[2][16][16];               // [parity][k][i]: the two 16 x 16 cosine matrices of its dense form (rg_mp3_dct32_split)
    float D[512];                         // synthesis window
};

void build_huff(int t, HuffLut &L) {
    const RgMp3HuffSpec &S = kMp3Huff[t];
    if (S.n == 0) return;
    const int cnt = S.n * S.n;
    int maxlen = 0;
    for (int i = 0; i < cnt; ++i) maxlen = S.len[i] > maxlen ? S.len[i] : maxlen;
    const int P = maxlen < 9 ? maxlen : 9;
    L.primary_bits = P;
    L.e.assign((size_t)1 << P, 0);
    // long codes grouped by their first P bits
    std::vector<int> group_max((size_t)1 << P, 0);
    for (int i = 0; i < cnt; ++i)
        if (S.len[i] > P) {
            const int pre = S.code[i] >> (S.len[i] - P);
            if (S.len[i] - P > group_max[pre]) group_max[pre] = S.len[i] - P;
        }
    for (size_t pre = 0; pre < group_max.size(); ++pre)
        if (group_max[pre]) {
            const size_t off = L.e.size();
            L.e.resize(off + ((size_t)1 << group_max[pre]), 0);
            L.e[pre] = 0x80000000u | (uint32_t)group_max[pre] | ((uint32_t)off << 8);
        }
    for (int i = 0; i < cnt; ++i) {
        const int len = S.len[i];
        const uint32_t xy = (uint32_t)(((i / S.n) << 4) | (i % S.n));
        if (len <= P) {
            const uint32_t base = (uint32_t)S.code[i] << (P - len);
            for (uint32_t k = 0; k < (1u << (P - len)); ++k) L.e[base + k] = (uint32_t)len | (xy << 8);
        } else {
            const int pre = S.code[i] >> (len - P);
            const uint32_t link = L.e[pre];
            const int sub = (int)(link & 0xFF), rem = len - P;
            const size_t off = (link >> 8) & 0x7FFFFF;
            const uint32_t base = ((uint32_t)S.code[i] & ((1u << rem) - 1)) << (sub - rem);
            for (uint32_t k = 0; k < (1u << (sub - rem)); ++k) L.e[off + base + k] = (uint32_t)rem | (xy << 8);
        }
    }
}

const Tables &tables() {
    static Tables *T = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        Tables *t = new Tables();
        for (int i = 0; i < 32; ++i) build_huff(i, t->huff[i]);
        memset(t->quadA, 0, sizeof t->quadA);
        for (int v = 0; v < 16; ++v) {
            const int len = kMp3QuadLenA[v];
            const uint32_t base = (uint32_t)kMp3QuadCodeA[v] << (6 - len);
            for (uint32_t k = 0; k < (1u << (6 - len)); ++k) t->quadA[base + k] = (uint8_t)((len << 4) | v);
        }
        for (int i = 0; i < 8208; ++i) t->pow43[i] = (float)pow((double)i, 4.0 / 3.0);
        for (int r = 0; r < 9; ++r) {
            t->sfb_long[r][0] = 0;
            for (int b = 0; b < 22; ++b) t->sfb_long[r][b + 1] = (uint16_t)(t->sfb_long[r][b] + kMp3BandLong[r][b]);
            t->sfb_short[r][0] = 0;
            for (int b = 0; b < 13; ++b) t->sfb_short[r][b + 1] = (uint16_t)(t->sfb_short[r][b] + kMp3BandShort[r][b]);
        }
        // 24 kHz: band 18 starts at line 332 in every encoder in use (and, as far as can be told without its source,
        // in the reference's decoder); the table the constants were read from has the standard's printed 330.
        // Taking 332 keeps real files right; the two lines differ only when bands 17 and 18 carry different scalefactors.
        t->sfb_long[4][18] = 332;
        for (int i = 0; i < 7; ++i) {
            if (i == 6) { t->is_ratio_l[i] = 1.0f; t->is_ratio_r[i] = 0.0f; continue; }
            const double r = tan((double)i * M_PI / 12.0);
            t->is_ratio_l[i] = (float)(r / (1.0 + r));
            t->is_ratio_r[i] = (float)(1.0 / (1.0 + r));
        }
        const double ci[8] = {-0.6, -0.535, -0.33, -0.185, -0.095, -0.041, -0.0142, -0.0037};
        for (int i = 0; i < 8; ++i) {
            const double s = sqrt(1.0 + ci[i] * ci[i]);
            t->cs[i] = (float)(1.0 / s);
            t->ca[i] = (float)(ci[i] / s);
        }
        for (int i = 0; i < 36; ++i) {
            const double sl = sin(M_PI / 36.0 * (i + 0.5));
            t->win[0][i] = (float)sl;
            t->win[1][i] = (float)(i < 18 ? sl : (i < 24 ? 1.0 : (i < 30 ? sin(M_PI / 12.0 * (i - 18 + 0.5)) : 0.0)));
            t->win[3][i] = (float)(i < 6 ? 0.0 : (i < 12 ? sin(M_PI / 12.0 * (i - 6 + 0.5)) : (i < 18 ? 1.0 : sl)));
            t->win[2][i] = (float)(i < 12 ? sin(M_PI / 12.0 * (i + 0.5)) : 0.0);
        }
        for (int i = 0; i < 36; ++i)
            for (int k = 0; k < 18; ++k) t->imdct36[i][k] = (float)cos(M_PI / 72.0 * (2.0 * i + 1.0 + 18.0) * (2.0 * k + 1.0));
        for (int i = 0; i < 12; ++i)
            for (int k = 0; k < 6; ++k) t->imdct12[i][k] = (float)cos(M_PI / 24.0 * (2.0 * i + 1.0 + 6.0) * (2.0 * k + 1.0));
        {
            int at = 0;
            for (int n = 32; n >= 2; n /= 2)
                for (int k = 0; k < n / 2; ++k) t->sec[at++] = (float)(1.0 / (2.0 * cos(M_PI * (2.0 * k + 1.0) / (2.0 * n))));
            t->sec[31] = 0.0f;
        }
        for (int k = 0; k < 16; ++k)
            for (int i = 0; i < 16; ++i) {
                t->dct16[0][k][i] = (float)cos(M_PI * (2.0 * i) * (2.0 * k + 1.0) / 64.0);
                t->dct16[1][k][i] = (float)cos(M_PI * (2.0 * i + 1.0) * (2.0 * k + 1.0) / 64.0);
            }
        for (int i = 0; i <= 256; ++i) t->D[i] = (float)((double)kMp3SynthWindowQ16[i] / 65536.0);
        for (int i = 1; i < 256; ++i) t->D[512 - i] = (i & 63) ? -t->D[i] : t->D[i];
        T = t;
    });
    return *T;
}

// ---------------------------------------------------------------------------------------------------------------
// side information (ISO 11172-3 2.4.1.7; 13818-3 2.4.1.7 for LSF)
// ---------------------------------------------------------------------------------------------------------------
struct Granule {
    int part2_3_length, big_values, global_gain, scalefac_compress;
    int window_switching, block_type, mixed;
    int table_select[3], subblock_gain[3];
    int region0_count, region1_count;
    int preflag, scalefac_scale, count1table;
    // derived
    int long_end, short_start;  // in scalefactor bands
    int region_end[3];          // big_values regions, in spectral lines
};

struct SideInfo {
    int main_data_begin;
    int scfsi[2][4];
    Granule g[2][2];  // [granule][channel]
};

bool parse_side_info(const uint8_t *p, const Header &h, const Tables &T, SideInfo *si) {
    uint8_t padded[32 + 8] = {0};  // the bit reader looks up to 3 bytes ahead; the frame may end with the side information
    memcpy(padded, p, (size_t)h.side_bytes);
    Bits b{padded, 0, (size_t)h.side_bytes * 8};
    const int nch = h.channels, ngr = h.lsf ? 1 : 2;
    memset(si, 0, sizeof *si);
    if (!h.lsf) {
        si->main_data_begin = (int)b.get(9);
        b.get(nch == 1 ? 5 : 3);
        for (int ch = 0; ch < nch; ++ch)
            for (int k = 0; k < 4; ++k) si->scfsi[ch][k] = (int)b.get1();
    } else {
        si->main_data_begin = (int)b.get(8);
        b.get(nch == 1 ? 1 : 2);
    }
    for (int gr = 0; gr < ngr; ++gr)
        for (int ch = 0; ch < nch; ++ch) {
            Granule &g = si->g[gr][ch];
            g.part2_3_length = (int)b.get(12);
            g.big_values = (int)b.get(9);
            g.global_gain = (int)b.get(8);
            g.scalefac_compress = (int)b.get(h.lsf ? 9 : 4);
            g.window_switching = (int)b.get1();
            if (g.big_values > 288) return false;
            if (g.window_switching) {
                g.block_type = (int)b.get(2);
                g.mixed = (int)b.get1();
                if (g.block_type == 0) return false;  // reserved
                for (int k = 0; k < 2; ++k) g.table_select[k] = (int)b.get(5);
                g.table_select[2] = 0;
                for (int k = 0; k < 3; ++k) g.subblock_gain[k] = (int)b.get(3);
                g.region0_count = (g.block_type == 2 && !g.mixed) ? 8 : 7;
                g.region1_count = 20 - g.region0_count;
            } else {
                g.block_type = 0;
                g.mixed = 0;
                for (int k = 0; k < 3; ++k) g.table_select[k] = (int)b.get(5);
                g.region0_count = (int)b.get(4);
                g.region1_count = (int)b.get(3);
            }
            g.preflag = h.lsf ? 0 : (int)b.get1();
            g.scalefac_scale = (int)b.get1();
            g.count1table = (int)b.get1();
            // ---- derived: which bands are long / short, and where the big_values regions end ----
            if (g.block_type == 2) {
                if (g.mixed) {
                    g.long_end = h.rate_row <= 2 ? 8 : 6;  // 36 lines of long bands
                    g.short_start = 3;
                } else {
                    g.long_end = 0;
                    g.short_start = 0;
                }
            } else {
                g.long_end = 22;
                g.short_start = 13;
            }
            const int bv2 = g.big_values * 2;
            int r0, r1;
            if (g.window_switching) {
                // region 0 = 9 short sub-bands (three scalefactor bands x three windows) of a short block, or the
                // first 8 long bands of a start / stop / mixed block; region 1 takes the rest
                if (g.block_type == 2 && !g.mixed) r0 = 3 * T.sfb_short[h.rate_row][3];
                else if (g.block_type == 2) r0 = 3 * T.sfb_short[h.rate_row][3];
                else r0 = T.sfb_long[h.rate_row][8];
                r1 = 576;
            } else {
                const int i0 = g.region0_count + 1, i1 = g.region0_count + g.region1_count + 2;
                r0 = T.sfb_long[h.rate_row][i0 > 22 ? 22 : i0];
                r1 = T.sfb_long[h.rate_row][i1 > 22 ? 22 : i1];
            }
            g.region_end[0] = r0 < bv2 ? r0 : bv2;
            g.region_end[1] = r1 < bv2 ? r1 : bv2;
            g.region_end[2] = bv2;
        }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// stage A: scalefactors and Huffman-coded spectrum
// ---------------------------------------------------------------------------------------------------------------
const uint8_t kSlen[2][16] = {{0, 0, 0, 0, 3, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4}, {0, 1, 2, 3, 0, 1, 2, 3, 1, 2, 3, 1, 2, 3, 2, 3}};
const uint8_t kPretab[22] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 3, 2, 0};
// 13818-3 Table B.? : number of scalefactor bands per slen partition, [set][block kind: long, short, mixed][partition]
const uint8_t kLsfPartitions[6][3][4] = {
    {{6, 5, 5, 5}, {9, 9, 9, 9}, {6, 9, 9, 9}},   {{6, 5, 7, 3}, {9, 9, 12, 6}, {6, 9, 12, 6}},
    {{11, 10, 0, 0}, {18, 18, 0, 0}, {15, 18, 0, 0}}, {{7, 7, 7, 0}, {12, 12, 12, 0}, {6, 15, 12, 0}},
    {{6, 6, 6, 3}, {12, 9, 9, 6}, {6, 12, 9, 6}},  {{8, 8, 5, 0}, {15, 12, 9, 0}, {6, 18, 9, 0}}};

// Scalefactors in one flat array (Channel::sf): long bands first (index = band), then short bands three windows
// per band (index = long_end + 3 * (band - short_start) + window).  `illegal[i]` marks intensity positions that mean
// "not intensity coded" in the LSF syntax (value == 2^slen - 1).
struct ChannelState {
    int sf[2][40];            // per granule (granule 1 may reuse granule 0's through scfsi)
    uint8_t illegal[40];
    float overlap[32][18];
    float V[1024];
    int v_off;
};

void read_scalefactors_v1(Bits &b, const Granule &g, const int scfsi[4], int gr, ChannelState &cs) {
    int *sf = cs.sf[gr];
    const int s1 = kSlen[0][g.scalefac_compress], s2 = kSlen[1][g.scalefac_compress];
    memset(cs.illegal, 0, sizeof cs.illegal);
    if (g.block_type == 2) {
        int i = 0;
        if (g.mixed) {
            for (; i < 8; ++i) sf[i] = (int)b.get(s1);
            for (int band = 3; band < 6; ++band)
                for (int w = 0; w < 3; ++w) sf[i++] = (int)b.get(s1);
            for (int band = 6; band < 12; ++band)
                for (int w = 0; w < 3; ++w) sf[i++] = (int)b.get(s2);
        } else {
            for (int band = 0; band < 6; ++band)
                for (int w = 0; w < 3; ++w) sf[i++] = (int)b.get(s1);
            for (int band = 6; band < 12; ++band)
                for (int w = 0; w < 3; ++w) sf[i++] = (int)b.get(s2);
        }
        for (; i < 40; ++i) sf[i] = 0;
    } else {
        static const int lo[5] = {0, 6, 11, 16, 21};
        for (int k = 0; k < 4; ++k) {
            const int bits = k < 2 ? s1 : s2;
            if (gr == 1 && scfsi[k]) {
                for (int band = lo[k]; band < lo[k + 1]; ++band) sf[band] = cs.sf[0][band];
            } else {
                for (int band = lo[k]; band < lo[k + 1]; ++band) sf[band] = (int)b.get(bits);
            }
        }
        for (int i = 21; i < 40; ++i) sf[i] = 0;
    }
}

void read_scalefactors_lsf(Bits &b, Granule &g, bool intensity_right, ChannelState &cs) {
    int *sf = cs.sf[0];
    int slen[4], set;
    int sfc = g.scalefac_compress;
    g.preflag = 0;
    if (!intensity_right) {
        if (sfc < 400) {
            slen[0] = (sfc >> 4) / 5; slen[1] = (sfc >> 4) % 5; slen[2] = (sfc & 15) >> 2; slen[3] = sfc & 3; set = 0;
        } else if (sfc < 500) {
            sfc -= 400;
            slen[0] = (sfc >> 2) / 5; slen[1] = (sfc >> 2) % 5; slen[2] = sfc & 3; slen[3] = 0; set = 1;
        } else {
            sfc -= 500;
            slen[0] = sfc / 3; slen[1] = sfc % 3; slen[2] = 0; slen[3] = 0; set = 2;
            g.preflag = 1;
        }
    } else {
        sfc >>= 1;
        if (sfc < 180) {
            slen[0] = sfc / 36; slen[1] = (sfc % 36) / 6; slen[2] = (sfc % 36) % 6; slen[3] = 0; set = 3;
        } else if (sfc < 244) {
            sfc -= 180;
            slen[0] = (sfc & 0x3F) >> 4; slen[1] = (sfc & 0xF) >> 2; slen[2] = sfc & 3; slen[3] = 0; set = 4;
        } else {
            sfc -= 244;
            slen[0] = sfc / 3; slen[1] = sfc % 3; slen[2] = 0; slen[3] = 0; set = 5;
        }
    }
    const int kind = g.block_type == 2 ? (g.mixed ? 2 : 1) : 0;
    int i = 0;
    for (int k = 0; k < 4; ++k) {
        const int n = kLsfPartitions[set][kind][k];
        for (int q = 0; q < n; ++q, ++i) {
            const int v = (int)b.get(slen[k]);
            sf[i] = v;
            // a partition without bits transmits position 0 for its bands, which is a legal position (equal gains)
            cs.illegal[i] = (uint8_t)(intensity_right && slen[k] > 0 && v == (1 << slen[k]) - 1 ? 1 : 0);
        }
    }
    for (; i < 40; ++i) { sf[i] = 0; cs.illegal[i] = 0; }
}

// Huffman-coded spectrum of one granule and channel: big_values pairs in up to three regions, then count1 quadruples
// until part2_3_length is used up.  Returns the number of lines decoded (the rest is zero).
int decode_spectrum(Bits &b, const Granule &g, const Tables &T, int is[576]) {
    int line = 0;
    for (int r = 0; r < 3; ++r) {
        const int end = g.region_end[r];
        const int t = g.table_select[r];
        const HuffLut &L = T.huff[t];
        if (L.primary_bits == 0) {
            for (; line < end; ++line) is[line] = 0;
            continue;
        }
        const int linbits = kMp3Linbits[t];
        const int P = L.primary_bits;
        while (line < end) {
            if (b.pos >= b.end) { for (; line < end; ++line) is[line] = 0; break; }
            uint32_t e = L.e[b.peek(P)];
            if (e & 0x80000000u) {
                b.pos += P;
                e = L.e[((e >> 8) & 0x7FFFFF) + b.peek((int)(e & 0xFF))];
            }
            b.pos += e & 0xFF;
            int x = (int)((e >> 12) & 15), y = (int)((e >> 8) & 15);
            if (x) {
                if (linbits && x == 15) x += (int)b.get(linbits);
                if (b.get1()) x = -x;
            }
            if (y) {
                if (linbits && y == 15) y += (int)b.get(linbits);
                if (b.get1()) y = -y;
            }
            is[line++] = x;
            is[line++] = y;
        }
    }
    // count1: quadruples of |value| <= 1
    while (line <= 572 && b.pos < b.end) {
        int v;
        if (g.count1table) {
            v = (int)(~b.get(4)) & 15;
        } else {
            const uint8_t q = T.quadA[b.peek(6)];
            b.pos += q >> 4;
            v = q & 15;
        }
        int q4[4];
        for (int k = 0; k < 4; ++k) {
            q4[k] = (v >> (3 - k)) & 1;
            if (q4[k] && b.get1()) q4[k] = -1;
        }
        if (b.pos > b.end) break;  // the quadruple ran past the granule's bits: it is stuffing, not data
        for (int k = 0; k < 4; ++k) is[line++] = q4[k];
    }
    const int nz = line;
    for (; line < 576; ++line) is[line] = 0;
    return nz;
}

// ---------------------------------------------------------------------------------------------------------------
// stage B: requantisation  xr = sign(is) |is|^(4/3) 2^((global_gain - 210) / 4) 2^(-(mult (sf + preflag pretab)))
//                                                  [short: 2^(-2 subblock_gain)]          (ISO 11172-3 2.4.3.4)
// ---------------------------------------------------------------------------------------------------------------
void requantize(const int is[576], int nz, const Granule &g, const int *sf, const Header &h, const Tables &T, float xr[576]) {
    const uint16_t *bl = T.sfb_long[h.rate_row], *bs = T.sfb_short[h.rate_row];
    const double mult = g.scalefac_scale ? 1.0 : 0.5;
    const double base = 0.25 * (g.global_gain - 210);
    auto band = [&](int lo, int hi, double e) {
        if (lo >= nz) { for (int i = lo; i < hi; ++i) xr[i] = 0.0f; return; }
        const float gain = (float)exp2(e);
        for (int i = lo; i < hi; ++i) {
            const int v = is[i];
            const int a = v < 0 ? -v : v;
            const float m = T.pow43[a > 8207 ? 8207 : a] * gain;
            xr[i] = v < 0 ? -m : m;
        }
    };
    int line_end_long = 0;
    for (int b = 0; b < g.long_end; ++b) {
        const int lo = bl[b], hi = bl[b + 1];
        band(lo, hi, base - mult * (sf[b] + (g.preflag ? kPretab[b] : 0)));
        line_end_long = hi;
    }
    if (g.block_type == 2) {
        int idx = g.long_end;
        int pos = g.mixed ? line_end_long : 0;
        for (int b = g.short_start; b < 13; ++b) {
            const int wd = bs[b + 1] - bs[b];
            for (int w = 0; w < 3; ++w, ++idx) {
                const int s = b < 12 ? sf[idx] : 0;
                band(pos, pos + wd, base - 2.0 * g.subblock_gain[w] - mult * s);
                pos += wd;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// stage C: joint stereo (ISO 11172-3 2.4.3.4 "Stereo processing"; 13818-3 2.4.3.2 for the LSF intensity scale)
// Spectra are still in bitstream order: short bands as [band][window][line].
// ---------------------------------------------------------------------------------------------------------------
void stereo(float xr[2][576], const Granule &g1, const int *sf1, const uint8_t *illegal1, int nz_l, int nz_r,
            const Header &h, const Tables &T) {
    const bool ms = (h.mode_ext & 2) != 0, is_on = (h.mode_ext & 1) != 0;
    float *L = xr[0], *R = xr[1];
    const float isq2 = 0.70710678118654752440f;
    auto ms_range = [&](int lo, int hi) {
        for (int i = lo; i < hi; ++i) {
            const float a = L[i], b = R[i];
            L[i] = (a + b) * isq2;
            R[i] = (a - b) * isq2;
        }
    };
    if (!is_on) {
        if (ms) {
            const int n = nz_l > nz_r ? nz_l : nz_r;
            ms_range(0, n);
        }
        return;
    }
    const uint16_t *bl = T.sfb_long[h.rate_row], *bs = T.sfb_short[h.rate_row];
    // LSF intensity ratios: io = 2^(-1/4) or 2^(-1/2) by the low bit of the right channel's scalefac_compress
    const double io_exp = (g1.scalefac_compress & 1) ? 0.5 : 0.25;
    auto apply_is = [&](int lo, int hi, int pos) {
        float kl, kr;
        if (!h.lsf) {
            kl = T.is_ratio_l[pos];
            kr = T.is_ratio_r[pos];
        } else if (pos == 0) {
            kl = kr = 1.0f;
        } else if (pos & 1) {
            kl = (float)exp2(-io_exp * ((pos + 1) >> 1));
            kr = 1.0f;
        } else {
            kl = 1.0f;
            kr = (float)exp2(-io_exp * (pos >> 1));
        }
        for (int i = lo; i < hi; ++i) {
            const float v = L[i];
            L[i] = v * kl;
            R[i] = v * kr;
        }
    };
    auto all_zero = [&](int lo, int hi) {
        for (int i = lo; i < hi; ++i)
            if (R[i] != 0.0f) return false;
        return true;
    };
    // walk the bands from the top: a band is intensity coded while every band above it (of the same window, for short
    // blocks) has an all-zero right channel and its own position is legal
    bool found_long = false;
    if (g1.block_type == 2) {
        bool found[3] = {false, false, false};
        // offsets of the short bands in bitstream order
        int start[14];
        int pos = g1.mixed ? bl[g1.long_end] : 0;
        for (int b = g1.short_start; b < 13; ++b) { start[b] = pos; pos += 3 * (bs[b + 1] - bs[b]); }
        for (int b = 12; b >= g1.short_start; --b) {
            const int wd = bs[b + 1] - bs[b];
            const int sb = b == 12 ? 11 : b;  // the last band has no scalefactor of its own: it takes the previous band's
            for (int w = 2; w >= 0; --w) {
                const int lo = start[b] + w * wd, hi = lo + wd;
                const int idx = g1.long_end + 3 * (sb - g1.short_start) + w;
                bool intensity = false;
                if (!found[w]) {
                    if (!all_zero(lo, hi)) {
                        found[w] = true;
                    } else {
                        const int p = sf1[idx];
                        intensity = h.lsf ? !illegal1[idx] : p < 7;
                        if (intensity) apply_is(lo, hi, p);
                    }
                }
                if (!intensity && ms) ms_range(lo, hi);
            }
        }
        found_long = found[0] || found[1] || found[2];
    }
    for (int b = g1.long_end - 1; b >= 0; --b) {
        if (g1.block_type == 2 && !g1.mixed) break;
        const int lo = bl[b], hi = bl[b + 1];
        const int sb = b == 21 ? 20 : b;
        bool intensity = false;
        if (!found_long) {
            if (!all_zero(lo, hi)) {
                found_long = true;
            } else {
                const int p = sf1[sb];
                intensity = h.lsf ? !illegal1[sb] : p < 7;
                if (intensity) apply_is(lo, hi, p);
            }
        }
        if (!intensity && ms) ms_range(lo, hi);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// stage D: reorder, alias reduction, IMDCT, overlap-add, frequency inversion -> 18 time slots x 32 subbands
// ---------------------------------------------------------------------------------------------------------------
void reorder_short(float xr[576], const Granule &g, const Header &h, const Tables &T) {
    const uint16_t *bl = T.sfb_long[h.rate_row], *bs = T.sfb_short[h.rate_row];
    float tmp[576];
    int pos = g.mixed ? bl[g.long_end] : 0;
    const int first = pos;
    for (int b = g.short_start; b < 13; ++b) {
        const int wd = bs[b + 1] - bs[b];
        for (int w = 0; w < 3; ++w)
            for (int i = 0; i < wd; ++i) tmp[pos + 3 * i + w] = xr[pos + w * wd + i];
        pos += 3 * wd;
    }
    memcpy(xr + first, tmp + first, sizeof(float) * (size_t)(pos - first));
}

void alias_reduce(float xr[576], int n_boundaries, const Tables &T) {
    for (int sb = 1; sb <= n_boundaries; ++sb) {
        float *lo = xr + sb * 18 - 1, *hi = xr + sb * 18;
        for (int i = 0; i < 8; ++i) {
            const float a = lo[-i], b = hi[i];
            lo[-i] = a * T.cs[i] - b * T.ca[i];
            hi[i] = b * T.cs[i] + a * T.ca[i];
        }
    }
}

void hybrid(const float xr[576], const Granule &g, ChannelState &cs, const Tables &T, float out[18][32]) {
    for (int sb = 0; sb < 32; ++sb) {
        const float *X = xr + sb * 18;
        float raw[36];
        const int bt = (g.block_type == 2 && g.mixed && sb < 2) ? 0 : g.block_type;
        if (bt != 2) {
            // x[17 - i] = -x[i] and x[35 - j] = x[18 + j]: an 18-point DCT-IV gives all 36 samples (rg_mp3_math.h, shared
            // with the device decoder)
            rg_mp3_imdct36_windowed(X, T.win[bt], raw);
        } else {
            for (int i = 0; i < 36; ++i) raw[i] = 0.0f;
            for (int w = 0; w < 3; ++w)
                for (int i = 0; i < 12; ++i) {
                    float s = 0.0f;
                    for (int k = 0; k < 6; ++k) s = rg_mp3_mac(X[3 * k + w], T.imdct12[i][k], s);
                    raw[6 + 6 * w + i] = rg_mp3_mac(s, T.win[2][i], raw[6 + 6 * w + i]);
                }
        }
        float *ov = cs.overlap[sb];
        for (int i = 0; i < 18; ++i) {
            float v = raw[i] + ov[i];
            ov[i] = raw[18 + i];
            if ((sb & 1) && (i & 1)) v = -v;  // frequency inversion of the polyphase filterbank
            out[i][sb] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// stage E: polyphase synthesis (ISO 11172-3 Figure A.2): matrixing into a 1024-entry ring, 512-tap window
// ---------------------------------------------------------------------------------------------------------------
void synth(const float S[18][32], ChannelState &cs, const Tables &T, float *pcm /* 576 */) {
    for (int t = 0; t < 18; ++t) {
        cs.v_off = (cs.v_off - 64) & 1023;
        float *V = cs.V;
        const int o = cs.v_off;
        struct Ring {  // the 64 new entries of the FIFO, at offset o of the 1024-entry ring
            float *V;
            int o;
            float &operator[](int i) { return V[(o + i) & 1023]; }
        } ring{V, o};
        rg_mp3_matrixing(S[t], ring, T.sec, T.dct16);
        float *dst = pcm + 32 * t;
        for (int j = 0; j < 32; ++j) {
            float s = 0.0f;
            for (int i = 0; i < 8; ++i) {
                s = rg_mp3_mac(V[(o + i * 128 + j) & 1023], T.D[i * 64 + j], s);
                s = rg_mp3_mac(V[(o + i * 128 + 96 + j) & 1023], T.D[i * 64 + 32 + j], s);
            }
            dst[j] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the frame loop
// ---------------------------------------------------------------------------------------------------------------
// where stage A's output goes when the back half runs elsewhere (rg_mp3_parse_units)
struct UnitSink {
    int16_t *is;
    rg_mp3_unit *units;
    uint64_t cap, count;
    bool overflow;
};

struct Decoder {
    ChannelState ch[2];
    std::vector<uint8_t> reservoir;  // main data of the frames so far (tail kept)
    UnitSink *sink = nullptr;
    Decoder() {
        memset(ch, 0, sizeof ch);
        reservoir.reserve(8192);
    }
};

// Decodes one frame into out[ch][samples]; false = the frame is dropped (no output).
// `wanted` = false: the frame only feeds the bit reservoir (its channel count is not the stream's)
bool decode_frame(Decoder &D, const uint8_t *f, const Header &h, const Tables &T, float *out0, float *out1, bool wanted = true) {
    const uint8_t *side = f + 4 + (h.crc ? 2 : 0);
    SideInfo si;
    const bool side_ok = parse_side_info(side, h, T, &si);
    const uint8_t *main = side + h.side_bytes;
    const int main_len = h.frame_bytes - (int)(main - f);
    const size_t have = D.reservoir.size();
    // the frame's own main data joins the reservoir whether or not the frame can be decoded
    D.reservoir.insert(D.reservoir.end(), main, main + main_len);
    bool ok = wanted && side_ok && (size_t)si.main_data_begin <= have;
    if (ok) {
        // A frame decodes as a whole or not at all: a second granule whose lengths do not add up must not leave the first
        // granule's overlap and filterbank history behind (the device routes decide per frame before they decode
        // anything, with this same function: rg_mp3_frame.h).  The frame's own channel count is accepted here; what the
        // callers do with a frame whose count differs from the stream's is theirs.
        uint8_t slot[RG_MP3_SLOT_BYTES] = {0};
        memcpy(slot, f, 4);
        memcpy(slot + 4, side, (size_t)h.side_bytes);
        RgMp3HuffRec recs[4];
        uint32_t mb = 0;
        ok = rg_mp3_frame_records(slot, have, h.channels, recs, &mb) != 0;
    }
    const uint64_t sink_mark = D.sink ? D.sink->count : 0;  // a frame that fails half-way leaves no units behind
    if (ok) {
        const size_t begin = have - (size_t)si.main_data_begin;
        const size_t total = D.reservoir.size() - begin;
        std::vector<uint8_t> buf(total + 64, 0);  // scalefactor / linbits reads may run a few bytes past a granule that lies about its length
        memcpy(buf.data(), D.reservoir.data() + begin, total);
        Bits b{buf.data(), 0, total * 8};
        const int ngr = h.lsf ? 1 : 2;
        size_t bit = 0;
        for (int gr = 0; gr < ngr && ok; ++gr) {
            float xr[2][576];
            int nz[2] = {0, 0};
            for (int c = 0; c < h.channels; ++c) {
                Granule &g = si.g[gr][c];
                b.pos = bit;
                b.end = bit + (size_t)g.part2_3_length;
                if (b.end > total * 8) { ok = false; break; }
                if (h.lsf) read_scalefactors_lsf(b, g, c == 1 && (h.mode == 1) && (h.mode_ext & 1), D.ch[c]);
                else read_scalefactors_v1(b, g, si.scfsi[c], gr, D.ch[c]);
                int is[576];
                if (b.pos > b.end) { ok = false; break; }
                nz[c] = decode_spectrum(b, g, T, is);
                bit += (size_t)g.part2_3_length;
                if (D.sink) {  // stage A only: hand the quantised values and the granule's parameters over
                    UnitSink &S = *D.sink;
                    if (S.count >= S.cap) { S.overflow = true; S.count += 1; continue; }
                    int16_t *dst = S.is + S.count * 576;
                    for (int i = 0; i < 576; ++i) dst[i] = (int16_t)(is[i] > 8207 ? 8207 : (is[i] < -8207 ? -8207 : is[i]));
                    rg_mp3_unit &u = S.units[S.count];
                    memset(&u, 0, sizeof u);
                    const int *sf = D.ch[c].sf[h.lsf ? 0 : gr];
                    for (int i = 0; i < 40; ++i) {
                        u.sf[i] = (uint8_t)sf[i];
                        if (D.ch[c].illegal[i]) u.illegal |= 1ull << i;
                    }
                    u.nz = (uint16_t)nz[c];
                    u.global_gain = (uint8_t)g.global_gain;
                    u.block_type = (uint8_t)g.block_type;
                    u.mixed = (uint8_t)g.mixed;
                    for (int k = 0; k < 3; ++k) u.subblock_gain[k] = (uint8_t)g.subblock_gain[k];
                    u.scalefac_scale = (uint8_t)g.scalefac_scale;
                    u.preflag = (uint8_t)g.preflag;
                    u.long_end = (uint8_t)g.long_end;
                    u.short_start = (uint8_t)g.short_start;
                    u.mode_ext = (uint8_t)(h.channels == 2 && h.mode == 1 ? h.mode_ext : 0);
                    u.intensity_scale = (uint8_t)(si.g[gr][h.channels - 1].scalefac_compress & 1);
                    S.count += 1;
                    continue;
                }
                requantize(is, nz[c], g, D.ch[c].sf[h.lsf ? 0 : gr], h, T, xr[c]);
            }
            if (!ok) break;
            if (D.sink) continue;
            if (h.channels == 2 && h.mode == 1)
                stereo(xr, si.g[gr][1], D.ch[1].sf[h.lsf ? 0 : gr], D.ch[1].illegal, nz[0], nz[1], h, T);
            for (int c = 0; c < h.channels; ++c) {
                const Granule &g = si.g[gr][c];
                if (g.block_type == 2) {
                    reorder_short(xr[c], g, h, T);
                    if (g.mixed) alias_reduce(xr[c], 1, T);
                } else {
                    alias_reduce(xr[c], 31, T);
                }
                float S[18][32];
                hybrid(xr[c], g, D.ch[c], T, S);
                synth(S, D.ch[c], T, (c == 0 ? out0 : out1) + 576 * gr);
            }
        }
    }
    if (D.sink && !ok) D.sink->count = sink_mark;
    // keep only what a later frame may still reach back to (main_data_begin < 512 bytes)
    if (D.reservoir.size() > 4096) D.reservoir.erase(D.reservoir.begin(), D.reservoir.end() - 2048);
    return ok;
}

struct Walk {
    rg_mp3_stream_info info;
    Header first;
};

// shared by scan and decode: iterate the audio frames of a stream
template <typename F>
int walk_frames(const uint8_t *d, size_t len, rg_mp3_stream_info *info, F &&on_frame) {
    memset(info, 0, sizeof *info);
    size_t pos = id3v2_size(d, len);
    info->id3v2_bytes = (uint32_t)pos;
    Header h;
    pos = find_sync(d, len, pos, &h, true);
    if (pos + 4 > len) return fail(RG_MP3DEC_ERR_NO_AUDIO, "no MPEG Layer III frame found");
    info->junk_bytes = (uint32_t)(pos - info->id3v2_bytes);
    info->first_frame_offset = pos;
    if (pos + (size_t)h.frame_bytes <= len && is_info_frame(d + pos, h)) {
        info->info_frame = 1;
        const size_t after = pos + (size_t)h.frame_bytes;
        Header h2;
        const size_t nx = find_sync(d, len, after, &h2, true);
        if (nx + 4 > len) return fail(RG_MP3DEC_ERR_NO_AUDIO, "no audio frame after the Xing/Info header frame");
        info->junk_bytes += (uint32_t)(nx - after);
        pos = nx;
        h = h2;
    }
    const Header first = h;
    info->sample_rate = first.rate;
    info->channels = (uint32_t)first.channels;
    info->mpeg_version = (uint32_t)first.version;
    info->samples_per_frame = (uint32_t)first.samples;
    while (pos + 4 <= len) {
        Header hh;
        if (!parse_header(d + pos, &hh) || hh.rate != first.rate || hh.version != first.version) {
            const size_t nx = find_sync(d, len, pos + 1, &hh, true);
            if (nx + 4 > len) break;
            if (hh.rate != first.rate || hh.version != first.version) { info->junk_bytes += (uint32_t)(nx + 1 - pos); pos = nx + 1; continue; }
            info->junk_bytes += (uint32_t)(nx - pos);
            pos = nx;
        }
        if (pos + (size_t)hh.frame_bytes > len) break;  // truncated last frame: the packet reader hits the end of the data
        on_frame(d + pos, hh);
        pos += (size_t)hh.frame_bytes;
    }
    return RG_MP3DEC_OK;
}

}  // namespace

extern "C" const char *rg_mp3dec_last_error(void) { return g_err; }

extern "C" int rg_mp3_scan(const void *data, size_t len, rg_mp3_stream_info *out) {
    RG_NEED_FMA();
    if (!data || !out) return fail(RG_MP3DEC_ERR_ARG, "null argument");
    g_err[0] = 0;
    uint32_t n = 0;
    const int rc = walk_frames((const uint8_t *)data, len, out, [&](const uint8_t *, const Header &) { ++n; });
    if (rc != RG_MP3DEC_OK) return rc;
    out->audio_frames = n;
    out->frames = (uint64_t)n * out->samples_per_frame;
    return RG_MP3DEC_OK;
}

extern "C" int rg_mp3_decode_f32(const void *data, size_t len, float *ch0, float *ch1, uint64_t capacity,
                                 rg_mp3_stream_info *out) {
    RG_NEED_FMA();
    if (!data || !out || !ch0) return fail(RG_MP3DEC_ERR_ARG, "null argument");
    g_err[0] = 0;
    const Tables &T = tables();
    Decoder *D = new Decoder();
    uint64_t produced = 0;
    uint32_t decoded = 0, skipped = 0;
    bool overflow = false;
    int stream_channels = 0;
    float tmp0[1152], tmp1[1152];
    const int rc = walk_frames((const uint8_t *)data, len, out, [&](const uint8_t *f, const Header &h) {
        if (!stream_channels) stream_channels = h.channels;
        // a frame whose channel count differs from the stream's is dropped (every route does; the reference reads plane 1 of
        // whatever its decoder returns and would not survive a mono frame in a stereo stream); its bytes still feed the
        // reservoir
        if (!decode_frame(*D, f, h, T, tmp0, tmp1, h.channels == stream_channels)) { ++skipped; return; }
        ++decoded;
        if (produced + (uint64_t)h.samples > capacity) { overflow = true; produced += (uint64_t)h.samples; return; }
        memcpy(ch0 + produced, tmp0, sizeof(float) * (size_t)h.samples);
        if (stream_channels == 2 && ch1) memcpy(ch1 + produced, tmp1, sizeof(float) * (size_t)h.samples);
        produced += (uint64_t)h.samples;
    });
    delete D;
    if (rc != RG_MP3DEC_OK) return rc;
    out->audio_frames = decoded;
    out->skipped_frames = skipped;
    out->frames = produced;
    if (stream_channels == 2 && !ch1) return fail(RG_MP3DEC_ERR_ARG, "stereo stream needs a second output channel");
    if (overflow) return fail(RG_MP3DEC_ERR_CAPACITY, "output capacity %llu < %llu frames", (unsigned long long)capacity, (unsigned long long)produced);
    return RG_MP3DEC_OK;
}

extern "C" int rg_mp3_parse_units(const void *data, size_t len, int16_t *is_out, rg_mp3_unit *units_out, uint64_t capacity_units,
                                  uint64_t *n_units, rg_mp3_stream_info *out) {
    RG_NEED_FMA();
    if (!data || !out || !n_units || (capacity_units && (!is_out || !units_out))) return fail(RG_MP3DEC_ERR_ARG, "null argument");
    g_err[0] = 0;
    const Tables &T = tables();
    Decoder *D = new Decoder();
    UnitSink sink{is_out, units_out, capacity_units, 0, false};
    D->sink = &sink;
    uint64_t produced = 0;
    uint32_t decoded = 0, skipped = 0;
    int stream_channels = 0;
    const int rc = walk_frames((const uint8_t *)data, len, out, [&](const uint8_t *f, const Header &h) {
        if (!stream_channels) stream_channels = h.channels;
        if (!decode_frame(*D, f, h, T, nullptr, nullptr, h.channels == stream_channels)) { ++skipped; return; }
        ++decoded;
        produced += (uint64_t)h.samples;
    });
    delete D;
    if (rc != RG_MP3DEC_OK) return rc;
    out->audio_frames = decoded;
    out->skipped_frames = skipped;
    out->frames = produced;
    *n_units = sink.count;
    if (sink.overflow) return fail(RG_MP3DEC_ERR_CAPACITY, "unit capacity %llu < %llu", (unsigned long long)capacity_units, (unsigned long long)sink.count);
    return RG_MP3DEC_OK;
}

// The device half's constants, from the tables above (rg_mp3dev.h): same numbers on both sides.
extern "C" void rg_mp3_fill_device_tables(RgMp3DevTables *o) {
    const Tables &T = tables();
    memset(o, 0, sizeof *o);
    memcpy(o->pow43, T.pow43, sizeof o->pow43);
    for (int q = RG_MP3_GAIN_Q_MIN; q <= RG_MP3_GAIN_Q_MAX; ++q) o->gain[q - RG_MP3_GAIN_Q_MIN] = (float)exp2((double)q / 4.0);
    for (int k = 0; k < 32; ++k) {
        o->lsf_is[0][k] = (float)exp2(-0.25 * k);
        o->lsf_is[1][k] = (float)exp2(-0.5 * k);
    }
    for (int i = 0; i < 7; ++i) { o->is_l[i] = T.is_ratio_l[i]; o->is_r[i] = T.is_ratio_r[i]; }
    memcpy(o->cs, T.cs, sizeof o->cs);
    memcpy(o->ca, T.ca, sizeof o->ca);
    memcpy(o->win, T.win, sizeof o->win);
    memcpy(o->imdct36, T.imdct36, sizeof o->imdct36);
    memcpy(o->imdct12, T.imdct12, sizeof o->imdct12);
    memcpy(o->sec, T.sec, sizeof o->sec);
    memcpy(o->dct16, T.dct16, sizeof o->dct16);
    memcpy(o->D, T.D, sizeof o->D);
    for (int r = 0; r < 9; ++r) {
        for (int b = 0; b < 23; ++b) o->sfb_long[r][b] = T.sfb_long[r][b];
        for (int b = 0; b < 14; ++b) o->sfb_short[r][b] = T.sfb_short[r][b];
        for (int b = 0; b < 22; ++b)
            for (int i = T.sfb_long[r][b]; i < T.sfb_long[r][b + 1]; ++i) o->long_band_of_line[r][i] = (uint8_t)b;
        int pos = 0;
        for (int b = 0; b < 13; ++b) {
            const int wd = T.sfb_short[r][b + 1] - T.sfb_short[r][b];
            for (int w = 0; w < 3; ++w)
                for (int i = 0; i < wd; ++i) {
                    o->short_idx_of_line[r][pos + w * wd + i] = (uint8_t)(3 * b + w);
                    o->short_reorder_src[r][pos + 3 * i + w] = (uint16_t)(pos + w * wd + i);
                }
            pos += 3 * wd;
        }
    }
    for (int b = 0; b < 22; ++b) o->pretab[b] = kPretab[b];
}

extern "C" void rg_mp3_fill_device_huff(RgMp3DevHuff *o) {
    const Tables &T = tables();
    memset(o, 0, sizeof *o);
    uint32_t at = 0;
    for (int t = 0; t < 32; ++t) {
        o->base[t] = at;
        o->primary_bits[t] = (uint8_t)T.huff[t].primary_bits;
        o->linbits[t] = kMp3Linbits[t];
        // tables 16..23 and 24..31 share their codes: one copy each
        if (t > 16 && t < 24) { o->base[t] = o->base[16]; continue; }
        if (t > 24) { o->base[t] = o->base[24]; continue; }
        const size_t n = T.huff[t].e.size();
        if (at + n > sizeof o->e / sizeof o->e[0]) { o->n_entries = 0xFFFFFFFFu; return; }  // (the caller's "do not fit": rg_mp3dev_host.hip ensure_tables)
        memcpy(o->e + at, T.huff[t].e.data(), n * sizeof(uint32_t));
        at += (uint32_t)n;
    }
    o->n_entries = at;
    if (at + 2 > RG_MP3_HUFF_LDS_ENTRIES) return;  // two zero entries follow the tables (rg_mp3dev.hip: a table that codes nothing); the caller refuses
    for (uint32_t i = 0; i < at; ++i) {  // the 16-bit image (rg_mp3dev.h)
        const uint32_t e = o->e[i];
        if (e & 0x80000000u) {
            const uint32_t sub = e & 0xFFu, off = (e >> 8) & 0x7FFFFFu;
            if (sub > 15 || off > 2047) { o->n_entries = 0xFFFFFFFFu; return; }  // does not fit the 16-bit image: the caller refuses
            o->e16[i] = (uint16_t)(0x8000u | sub | (off << 4));
        } else {
            const uint32_t len = e & 0xFFu, xy = (e >> 8) & 0xFFu;
            if (len > 15) { o->n_entries = 0xFFFFFFFFu; return; }
            o->e16[i] = (uint16_t)(len | (xy << 4));
        }
    }
    memcpy(o->quadA, T.quadA, sizeof o->quadA);
}


int rg_mp3_index_stream(const void *data, size_t len, std::vector<uint8_t> *main_stream, std::vector<RgMp3HuffRec> *recs,
                        rg_mp3_stream_info *out) {
    RG_NEED_FMA();
    if (!data || !main_stream || !recs || !out) return fail(RG_MP3DEC_ERR_ARG, "null argument");
    g_err[0] = 0;
    main_stream->clear();
    recs->clear();
    uint64_t produced = 0;
    uint32_t decoded = 0, skipped = 0;
    int stream_channels = 0;
    const int rc = walk_frames((const uint8_t *)data, len, out, [&](const uint8_t *f, const Header &h) {
        if (!stream_channels) stream_channels = h.channels;
        uint8_t slot[RG_MP3_SLOT_BYTES] = {0};
        const uint8_t *side = f + 4 + (h.crc ? 2 : 0);
        memcpy(slot, f, 4);
        memcpy(slot + 4, side, (size_t)h.side_bytes);
        const uint8_t *main = side + h.side_bytes;
        const int main_len = h.frame_bytes - (int)(main - f);
        const uint64_t have = main_stream->size();
        main_stream->insert(main_stream->end(), main, main + main_len);  // every frame feeds the reservoir
        RgMp3HuffRec r[4];
        uint32_t mb = 0;
        const int n = rg_mp3_frame_records(slot, have, stream_channels, r, &mb);  // rg_mp3_frame.h: the device runs the same code
        if (n == 0) { ++skipped; return; }
        recs->insert(recs->end(), r, r + n);
        ++decoded;
        produced += (uint64_t)h.samples;
    });
    if (rc != RG_MP3DEC_OK) return rc;
    out->audio_frames = decoded;
    out->skipped_frames = skipped;
    out->frames = produced;
    return RG_MP3DEC_OK;
}

// Tuning key 6 = 3: the host does not look inside the frames at all.  `data` is compacted IN PLACE into the stream's main
// data (every walked frame's bytes after header, CRC and side information, back to back from data[0]: the destination
// never overtakes the walk), and each walked frame leaves one slot (rg_mp3_frame.h: header + side information) in `slots`.
// Which frames decode, and to what, is the device's business (rg_mp3_frames_kernel).
int rg_mp3_compact_stream(uint8_t *data, size_t len, std::vector<uint8_t> *slots, std::vector<uint64_t> *tiles, uint64_t *main_len_out,
                          rg_mp3_stream_info *out) {
    RG_NEED_FMA();
    if (!data || !slots || !tiles || !main_len_out || !out) return fail(RG_MP3DEC_ERR_ARG, "null argument");
    g_err[0] = 0;
    slots->clear();
    tiles->clear();
    uint64_t at = 0;
    uint32_t nframes = 0;
    const int rc = walk_frames(data, len, out, [&](const uint8_t *f, const Header &h) {
        if (nframes % RG_MP3_FRAME_TILE == 0) tiles->push_back(at);  // where the tile's first frame sits in the bit reservoir
        const size_t so = slots->size();
        slots->resize(so + RG_MP3_SLOT_BYTES, 0);
        const uint8_t *side = f + 4 + (h.crc ? 2 : 0);
        memcpy(slots->data() + so, f, 4);
        memcpy(slots->data() + so + 4, side, (size_t)h.side_bytes);
        const uint8_t *main = side + h.side_bytes;
        const size_t main_len = (size_t)h.frame_bytes - (size_t)(main - f);
        memmove(data + at, main, main_len);
        at += main_len;
        ++nframes;
    });
    if (rc != RG_MP3DEC_OK) return rc;
    *main_len_out = at;
    out->audio_frames = nframes;                      // walked; the device decides how many of them decode
    out->frames = (uint64_t)nframes * out->samples_per_frame;  // upper bound
    return RG_MP3DEC_OK;
}

// Test hook: the two frame indexers must tell the same story.  Runs rg_mp3_index_stream (slots interpreted during the
// walk) and rg_mp3_compact_stream + rg_mp3_frame_records over the slots afterwards (what rg_mp3_frames_kernel does on the
// device) and compares main data and records.  0 = identical, 1 = different, < 0 = the stream has no audio.
extern "C" int rg_mp3_index_selfcheck(const void *data, size_t len) {
    RG_NEED_FMA();
    std::vector<uint8_t> main_a, slots;
    std::vector<uint64_t> tiles;
    std::vector<RgMp3HuffRec> recs_a, recs_b;
    rg_mp3_stream_info ia, ib;
    const int rc = rg_mp3_index_stream(data, len, &main_a, &recs_a, &ia);
    std::vector<uint8_t> copy((const uint8_t *)data, (const uint8_t *)data + len);
    uint64_t main_len = 0;
    const int rc2 = rg_mp3_compact_stream(copy.data(), len, &slots, &tiles, &main_len, &ib);
    if (rc != rc2) return 1;
    if (rc != RG_MP3DEC_OK) return rc;
    if (main_len != main_a.size() || (main_len != 0 && memcmp(copy.data(), main_a.data(), main_len) != 0)) return 1;
    uint64_t have = 0;
    uint32_t decoded = 0;
    const size_t nframes = slots.size() / RG_MP3_SLOT_BYTES;
    if (tiles.size() != (nframes + RG_MP3_FRAME_TILE - 1) / RG_MP3_FRAME_TILE) return 1;
    for (size_t f = 0; f < nframes; ++f) {
        RgMp3HuffRec r[4];
        uint32_t mb = 0;
        if (f % RG_MP3_FRAME_TILE == 0 && tiles[f / RG_MP3_FRAME_TILE] != have) return 1;
        const int n = rg_mp3_frame_records(slots.data() + f * RG_MP3_SLOT_BYTES, have, (int)ib.channels, r, &mb);
        have += mb;
        if (n) { recs_b.insert(recs_b.end(), r, r + n); ++decoded; }
    }
    if (have != main_len || decoded != ia.audio_frames || recs_a.size() != recs_b.size()) return 1;
    if (!recs_a.empty() && memcmp(recs_a.data(), recs_b.data(), recs_a.size() * sizeof(RgMp3HuffRec)) != 0) return 1;
    if (ia.sample_rate != ib.sample_rate || ia.channels != ib.channels || ia.mpeg_version != ib.mpeg_version || ib.audio_frames != nframes) return 1;
    return 0;
}

extern "C" int rg_mp3_index_units(const void *data, size_t len, uint64_t *n_units, rg_mp3_stream_info *out) {
    RG_NEED_FMA();
    if (!n_units) return fail(RG_MP3DEC_ERR_ARG, "null argument");
    std::vector<uint8_t> main_stream;
    std::vector<RgMp3HuffRec> recs;
    const int rc = rg_mp3_index_stream(data, len, &main_stream, &recs, out);
    if (rc == RG_MP3DEC_OK) *n_units = recs.size();
    return rc;
}
