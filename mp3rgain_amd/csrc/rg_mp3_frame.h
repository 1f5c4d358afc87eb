// rg_mp3_frame.h -- one Layer III frame's header and side information -> the records the device Huffman stage works from.
//
// Shared VERBATIM by the host (rg_mp3dec.cpp: rg_mp3_index_stream, the frame walk of tuning key 6 = 2) and the device
// (rg_mp3dev.hip: rg_mp3_frames_kernel, tuning key 6 = 3, where the host does not even look at the side information):
// which frames are decodable and where each granule's bits are is decided by this one piece of integer code on both
// sides.  The rules are those of the one-shot host decoder (rg_mp3dec.cpp: parse_header, parse_side_info, decode_frame);
// tests/test_mp3dec.py holds the two to the same answer on damaged streams.
#pragma once

#include <stdint.h>

#include "rg_mp3_math.h"  // RG_MP3_HD
#include "rg_mp3dev.h"

// A frame as the host's walk hands it to the device: the 4 header bytes, then the side information (9 / 17 / 32 bytes;
// a CRC word between the two is left out), zero padded (the side-information reader looks two bytes ahead).
#define RG_MP3_SLOT_BYTES 40
// The device parses the frames in tiles of this many; the host's walk notes where each tile starts in the bit reservoir.
#define RG_MP3_FRAME_TILE 256

struct RgMp3FrameHdr {
    uint8_t lsf;        // MPEG-2 / 2.5
    uint8_t crc;
    uint8_t channels;
    uint8_t mode;       // 0 stereo, 1 joint stereo, 2 dual channel, 3 mono
    uint8_t mode_ext;
    uint8_t side_bytes;
    uint8_t rate_row;   // row of the band tables
    uint8_t version;    // 1, 2, 25
    uint32_t rate;
    uint32_t frame_bytes;
};

// rg_mp3dec.cpp: parse_header (same validity rules: Layer III only, no free format, no reserved fields)
RG_MP3_HD bool rg_mp3_frame_header(const uint8_t *p, RgMp3FrameHdr *h) {
    static const uint16_t br_v1[16] = {0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 0};
    static const uint16_t br_v2[16] = {0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160, 0};
    static const uint32_t rates[3] = {44100, 48000, 32000};
    if (p[0] != 0xFF || (p[1] & 0xE0) != 0xE0) return false;
    const int ver = (p[1] >> 3) & 3, layer = (p[1] >> 1) & 3;
    if (ver == 1 || layer != 1) return false;
    const int br = p[2] >> 4, sr = (p[2] >> 2) & 3;
    if (br == 0 || br == 15 || sr == 3) return false;
    h->version = (uint8_t)(ver == 3 ? 1 : (ver == 2 ? 2 : 25));
    h->lsf = ver != 3;
    h->crc = (p[1] & 1) == 0;
    const int kbps = h->lsf ? br_v2[br] : br_v1[br];
    h->rate = rates[sr] >> (ver == 3 ? 0 : (ver == 2 ? 1 : 2));
    h->rate_row = (uint8_t)(sr + (ver == 3 ? 0 : (ver == 2 ? 3 : 6)));
    const int padding = (p[2] >> 1) & 1;
    h->mode = (uint8_t)(p[3] >> 6);
    h->mode_ext = (uint8_t)((p[3] >> 4) & 3);
    h->channels = (uint8_t)(h->mode == 3 ? 1 : 2);
    h->frame_bytes = (uint32_t)((h->lsf ? 72 : 144) * kbps * 1000 / (int)h->rate + padding);
    h->side_bytes = (uint8_t)(h->lsf ? (h->channels == 1 ? 9 : 17) : (h->channels == 1 ? 17 : 32));
    return h->frame_bytes >= 4u + (h->crc ? 2u : 0u) + h->side_bytes;
}

// bytes of main data a frame carries (everything after header, CRC and side information)
RG_MP3_HD uint32_t rg_mp3_frame_main_bytes(const RgMp3FrameHdr &h) { return h.frame_bytes - 4u - (h.crc ? 2u : 0u) - h.side_bytes; }

struct RgMp3SideBits {
    const uint8_t *p;
    uint32_t pos;
    RG_MP3_HD uint32_t get(int n) {  // n <= 16, MSB first; the slot's zero padding covers the look-ahead
        const uint32_t byte = pos >> 3, sh = pos & 7;
        const uint32_t w = ((uint32_t)p[byte] << 16) | ((uint32_t)p[byte + 1] << 8) | (uint32_t)p[byte + 2];
        pos += (uint32_t)n;
        return (w >> (24 - sh - n)) & ((1u << n) - 1u);
    }
};

// bits of part 2 (the scalefactors) of a granule, from the side information alone (rg_mp3dec.cpp: read_scalefactors_*)
RG_MP3_HD int rg_mp3_part2_bits(bool lsf, int scalefac_compress, int block_type, int mixed, int scfsi_mask, int gr, bool intensity_right) {
    static const uint8_t slen_tab[2][16] = {{0, 0, 0, 0, 3, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4}, {0, 1, 2, 3, 0, 1, 2, 3, 1, 2, 3, 1, 2, 3, 2, 3}};
    static const uint8_t parts[6][3][4] = {
        {{6, 5, 5, 5}, {9, 9, 9, 9}, {6, 9, 9, 9}},   {{6, 5, 7, 3}, {9, 9, 12, 6}, {6, 9, 12, 6}},
        {{11, 10, 0, 0}, {18, 18, 0, 0}, {15, 18, 0, 0}}, {{7, 7, 7, 0}, {12, 12, 12, 0}, {6, 15, 12, 0}},
        {{6, 6, 6, 3}, {12, 9, 9, 6}, {6, 12, 9, 6}},  {{8, 8, 5, 0}, {15, 12, 9, 0}, {6, 18, 9, 0}}};
    if (!lsf) {
        const int s1 = slen_tab[0][scalefac_compress], s2 = slen_tab[1][scalefac_compress];
        if (block_type == 2) return mixed ? 17 * s1 + 18 * s2 : 18 * s1 + 18 * s2;
        int bits = 0;
        if (!(gr == 1 && (scfsi_mask & 1))) bits += 6 * s1;
        if (!(gr == 1 && (scfsi_mask & 2))) bits += 5 * s1;
        if (!(gr == 1 && (scfsi_mask & 4))) bits += 5 * s2;
        if (!(gr == 1 && (scfsi_mask & 8))) bits += 5 * s2;
        return bits;
    }
    int slen[4], set;
    int sfc = scalefac_compress;
    if (!intensity_right) {
        if (sfc < 400) { slen[0] = (sfc >> 4) / 5; slen[1] = (sfc >> 4) % 5; slen[2] = (sfc & 15) >> 2; slen[3] = sfc & 3; set = 0; }
        else if (sfc < 500) { sfc -= 400; slen[0] = (sfc >> 2) / 5; slen[1] = (sfc >> 2) % 5; slen[2] = sfc & 3; slen[3] = 0; set = 1; }
        else { sfc -= 500; slen[0] = sfc / 3; slen[1] = sfc % 3; slen[2] = 0; slen[3] = 0; set = 2; }
    } else {
        sfc >>= 1;
        if (sfc < 180) { slen[0] = sfc / 36; slen[1] = (sfc % 36) / 6; slen[2] = (sfc % 36) % 6; slen[3] = 0; set = 3; }
        else if (sfc < 244) { sfc -= 180; slen[0] = (sfc & 0x3F) >> 4; slen[1] = (sfc & 0xF) >> 2; slen[2] = sfc & 3; slen[3] = 0; set = 4; }
        else { sfc -= 244; slen[0] = sfc / 3; slen[1] = sfc % 3; slen[2] = 0; slen[3] = 0; set = 5; }
    }
    const int kind = block_type == 2 ? (mixed ? 2 : 1) : 0;
    int bits = 0;
    for (int k = 0; k < 4; ++k) bits += parts[set][kind][k] * slen[k];
    return bits;
}

// The frame in `slot` (RG_MP3_SLOT_BYTES), whose main data follows `have` bytes of earlier main data in the track's
// stream: writes one record per granule and channel in decode order (granule-major) and returns how many, or 0 when the
// frame is dropped -- invalid header or side information, a channel count other than the stream's, main_data_begin
// reaching back before the stream's first byte, part2_3 lengths that overrun the frame's bits or are shorter than
// their own scalefactors.  *main_bytes = the frame's contribution to the stream (it counts even when the frame is
// dropped: the bit reservoir is made of every frame's bytes).
RG_MP3_HD int rg_mp3_frame_records(const uint8_t *slot, uint64_t have, int stream_channels, RgMp3HuffRec *out, uint32_t *main_bytes) {
    RgMp3FrameHdr h;
    *main_bytes = 0;
    if (!rg_mp3_frame_header(slot, &h)) return 0;
    const uint32_t main_len = rg_mp3_frame_main_bytes(h);
    *main_bytes = main_len;
    if ((int)h.channels != stream_channels) return 0;
    RgMp3SideBits b{slot + 4, 0};
    const int nch = h.channels, ngr = h.lsf ? 1 : 2;
    uint32_t main_data_begin;
    int scfsi[2] = {0, 0};
    if (!h.lsf) {
        main_data_begin = b.get(9);
        b.get(nch == 1 ? 5 : 3);
        for (int ch = 0; ch < nch; ++ch)
            for (int k = 0; k < 4; ++k) scfsi[ch] |= (int)b.get(1) << k;
    } else {
        main_data_begin = b.get(8);
        b.get(nch == 1 ? 1 : 2);
    }
    if ((uint64_t)main_data_begin > have) return 0;
    const uint64_t begin_bit = (have - main_data_begin) * 8;
    const uint64_t total_bits = ((uint64_t)main_data_begin + main_len) * 8;
    const uint64_t frame_end_bit = (have + main_len) * 8;
    uint64_t bit = 0;
    int n = 0;
    for (int gr = 0; gr < ngr; ++gr)
        for (int ch = 0; ch < nch; ++ch) {
            RgMp3HuffRec r;
            r.bit_off = begin_bit + bit;
            r.frame_end_bit = frame_end_bit;
            r.part2_3_length = (uint16_t)b.get(12);
            r.big_values = (uint16_t)b.get(9);
            r.global_gain = (uint8_t)b.get(8);
            r.scalefac_compress = (uint16_t)b.get(h.lsf ? 9 : 4);
            const uint32_t window_switching = b.get(1);
            if (r.big_values > 288) return 0;
            if (window_switching) {
                r.block_type = (uint8_t)b.get(2);
                r.mixed = (uint8_t)b.get(1);
                if (r.block_type == 0) return 0;  // reserved
                r.table_select[0] = (uint8_t)b.get(5);
                r.table_select[1] = (uint8_t)b.get(5);
                r.table_select[2] = 0;
                for (int k = 0; k < 3; ++k) r.subblock_gain[k] = (uint8_t)b.get(3);
                r.region0_count = (uint8_t)((r.block_type == 2 && !r.mixed) ? 8 : 7);
                r.region1_count = (uint8_t)(20 - r.region0_count);
            } else {
                r.block_type = 0;
                r.mixed = 0;
                for (int k = 0; k < 3; ++k) r.table_select[k] = (uint8_t)b.get(5);
                for (int k = 0; k < 3; ++k) r.subblock_gain[k] = 0;
                r.region0_count = (uint8_t)b.get(4);
                r.region1_count = (uint8_t)b.get(3);
            }
            r.preflag = (uint8_t)(h.lsf ? 0 : b.get(1));
            r.scalefac_scale = (uint8_t)b.get(1);
            r.count1table = (uint8_t)b.get(1);
            r.scfsi = (uint8_t)scfsi[ch];
            r.gr = (uint8_t)gr;
            r.mode_ext = (uint8_t)(nch == 2 && h.mode == 1 ? h.mode_ext : 0);
            const bool ir = h.lsf && ch == 1 && h.mode == 1 && (h.mode_ext & 1);
            r.intensity_right = (uint8_t)(ir ? 1 : 0);
            r.intensity_scale = 0;  // the right channel's scalefac_compress: filled in below
            for (int k = 0; k < 7; ++k) r.pad_[k] = 0;
            out[n++] = r;
            bit += r.part2_3_length;
        }
    // the checks the host decoder makes granule by granule while it decodes (decode_frame): a granule that fails drops
    // the whole frame
    bit = 0;
    for (int i = 0; i < n; ++i) {
        const RgMp3HuffRec &r = out[i];
        if (bit + r.part2_3_length > total_bits) return 0;
        if (rg_mp3_part2_bits(h.lsf, r.scalefac_compress, r.block_type, r.mixed, r.scfsi, r.gr, r.intensity_right != 0) > (int)r.part2_3_length) return 0;
        bit += r.part2_3_length;
    }
    for (int i = 0; i < n; ++i) out[i].intensity_scale = (uint8_t)(out[(i / nch) * nch + nch - 1].scalefac_compress & 1);
    return n;
}
