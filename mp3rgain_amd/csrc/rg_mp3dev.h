// rg_mp3dev.h -- data shared by the host half (rg_mp3dec.cpp: stage A, table construction) and the device half
// (rg_mp3dev.hip: stages B-E) of the split MP3 decoder.  See include/mp3rgain_amd_dec.h for the unit format.
#pragma once

#include <stdint.h>

#include "../../include/mp3rgain_amd_dec.h"

#define RG_MP3_GAIN_Q_MIN (-512)   // requantisation gains 2^(q/4) are tabulated for q in [RG_MP3_GAIN_Q_MIN, RG_MP3_GAIN_Q_MAX]
#define RG_MP3_GAIN_Q_MAX 64
#define RG_MP3_HUFF_LDS_ENTRIES 7808  // room for the flattened Huffman tables in the Huffman kernel's LDS (they have 7752 entries)
#define RG_MP3_RUN 32             // consecutive granules per block of the back-half kernel (both channels)
// The quantised spectra between the device Huffman stage and the back half: one row of RG_MP3_ROW_BYTES per unit, sign and
// magnitude in two planes of a byte per line -- at i: sign << 7 | |v| & 127, at 576 + i: |v| >> 7, the second plane written
// only for the first 4 * rg_mp3_unit::reserved[0] lines (no other line holds a value above 127).  (Spectra parsed on the
// host -- rg_mp3_parse_units -- are rows of 576 int16 in the same RG_MP3_ROW_BYTES.)
#define RG_MP3_ROW_BYTES 1152
// The Huffman kernel's bit reader (rg_mp3dev.hip: BitRing) opens on a 32-byte-aligned group and stays up to a ring (16 words)
// plus a feed (8 words) ahead of the position: behind the last stream of a buffer it reads up to 4 * (16 + 8) + 32 = 128 bytes
// past the data.  What it finds there never reaches a decoded value (everything past a frame's data is masked), but the
// bytes must be mapped: every buffer the reader walks is reserved with this much behind its payload.
#define RG_MP3_READ_AHEAD_BYTES 256
#define RG_MP3_SORT_BUCKETS 96    // lane sort of the Huffman stage: big_values >> 2, heaviest first (73 buckets in use)
#define RG_MP3_SORT_WORDS 256     // per staging set: [0, 96) histogram (zero between uses), [96, 192) bucket cursors, [192] units that decode
#define RG_MP3_SORT_NVALID 192

// Every constant the device stages use, built once on the host from the very tables the host decoder uses, so that
// the two halves work with identical numbers.
struct RgMp3DevTables {
    float pow43[8208];
    float gain[RG_MP3_GAIN_Q_MAX - RG_MP3_GAIN_Q_MIN + 1];  // (float)exp2(q / 4.0)
    float lsf_is[2][32];               // (float)exp2(-io_exp * k), io_exp = 0.25 / 0.5, k = 0..31
    float is_l[8], is_r[8];            // MPEG-1 intensity ratios, positions 0..6
    float cs[8], ca[8];
    float win[4][36];
    float imdct36[36][18];
    float imdct12[12][6];
    float sec[32];                     // secants of the 32-point DCT behind the matrixing (rg_mp3_math.h, Lee's form: measurement builds)
    float dct16[2][16][16];            // [parity][k][i]: the cosine matrices of its dense form, the matrix cores' A operands
    float D[512];
    uint16_t sfb_long[9][24];
    uint16_t sfb_short[9][16];
    uint8_t long_band_of_line[9][576];     // long scalefactor band of a spectral line
    uint8_t short_idx_of_line[9][576];     // 3 * band + window of a line of a pure short block, bitstream order
    uint16_t short_reorder_src[9][576];    // pure short block: line of the reordered spectrum -> bitstream position
    uint8_t pretab[24];
};

// Flattened Huffman look-up tables as the host decoder builds them (rg_mp3dec.cpp: HuffLut): per table a primary
// array of 2^primary_bits entries, long codes continue in secondary arrays.  Entry: leaf = len | xy << 8,
// link = 0x80000000 | sub_bits | offset << 8 (offset relative to the table's own base).
struct RgMp3DevHuff {
    uint32_t base[32];          // first entry of table t in `e`
    uint8_t primary_bits[32];   // 0 = the table is empty (all zeros)
    uint8_t linbits[32];
    uint8_t quadA[64];          // count1 table A: 6 peeked bits -> len << 4 | vwxy
    uint32_t n_entries;
    uint32_t e[14000];
    // the same entries in 16 bits, what the Huffman kernel keeps in LDS: leaf = len | y << 4 | x << 8 (len <= 10),
    // link = 0x8000 | sub_bits | offset << 4 (sub_bits <= 10, offset < 2048; rg_mp3_fill_device_huff checks)
    uint16_t e16[RG_MP3_HUFF_LDS_ENTRIES];
};

// One granule of one channel of a decodable frame, as the host's frame indexer hands it to the device Huffman
// stage: where its bits are in the track's contiguous main-data stream, and its side information.
struct RgMp3HuffRec {
    uint64_t bit_off;          // first bit of part 2 (scalefactors), relative to the batch's main-data buffer
    uint64_t frame_end_bit;    // end of the frame's own main data: bits at or past it read as zero (the host decoder
                               // works on a copy that ends there)
    uint16_t part2_3_length;
    uint16_t big_values;
    uint16_t scalefac_compress;
    uint8_t global_gain;
    uint8_t block_type;
    uint8_t mixed;
    uint8_t table_select[3];
    uint8_t subblock_gain[3];
    uint8_t region0_count, region1_count;
    uint8_t preflag, scalefac_scale, count1table;
    uint8_t scfsi;             // bit k: group k of granule 1 reuses granule 0's scalefactors (MPEG-1)
    uint8_t gr;
    uint8_t mode_ext;          // joint stereo only, else 0
    uint8_t intensity_right;   // LSF: this is the right channel of an intensity-stereo frame
    uint8_t intensity_scale;   // low bit of the right channel's scalefac_compress
    uint8_t pad_[7];
};                             // 48 bytes

struct RgMp3DevTrack {
    uint64_t unit_base;      // first unit of the track in the batch's unit / spectrum arrays
    uint32_t granule_base;   // first granule of the track in the batch's granule numbering
    uint32_t n_granules;
    uint32_t channels;
    uint32_t rate_row;
    uint32_t lsf;
    uint32_t run_base;       // first block of the track in the back-half kernel's grid (one block per run of RG_MP3_RUN granules)
    float *ch0;              // PCM outputs (planar); 576 frames per granule
    float *ch1;
    uint64_t main_base;      // device Huffman stage: byte offset of the track's main-data stream in the chunk buffer
    uint32_t all_frames_decode;  // written by the frame parser's scan pass: no frame of the track was dropped (its records are in place)
    uint32_t n_frames;       // tuning key 6 = 3: frames the host walked (slots); the device decides which decode
    uint64_t slots_base;     //   byte offset of the track's slots (rg_mp3_frame.h) in the chunk buffer
    uint32_t result_index;   //   where the frame parser reports the granules it found decodable
    uint32_t tile_base;      //   first tile (RG_MP3_FRAME_TILE frames) of the track in the chunk's tile numbering
    uint64_t tiles_base;     //   byte offset of the track's tile table (uint64 per tile: main-data bytes before the tile)
};
// With the device-side frame parser (rg_mp3_frames_kernel) unit_base / granule_base / run_base and the grids
// are laid out for the upper bound "every walked frame decodes"; the kernel then overwrites n_granules (and ch1, which
// follows the decoded length) with what it found, and the later stages skip the units past it.

#ifdef __cplusplus
#include <vector>
// host (rg_mp3dec.cpp): the frame walk and side information only.  Appends every frame's main data to `main_stream`
// (what the bit reservoir is made of) and, for the frames the one-shot decoder would decode, one record per granule and
// channel in decode order.  `bit_base` = bit position of main_stream's first byte in the batch buffer.
int rg_mp3_index_stream(const void *data, size_t len, std::vector<uint8_t> *main_stream, std::vector<RgMp3HuffRec> *recs,
                        rg_mp3_stream_info *info);
// host (rg_mp3dec.cpp), tuning key 6 = 3: compacts `data` in place into the stream's main data and leaves one slot
// (rg_mp3_frame.h) per walked frame; info->frames is an upper bound, the device decides which frames decode.
// `tiles`: for every RG_MP3_FRAME_TILE frames, the main-data bytes that precede the tile's first frame.
int rg_mp3_compact_stream(uint8_t *data, size_t len, std::vector<uint8_t> *slots, std::vector<uint64_t> *tiles, uint64_t *main_len,
                          rg_mp3_stream_info *info);
extern "C" {
#endif
// host: fill the table blocks (rg_mp3dec.cpp)
void rg_mp3_fill_device_tables(RgMp3DevTables *out);
void rg_mp3_fill_device_huff(RgMp3DevHuff *out);
#ifdef __cplusplus
}
#endif
