// rg_mp3dev.h -- data shared by the host half (rg_mp3dec.cpp: stage A, table construction) and the device half
// (rg_mp3dev.hip: stages B-E) of the split MP3 decoder.  See include/mp3rgain_amd_dec.h for the unit format.
#pragma once

#include <stdint.h>

#include "../../include/mp3rgain_amd_dec.h"

#define RG_MP3_GAIN_Q_MIN (-512)   // requantisation gains 2^(q/4) are tabulated for q in [RG_MP3_GAIN_Q_MIN, RG_MP3_GAIN_Q_MAX]
#define RG_MP3_GAIN_Q_MAX 64

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
    float matrix[64][32];
    float D[512];
    uint16_t sfb_long[9][24];
    uint16_t sfb_short[9][16];
    uint8_t long_band_of_line[9][576];     // long scalefactor band of a spectral line
    uint8_t short_idx_of_line[9][576];     // 3 * band + window of a line of a pure short block, bitstream order
    uint16_t short_reorder_src[9][576];    // pure short block: line of the reordered spectrum -> bitstream position
    uint8_t pretab[24];
};

struct RgMp3DevTrack {
    uint64_t unit_base;      // first unit of the track in the batch's unit / spectrum arrays
    uint32_t granule_base;   // first granule of the track in the batch's granule numbering
    uint32_t n_granules;
    uint32_t channels;
    uint32_t rate_row;
    uint32_t lsf;
    uint32_t pad_;
    float *ch0;              // PCM outputs (planar); 576 frames per granule
    float *ch1;
};

#ifdef __cplusplus
extern "C" {
#endif
// host: fill the table block (rg_mp3dec.cpp)
void rg_mp3_fill_device_tables(RgMp3DevTables *out);
#ifdef __cplusplus
}
#endif
