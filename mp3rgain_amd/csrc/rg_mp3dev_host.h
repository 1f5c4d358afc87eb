// rg_mp3dev_host.h -- internal: run the device half of the split MP3 decoder for a list of parsed streams.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/mp3rgain_amd_dec.h"

struct rg_ctx;

struct RgMp3HuffRec;
struct RgMp3SplitItem {
    // either stage A's output from the host ...
    const int16_t *is;         // host: [n_units][576]
    const rg_mp3_unit *units;  // host: [n_units]
    // ... or, when `recs` is set, the frame index only: scalefactors and Huffman run on the device too
    const RgMp3HuffRec *recs;  // host: [n_units]
    const uint8_t *main;       // host: the track's main-data stream
    uint64_t main_len;
    uint64_t n_units;
    uint32_t channels, rate_row, lsf;
    float *d_ch0, *d_ch1;      // device outputs, (n_units / channels) * 576 frames each
};

int rg_mp3_rate_row(uint32_t sample_rate);
// stages B-E on the device, results in d_ch0 / d_ch1 when this returns (stream `s` is synchronised)
int rg_mp3dev_decode(rg_ctx *c, const RgMp3SplitItem *items, size_t n, hipStream_t s);
