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

// ---- tuning key 6 = 3: the device parses the frames too --------------------------------------------------------------
// One compacted stream inside a chunk's pinned staging block (rg_mp3_compact_stream's outputs copied there).
struct RgMp3StreamItem {
    uint64_t main_off;     // staging block: the stream's main data (8 readable bytes follow it)
    uint64_t slots_off;    // staging block: n_frames slots of RG_MP3_SLOT_BYTES, 8-byte aligned
    uint64_t tiles_off;    // staging block: one uint64 per RG_MP3_FRAME_TILE frames (rg_mp3_compact_stream's `tiles`)
    uint32_t n_frames;     // frames the host walked
    uint32_t channels, rate_row, lsf;
    uint32_t result_index; // h_mp3_results[result_index] = granules decoded, valid once stream `s` has been synchronised
    float *d_ch0;          // device PCM output, sized for n_frames decodable frames; channel 1 follows the decoded length
};
// bytes of the descriptor array the staging block must have room for behind the streams (8-byte aligned offset `tracks_off`)
size_t rg_mp3dev_track_bytes(size_t n_items);
// One chunk: H2D of staging[0, bytes) on the context's copy stream into device set `set` (0 / 1), then frame parser,
// Huffman and back-half kernels on `s`.  `staged` is recorded on the copy stream behind the H2D: the staging block
// may be refilled once it has completed.  Nothing is synchronised.  `counts_out` (pinned, n_counts words; may be null): a copy
// of the per-file granule counts as they stand behind this chunk's frame parser, `counts_ev` recorded behind it.
int rg_mp3dev_enqueue_chunk(rg_ctx *c, int set, uint8_t *staging, size_t bytes, size_t tracks_off, hipEvent_t staged,
                            const RgMp3StreamItem *items, size_t n, hipStream_t s, uint32_t *counts_out = nullptr, size_t n_counts = 0,
                            hipEvent_t counts_ev = nullptr);
// granules decoded per result_index: rg_mp3dev_fetch_results enqueues the D2H on `s`; read after synchronising `s`
int rg_mp3dev_reserve_results(rg_ctx *c, size_t n, hipStream_t s);  // zeroed on `s`, which the chunks' kernels must follow
int rg_mp3dev_fetch_results(rg_ctx *c, size_t n, hipStream_t s);
const uint32_t *rg_mp3dev_results(rg_ctx *c);
