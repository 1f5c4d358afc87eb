// rg_device.h -- device-side data layout shared by the host orchestration and the gfx950 kernels.
//
// HBM layout of one batch (all arrays are ctx-owned except the PCM arena):
//   PCM arena (caller)      planar samples; track t, channel c at pcm_base + offset + c*frames*bps
//   RgTrackDev  [n]         per-track launch descriptor (below), uploaded once per enqueue
//   uint32_t    [n][12000]  per-track loudness histograms (LoudnessHistogram, replaygain.rs:644-683)
//   uint64_t    [n]         per-track peak as the bit pattern of a non-negative double (atomicMax-able)
//   rg_track_result [n]     ReplayGainResult per track (replaygain.rs:57-68)
//   uint32_t    [12000]     album histogram, double[1] album peak (replaygain.rs:1048-1066)
#pragma once

#include <stdint.h>

#include "../../include/mp3rgain_amd.h"
#include "../../include/rg_coeffs.h"

#define RG_MAX_TAPS 11

// filter constants of one sample rate as the kernels consume them
struct RgCoefDev {
    double ya[11];  // yule a[0..10]  (a[0] == 1, unused)
    double yb[11];  // yule b[0..10]
    double ba[3];   // butter a[0..2]
    double bb[3];   // butter b[0..2]
};

struct RgTrackDev {
    const void *ch0;       // device pointer to channel 0
    const void *ch1;       // device pointer to channel 1, or nullptr for mono
    uint64_t frames;
    uint32_t window;       // 50 ms window in frames: sr*50/1000 (replaygain.rs:704)
    uint32_t n_windows;    // ceil(frames / window): the last one may be partial (replaygain.rs:907)
    uint32_t seg_windows;  // windows owned by one work item
    uint32_t n_segments;   // ceil(n_windows / seg_windows)
    uint32_t halo;         // warm-up frames run before a segment (0 = from track start)
    uint32_t coef_idx;     // row of RG_RATE_TABLE
    uint32_t format;       // rg_sample_format
    uint32_t item_base;    // exclusive prefix sum of work items over the batch
    uint32_t sample_rate;
    uint32_t file_type;    // rg_file_type, carried through to the result
    uint32_t track_index;  // row of this track in the histogram / peak / result arrays
    uint32_t pad_;
};

// 1.0 - RMS_PERCENTILE evaluated in f64 exactly as the reference does (replaygain.rs:671):
// 1.0 - 0.95 == 0.050000000000000044
#define RG_ONE_MINUS_PERCENTILE (1.0 - 0.95)
