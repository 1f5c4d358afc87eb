// rg_device_inl.h -- small device-inline pieces shared by the kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <limits.h>

#include "rg_device.h"

// finish_window's arithmetic, src/replaygain.rs:749-759: mean square of the window ->
// STEPS_PER_DB*10*log10(ms + 1e-37) -> Rust `as i32` (truncate, saturate, NaN -> 0) ->
// + HISTOGRAM_OFFSET with a wrapping i32 add -> valid iff 0 <= idx < HISTOGRAM_SIZE.
// Returns the histogram index or -1 for a dropped window.
static __device__ __forceinline__ int rg_window_bin(double lsum, double rsum, uint32_t n) {
    const double mean_square = (lsum + rsum) / (double)n * 0.5;
    const double val = (100.0 * 10.0) * log10(mean_square + 1e-37);
    int iv;
    if (val != val) iv = 0;
    else if (val >= 2147483647.0) iv = INT_MAX;
    else if (val <= -2147483648.0) iv = INT_MIN;
    else iv = (int)val;
    const int idx = (int)((unsigned)iv + (unsigned)RG_HISTOGRAM_OFFSET);
    return (idx >= 0 && idx < RG_HISTOGRAM_SIZE) ? idx : -1;
}

// ---------------------------------------------------------------------------------------------
// LoudnessHistogram::get_loudness (src/replaygain.rs:665-682) for one histogram per workgroup of 256
// threads, followed by the tail of analyze_track_internal (src/replaygain.rs:910-918): gain =
// PINK_REF - loudness, gain_steps = round(gain / 1.5).
//   total     = sum of bins (u64)
//   threshold = ceil(total as f64 * (1.0 - 0.95)) as u64
//   scan i = 11999 .. 0 accumulating; first i with count >= threshold -> (i - 2000) / 100.0
// ---------------------------------------------------------------------------------------------
#define RG_PCT_THREADS 256
#define RG_PCT_CHUNK 48    // bins per owner thread: twelve 16-byte loads
#define RG_PCT_OWNERS 250  // 250 * 48 = 12000

struct RgLoudness {
    double loudness_db;
    uint64_t total;
};

// All 256 threads cooperate, and every step is one round of independent loads (the finisher of a track is
// on the critical path of a whole batch): threads 0..249 own 48 consecutive bins each (twelve 16-byte loads),
// a suffix scan over the 250 chunk sums (wave shuffles + one LDS hop) finds the one chunk in which the
// running count from the top bin down first reaches the threshold, and wave 0 resolves the bin inside that
// chunk with one load per lane and a ballot.  The result is exactly the sequential scan's.
// LDS: scan[256] is used for the four wave totals and the crossing chunk.
static __device__ __forceinline__ RgLoudness rg_block_loudness(const uint32_t *__restrict__ h, uint64_t *scan /* LDS[256] */) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint64_t s = 0;
    if (t < RG_PCT_OWNERS) {
        if (((uintptr_t)h & 15) == 0) {
            const uint4 *__restrict__ h4 = reinterpret_cast<const uint4 *>(h) + t * (RG_PCT_CHUNK / 4);
            uint4 v[RG_PCT_CHUNK / 4];
#pragma unroll
            for (int i = 0; i < RG_PCT_CHUNK / 4; ++i) v[i] = h4[i];
#pragma unroll
            for (int i = 0; i < RG_PCT_CHUNK / 4; ++i) s += (uint64_t)v[i].x + v[i].y + v[i].z + v[i].w;
        } else {
#pragma unroll 8
            for (int i = 0; i < RG_PCT_CHUNK; ++i) s += h[t * RG_PCT_CHUNK + i];
        }
    }
    // inclusive suffix sum over threads: suffix[t] = sum_{u >= t} chunk[u]
    uint64_t suffix = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = __shfl_down(suffix, d, 64);
        if (lane + d < 64) suffix += o;
    }
    if (lane == 0) scan[wave] = suffix;  // wave totals
    __syncthreads();
    uint64_t higher = 0;  // chunks of the waves above this one
#pragma unroll
    for (int w = 1; w < RG_PCT_THREADS / 64; ++w) higher += w > wave ? scan[w] : 0;
    const uint64_t total = scan[0] + scan[1] + scan[2] + scan[3];
    suffix += higher;
    const uint64_t above = suffix - s;  // count of all bins above this chunk
    __shared__ RgLoudness res;  // one instance per kernel: the helper is called once per block
    __shared__ int cross_chunk;
    __shared__ uint64_t cross_above;
    if (t == 0) {
        res.loudness_db = -20.0;  // empty histogram, or the fall-through of src/replaygain.rs:681
        res.total = total;
    }
    const uint64_t threshold = (uint64_t)ceil((double)total * RG_ONE_MINUS_PERCENTILE);
    if (total != 0 && suffix >= threshold && above < threshold) {  // exactly one thread
        cross_chunk = t;
        cross_above = above;
    }
    __syncthreads();
    if (total != 0 && wave == 0) {
        const int b = cross_chunk * RG_PCT_CHUNK + lane;
        uint64_t c = lane < RG_PCT_CHUNK ? h[b] : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = __shfl_down(c, d, 64);
            if (lane + d < 64) c += o;
        }
        // the count from the top reaches the threshold at the highest bin whose suffix does
        const unsigned long long m = __ballot(lane < RG_PCT_CHUNK && cross_above + c >= threshold);
        if (lane == 0) res.loudness_db = (double)(cross_chunk * RG_PCT_CHUNK + (63 - __clzll((long long)m)) - RG_HISTOGRAM_OFFSET) / 100.0;
    }
    __syncthreads();
    return res;
}

static __device__ __forceinline__ int32_t rg_round_steps(double gain_db) {
    const double r = round(gain_db / RG_GAIN_STEP_DB);  // Rust f64::round: half away from zero
    if (r != r) return 0;
    if (r >= 2147483647.0) return 2147483647;
    if (r <= -2147483648.0) return (int32_t)0x80000000;
    return (int32_t)r;
}


static __device__ __forceinline__ void rg_store_track_result(rg_track_result *out, const RgLoudness &l, double peak,
                                                             uint32_t sample_rate, uint32_t file_type, uint32_t flags = 0) {
    rg_track_result r;
    r.loudness_db = l.loudness_db;
    r.gain_db = RG_PINK_REF - l.loudness_db;
    r.peak = peak;
    r.sample_rate = sample_rate;
    r.gain_steps = rg_round_steps(r.gain_db);
    r.windows = (uint32_t)l.total;
    r.file_type = file_type;
    r.flags = flags;
    r.reserved = 0;
    *out = r;
}
