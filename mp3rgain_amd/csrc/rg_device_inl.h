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
#define RG_PCT_CHUNK 47  // 256 * 47 = 12032 >= 12000

struct RgLoudness {
    double loudness_db;
    uint64_t total;
};

// All 256 threads cooperate: thread t owns bins [47t, 47t+47), a block-wide suffix scan of the 256 chunk
// sums finds the one chunk in which the running count (from the top bin down) first reaches the threshold,
// and that thread alone walks its 47 bins from the top.  The result is exactly the sequential scan's.
// LDS: 2 KiB; no per-thread bin array (the kernels that inline this keep their register budget).
static __device__ __forceinline__ RgLoudness rg_block_loudness(const uint32_t *__restrict__ h, uint64_t *scan /* LDS[256] */) {
    const int t = threadIdx.x;
    uint64_t s = 0;
#pragma unroll 8
    for (int i = 0; i < RG_PCT_CHUNK; ++i) {
        const int b = t * RG_PCT_CHUNK + i;
        s += b < RG_HISTOGRAM_SIZE ? h[b] : 0u;
    }
    // inclusive suffix sum over threads: suffix[t] = sum_{u >= t} chunk[u]
    scan[t] = s;
    __syncthreads();
    for (int d = 1; d < RG_PCT_THREADS; d <<= 1) {
        const uint64_t add = t + d < RG_PCT_THREADS ? scan[t + d] : 0;
        __syncthreads();
        scan[t] += add;
        __syncthreads();
    }
    const uint64_t total = scan[0];
    const uint64_t suffix = scan[t];
    const uint64_t above = t + 1 < RG_PCT_THREADS ? scan[t + 1] : 0;  // count of all bins above this chunk
    __shared__ RgLoudness res;  // one instance per kernel: the helper is called once per block
    if (t == 0) {
        res.loudness_db = -20.0;  // empty histogram, or the fall-through of src/replaygain.rs:681
        res.total = total;
    }
    __syncthreads();
    if (total != 0) {
        const uint64_t threshold = (uint64_t)ceil((double)total * RG_ONE_MINUS_PERCENTILE);
        if (suffix >= threshold && above < threshold) {  // exactly one thread: re-read its 47 bins from the top
            uint64_t count = above;
            for (int i = RG_PCT_CHUNK - 1; i >= 0; --i) {
                const int b = t * RG_PCT_CHUNK + i;
                count += b < RG_HISTOGRAM_SIZE ? h[b] : 0u;
                if (count >= threshold) {
                    res.loudness_db = (double)(b - RG_HISTOGRAM_OFFSET) / 100.0;
                    break;
                }
            }
        }
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
                                                             uint32_t sample_rate, uint32_t file_type) {
    rg_track_result r;
    r.loudness_db = l.loudness_db;
    r.gain_db = RG_PINK_REF - l.loudness_db;
    r.peak = peak;
    r.sample_rate = sample_rate;
    r.gain_steps = rg_round_steps(r.gain_db);
    r.windows = (uint32_t)l.total;
    r.file_type = file_type;
    *out = r;
}
