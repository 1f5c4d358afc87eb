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
