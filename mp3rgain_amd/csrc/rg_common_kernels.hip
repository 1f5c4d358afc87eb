// rg_common_kernels.hip -- small kernels around the IIR kernel: histogram percentile, album merge,
// all-channel peak scan and the synthetic-PCM generator.
#include <hip/hip_runtime.h>

#include "../../include/rg_synth.h"
#include "rg_device.h"
#include "rg_device_inl.h"

// ---------------------------------------------------------------------------------------------
// LoudnessHistogram::get_loudness (src/replaygain.rs:665-682) for one histogram per workgroup,
// followed by the tail of analyze_track_internal (src/replaygain.rs:910-918): gain = PINK_REF -
// loudness, gain_steps = round(gain / 1.5).
//   total     = sum of bins (u64)
//   threshold = ceil(total as f64 * (1.0 - 0.95)) as u64
//   scan i = 11999 .. 0 accumulating; first i with count >= threshold -> (i - 2000) / 100.0
// 256 threads own 47 consecutive bins each; one thread then walks the 256 chunk sums from the top
// and finishes inside the crossing chunk, so the scan order and the result are exactly the
// sequential ones.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RG_PCT_THREADS)
rg_track_result_kernel(const uint32_t *__restrict__ hist, const unsigned long long *__restrict__ peak_bits,
                       const RgTrackDev *__restrict__ list /* the tracks to finish, any subset */,
                       const unsigned long long *__restrict__ first_bad /* variant 1: first non-finite frame, or ~0 */,
                       rg_track_result *__restrict__ out) {
    __shared__ uint64_t scan[RG_PCT_THREADS];
    const RgTrackDev tr = list[blockIdx.x];
    const uint32_t t = tr.track_index;
    const RgLoudness l = rg_block_loudness(hist + (size_t)t * RG_HISTOGRAM_SIZE, scan);
    if (threadIdx.x == 0)
        rg_store_track_result(out + t, l, __longlong_as_double((long long)peak_bits[t]), tr.sample_rate, tr.file_type,
                              first_bad[t] != ~0ull ? RG_TRACK_FLAG_NONFINITE : 0u);
}

__global__ void __launch_bounds__(RG_PCT_THREADS)
rg_album_result_kernel(const uint32_t *__restrict__ album_hist, const double *__restrict__ album_peak,
                       rg_album_result *__restrict__ out) {
    __shared__ uint64_t scan[RG_PCT_THREADS];
    const RgLoudness l = rg_block_loudness(album_hist, scan);
    if (threadIdx.x == 0) {
        rg_album_result r;
        r.album_loudness_db = l.loudness_db;
        r.album_gain_db = RG_PINK_REF - l.loudness_db;
        r.album_peak = *album_peak;
        r.album_gain_steps = rg_round_steps(r.album_gain_db);
        r.windows = (uint32_t)l.total;
        *out = r;
    }
}

// ---------------------------------------------------------------------------------------------
// LoudnessHistogram::accumulate over this GPU's tracks + album_peak = max (src/replaygain.rs:1056-1059).
// Bins are u32 and the adds wrap, as the reference's release build does.
// blocks [0, 47): 256 bins each; block 47: peak.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
rg_album_merge_kernel(const uint32_t *__restrict__ hist, const unsigned long long *__restrict__ peak_bits,
                      uint32_t n_tracks, uint32_t *__restrict__ album_hist, double *__restrict__ album_peak) {
    const int nb = (RG_HISTOGRAM_SIZE + 255) / 256;
    if ((int)blockIdx.x < nb) {
        const int b = blockIdx.x * 256 + threadIdx.x;
        if (b < RG_HISTOGRAM_SIZE) {
            uint32_t s = 0;
            for (uint32_t t = 0; t < n_tracks; ++t) s += hist[(size_t)t * RG_HISTOGRAM_SIZE + b];
            album_hist[b] = s;
        }
    } else {
        __shared__ unsigned long long m[256];
        unsigned long long v = 0;
        for (uint32_t t = threadIdx.x; t < n_tracks; t += 256) {
            const unsigned long long p = peak_bits[t];
            v = p > v ? p : v;
        }
        m[threadIdx.x] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) m[threadIdx.x] = m[threadIdx.x + s] > m[threadIdx.x] ? m[threadIdx.x + s] : m[threadIdx.x];
            __syncthreads();
        }
        if (threadIdx.x == 0) *album_peak = __longlong_as_double((long long)m[0]);
    }
}

// ---------------------------------------------------------------------------------------------
// The same merge one level up: every rank's [album histogram | album peak] pack (12000 + 2 words) after
// an all-gather, `world` packs back to back -> this rank's album histogram / peak.  One collective
// instead of an all-reduce(sum) plus an all-reduce(max).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
rg_album_reduce_gathered_kernel(const uint32_t *__restrict__ g, uint32_t world, uint32_t *__restrict__ album_hist,
                                double *__restrict__ album_peak) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    const size_t pack = RG_HISTOGRAM_SIZE + 2;
    if (b < RG_HISTOGRAM_SIZE) {
        uint32_t s = 0;
        for (uint32_t r = 0; r < world; ++r) s += g[(size_t)r * pack + b];
        album_hist[b] = s;
    } else if (b == RG_HISTOGRAM_SIZE) {
        double m = 0.0;
        for (uint32_t r = 0; r < world; ++r) {
            const double p = *reinterpret_cast<const double *>(g + (size_t)r * pack + RG_HISTOGRAM_SIZE);
            m = p > m ? p : m;
        }
        *album_peak = m;
    }
}

// ---------------------------------------------------------------------------------------------
// find_peak_amplitude's scan (src/replaygain.rs:1210-1241): max |x| over ALL channels, normalised.
// `total` samples = channels * frames, contiguous because the layout is planar.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
rg_peak_all_kernel(const void *__restrict__ base, uint64_t total, uint32_t fmt, unsigned long long *__restrict__ peak_bits) {
    double peak = 0.0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        double v;
        if (fmt == RG_FMT_F32_PLANAR) v = (double)fabsf(((const float *)base)[i]);
        else if (fmt == RG_FMT_S16_PLANAR) v = fabs((double)((const int16_t *)base)[i]) / 32768.0;
        else v = fabs((double)((const int32_t *)base)[i]) / 2147483648.0;
        if (v > peak) peak = v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(peak, off, 64);
        if (o > peak) peak = o;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(peak_bits, (unsigned long long)__double_as_longlong(peak));
}

// ---------------------------------------------------------------------------------------------
// synthetic PCM straight into HBM (include/rg_synth.h); 4 consecutive frames per lane, 16-B stores
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
rg_synth_fill_kernel(float *__restrict__ dst, uint64_t seed, uint32_t channel, uint32_t sample_rate,
                     uint64_t first_frame, uint64_t frames) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 4;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < frames; i += stride) {
        if (i + 4 <= frames && (((uintptr_t)(dst + i)) & 15) == 0) {
            float4 v;
            v.x = rg_synth_sample_f32(seed, channel, sample_rate, first_frame + i);
            v.y = rg_synth_sample_f32(seed, channel, sample_rate, first_frame + i + 1);
            v.z = rg_synth_sample_f32(seed, channel, sample_rate, first_frame + i + 2);
            v.w = rg_synth_sample_f32(seed, channel, sample_rate, first_frame + i + 3);
            *reinterpret_cast<float4 *>(dst + i) = v;
        } else {
            for (uint64_t k = i; k < frames && k < i + 4; ++k)
                dst[k] = rg_synth_sample_f32(seed, channel, sample_rate, first_frame + k);
        }
    }
}

// ---- launch wrappers (plain C linkage so the host TU needs no kernel declarations) ---------------
extern "C" hipError_t rg_launch_track_results(const uint32_t *d_hist, const unsigned long long *d_peak_bits,
                                              const RgTrackDev *d_tracks, const unsigned long long *d_first_bad,
                                              rg_track_result *d_out, uint32_t n_tracks, hipStream_t s) {
    if (n_tracks == 0) return hipSuccess;
    hipLaunchKernelGGL(rg_track_result_kernel, dim3(n_tracks), dim3(RG_PCT_THREADS), 0, s, d_hist, d_peak_bits,
                       d_tracks, d_first_bad, d_out);
    return hipGetLastError();
}

extern "C" hipError_t rg_launch_album_merge(const uint32_t *d_hist, const unsigned long long *d_peak_bits,
                                            uint32_t n_tracks, uint32_t *d_album_hist, double *d_album_peak,
                                            hipStream_t s) {
    const int nb = (RG_HISTOGRAM_SIZE + 255) / 256;
    hipLaunchKernelGGL(rg_album_merge_kernel, dim3(nb + 1), dim3(256), 0, s, d_hist, d_peak_bits, n_tracks,
                       d_album_hist, d_album_peak);
    return hipGetLastError();
}

extern "C" hipError_t rg_launch_album_reduce_gathered(const uint32_t *d_gathered, uint32_t world, uint32_t *d_album_hist,
                                                      double *d_album_peak, hipStream_t s) {
    const int nb = (RG_HISTOGRAM_SIZE + 1 + 255) / 256;
    hipLaunchKernelGGL(rg_album_reduce_gathered_kernel, dim3(nb), dim3(256), 0, s, d_gathered, world, d_album_hist,
                       d_album_peak);
    return hipGetLastError();
}

extern "C" hipError_t rg_launch_album_result(const uint32_t *d_album_hist, const double *d_album_peak,
                                             rg_album_result *d_out, hipStream_t s) {
    hipLaunchKernelGGL(rg_album_result_kernel, dim3(1), dim3(RG_PCT_THREADS), 0, s, d_album_hist, d_album_peak,
                       d_out);
    return hipGetLastError();
}

extern "C" hipError_t rg_launch_peak_all(const void *d_base, uint64_t total, uint32_t fmt,
                                         unsigned long long *d_peak_bits, hipStream_t s) {
    if (total == 0) return hipSuccess;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(rg_peak_all_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, d_base, total, fmt, d_peak_bits);
    return hipGetLastError();
}

extern "C" hipError_t rg_launch_synth_fill(float *d_dst, uint64_t seed, uint32_t channel, uint32_t sample_rate,
                                           uint64_t first_frame, uint64_t frames, hipStream_t s) {
    if (frames == 0) return hipSuccess;
    uint64_t blocks = (frames / 4 + 255) / 256 + 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rg_synth_fill_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, d_dst, seed, channel,
                       sample_rate, first_frame, frames);
    return hipGetLastError();
}

// One wave that keeps its queue busy for `ticks` of the 100 MHz wall clock: rg_create launches it on all pipeline streams at once
// to see whether they run beside each other (rg_capi.hip: check_hw_queues).
__global__ void __launch_bounds__(64) rg_spin_kernel(uint64_t ticks) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
extern "C" hipError_t rg_launch_spin(uint64_t ticks, hipStream_t s) {
    hipLaunchKernelGGL(rg_spin_kernel, dim3(1), dim3(64), 0, s, ticks);
    return hipGetLastError();
}
