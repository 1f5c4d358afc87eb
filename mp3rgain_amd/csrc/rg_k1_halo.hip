// rg_k1_halo.hip -- variant 1: halo-tiled, operation-order-faithful IIR + RMS + histogram kernel.
//
// COMPILED WITH -ffp-contract=off.  One work item (lane) owns `seg_windows` consecutive 50 ms
// windows of one track, both channels.  It first runs the filter cascade over `halo` frames
// before its segment from a zero state (or from the true track start when the segment is
// closer than `halo` to it, in which case the state is exact), then processes its segment
// exactly as the reference's per-sample loop does:
//   process_audio_buffer           src/replaygain.rs:953-1029  (sample conversion, peak)
//   EqualLoudnessFilter::process   src/replaygain.rs:586-616   (direct form I, same sum order)
//   add_sample / add_mono_sample   src/replaygain.rs:720-740
//   finish_window                  src/replaygain.rs:743-765   (mean square -> dB*100 -> bin)
// Twelve frames per loop turn: the histories rotate through twelve physical slots instead of shifting (same
// operations on the same values, in the same order), and the turn's PCM arrives as three 16-byte loads per channel
// issued one turn ahead.
// With halo == UINT32_MAX and one segment per track the kernel is the reference's sequential
// algorithm verbatim (used for the unstable 88.2 kHz coefficient row and as a parity anchor).
// This variant is the correctness anchor; variant 2 (rg_k2_tm.hip) is the fast path.
#include <hip/hip_runtime.h>
#include <limits.h>

#include <type_traits>

#include "rg_device.h"
#include "rg_device_inl.h"

// f(integral_constant<int, 0>) ... f(integral_constant<int, 10>)
#define RG_K1_ELEVEN(f)                                                                                       \
    do {                                                                                                      \
        f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{}); f(std::integral_constant<int, 2>{});   \
        f(std::integral_constant<int, 3>{});                                                                             \
        __builtin_amdgcn_sched_barrier(0); /* keeps the loads of later frames from being hoisted all the way up */      \
        f(std::integral_constant<int, 4>{}); f(std::integral_constant<int, 5>{}); f(std::integral_constant<int, 6>{});   \
        f(std::integral_constant<int, 7>{});                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        f(std::integral_constant<int, 8>{}); f(std::integral_constant<int, 9>{}); f(std::integral_constant<int, 10>{}); \
    } while (0)

// f(integral_constant<int, 0>) ... f(integral_constant<int, 11>), in three groups of four (one 16-byte piece each)
#define RG_K1_TWELVE(f)                                                                                       \
    do {                                                                                                      \
        f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{}); f(std::integral_constant<int, 2>{});   \
        f(std::integral_constant<int, 3>{});                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        f(std::integral_constant<int, 4>{}); f(std::integral_constant<int, 5>{}); f(std::integral_constant<int, 6>{});   \
        f(std::integral_constant<int, 7>{});                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        f(std::integral_constant<int, 8>{}); f(std::integral_constant<int, 9>{}); f(std::integral_constant<int, 10>{}); \
        f(std::integral_constant<int, 11>{});                                                                            \
    } while (0)

namespace {

// twelve physical slots for the eleven-entry histories: one spare, so that the rotation period is twelve frames --
// three 16-byte pieces of PCM -- instead of eleven
struct Df1State {
    double yx[12], yy[12], bx[3], by[3];
};

__device__ __forceinline__ void df1_reset(Df1State &f) {
#pragma unroll
    for (int i = 0; i < 12; ++i) { f.yx[i] = 0.0; f.yy[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { f.bx[i] = 0.0; f.by[i] = 0.0; }
}

// EqualLoudnessFilter::process, src/replaygain.rs:586-616 -- same shifts, same fold order.
__device__ __forceinline__ double df1_process(Df1State &f, const RgCoefDev &c, double s) {
#pragma unroll
    for (int i = 10; i >= 1; --i) { f.yx[i] = f.yx[i - 1]; f.yy[i] = f.yy[i - 1]; }
    f.yx[0] = s;
    double acc = 0.0;
#pragma unroll
    for (int i = 1; i < 11; ++i) {
        double t = c.yb[i] * f.yx[i] - c.ya[i] * f.yy[i];
        acc = acc + t;
    }
    const double y = (1e-10 + c.yb[0] * f.yx[0]) + acc;
    f.yy[0] = y;

    f.bx[2] = f.bx[1]; f.bx[1] = f.bx[0];
    f.by[2] = f.by[1]; f.by[1] = f.by[0];
    f.bx[0] = y;
    double acc2 = 0.0;
#pragma unroll
    for (int i = 1; i < 3; ++i) {
        double t = c.bb[i] * f.bx[i] - c.ba[i] * f.by[i];
        acc2 = acc2 + t;
    }
    const double z = (1e-10 + c.bb[0] * f.bx[0]) + acc2;
    f.by[0] = z;
    return z;
}

// The same step with the Yule-Walker histories rotated instead of shifted.  K = 0..11 counts the frames since the
// histories were last in their natural order (logical entry i in physical slot i).  The shift of a frame would move
// x_buf[i-1] to x_buf[i]; after K + 1 of them the value the reference holds in x_buf[i] / y_buf[i] lives in physical
// slot (i - K - 1) mod 12 without having been moved, and the new sample goes to slot (11 - K).  The twenty moves of
// the shift disappear when twelve consecutive frames are unrolled with K = 0..11 -- and after the twelfth the
// histories are in natural order again (slot 11 then holds the entry the reference has just dropped).
// Same products, same subtraction, same fold order: bit-identical to df1_process.
template <int K>
__device__ __forceinline__ double df1_process_rot(Df1State &f, const RgCoefDev &c, double s) {
    constexpr int P0 = (11 - K) % 12;
    f.yx[P0] = s;
    double acc = 0.0;
#pragma unroll
    for (int i = 1; i < 11; ++i) {
        const int p = (i - K - 1 + 24) % 12;
        double t = c.yb[i] * f.yx[p] - c.ya[i] * f.yy[p];
        acc = acc + t;
    }
    const double y = (1e-10 + c.yb[0] * f.yx[P0]) + acc;
    f.yy[P0] = y;

    f.bx[2] = f.bx[1]; f.bx[1] = f.bx[0];
    f.by[2] = f.by[1]; f.by[1] = f.by[0];
    f.bx[0] = y;
    double acc2 = 0.0;
#pragma unroll
    for (int i = 1; i < 3; ++i) {
        double t = c.bb[i] * f.bx[i] - c.ba[i] * f.by[i];
        acc2 = acc2 + t;
    }
    const double z = (1e-10 + c.bb[0] * f.bx[0]) + acc2;
    f.by[0] = z;
    return z;
}

// sample conversion arms of process_audio_buffer (src/replaygain.rs:959-1024):
// returns the filter input, writes the normalised magnitude used for the peak.
__device__ __forceinline__ double load_input(const void *base, uint64_t i, uint32_t fmt, double &mag) {
    if (fmt == RG_FMT_F32_PLANAR) {
        const double xn = (double)((const float *)base)[i];
        mag = fabs(xn);
        return xn * 32768.0;
    } else if (fmt == RG_FMT_S16_PLANAR) {
        const double v = (double)((const int16_t *)base)[i];
        mag = fabs(v / 32768.0);
        return v;
    } else {
        const double v = (double)((const int32_t *)base)[i] * (32768.0 / 2147483648.0);
        mag = fabs(v / 32768.0);
        return v;
    }
}

// the same conversion from a staged 32-bit word (float bits, or a sign-extended integer)
__device__ __forceinline__ double cvt_input(uint32_t w, uint32_t fmt, double &mag) {
    if (fmt == RG_FMT_F32_PLANAR) {
        const double xn = (double)__uint_as_float(w);
        mag = fabs(xn);
        return xn * 32768.0;
    } else if (fmt == RG_FMT_S16_PLANAR) {
        const double v = (double)(int32_t)w;
        mag = fabs(v / 32768.0);
        return v;
    } else {
        const double v = (double)(int32_t)w * (32768.0 / 2147483648.0);
        mag = fabs(v / 32768.0);
        return v;
    }
}

typedef uint32_t __attribute__((ext_vector_type(4), aligned(4))) k1_u32x4u;  // 16-byte load, 4-byte aligned
typedef short __attribute__((ext_vector_type(4), aligned(2))) k1_s16x4u;     // 8-byte load, 2-byte aligned

// four consecutive samples of one channel as 32-bit words
__device__ __forceinline__ uint4 load_piece(const void *base, uint64_t i, uint32_t fmt) {
    if (fmt == RG_FMT_S16_PLANAR) {
        const k1_s16x4u v = *(const k1_s16x4u *)((const int16_t *)base + i);
        return make_uint4((uint32_t)(int32_t)v.x, (uint32_t)(int32_t)v.y, (uint32_t)(int32_t)v.z, (uint32_t)(int32_t)v.w);
    }
    const k1_u32x4u v = *(const k1_u32x4u *)((const uint32_t *)base + i);
    return make_uint4(v.x, v.y, v.z, v.w);
}

}  // namespace

// A sample that is not finite (float PCM only) leaves the reference's filter state NaN for the rest of the track
// (src/replaygain.rs:586-616 never resets it), so every window from that frame on is a NaN window.  A lane that
// warms up from zero cannot know about a NaN before its halo: one streaming pass finds the first such frame of
// every track (channels 0 and 1, the ones that are analysed) before the lanes run.
__global__ void __launch_bounds__(256) rg_k1_first_nonfinite_kernel(const RgTrackDev *__restrict__ tracks,
                                                                    unsigned long long *__restrict__ first_bad, uint32_t n_tracks) {
  // gridDim.y is capped at 65535: a block row walks the tracks t, t + gridDim.y, ...
  for (uint32_t t = blockIdx.y; t < n_tracks; t += gridDim.y) {
    const RgTrackDev tr = tracks[t];
    if (tr.format != RG_FMT_F32_PLANAR) continue;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long best = ~0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tr.frames; i += stride) {
        const float a = ((const float *)tr.ch0)[i];
        const float b = tr.ch1 ? ((const float *)tr.ch1)[i] : 0.0f;
        if (!(fabsf(a) <= 3.402823466e38f) || !(fabsf(b) <= 3.402823466e38f)) { best = i; break; }  // ascending per thread
    }
    if (best != ~0ull) atomicMin(&first_bad[tr.track_index], best);
  }
}

__global__ void __launch_bounds__(256) rg_k1_halo_kernel(const RgTrackDev *__restrict__ tracks, uint32_t n_tracks,
                                                         uint32_t total_items, const RgCoefDev *__restrict__ coefs,
                                                         uint32_t *__restrict__ hist,
                                                         unsigned long long *__restrict__ peak_bits,
                                                         const unsigned long long *__restrict__ first_bad) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total_items) return;

    // largest t with item_base[t] <= g (tracks with zero items share a base with their successor)
    uint32_t lo = 0, hi = n_tracks - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (tracks[mid].item_base <= g) lo = mid; else hi = mid - 1;
    }
    const uint32_t t = lo;
    const RgTrackDev tr = tracks[t];
    const uint32_t seg = g - tr.item_base;
    if (seg >= tr.n_segments) return;
    // every wave holds items of one sample rate (rg_enqueue.hip pads the item list to a wave boundary where the rate
    // changes), so the 28 filter constants are wave-uniform: scalar loads into SGPRs instead of 56 VGPRs per lane
    const RgCoefDev c = coefs[__builtin_amdgcn_readfirstlane(tr.coef_idx)];

    const uint64_t W = tr.window;
    const uint64_t first = (uint64_t)seg * tr.seg_windows * W;
    uint64_t last = first + (uint64_t)tr.seg_windows * W;
    if (last > tr.frames) last = tr.frames;
    const uint64_t warm = (tr.halo != 0xFFFFFFFFu && first > tr.halo) ? first - tr.halo : 0;
    const bool stereo = tr.ch1 != nullptr;
    const uint32_t fmt = tr.format;

    Df1State fl, fr;
    df1_reset(fl);
    df1_reset(fr);
    double mag;

    double peak = 0.0, lsum = 0.0, rsum = 0.0;
    uint32_t n = 0;
    uint32_t *const h = hist + (size_t)tr.track_index * RG_HISTOGRAM_SIZE;
    const uint64_t bad_from = first_bad[tr.track_index];  // ~0 when every sample is finite
    const double qnan = __longlong_as_double(0x7FF8000000000000ll);
    // One frame.  Frames before `first` are the warm-up halo: the cascade runs, nothing is counted.
    // ROT >= 0 selects the rotated step of that phase, -1 the shifting one.
    auto frame = [&](const uint64_t i, const double xl, const double ml, const double xr, const double mr, auto rot) {
        constexpr int ROT = decltype(rot)::value;
        double lf, rf = 0.0;
        if constexpr (ROT >= 0) lf = df1_process_rot<ROT>(fl, c, xl);
        else lf = df1_process(fl, c, xl);
        if (stereo) {
            if constexpr (ROT >= 0) rf = df1_process_rot<ROT>(fr, c, xr);
            else rf = df1_process(fr, c, xr);
        }
        if (i < first) return;
        if (ml > peak) peak = ml;
        if (stereo) {
            if (mr > peak) peak = mr;
            lsum += lf * lf;
            rsum += rf * rf;
        } else {
            const double sq = lf * lf;
            lsum += sq;
            rsum += sq;
        }
        if (++n >= W) {
            // A window that starts after the first non-finite sample is a NaN window in the reference (the filter state
            // never recovers) whatever this lane's zero-warmed state says.  The window that HOLDS that sample went through
            // this lane's own cascade, in the reference's order: its sum is the reference's -- NaN, or +Inf when an Inf is
            // its very last frame (then `val as i32` saturates, the index wraps and the window is dropped, :749-759).
            if (i + 1 - n > bad_from) lsum = qnan;
            const int idx = rg_window_bin(lsum, rsum, n);
            if (idx >= 0) atomicAdd(&h[idx], 1u);
            lsum = 0.0; rsum = 0.0; n = 0;
        }
    };
    {
        // Twelve frames per turn: rotated histories (no shifts), and the PCM of a turn is three 16-byte loads per
        // channel issued one turn ahead (per-sample loads from 64 different rows cost a cache line each and sat on
        // the critical path: a lone 3-second track took 4.3 ms, 18x its bytes came from HBM).
        uint64_t i = warm;
        uint4 pl[3], pr[3];  // the three pieces of the current turn; piece p is refilled for the next turn as soon as
                             // its four frames are done, so one set of registers serves as the prefetch buffer
        if (i + 12 <= last) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                pl[p] = load_piece(tr.ch0, i + 4u * p, fmt);
                if (stereo) pr[p] = load_piece(tr.ch1, i + 4u * p, fmt);
            }
        }
        for (; i + 12 <= last; i += 12) {
            const bool more = i + 24 <= last;
            auto four = [&](auto piece) {
                constexpr int P = decltype(piece)::value;
                const uint32_t wl[4] = {pl[P].x, pl[P].y, pl[P].z, pl[P].w};
                const uint32_t wr[4] = {pr[P].x, pr[P].y, pr[P].z, pr[P].w};
                if (more) {
                    pl[P] = load_piece(tr.ch0, i + 12u + 4u * P, fmt);
                    if (stereo) pr[P] = load_piece(tr.ch1, i + 12u + 4u * P, fmt);
                }
                auto step = [&](auto k) {
                    constexpr int K = decltype(k)::value;
                    double ml, mr = 0.0, xr = 0.0;
                    const double xl = cvt_input(wl[K & 3], fmt, ml);
                    if (stereo) xr = cvt_input(wr[K & 3], fmt, mr);
                    frame(i + K, xl, ml, xr, mr, k);
                };
                step(std::integral_constant<int, 4 * P>{});
                step(std::integral_constant<int, 4 * P + 1>{});
                step(std::integral_constant<int, 4 * P + 2>{});
                step(std::integral_constant<int, 4 * P + 3>{});
                __builtin_amdgcn_sched_barrier(0);
            };
            four(std::integral_constant<int, 0>{});
            four(std::integral_constant<int, 1>{});
            four(std::integral_constant<int, 2>{});
        }
        for (; i < last; ++i) {
            double ml, mr = 0.0, xr = 0.0;
            const double xl = load_input(tr.ch0, i, fmt, ml);
            if (stereo) xr = load_input(tr.ch1, i, fmt, mr);
            frame(i, xl, ml, xr, mr, std::integral_constant<int, -1>{});
        }
    }
    (void)mag;
    if (n > 0) {  // final partial window, src/replaygain.rs:907
        if (last - n > bad_from) lsum = qnan;
        const int idx = rg_window_bin(lsum, rsum, n);
        if (idx >= 0) atomicAdd(&h[idx], 1u);
    }
    atomicMax(&peak_bits[tr.track_index], (unsigned long long)__double_as_longlong(peak));
}

extern "C" hipError_t rg_launch_k1_halo(const RgTrackDev *d_tracks, uint32_t n_tracks, uint32_t total_items,
                                        const RgCoefDev *d_coefs, uint32_t *d_hist,
                                        unsigned long long *d_peak_bits, unsigned long long *d_first_bad /* preset to ~0 */,
                                        hipStream_t stream) {
    if (total_items == 0) return hipSuccess;
    hipLaunchKernelGGL(rg_k1_first_nonfinite_kernel, dim3(512, n_tracks < 65535u ? n_tracks : 65535u), dim3(256), 0, stream, d_tracks,
                       d_first_bad, n_tracks);
    const uint32_t block = 64;  // one wave per workgroup: spreads a small item count over all CUs
    const uint32_t grid = (total_items + block - 1) / block;
    hipLaunchKernelGGL(rg_k1_halo_kernel, dim3(grid), dim3(block), 0, stream, d_tracks, n_tracks, total_items,
                       d_coefs, d_hist, d_peak_bits, d_first_bad);
    return hipGetLastError();
}
