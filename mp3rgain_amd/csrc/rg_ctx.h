// rg_ctx.h -- the context object behind rg_ctx* and small host helpers shared by the host
// translation units (rg_capi.hip: API surface; rg_enqueue.hip: batch set-up and launches).
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "rg_design.h"
#include "rg_device.h"
#include "rg_tm.h"

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;  // elements
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 16;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

template <typename T>
struct PinnedBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 16;
        hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

// device-resident tables of one (sample rate, segment length) pair for variant 2
struct RgTmDeviceTables {
    RgTmDesign design;       // host copy (vectors kept for diagnostics)
    double *d_blob = nullptr;  // one allocation: T | Gp | PhiY | PhiB | X | sigma0
    RgTmGeom geom{};
    RgTmFixTables fix{};
};

enum { RG_TUNE_TM_SEGMENT = 1, RG_TUNE_TM_TARGET_LANES = 2 };

struct rg_ctx {
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    int kernel_variant = 0;      // 0 auto (= 2), 1 halo/reference-order kernel, 2 transient-moment kernels
    uint32_t tune_tm_segment = 0;          // 0 = choose from the workload
    uint64_t tune_tm_target_lanes = 0;     // 0 = default

    RgRateDesign design[RG_NUM_RATES];
    DevBuf<RgCoefDev> d_coefs;

    DevBuf<RgTrackDev> d_tracks;       // all tracks of the batch, index == track index
    PinnedBuf<RgTrackDev> h_tracks;
    DevBuf<RgTrackDev> d_k1_tracks;    // the tracks variant 1 processes (a subset under variant 2)
    PinnedBuf<RgTrackDev> h_k1_tracks;
    DevBuf<RgTmTrack> d_tm_tracks;     // variant 2 launch lists, all groups back to back
    PinnedBuf<RgTmTrack> h_tm_tracks;
    DevBuf<double> d_tm_rec;           // segment records (rg_tm.h)
    std::map<uint32_t, RgTmDeviceTables *> tm_tables;  // key = rate_idx << 16 | L
    hipEvent_t staging_done = nullptr; // H2D copies out of the pinned staging buffers have finished
    bool staging_pending = false;

    DevBuf<uint32_t> d_hist;
    DevBuf<unsigned long long> d_peak_bits;
    DevBuf<rg_track_result> d_results;
    PinnedBuf<rg_track_result> h_results;
    DevBuf<uint32_t> d_album_hist;
    DevBuf<double> d_album_peak;
    DevBuf<rg_album_result> d_album_result;
    PinnedBuf<rg_album_result> h_album_result;
    DevBuf<unsigned char> d_arena;  // staging for host PCM

    size_t n_enqueued = 0;
    bool album_ready = false;

    // timing of the dominant kernel
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    double timing_sum_ms = 0.0;
    uint64_t timing_count = 0;
};

int rg_set_err(rg_ctx *c, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
int rg_rate_index(uint32_t sr);
static inline size_t rg_bytes_per_sample(uint32_t fmt) { return fmt == RG_FMT_S16_PLANAR ? 2 : 4; }
int rg_bind_device(rg_ctx *c);
// the whole analysis of one batch, enqueued on c->stream (rg_enqueue.hip)
int rg_enqueue_impl(rg_ctx *c, const rg_track_desc *tracks, size_t n, const void *d_pcm_base, size_t pcm_bytes,
                    int album);
void rg_tm_tables_release(rg_ctx *c);

#define RG_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return rg_set_err((ctx), RG_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e__)); \
    } while (0)
