// rg_capi.hip -- host orchestration behind include/mp3rgain_amd.h.
//
// Mirrors analyze_track_internal from the filters onwards (src/replaygain.rs:866-925) and
// analyze_album_with_index (src/replaygain.rs:1044-1074) on decoded planar PCM.  Everything after
// the H2D copy (or nothing at all when the PCM is already in HBM) runs on one HIP stream:
//   memset(hist, peak) -> K1 IIR+RMS+histogram+peak -> per-track percentile/result
//   [album] -> merge -> (caller's all-reduce) -> album percentile
// There is no CPU compute path in this file by design.
#include <atomic>
#include <chrono>
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "rg_ctx.h"

// ---- kernel launchers (defined in the kernel translation units) --------------------------------
extern "C" {
hipError_t rg_launch_k1_halo(const RgTrackDev *, uint32_t, uint32_t, const RgCoefDev *, uint32_t *,
                             unsigned long long *, unsigned long long *, hipStream_t);
hipError_t rg_launch_track_results(const uint32_t *, const unsigned long long *, const RgTrackDev *,
                                   const unsigned long long *, rg_track_result *, uint32_t, hipStream_t);
hipError_t rg_launch_album_merge(const uint32_t *, const unsigned long long *, uint32_t, uint32_t *, double *,
                                 hipStream_t);
hipError_t rg_launch_album_result(const uint32_t *, const double *, rg_album_result *, hipStream_t);
hipError_t rg_launch_album_reduce_gathered(const uint32_t *, uint32_t, uint32_t *, double *, hipStream_t);
hipError_t rg_launch_peak_all(const void *, uint64_t, uint32_t, unsigned long long *, hipStream_t);
hipError_t rg_launch_synth_fill(float *, uint64_t, uint32_t, uint32_t, uint64_t, uint64_t, hipStream_t);
}

namespace {
thread_local std::string g_create_error;
}

int rg_set_err(rg_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

int rg_rate_index(uint32_t sr) {
    for (int i = 0; i < RG_NUM_RATES; ++i)
        if (RG_RATE_TABLE[i].sample_rate == sr) return i;
    return -1;
}

int rg_bind_device(rg_ctx *c) {
    RG_HIP(c, hipSetDevice(c->device));
    return RG_OK;
}

namespace {

int32_t round_to_i32(double v) {  // Rust: f64::round() as i32
    double r = round(v);
    if (r != r) return 0;
    if (r >= 2147483647.0) return INT32_MAX;
    if (r <= -2147483648.0) return INT32_MIN;
    return (int32_t)r;
}

// flush finished timing events into the running sum (requires the stream to be idle)
int drain_timing(rg_ctx *c) {
    for (int k = 0; k < RG_MAX_SLOTS; ++k) {
        RgSlot &S = c->slots[k];
        for (size_t i = 0; i < S.ev_used; ++i) {
            float ms = 0.f;
            RG_HIP(c, hipEventElapsedTime(&ms, S.ev_pool[i].first, S.ev_pool[i].second));
            c->timing_sum_ms += ms;
            c->timing_count += 1;
            if (c->timing_first) {
                float sp = 0.f;
                RG_HIP(c, hipEventElapsedTime(&sp, c->timing_first, S.ev_pool[i].second));
                if (sp > c->timing_span_ms) c->timing_span_ms = sp;
            }
        }
        S.ev_used = 0;
    }
    return RG_OK;
}

int sync_all(rg_ctx *c) {
    for (int k = 0; k < RG_MAX_SLOTS; ++k)
        if (c->slots[k].stream) RG_HIP(c, hipStreamSynchronize(c->slots[k].stream));
    if (c->user_attached) RG_HIP(c, hipStreamSynchronize(c->user_stream));
    return RG_OK;
}

}  // namespace

// ================================ pure helpers =====================================================
extern "C" int rg_abi_version(void) { return RG_ABI_VERSION; }
extern "C" int rg_is_available(void) { return 1; }
extern "C" int rg_supported_rate(uint32_t sr) { return rg_rate_index(sr) >= 0 ? 1 : 0; }
extern "C" uint32_t rg_window_samples(uint32_t sr) { return (uint32_t)(((uint64_t)sr * 50u) / 1000u); }

extern "C" double rg_hist_loudness(const uint32_t *hist) {
    if (!hist) return -20.0;
    uint64_t total = 0;
    for (int i = 0; i < RG_HISTOGRAM_SIZE; ++i) total += hist[i];
    if (total == 0) return -20.0;
    const uint64_t threshold = (uint64_t)ceil((double)total * RG_ONE_MINUS_PERCENTILE);
    uint64_t count = 0;
    for (int i = RG_HISTOGRAM_SIZE - 1; i >= 0; --i) {
        count += hist[i];
        if (count >= threshold) return (double)(i - RG_HISTOGRAM_OFFSET) / 100.0;
    }
    return -20.0;
}

extern "C" double rg_gain_from_loudness(double l) { return RG_PINK_REF - l; }
extern "C" int32_t rg_gain_steps(double gain_db) { return round_to_i32(gain_db / RG_GAIN_STEP_DB); }
extern "C" int32_t rg_db_to_steps(double db) { return round_to_i32(db / RG_GAIN_STEP_DB); }
extern "C" double rg_steps_to_db(int32_t steps) { return (double)steps * RG_GAIN_STEP_DB; }

extern "C" int32_t rg_clip_limit_steps(int32_t steps, double gain_db, double peak, int prevent_clipping, int wrap) {
    int32_t actual = steps;
    if (steps > 0 && !wrap) {
        const double new_peak = peak * pow(10.0, gain_db / 20.0);
        if (new_peak > 1.0 && prevent_clipping) {
            const int32_t safe = rg_db_to_steps(-20.0 * log10(peak));
            actual = safe > 0 ? safe : 0;
        }
    }
    return actual;
}

extern "C" int rg_rate_design_info(uint32_t sr, int *stable, uint32_t *halo, double *decay) {
    const int ri = rg_rate_index(sr);
    if (ri < 0) return RG_ERR_UNSUPPORTED_RATE;
    RgRateDesign d;
    rg_design_rate(RG_RATE_TABLE[ri], &d);
    if (stable) *stable = d.stable ? 1 : 0;
    if (halo) *halo = d.halo_frames;
    if (decay) *decay = d.pole_radius;
    return RG_OK;
}

// ================================ context ==========================================================
extern "C" const char *rg_last_error(const rg_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

// The environment's routing / tracing knobs (INTEGRATION.md lists them), read once per context: the file layer used to call
// getenv per call, which races with a host application's setenv; a value that does not parse keeps the default and says so
// on stderr (a silent "nan" used to switch the album parts off).
static void read_env_defaults(rg_ctx *c) {
    auto number = [](const char *name, double lo, double *out) {
        const char *e = getenv(name);
        if (!e || !*e) return;
        char *end = nullptr;
        const double v = strtod(e, &end);
        if (end == e || *end != '\0' || !(v >= lo) || !(v < 1e18)) {
            fprintf(stderr, "mp3rgain_amd: %s=\"%s\" is not a number >= %g: ignored\n", name, e, lo);
            return;
        }
        *out = v;
    };
    if (const char *e = getenv("RG_ALBUM_PARTS")) c->env_parts_on = !(e[0] == '0');
    number("RG_PARTS_MIN_BYTES_PER_UNIT", 0.0, &c->env_parts_min_bpu);
    double v = 0.0;
    number("RG_MP3_STAGE_BYTES", 4096.0, &v);
    c->env_stage_bytes = (size_t)v;
    v = 0.0;
    number("RG_TRACKS_GROUP_BYTES", 1.0, &v);
    c->env_group_bytes = (size_t)v;
    c->trace_files = getenv("RG_TRACE_FILES") != nullptr;
    if (const char *e = getenv("RG_TRACE_TM")) c->trace_tm = e[0] == '2' ? 2 : 1;
}

extern "C" hipError_t rg_launch_spin(uint64_t ticks, hipStream_t s);
// Do the context's pipeline streams own a hardware queue each?  (a batch's fix-up runs under the next batch's
// main kernel only then; the runtime has GPU_MAX_HW_QUEUES = 4 queues per process by default and hands them to streams in the
// order of their creation.)  A wave that spins for 100 us goes to every stream at once: four queues finish together, shared ones
// one after the other.  Costs 0.4 ms per context; the finding is a line on stderr (once per process) and the text of
// rg_last_error until a real error replaces it -- results do not depend on it.
static void check_hw_queues(rg_ctx *c) {
    auto run = [&](int n_streams) -> double {
        for (int k = 0; k < n_streams; ++k)
            if (rg_launch_spin(1, c->slots[k].stream) != hipSuccess) return -1.0;
        for (int k = 0; k < n_streams; ++k)
            if (hipStreamSynchronize(c->slots[k].stream) != hipSuccess) return -1.0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < n_streams; ++k)
            if (rg_launch_spin(10000, c->slots[k].stream) != hipSuccess) return -1.0;
        for (int k = 0; k < n_streams; ++k)
            if (hipStreamSynchronize(c->slots[k].stream) != hipSuccess) return -1.0;
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    };
    const double one = run(1), all = run(RG_SLOT_STREAMS);
    if (one <= 0.0 || all <= 0.0) return;
    c->hw_queue_serial = all / one;
    if (all > 1.6 * one) {
        char buf[400];
        snprintf(buf, sizeof buf, "mp3rgain_amd: the context's %d pipeline streams share hardware queues (%.0f us for one spinning kernel, %.0f us for one on "
                 "each stream): create the context before other HIP streams and keep GPU_MAX_HW_QUEUES >= 4 (INTEGRATION.md); results are "
                 "unaffected, pipelined batches overlap less", RG_SLOT_STREAMS, one, all);
        c->err = buf;
        static std::atomic<bool> said{false};
        if (!said.exchange(true)) fprintf(stderr, "%s\n", buf);
    }
}

extern "C" rg_ctx *rg_create(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        rg_set_err(nullptr, RG_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= count) {
        rg_set_err(nullptr, RG_ERR_INVALID_ARG, "device ordinal %d out of range (have %d)", device, count);
        return nullptr;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
        rg_set_err(nullptr, RG_ERR_DEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
        return nullptr;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        rg_set_err(nullptr, RG_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device,
                prop.gcnArchName);
        return nullptr;
    }
    rg_ctx *c = new rg_ctx();
    c->device = device;
    read_env_defaults(c);
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->user_ev, hipEventDisableTiming);
    for (int k = 0; k < RG_MAX_SLOTS && e == hipSuccess; ++k) {
        RgSlot &S = c->slots[k];
        // RG_SLOT_STREAMS streams carry the batches; the slots beyond them are further BUFFER sets on the same
        // streams (slot k runs on stream k mod RG_SLOT_STREAMS).  A slot's buffers are reused n_slots batches
        // later, so with more slots than streams a batch never waits in its queue for the album tail (collective
        // + percentile on the caller's stream) of the batch that had the buffers before it.
        if (k < RG_SLOT_STREAMS) e = hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking);
        else S.stream = c->slots[k % RG_SLOT_STREAMS].stream;
        if (e == hipSuccess) e = hipEventCreateWithFlags(&S.staging_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&S.batch_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&S.album_done, hipEventDisableTiming);
        // [album histogram 12000 x u32 | album peak f64]: one contiguous pack, so that one all-gather moves both
        if (e == hipSuccess) e = S.d_album_hist.reserve(RG_HISTOGRAM_SIZE + 2);
        if (e == hipSuccess) S.d_album_peak.p = reinterpret_cast<double *>(S.d_album_hist.p + RG_HISTOGRAM_SIZE);
        if (e == hipSuccess) e = S.d_album_result.reserve(1);
        if (e == hipSuccess) e = S.h_album_result.reserve(1);
    }
    if (e != hipSuccess) {
        rg_set_err(nullptr, RG_ERR_DEVICE, "stream / buffer creation: %s", hipGetErrorString(e));
        rg_destroy(c);
        return nullptr;
    }

    std::vector<RgCoefDev> coefs(RG_NUM_RATES);
    for (int i = 0; i < RG_NUM_RATES; ++i) {
        rg_design_rate(RG_RATE_TABLE[i], &c->design[i]);
        memcpy(coefs[i].ya, RG_RATE_TABLE[i].yule_a, sizeof coefs[i].ya);
        memcpy(coefs[i].yb, RG_RATE_TABLE[i].yule_b, sizeof coefs[i].yb);
        memcpy(coefs[i].ba, RG_RATE_TABLE[i].butter_a, sizeof coefs[i].ba);
        memcpy(coefs[i].bb, RG_RATE_TABLE[i].butter_b, sizeof coefs[i].bb);
    }
    if ((e = c->d_coefs.reserve(RG_NUM_RATES)) != hipSuccess ||
        (e = hipMemcpy(c->d_coefs.p, coefs.data(), sizeof(RgCoefDev) * RG_NUM_RATES, hipMemcpyHostToDevice)) !=
            hipSuccess) {
        rg_set_err(nullptr, RG_ERR_DEVICE, "context allocation: %s", hipGetErrorString(e));
        rg_destroy(c);
        return nullptr;
    }
    check_hw_queues(c);
    return c;
}

extern "C" void rg_destroy(rg_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)sync_all(c);
    (void)rg_comm_destroy(c);
    for (int k = 0; k < RG_MAX_SLOTS; ++k) {
        RgSlot &S = c->slots[k];
        for (auto &p : S.ev_pool) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
        S.d_desc.release();
        S.h_desc.release();
        S.d_tm_rec.release();
        S.d_tm_win.release();
        S.d_hist.release();
        S.d_nonfinite.release();
        S.d_imprecise.release();
        S.d_k1_bad.release();
        S.d_results.release();
        S.h_results.release();
        S.d_album_hist.release();
        S.d_album_peak.p = nullptr;  // lives inside d_album_hist
        S.d_album_result.release();
        S.d_gather.release();
        S.h_album_result.release();
        if (S.staging_done) (void)hipEventDestroy(S.staging_done);
        if (S.batch_done) (void)hipEventDestroy(S.batch_done);
        if (S.album_done) (void)hipEventDestroy(S.album_done);
        if (S.stream && k < RG_SLOT_STREAMS) (void)hipStreamDestroy(S.stream);
    }
    if (c->user_ev) (void)hipEventDestroy(c->user_ev);
    c->d_coefs.release();
    c->d_peak_bits.release();
    c->d_arena.release();
    c->d_wav.release();
    if (c->file_pool && c->file_pool_free) c->file_pool_free(c->file_pool);
    c->file_pool = nullptr;
    c->d_mp3_tab.release();
    c->d_mp3_is.release();
    c->d_mp3_units.release();
    c->d_mp3_tracks.release();
    c->d_mp3_huff.release();
    c->d_mp3_recs.release();
    c->d_mp3_main.release();
    if (c->mp3_pipe && c->mp3_pipe_free) c->mp3_pipe_free(c->mp3_pipe);
    c->mp3_pipe = nullptr;
    c->d_mp3_stage[0].release();
    c->d_mp3_stage[1].release();
    c->d_mp3_results.release();
    c->d_mp3_tiles.release();
    for (int k = 0; k < 2; ++k) {
        c->d_mp3_recs_set[k].release();
        c->d_mp3_tiles_set[k].release();
        c->d_mp3_perm_set[k].release();
        c->d_mp3_sortw_set[k].release();
    }
    c->h_mp3_results.release();
    c->h_mp3_part_counts.release();
    c->h_part_results.release();
    for (int k = 0; k < 2; ++k)
        if (c->mp3_set_free[k]) (void)hipEventDestroy(c->mp3_set_free[k]);
    c->d_ingest[0].release();
    c->d_ingest[1].release();
    c->d_album_packs.release();
    if (c->ingest_stream) {
        (void)hipStreamDestroy(c->ingest_stream);
        for (int k = 0; k < 2; ++k) {
            (void)hipEventDestroy(c->ingest_copied[k]);
            (void)hipEventDestroy(c->ingest_free[k]);
        }
    }
    rg_tm_tables_release(c);
    delete c;
}

extern "C" int rg_set_stream(rg_ctx *c, void *s, int attach) {
    if (!c) return RG_ERR_INVALID_ARG;
    if (rg_bind_device(c) != RG_OK) return RG_ERR_DEVICE;
    int rc = sync_all(c);
    if (rc != RG_OK) return rc;
    c->user_stream = attach ? (hipStream_t)s : nullptr;
    c->user_attached = attach != 0;
    c->user_dirty = c->user_attached;
    return RG_OK;
}

extern "C" void *rg_batch_stream(rg_ctx *c) { return c ? (void *)c->slot().stream : nullptr; }

extern "C" int rg_wait_user_stream(rg_ctx *c) {
    if (!c) return RG_ERR_INVALID_ARG;
    c->user_dirty = c->user_attached;
    return RG_OK;
}

extern "C" int rg_set_kernel(rg_ctx *c, int variant) {
    if (!c) return RG_ERR_INVALID_ARG;
    if (variant < 0 || variant > 2) return rg_set_err(c, RG_ERR_INVALID_ARG, "unknown kernel variant %d", variant);
    c->kernel_variant = variant;
    return RG_OK;
}

extern "C" int rg_set_tuning(rg_ctx *c, int key, int64_t value) {
    if (!c) return RG_ERR_INVALID_ARG;
    if (value < 0) return rg_set_err(c, RG_ERR_INVALID_ARG, "negative tuning value");
    switch (key) {
        case RG_TUNE_TM_SEGMENT: c->tune_tm_segment = (uint32_t)value; return RG_OK;
        case RG_TUNE_TM_TARGET_LANES: c->tune_tm_target_lanes = (uint64_t)value; return RG_OK;
        case RG_TUNE_TM_WINDOWS: c->tune_tm_windows = (uint32_t)(value > 255 ? 255 : value); return RG_OK;
        case RG_TUNE_INGEST_CHUNK_KIB: c->tune_ingest_chunk_kib = (uint64_t)value; return RG_OK;
        case RG_TUNE_GPU_MP3_DECODE: c->gpu_mp3_decode = value > 3 ? 3 : (int)value; return RG_OK;
        case RG_TUNE_LOADER_THREADS: c->loader_threads = (unsigned)(value > 1024 ? 1024 : value); return RG_OK;
        case RG_TUNE_ALBUM_PARTS:
            if (value > 3) return rg_set_err(c, RG_ERR_INVALID_ARG, "tuning key 10 takes 0 (default), 1 (never), 2 (on) or 3 (on, copy-bound chunks only)");
            c->tune_album_parts = (int)value;
            return RG_OK;
        case RG_TUNE_PARTS_MIN_BPU: c->tune_parts_min_bpu = value; return RG_OK;
        case RG_TUNE_STAGE_BYTES:
            if (value != 0 && value < 4096) return rg_set_err(c, RG_ERR_INVALID_ARG, "a staging block holds at least 4096 bytes");
            c->tune_stage_bytes = (size_t)value;
            return RG_OK;
        case RG_TUNE_GROUP_BYTES: c->tune_group_bytes = (size_t)value; return RG_OK;
        case RG_TUNE_PIPELINE_SLOTS: {
            if (sync_all(c) != RG_OK) return RG_ERR_DEVICE;
            c->n_slots = value == 0 ? RG_DEFAULT_SLOTS : (value > RG_MAX_SLOTS ? RG_MAX_SLOTS : (int)value);
            c->cur = 0;
            return RG_OK;
        }
        default: return rg_set_err(c, RG_ERR_INVALID_ARG, "unknown tuning key %d", key);
    }
}

extern "C" int rg_tm_design_info(uint32_t sr, uint32_t L, uint32_t *H10, uint32_t *rounds, uint32_t *rounds_fast,
                                 double *resid, double *T_out, double *gram_last_out) {
    const int ri = rg_rate_index(sr);
    if (ri < 0) return RG_ERR_UNSUPPORTED_RATE;
    RgTmDesign d;
    rg_tm_design(RG_RATE_TABLE[ri], L, &d);
    if (!d.ok) return RG_ERR_INVALID_ARG;
    if (H10) *H10 = d.H10;
    if (rounds) *rounds = d.rounds;
    if (rounds_fast) *rounds_fast = d.rounds_fast;
    if (resid) *resid = d.resid;
    if (T_out) memcpy(T_out, d.T.data(), d.T.size() * sizeof(double));
    if (gram_last_out) memcpy(gram_last_out, d.Gp.data() + (size_t)(L - 1) * RG_TM_GRAM, RG_TM_GRAM * sizeof(double));
    return RG_OK;
}

// diagnostic (host only): the affine side of variant 2's design -- which form the Butterworth stage runs in, its constants,
// the constant output offset of the reference's "+1e-10" terms and the track-start state in the carried coordinates
extern "C" int rg_tm_design_affine(uint32_t sr, uint32_t L, int *servo, double *alpha, double *beta, double *g, double *d_inf,
                                   double *sigma0_out) {
    const int ri = rg_rate_index(sr);
    if (ri < 0) return RG_ERR_UNSUPPORTED_RATE;
    RgTmDesign d;
    rg_tm_design(RG_RATE_TABLE[ri], L, &d);
    if (!d.ok) return RG_ERR_INVALID_ARG;
    if (servo) *servo = d.servo ? 1 : 0;
    if (alpha) *alpha = d.alpha;
    if (beta) *beta = d.beta;
    if (g) *g = d.g;
    if (d_inf) *d_inf = d.dinf;
    if (sigma0_out) memcpy(sigma0_out, d.sigma0, sizeof d.sigma0);
    return RG_OK;
}

extern "C" int rg_timing_enable(rg_ctx *c, int on) {
    if (!c) return RG_ERR_INVALID_ARG;
    c->timing = on != 0;
    return RG_OK;
}

extern "C" int rg_timing_read(rg_ctx *c, double *sum_ms, uint64_t *launches, double *span_ms, int reset) {
    if (!c) return RG_ERR_INVALID_ARG;
    if (rg_bind_device(c) != RG_OK) return RG_ERR_DEVICE;
    int rc = sync_all(c);
    if (rc != RG_OK) return rc;
    rc = drain_timing(c);
    if (rc != RG_OK) return rc;
    if (sum_ms) *sum_ms = c->timing_sum_ms;
    if (launches) *launches = c->timing_count;
    if (span_ms) *span_ms = c->timing_span_ms;
    if (reset) {
        c->timing_sum_ms = 0.0;
        c->timing_count = 0;
        c->timing_span_ms = 0.0;
        c->timing_first = nullptr;
    }
    return RG_OK;
}

namespace {

// copy a host PCM arena to the device staging buffer
int stage_pcm(rg_ctx *c, const void *pcm_base, size_t pcm_bytes, int on_device, const void **d_base) {
    if (on_device) {
        *d_base = pcm_base;
        return RG_OK;
    }
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    rc = sync_all(c);  // the arena may still be read by a previous batch
    if (rc != RG_OK) return rc;
    RG_HIP(c, c->d_arena.reserve(pcm_bytes ? pcm_bytes : 1));
    if (pcm_bytes) RG_HIP(c, hipMemcpy(c->d_arena.p, pcm_base, pcm_bytes, hipMemcpyHostToDevice));
    *d_base = c->d_arena.p;
    return RG_OK;
}

}  // namespace

extern "C" int rg_enqueue_pcm_batch(rg_ctx *c, const rg_track_desc *tracks, size_t n, const void *d_pcm_base,
                                    size_t pcm_bytes, int album) {
    return rg_enqueue_impl(c, tracks, n, d_pcm_base, pcm_bytes, album);
}

extern "C" int rg_device_view_get(rg_ctx *c, rg_device_view *v) {
    if (!c || !v) return RG_ERR_INVALID_ARG;
    v->d_track_hist = c->slot().d_hist.p;
    v->d_track_result = c->slot().d_results.p;
    v->d_album_hist = c->slot().d_album_hist.p;
    v->d_album_peak = c->slot().d_album_peak.p;
    v->n_tracks = c->slot().n_enqueued;
    return RG_OK;
}

extern "C" int rg_collect(rg_ctx *c, rg_track_result *out, uint32_t *hist_out) {
    if (!c) return RG_ERR_INVALID_ARG;
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    const size_t n = c->slot().n_enqueued;
    if (n && out)
        RG_HIP(c, hipMemcpyAsync(c->slot().h_results.p, c->slot().d_results.p, n * sizeof(rg_track_result), hipMemcpyDeviceToHost,
                                 c->slot().stream));
    if (n && hist_out)
        RG_HIP(c, hipMemcpyAsync(hist_out, c->slot().d_hist.p, n * (size_t)RG_HISTOGRAM_SIZE * sizeof(uint32_t),
                                 hipMemcpyDeviceToHost, c->slot().stream));
    RG_HIP(c, hipStreamSynchronize(c->slot().stream));
    if (n && out) memcpy(out, c->slot().h_results.p, n * sizeof(rg_track_result));
    return RG_OK;
}

extern "C" int rg_album_reduce_gathered(rg_ctx *c, const void *d_gathered, uint32_t world) {
    if (!c || !d_gathered || world == 0) return RG_ERR_INVALID_ARG;
    if (!c->slot().album_ready) return rg_set_err(c, RG_ERR_STATE, "rg_album_reduce_gathered without an album enqueue");
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    RG_HIP(c, rg_launch_album_reduce_gathered((const uint32_t *)d_gathered, world, c->slot().d_album_hist.p,
                                              c->slot().d_album_peak.p, c->album_stream()));
    if (c->user_attached) {
        RG_HIP(c, hipEventRecord(c->slot().album_done, c->user_stream));
        c->slot().album_pending = true;
    }
    return RG_OK;
}

extern "C" int rg_album_result_enqueue(rg_ctx *c) {
    if (!c) return RG_ERR_INVALID_ARG;
    if (!c->slot().album_ready) return rg_set_err(c, RG_ERR_STATE, "rg_album_result_enqueue without an album enqueue");
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    RG_HIP(c, rg_launch_album_result(c->slot().d_album_hist.p, c->slot().d_album_peak.p, c->slot().d_album_result.p, c->album_stream()));
    if (c->user_attached) {
        RG_HIP(c, hipEventRecord(c->slot().album_done, c->user_stream));
        c->slot().album_pending = true;
    }
    return RG_OK;
}

extern "C" int rg_album_finish(rg_ctx *c, rg_album_result *album_out, uint32_t *album_hist_out) {
    if (!c) return RG_ERR_INVALID_ARG;
    if (!c->slot().album_ready) return rg_set_err(c, RG_ERR_STATE, "rg_album_finish without an album enqueue");
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    RG_HIP(c, rg_launch_album_result(c->slot().d_album_hist.p, c->slot().d_album_peak.p, c->slot().d_album_result.p, c->album_stream()));
    RG_HIP(c, hipMemcpyAsync(c->slot().h_album_result.p, c->slot().d_album_result.p, sizeof(rg_album_result), hipMemcpyDeviceToHost,
                             c->album_stream()));
    if (album_hist_out)
        RG_HIP(c, hipMemcpyAsync(album_hist_out, c->slot().d_album_hist.p, RG_HISTOGRAM_SIZE * sizeof(uint32_t),
                                 hipMemcpyDeviceToHost, c->album_stream()));
    RG_HIP(c, hipStreamSynchronize(c->album_stream()));
    if (album_out) *album_out = *c->slot().h_album_result.p;
    return RG_OK;
}

// ---- RCCL, resolved at run time so that a host that already loaded RCCL (e.g. through
// torch.distributed) shares its copy -------------------------------------------------------------
namespace {
typedef int (*nccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_group_fn)(void);
const int kNcclUint32 = 3, kNcclFloat64 = 8, kNcclSum = 0, kNcclMax = 2;  // rccl.h enum values

void *g_rccl = nullptr;  // rg_comm_library

void *resolve(const char *name) {
    if (g_rccl) {
        void *p = dlsym(g_rccl, name);
        if (p) return p;
    }
    void *p = dlsym(RTLD_DEFAULT, name);
    if (p) return p;
    static void *h = nullptr;
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    return h ? dlsym(h, name) : nullptr;
}

struct NcclId { char bytes[RG_COMM_ID_BYTES]; };  // ncclUniqueId
typedef int (*nccl_get_id_fn)(NcclId *);
typedef int (*nccl_init_rank_fn)(void **, int, NcclId, int);
typedef int (*nccl_destroy_fn)(void *);
typedef int (*nccl_allgather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
typedef const char *(*nccl_errstr_fn)(int);

const char *nccl_error(int r) {
    nccl_errstr_fn f = (nccl_errstr_fn)resolve("ncclGetErrorString");
    return f ? f(r) : "?";
}
}  // namespace

// ---- a communicator of the library's own: the album exchange then runs on the batch's stream, with no
// cross-stream event per step (what a torch.distributed collective costs twice: into its RCCL stream and back)
extern "C" int rg_comm_library(const char *path) {
    if (!path) return RG_ERR_INVALID_ARG;
    // Only a library called librccl.so[.N] is taken in normal operation (PyTorch's copy, or the system's).  Anything else
    // is a test seam -- tests/standin_rccl lets several ranks share one GPU -- and needs MP3RGAIN_AMD_TEST_SEAMS=1.
    {
        const char *base = strrchr(path, '/');
        base = base ? base + 1 : path;
        const bool rccl = strncmp(base, "librccl.so", 10) == 0 && (base[10] == 0 || base[10] == '.');
        const char *seams = getenv("MP3RGAIN_AMD_TEST_SEAMS");
        if (!rccl && !(seams && seams[0] == '1')) return RG_ERR_REFUSED;  // rg_status: not a librccl.so[.N]
    }
    void *h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return RG_ERR_COLLECTIVE;
    g_rccl = h;
    return RG_OK;
}

extern "C" int rg_comm_unique_id(void *id_out) {
    if (!id_out) return RG_ERR_INVALID_ARG;
    nccl_get_id_fn f = (nccl_get_id_fn)resolve("ncclGetUniqueId");
    if (!f) return RG_ERR_COLLECTIVE;
    NcclId id;
    memset(&id, 0, sizeof id);
    if (f(&id) != 0) return RG_ERR_COLLECTIVE;
    memcpy(id_out, &id, sizeof id);
    return RG_OK;
}

extern "C" int rg_comm_destroy(rg_ctx *c) {
    if (!c) return RG_ERR_INVALID_ARG;
    if (c->comm) {
        (void)sync_all(c);
        nccl_destroy_fn f = (nccl_destroy_fn)resolve("ncclCommDestroy");
        if (f) (void)f(c->comm);
        c->comm = nullptr;
        c->comm_world = 1;
    }
    return RG_OK;
}

// what the context's communicator is: ranks it spans (0 = none: single GPU, rg_album_exchange is a no-op) and the version the
// resolved library reports (ncclGetVersion; 0 when it has no such entry point, as the tests' stand-in)
extern "C" int rg_comm_info(rg_ctx *c, int *world_out, int *version_out) {
    if (!c) return RG_ERR_INVALID_ARG;
    if (world_out) *world_out = c->comm ? c->comm_world : 0;
    if (version_out) {
        *version_out = 0;
        typedef int (*nccl_version_fn)(int *);
        nccl_version_fn f = c->comm ? (nccl_version_fn)resolve("ncclGetVersion") : nullptr;
        int v = 0;
        if (f && f(&v) == 0) *version_out = v;
    }
    return RG_OK;
}

extern "C" int rg_comm_init(rg_ctx *c, const void *id, int world, int rank) {
    if (!c || !id || world < 1 || rank < 0 || rank >= world) return RG_ERR_INVALID_ARG;
    nccl_init_rank_fn f = (nccl_init_rank_fn)resolve("ncclCommInitRank");
    if (!f) return rg_set_err(c, RG_ERR_COLLECTIVE, "RCCL entry points not found (librccl.so)");
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    (void)rg_comm_destroy(c);
    NcclId nid;
    memcpy(&nid, id, sizeof nid);
    void *comm = nullptr;
    const int r = f(&comm, world, nid, rank);
    if (r != 0 || !comm) return rg_set_err(c, RG_ERR_COLLECTIVE, "ncclCommInitRank(%d of %d) failed: %s", rank, world, nccl_error(r));
    c->comm = comm;
    c->comm_world = world;
    return RG_OK;
}

// One communicator per context of this process (ncclCommInitAll: ranks = positions in `ctxs`), for a node that drives
// all its GPUs from one process (rg_node.hip).
int rg_comm_init_all(rg_ctx **ctxs, size_t n) {
    if (!ctxs || n == 0) return RG_ERR_INVALID_ARG;
    typedef int (*nccl_init_all_fn)(void **, int, const int *);
    nccl_init_all_fn f = (nccl_init_all_fn)resolve("ncclCommInitAll");
    if (!f) return rg_set_err(ctxs[0], RG_ERR_COLLECTIVE, "RCCL entry points not found (librccl.so)");
    std::vector<int> devs(n);
    std::vector<void *> comms(n, nullptr);
    for (size_t i = 0; i < n; ++i) {
        devs[i] = ctxs[i]->device;
        (void)rg_comm_destroy(ctxs[i]);
    }
    const int r = f(comms.data(), (int)n, devs.data());
    if (r != 0) return rg_set_err(ctxs[0], RG_ERR_COLLECTIVE, "ncclCommInitAll(%zu devices) failed: %s", n, nccl_error(r));
    for (size_t i = 0; i < n; ++i) {
        ctxs[i]->comm = comms[i];
        ctxs[i]->comm_world = (int)n;
    }
    return RG_OK;
}

int rg_comm_adopt(rg_ctx *c, void *comm, int world) {
    if (!c || !comm || world < 1) return RG_ERR_INVALID_ARG;
    (void)rg_comm_destroy(c);
    c->comm = comm;
    c->comm_world = world;
    return RG_OK;
}

// LoudnessHistogram::accumulate + album_peak.max across ranks (src/replaygain.rs:1056-1059): all-gather of the
// [histogram | peak] packs and the device fold, on the stream of the batch
extern "C" int rg_album_exchange(rg_ctx *c) {
    if (!c) return RG_ERR_INVALID_ARG;
    RgSlot &S = c->slot();
    if (!S.album_ready) return rg_set_err(c, RG_ERR_STATE, "rg_album_exchange without an album enqueue");
    if (!c->comm) return RG_OK;  // single GPU: nothing to exchange
    nccl_allgather_fn ag = (nccl_allgather_fn)resolve("ncclAllGather");
    if (!ag) return rg_set_err(c, RG_ERR_COLLECTIVE, "RCCL entry points not found (librccl.so)");
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    RG_HIP(c, S.d_gather.reserve((size_t)c->comm_world * RG_ALBUM_PACK_WORDS));
    hipStream_t s = c->album_stream();
    const int r = ag(S.d_album_hist.p, S.d_gather.p, RG_ALBUM_PACK_WORDS, kNcclUint32, c->comm, s);
    if (r != 0) return rg_set_err(c, RG_ERR_COLLECTIVE, "ncclAllGather failed: %s", nccl_error(r));
    RG_HIP(c, rg_launch_album_reduce_gathered(S.d_gather.p, (uint32_t)c->comm_world, S.d_album_hist.p, S.d_album_peak.p, s));
    if (c->user_attached) {
        RG_HIP(c, hipEventRecord(S.album_done, c->user_stream));
        S.album_pending = true;
    }
    return RG_OK;
}

extern "C" int rg_album_allreduce(rg_ctx *c, void *comm) {
    if (!c) return RG_ERR_INVALID_ARG;
    if (!c->slot().album_ready) return rg_set_err(c, RG_ERR_STATE, "rg_album_allreduce without an album enqueue");
    if (!comm) return RG_OK;  // single GPU: nothing to exchange
    nccl_allreduce_fn ar = (nccl_allreduce_fn)resolve("ncclAllReduce");
    nccl_group_fn gs = (nccl_group_fn)resolve("ncclGroupStart");
    nccl_group_fn ge = (nccl_group_fn)resolve("ncclGroupEnd");
    if (!ar || !gs || !ge) return rg_set_err(c, RG_ERR_COLLECTIVE, "RCCL entry points not found (librccl.so)");
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    int r = gs();
    if (r == 0) r = ar(c->slot().d_album_hist.p, c->slot().d_album_hist.p, RG_HISTOGRAM_SIZE, kNcclUint32, kNcclSum, comm, c->album_stream());
    if (r == 0) r = ar(c->slot().d_album_peak.p, c->slot().d_album_peak.p, 1, kNcclFloat64, kNcclMax, comm, c->album_stream());
    int r2 = ge();
    if (r != 0 || r2 != 0) return rg_set_err(c, RG_ERR_COLLECTIVE, "ncclAllReduce failed (%d/%d)", r, r2);
    if (c->user_attached) {
        RG_HIP(c, hipEventRecord(c->slot().album_done, c->user_stream));
        c->slot().album_pending = true;
    }
    return RG_OK;
}

// ================================ synchronous API ====================================================
namespace {
// Variant 2 marks tracks with a window it could not resolve reliably (RG_TRACK_FLAG_IMPRECISE).  In auto mode the
// synchronous entry points then repeat the batch with those tracks routed to the order-faithful kernel (exact,
// ~20x slower) and everything else on the fast path again.
bool needs_exact_pass(rg_ctx *c, const rg_track_result *res, size_t n) {
    if (c->kernel_variant != 0 || !res) return false;
    bool any = false;
    c->force_exact.assign(n, 0);
    for (size_t i = 0; i < n; ++i)
        if (res[i].flags & RG_TRACK_FLAG_IMPRECISE) c->force_exact[i] = 1, any = true;
    if (!any) c->force_exact.clear();
    return any;
}
struct OneShot {  // scope of a synchronous entry point's enqueue (rg_ctx.h: one_shot)
    rg_ctx *c;
    bool was;
    explicit OneShot(rg_ctx *ctx) : c(ctx), was(ctx->one_shot) { c->one_shot = true; }
    ~OneShot() { c->one_shot = was; }
};
struct ExactPass {  // scope of the repeat: the per-track routing mask is dropped afterwards
    rg_ctx *c;
    explicit ExactPass(rg_ctx *ctx) : c(ctx) {}
    ~ExactPass() { c->force_exact.clear(); }
};
}  // namespace

namespace {
// ---- streamed host ingest ------------------------------------------------------------------------------------------
// A host arena does not have to fit HBM, and its copy (PCIe: ~55 GB/s against 3.4 TB/s of analysis) is the whole cost
// of the call: the batch is cut at track boundaries into sub-batches of at most `chunk` bytes, two device arenas take
// turns -- sub-batch i+1 is copied on the ingest stream while the kernels of sub-batch i run on a pipeline stream -- and
// results come back per sub-batch, so the device holds two chunks at any time whatever the album's size.  Pageable
// memory goes through the runtime's own pinned staging (hipMemcpyAsync returns when the host buffer has been read);
// pinned memory (hipHostMalloc / hipHostRegister by the caller) is DMA'd in place.
// Bits are those of the one-shot path: tracks are independent, per-track results are written to their input positions,
// and the album histogram is the sum of the sub-batches' histograms (u32 adds commute), the peak their maximum.
constexpr uint64_t kDefaultIngestChunk = 2ull << 30;

int ingest_setup(rg_ctx *c) {
    if (c->ingest_stream) return RG_OK;
    RG_HIP(c, hipStreamCreateWithFlags(&c->ingest_stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        RG_HIP(c, hipEventCreateWithFlags(&c->ingest_copied[k], hipEventDisableTiming));
        RG_HIP(c, hipEventCreateWithFlags(&c->ingest_free[k], hipEventDisableTiming));
    }
    return RG_OK;
}

struct SubBatch {
    size_t first, count;  // tracks [first, first + count)
    size_t bytes;
};

int analyze_host_streamed(rg_ctx *c, const rg_track_desc *tracks, size_t n, const unsigned char *host, int album,
                          rg_track_result *out, uint32_t *hist_out, rg_album_result *album_out, uint32_t *album_hist_out,
                          uint64_t chunk) {
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    rc = ingest_setup(c);
    if (rc != RG_OK) return rc;
    auto track_bytes = [&](size_t t) { return (size_t)tracks[t].channels * tracks[t].frames * rg_bytes_per_sample(tracks[t].format); };
    std::vector<SubBatch> subs;
    for (size_t t = 0; t < n;) {
        SubBatch sb{t, 0, 0};
        while (t < n) {
            const size_t b = (track_bytes(t) + 15) & ~(size_t)15;
            if (sb.count && sb.bytes + b > chunk) break;
            sb.bytes += b;
            ++sb.count;
            ++t;
        }
        subs.push_back(sb);
    }
    size_t max_bytes = 16;
    for (const SubBatch &sb : subs) max_bytes = std::max(max_bytes, sb.bytes);
    rc = sync_all(c);
    if (rc != RG_OK) return rc;
    for (int k = 0; k < 2; ++k) RG_HIP(c, c->d_ingest[k].reserve(max_bytes));
    if (album) RG_HIP(c, c->d_album_packs.reserve(subs.size() * (size_t)RG_ALBUM_PACK_WORDS));
    std::vector<rg_track_result> res(n ? n : 1);
    std::vector<std::vector<rg_track_desc>> descs(subs.size());

    auto copy_sub = [&](size_t i) -> int {  // H2D of sub-batch i into arena i & 1, compacted, on the ingest stream
        const SubBatch &sb = subs[i];
        const int k = (int)(i & 1);
        if (i >= 2) RG_HIP(c, hipStreamWaitEvent(c->ingest_stream, c->ingest_free[k], 0));  // kernels of sub-batch i-2 are done with it
        descs[i].resize(sb.count);
        size_t off = 0;
        for (size_t q = 0; q < sb.count; ++q) {
            const rg_track_desc &d = tracks[sb.first + q];
            const size_t b = track_bytes(sb.first + q);
            if (b) RG_HIP(c, hipMemcpyAsync(c->d_ingest[k].p + off, host + d.offset_bytes, b, hipMemcpyHostToDevice, c->ingest_stream));
            descs[i][q] = d;
            descs[i][q].offset_bytes = off;
            off += (b + 15) & ~(size_t)15;
        }
        RG_HIP(c, hipEventRecord(c->ingest_copied[k], c->ingest_stream));
        return RG_OK;
    };
    auto enqueue_sub = [&](size_t i) -> int {
        const SubBatch &sb = subs[i];
        const int k = (int)(i & 1);
        // every pipeline stream may run this batch: they all wait for the copy (one event wait each, per sub-batch)
        for (int s = 0; s < c->n_slots; ++s) RG_HIP(c, hipStreamWaitEvent(c->slots[s].stream, c->ingest_copied[k], 0));
        return rg_enqueue_impl(c, descs[i].data(), sb.count, c->d_ingest[k].p, sb.bytes, album);
    };
    auto finish_sub = [&](size_t i) -> int {  // results of sub-batch i (the most recent enqueue), exact repeat included
        const SubBatch &sb = subs[i];
        const int k = (int)(i & 1);
        uint32_t *h = hist_out ? hist_out + sb.first * (size_t)RG_HISTOGRAM_SIZE : nullptr;
        int r = rg_collect(c, res.data() + sb.first, h);
        if (r != RG_OK) return r;
        if (needs_exact_pass(c, res.data() + sb.first, sb.count)) {
            ExactPass exact(c);
            r = rg_enqueue_impl(c, descs[i].data(), sb.count, c->d_ingest[k].p, sb.bytes, album);
            if (r != RG_OK) return r;
            r = rg_collect(c, res.data() + sb.first, h);
            if (r != RG_OK) return r;
        }
        RgSlot &S = c->slot();
        if (album)
            RG_HIP(c, hipMemcpyAsync(c->d_album_packs.p + i * (size_t)RG_ALBUM_PACK_WORDS, S.d_album_hist.p,
                                     (size_t)RG_ALBUM_PACK_WORDS * sizeof(uint32_t), hipMemcpyDeviceToDevice, S.stream));
        RG_HIP(c, hipEventRecord(c->ingest_free[k], S.stream));
        return RG_OK;
    };

    // copy(0); then per step: copy(i+1) goes out BEFORE the results of i are waited for, so the copy engine never idles
    rc = copy_sub(0);
    if (rc != RG_OK) return rc;
    for (size_t i = 0; i < subs.size(); ++i) {
        rc = enqueue_sub(i);
        if (rc != RG_OK) return rc;
        if (i + 1 < subs.size()) {
            rc = copy_sub(i + 1);
            if (rc != RG_OK) return rc;
        }
        rc = finish_sub(i);
        if (rc != RG_OK) return rc;
    }
    if (out)
        for (size_t t = 0; t < n; ++t) out[t] = res[t];
    if (album) {
        RgSlot &S = c->slot();
        RG_HIP(c, rg_launch_album_reduce_gathered(c->d_album_packs.p, (uint32_t)subs.size(), S.d_album_hist.p, S.d_album_peak.p, S.stream));
        S.album_ready = true;
        return rg_album_finish(c, album_out, album_hist_out);
    }
    return RG_OK;
}

// host input larger than one ingest chunk takes the streamed route
bool wants_streaming(const rg_ctx *c, size_t n, size_t pcm_bytes, int on_device, uint64_t *chunk) {
    *chunk = c->tune_ingest_chunk_kib ? c->tune_ingest_chunk_kib * 1024ull : kDefaultIngestChunk;
    return !on_device && n > 1 && pcm_bytes > *chunk && !c->user_attached;
}
}  // namespace

extern "C" int rg_analyze_pcm_batch(rg_ctx *c, const rg_track_desc *tracks, size_t n, const void *pcm_base,
                                    size_t pcm_bytes, int on_device, rg_track_result *out, uint32_t *hist_out) {
    if (!c) return RG_ERR_INVALID_ARG;
    if (n && !pcm_base) return rg_set_err(c, RG_ERR_INVALID_ARG, "null pcm_base");
    uint64_t chunk = 0;
    if (wants_streaming(c, n, pcm_bytes, on_device, &chunk)) {
        int v = rg_validate_batch(c, tracks, n, pcm_bytes);
        if (v != RG_OK) return v;
        return analyze_host_streamed(c, tracks, n, (const unsigned char *)pcm_base, 0, out, hist_out, nullptr, nullptr, chunk);
    }
    const void *d_base = nullptr;
    int rc = stage_pcm(c, pcm_base, pcm_bytes, on_device, &d_base);
    if (rc != RG_OK) return rc;
    OneShot one(c);
    rc = rg_enqueue_impl(c, tracks, n, d_base, pcm_bytes, 0);
    if (rc != RG_OK) return rc;
    rc = rg_collect(c, out, hist_out);
    if (rc != RG_OK || !needs_exact_pass(c, out, n)) return rc;
    ExactPass exact(c);
    rc = rg_enqueue_impl(c, tracks, n, d_base, pcm_bytes, 0);
    if (rc != RG_OK) return rc;
    return rg_collect(c, out, hist_out);
}

// rg_collect for callers of the asynchronous pair that want the synchronous calls' guarantee: the batch most recently enqueued
// with exactly these descriptors is collected, and if the fast kernels flagged a track whose bins rounding could have moved, the
// batch is run once more with those tracks on the order-faithful kernel (auto mode only; track mode -- an album's pack has
// already been produced from the first run).  The PCM must still be where it was.
extern "C" int rg_collect_exact(rg_ctx *c, const rg_track_desc *tracks, size_t n, const void *d_pcm_base, size_t pcm_bytes,
                                rg_track_result *out, uint32_t *hist_out) {
    if (!c || (n && (!tracks || !d_pcm_base || !out))) return RG_ERR_INVALID_ARG;
    if (c->slot().n_enqueued != n) return rg_set_err(c, RG_ERR_STATE, "rg_collect_exact: the last enqueue held %zu tracks, not %zu", c->slot().n_enqueued, n);
    if (c->slot().enq_album) return rg_set_err(c, RG_ERR_STATE, "rg_collect_exact: the last enqueue was an album enqueue (track mode only)");
    if (c->slot().enq_base != d_pcm_base || c->slot().enq_bytes != pcm_bytes)
        return rg_set_err(c, RG_ERR_STATE, "rg_collect_exact: not the PCM arena of the last enqueue");
    int rc = rg_collect(c, out, hist_out);
    if (rc != RG_OK || !needs_exact_pass(c, out, n)) return rc;
    ExactPass exact(c);
    OneShot one(c);
    rc = rg_enqueue_impl(c, tracks, n, d_pcm_base, pcm_bytes, 0);
    if (rc != RG_OK) return rc;
    return rg_collect(c, out, hist_out);
}

extern "C" int rg_analyze_album_pcm(rg_ctx *c, const rg_track_desc *tracks, size_t n, const void *pcm_base,
                                    size_t pcm_bytes, int on_device, rg_track_result *tracks_out,
                                    rg_album_result *album_out, uint32_t *album_hist_out) {
    if (!c) return RG_ERR_INVALID_ARG;
    if (n && !pcm_base) return rg_set_err(c, RG_ERR_INVALID_ARG, "null pcm_base");
    uint64_t chunk = 0;
    if (wants_streaming(c, n, pcm_bytes, on_device, &chunk)) {
        int v = rg_validate_batch(c, tracks, n, pcm_bytes);
        if (v != RG_OK) return v;
        return analyze_host_streamed(c, tracks, n, (const unsigned char *)pcm_base, 1, tracks_out, nullptr, album_out, album_hist_out, chunk);
    }
    int rc = rg_album_local_pcm(c, tracks, n, pcm_base, pcm_bytes, on_device, tracks_out);
    if (rc != RG_OK) return rc;
    return rg_album_finish(c, album_out, album_hist_out);
}

// rg_ctx.h: the album of this context's tracks up to, not including, the album percentile: per-track results (exact
// repeat of flagged tracks included) on the host, the album's [histogram | peak] pack ready on the device.  What follows
// is rg_album_finish, or first rg_album_exchange when other GPUs hold more of the album (rg_node.cpp).
int rg_album_local_pcm(rg_ctx *c, const rg_track_desc *tracks, size_t n, const void *pcm_base, size_t pcm_bytes, int on_device,
                       rg_track_result *tracks_out) {
    const void *d_base = nullptr;
    int rc = stage_pcm(c, pcm_base, pcm_bytes, on_device, &d_base);
    if (rc != RG_OK) return rc;
    OneShot one(c);
    rc = rg_enqueue_impl(c, tracks, n, d_base, pcm_bytes, 1);
    if (rc != RG_OK) return rc;
    std::vector<rg_track_result> probe;
    rg_track_result *res = tracks_out;
    if (!res && n) {  // the flags are needed even when the caller does not want the per-track results
        probe.resize(n);
        res = probe.data();
    }
    rc = rg_collect(c, res, nullptr);
    if (rc != RG_OK) return rc;
    if (needs_exact_pass(c, res, n)) {
        ExactPass exact(c);
        rc = rg_enqueue_impl(c, tracks, n, d_base, pcm_bytes, 1);
        if (rc != RG_OK) return rc;
        rc = rg_collect(c, tracks_out, nullptr);
        if (rc != RG_OK) return rc;
    }
    return RG_OK;
}

// rg_ctx.h: an album in parts (the file layer's albums larger than the device); same folding as the streamed host ingest
int rg_album_part(rg_ctx *c, const rg_track_desc *tracks, size_t n, const void *d_base, size_t bytes, size_t index, size_t parts,
                  rg_track_result *out) {
    if (index == 0) {
        int rc = sync_all(c);
        if (rc != RG_OK) return rc;
        RG_HIP(c, c->d_album_packs.reserve(parts * (size_t)RG_ALBUM_PACK_WORDS));
    }
    OneShot one(c);
    int rc = rg_enqueue_impl(c, tracks, n, d_base, bytes, 1);
    if (rc != RG_OK) return rc;
    rc = rg_collect(c, out, nullptr);
    if (rc != RG_OK) return rc;
    if (needs_exact_pass(c, out, n)) {
        ExactPass exact(c);
        rc = rg_enqueue_impl(c, tracks, n, d_base, bytes, 1);
        if (rc != RG_OK) return rc;
        rc = rg_collect(c, out, nullptr);
        if (rc != RG_OK) return rc;
    }
    RgSlot &S = c->slot();
    RG_HIP(c, hipMemcpyAsync(c->d_album_packs.p + index * (size_t)RG_ALBUM_PACK_WORDS, S.d_album_hist.p,
                             (size_t)RG_ALBUM_PACK_WORDS * sizeof(uint32_t), hipMemcpyDeviceToDevice, S.stream));
    RG_HIP(c, hipStreamSynchronize(S.stream));  // the arena is the next part's
    return RG_OK;
}

int rg_album_parts_fold(rg_ctx *c, size_t parts) {
    RgSlot &S = c->slot();
    RG_HIP(c, rg_launch_album_reduce_gathered(c->d_album_packs.p, (uint32_t)parts, S.d_album_hist.p, S.d_album_peak.p, S.stream));
    S.album_ready = true;
    return RG_OK;
}

int rg_album_parts_finish(rg_ctx *c, size_t parts, rg_album_result *album_out) {
    const int rc = rg_album_parts_fold(c, parts);
    if (rc != RG_OK) return rc;
    return rg_album_finish(c, album_out, nullptr);
}

extern "C" int rg_find_peak_pcm(rg_ctx *c, const rg_track_desc *track, const void *pcm_base, size_t pcm_bytes,
                                int on_device, rg_peak_result *out) {
    if (!c || !track || !out) return RG_ERR_INVALID_ARG;
    if (track->format > RG_FMT_S32_PLANAR) return rg_set_err(c, RG_ERR_INVALID_ARG, "unknown format");
    const size_t bps = rg_bytes_per_sample(track->format);
    const uint64_t total = (uint64_t)track->channels * track->frames;
    if (track->offset_bytes + total * bps > pcm_bytes) return rg_set_err(c, RG_ERR_INVALID_ARG, "track extends past arena");
    const void *d_base = nullptr;
    int rc = stage_pcm(c, pcm_base, pcm_bytes, on_device, &d_base);
    if (rc != RG_OK) return rc;
    RG_HIP(c, c->d_peak_bits.reserve(1));
    RG_HIP(c, hipMemsetAsync(c->d_peak_bits.p, 0, sizeof(unsigned long long), (c->user_attached ? c->user_stream : c->slot().stream)));
    RG_HIP(c, rg_launch_peak_all((const unsigned char *)d_base + track->offset_bytes, total, track->format,
                                 c->d_peak_bits.p, (c->user_attached ? c->user_stream : c->slot().stream)));
    unsigned long long bits = 0;
    RG_HIP(c, hipMemcpyAsync(&bits, c->d_peak_bits.p, sizeof bits, hipMemcpyDeviceToHost, (c->user_attached ? c->user_stream : c->slot().stream)));
    RG_HIP(c, hipStreamSynchronize((c->user_attached ? c->user_stream : c->slot().stream)));
    double pk;
    memcpy(&pk, &bits, sizeof pk);
    out->peak = pk;
    out->peak_pcm = pk * 32768.0;  // src/replaygain.rs:1246
    out->sample_rate = track->sample_rate;
    out->reserved = 0;
    return RG_OK;
}

extern "C" int rg_synth_fill_device(rg_ctx *c, void *d_dst, uint64_t seed, uint32_t channel, uint32_t sample_rate,
                                    uint64_t first_frame, uint64_t frames) {
    if (!c || (!d_dst && frames)) return RG_ERR_INVALID_ARG;
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    hipStream_t fs = c->user_attached ? c->user_stream : c->slot().stream;
    RG_HIP(c, rg_launch_synth_fill((float *)d_dst, seed, channel, sample_rate, first_frame, frames, fs));
    // every pipeline stream must see the generated PCM: the next enqueue waits for this point
    if (!c->user_attached) RG_HIP(c, hipEventRecord(c->user_ev, fs));
    c->user_dirty = true;
    return RG_OK;
}
