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

enum { RG_TUNE_TM_SEGMENT = 1, RG_TUNE_TM_TARGET_LANES = 2, RG_TUNE_PIPELINE_SLOTS = 3, RG_TUNE_TM_WINDOWS = 4, RG_TUNE_INGEST_CHUNK_KIB = 5,
       RG_TUNE_GPU_MP3_DECODE = 6, RG_TUNE_LOADER_THREADS = 7, RG_TUNE_ALBUM_PARTS = 10, RG_TUNE_PARTS_MIN_BPU = 11,
       RG_TUNE_STAGE_BYTES = 12, RG_TUNE_GROUP_BYTES = 13 };

#define RG_MAX_SLOTS 8
#define RG_SLOT_STREAMS 4   // HIP streams the slots are spread over (the runtime has 4 hardware queues by default)
#define RG_DEFAULT_SLOTS 8

// Everything one enqueued batch owns.  A context rotates through several slots, each with its own
// HIP stream, so that consecutive batches overlap on the GPU: the latency-bound fix-up / percentile
// kernels of batch i run under the main kernel of batch i+1, and two main kernels of small batches
// share the chip (one 10-minute track alone leaves it at under two waves per SIMD).
struct RgSlot {
    hipStream_t stream = nullptr;
    hipEvent_t staging_done = nullptr;  // H2D copy out of the pinned staging buffer has finished
    hipEvent_t batch_done = nullptr;    // everything enqueued for the batch has finished
    hipEvent_t album_done = nullptr;    // the album tail on the caller's stream is done with this slot's buffers
    bool album_pending = false;
    bool staging_pending = false;
    // all launch descriptors of a batch travel as one blob: [RgTrackDev x n | RgTrackDev x n_k1 | RgTmTrack x m];
    // an unchanged blob (the same batch enqueued again) is not copied again
    DevBuf<unsigned char> d_desc;
    PinnedBuf<unsigned char> h_desc;
    std::vector<unsigned char> desc_shadow;  // what d_desc currently holds
    DevBuf<double> d_tm_rec;                 // segment records (rg_tm.h)
    DevBuf<double> d_tm_win;                 // per-window energies of multi-window segments: [channels][windows of the group]
    DevBuf<uint32_t> d_hist;                 // [hist n*12000 | peak n*2 | done n] words
    DevBuf<unsigned long long> d_k1_bad;     // variant 1: first non-finite frame per track (~0 = none), set per batch
    DevBuf<uint32_t> d_imprecise;            // per track: set by fix-up blocks that saw a cancelled window, cleared by the finisher
    DevBuf<uint32_t> d_nonfinite;            // per track: 0, or 0xFFFFFFFF - (first segment whose output is not finite);
                                             // zero between batches (the fix-up kernel's finisher resets what the main kernel set)
    unsigned long long *peak_ptr = nullptr;
    DevBuf<rg_track_result> d_results;
    PinnedBuf<rg_track_result> h_results;
    DevBuf<uint32_t> d_album_hist;
    DevBuf<double> d_album_peak;
    DevBuf<rg_album_result> d_album_result;
    DevBuf<uint32_t> d_gather;               // rg_album_exchange: every rank's [histogram | peak] pack
    PinnedBuf<rg_album_result> h_album_result;
    size_t n_enqueued = 0;
    int enq_album = 0;                        // what the slot's last enqueue was given (rg_collect_exact repeats it)
    const void *enq_base = nullptr;
    size_t enq_bytes = 0;
    bool album_ready = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;  // timing of the dominant kernel
    size_t ev_used = 0;
};

struct rg_ctx {
    int device = -1;
    std::string err;
    int kernel_variant = 0;      // 0 auto (= 2), 1 halo/reference-order kernel, 2 transient-moment kernels
    uint32_t tune_tm_segment = 0;          // 0 = choose from the workload
    uint32_t tune_tm_windows = 0;          // windows per segment: 0 = choose from the workload, 1 = never more than one
    uint64_t tune_tm_target_lanes = 0;     // 0 = cost model
    int n_slots = RG_DEFAULT_SLOTS;

    RgSlot slots[RG_MAX_SLOTS];
    int cur = 0;                        // slot of the most recent enqueue
    hipStream_t user_stream = nullptr;  // rg_set_stream: the album tail (collectives, album percentile) runs on it
    bool user_attached = false;         // user_stream may legitimately be the HIP default stream (nullptr)
    hipEvent_t user_ev = nullptr;
    bool user_dirty = false;            // the next enqueue must first wait for what was submitted to user_stream
    RgSlot &slot() { return slots[cur]; }
    // stream on which the album tail (all-reduce, album percentile, its D2H) runs
    hipStream_t album_stream() { return user_attached ? user_stream : slots[cur].stream; }

    RgRateDesign design[RG_NUM_RATES];
    DevBuf<RgCoefDev> d_coefs;
    std::map<uint32_t, RgTmDeviceTables *> tm_tables;  // key = rate_idx << 24 | m << 16 | L, shared by all slots (read only)

    std::map<uint64_t, std::pair<uint32_t, uint32_t>> tm_choice;  // memo of choose_tm_tables: batch-shape signature -> (L, m)

    // host scratch for building one batch's descriptors
    std::vector<RgTrackDev> h_tracks, h_k1_tracks;
    std::vector<RgTmTrack> h_tm_tracks;

    DevBuf<unsigned long long> d_peak_bits;  // rg_find_peak_pcm
    // split MP3 decode (rg_mp3dev.hip): constants, and one chunk's spectra / units / IMDCT halves / track descriptors
    DevBuf<unsigned char> d_mp3_tab;
    DevBuf<int16_t> d_mp3_is;
    DevBuf<unsigned char> d_mp3_units;
    DevBuf<unsigned char> d_mp3_tracks;
    DevBuf<unsigned char> d_mp3_huff;        // Huffman look-up tables (device Huffman stage)
    DevBuf<unsigned char> d_mp3_recs;
    DevBuf<unsigned char> d_mp3_main;
    bool mp3_tab_ready = false;
    // tuning key 6 = 3 (rg_mp3dev_enqueue_chunk): two device copies of staging blocks, the per-file results
    DevBuf<unsigned char> d_mp3_stage[2];
    hipStream_t mp3_copy_stream = nullptr;   // H2D of staging blocks, beside the kernels of the chunk before: the second pipeline stream (not owned)
    hipEvent_t mp3_set_free[2] = {nullptr, nullptr};
    bool mp3_set_used[2] = {false, false};
    DevBuf<uint32_t> d_mp3_results;
    DevBuf<uint32_t> d_mp3_tiles;            // frame parser: per tile, granule-channels found and their prefix
    // The frame parser of chunk k + 1 runs on the copy stream, behind its staging block's H2D and beside the Huffman / back-half
    // kernels of chunk k: what it writes (records, tile sums) is per staging set.
    DevBuf<unsigned char> d_mp3_recs_set[2];
    DevBuf<uint32_t> d_mp3_tiles_set[2];
    // the Huffman stage's lane sort (rg_mp3_sort_*): per staging set the units ordered by big_values and the sort's working words
    // (RG_MP3_SORT_WORDS, allocated once and zeroed: the kernels leave the histogram zero behind them)
    DevBuf<uint32_t> d_mp3_perm_set[2];
    DevBuf<uint32_t> d_mp3_sortw_set[2];
    PinnedBuf<uint32_t> h_mp3_results;
    // An album of MPEG streams is analysed chunk by chunk while later chunks are still being copied and decoded (rg_files.hip:
    // album parts): the batch enqueued next waits for `enqueue_wait_ev` (the chunk's decode) on its own stream; per chunk a copy
    // of the frame parser's counts, and every part's per-track results, land in pinned memory without a host synchronise
    hipEvent_t enqueue_wait_ev = nullptr;
    hipStream_t enqueue_stream = nullptr;    // the next enqueue's launches go here instead of its slot's stream (the slot's BUFFERS are used)
    PinnedBuf<uint32_t> h_mp3_part_counts;
    PinnedBuf<rg_track_result> h_part_results;
    hipEvent_t *mp3_bench_ev = nullptr;      // rg_mp3_decode_bench: four events recorded around the three decode stages of a chunk
    void *mp3_pipe = nullptr;                // rg_files.hip: pinned staging blocks of the loader pipeline
    void (*mp3_pipe_free)(void *) = nullptr;
    // host buffers of the file layer (rg_files.hip), kept between calls: freeing and re-mapping hundreds of MB that
    // were the source of H2D copies cost more than decoding them (munmap of such pages: 0.4 ms per MB)
    void *file_pool = nullptr;
    std::vector<std::string> file_errors;    // rg_analyze_tracks: message per file of the last call
    void (*file_pool_free)(void *) = nullptr;
    int gpu_mp3_decode = 3;                  // tuning key 6: 0 = host decoder, 1 = stages B-E of MP3 decoding run on the device,
                                             // 2 = scalefactors + Huffman too, 3 (default) = side-information parsing too: the host
                                             // only finds the frames and strips their headers (loader pipeline, rg_files.hip)
    int32_t file_track_index = -1;           // Some(idx) of the file-level call in progress (src/replaygain.rs:838-851); -1 = None
    unsigned loader_threads = 0;             // tuning key 7: host threads of the file loaders; 0 = every core this process may use
    // Routing knobs of the file layer.  The environment is read ONCE, at rg_create (rg_capi.hip: read_env_defaults; getenv is
    // not safe against a host application's setenv, and a value that does not parse is ignored with a message in
    // rg_last_error's place); tuning keys 10-13 override per context, 0 = the default read then.
    bool env_parts_on = true;                // RG_ALBUM_PARTS != "0"
    double env_parts_min_bpu = 120.0;        // RG_PARTS_MIN_BYTES_PER_UNIT: compressed bytes per unit from which a chunk is copy-bound
    size_t env_stage_bytes = 0;              // RG_MP3_STAGE_BYTES (>= 4096), 0 = the album's size, at most 128 MB
    size_t env_group_bytes = 0;              // RG_TRACKS_GROUP_BYTES, 0 = a third of the free device memory
    double hw_queue_serial = 0.0;            // rg_create's finding: time of a spinning kernel on every pipeline stream / on one (1 = own queues, 4 = one queue)
    bool trace_files = false;                // RG_TRACE_FILES
    int trace_tm = 0;                        // RG_TRACE_TM: 1 = the segment chooser's decision, 2 = every candidate
    int tune_album_parts = 0;                // key 10: 0 = default, 1 = never, 2 = whenever a rule allows, 3 = the copy-bound rule only
    int64_t tune_parts_min_bpu = 0;          // key 11: threshold + 1 (1 = every chunk is a part), 0 = default
    size_t tune_stage_bytes = 0;             // key 12
    size_t tune_group_bytes = 0;             // key 13
    bool parts_on() const { return tune_album_parts ? tune_album_parts >= 2 : env_parts_on; }
    bool parts_when_starved() const { return tune_album_parts != 3; }  // a chunk the device had to wait for is a part as well
    double parts_min_bpu() const { return tune_parts_min_bpu ? (double)(tune_parts_min_bpu - 1) : env_parts_min_bpu; }
    size_t stage_bytes() const { return tune_stage_bytes ? tune_stage_bytes : env_stage_bytes; }
    size_t group_bytes() const { return tune_group_bytes ? tune_group_bytes : env_group_bytes; }
    DevBuf<unsigned char> d_arena;           // staging for host PCM (synchronous API)
    DevBuf<unsigned char> d_ingest[2];       // streamed host ingest: two sub-batch arenas, one filling while the other is analysed
    DevBuf<uint32_t> d_album_packs;          // streamed album: one [histogram | peak] pack per sub-batch, folded at the end
    hipStream_t ingest_stream = nullptr;     // H2D copies of the streamed ingest
    hipEvent_t ingest_copied[2] = {nullptr, nullptr}, ingest_free[2] = {nullptr, nullptr};
    uint64_t tune_ingest_chunk_kib = 0;      // 0 = default (2 GiB)
    DevBuf<unsigned char> d_wav;             // interleaved WAV samples awaiting de-interleave (rg_files.hip)
    std::string decoder_cmd;                 // rg_set_decoder_command
    std::vector<unsigned char> force_exact;  // per track of the next enqueue: 1 = use variant 1 (exact repeat of flagged tracks)
    bool one_shot = false;  // the enqueue is a synchronous entry point's: ONE batch in flight, not one per pipeline stream (cost model)
    void *comm = nullptr;                    // ncclComm_t of rg_comm_init (owned)
    int comm_world = 1;

    bool timing = false;
    double timing_sum_ms = 0.0;
    uint64_t timing_count = 0;
    hipEvent_t timing_first = nullptr;  // start event of the first bracketed launch since the last reset
    double timing_span_ms = 0.0;        // first start -> last end over all bracketed launches
};

int rg_set_err(rg_ctx *c, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
int rg_rate_index(uint32_t sr);
static inline size_t rg_bytes_per_sample(uint32_t fmt) { return fmt == RG_FMT_S16_PLANAR ? 2 : 4; }
int rg_bind_device(rg_ctx *c);
// the whole analysis of one batch, enqueued on c->stream (rg_enqueue.hip)
int rg_enqueue_impl(rg_ctx *c, const rg_track_desc *tracks, size_t n, const void *d_pcm_base, size_t pcm_bytes,
                    int album);
void rg_tm_tables_release(rg_ctx *c);
// An album whose PCM comes to the device part by part (rg_files.hip: more files than fit): rg_album_part analyses part
// `index` of `parts` from a device arena (per-track results in `out`, exact repeat included) and keeps its
// [histogram | peak] pack; rg_album_parts_finish folds the packs and produces the album result.
int rg_album_part(rg_ctx *c, const rg_track_desc *tracks, size_t n, const void *d_base, size_t bytes, size_t index, size_t parts,
                  rg_track_result *out);
int rg_album_parts_finish(rg_ctx *c, size_t parts, rg_album_result *album_out);
int rg_album_parts_fold(rg_ctx *c, size_t parts);  // the fold alone: the album pack is ready on the device, no percentile yet
// rg_analyze_album_pcm up to, not including, the album percentile (rg_capi.hip)
int rg_album_local_pcm(rg_ctx *c, const rg_track_desc *tracks, size_t n, const void *pcm_base, size_t pcm_bytes, int on_device,
                       rg_track_result *tracks_out);
// adopt a communicator made elsewhere (ncclCommInitAll in rg_node.hip); the context owns it from here on
int rg_comm_adopt(rg_ctx *c, void *comm, int world);
int rg_comm_init_all(rg_ctx **ctxs, size_t n);
unsigned rg_usable_cores();  // rg_files.hip: the affinity mask cut by the cgroup CPU quota
int rg_validate_batch(rg_ctx *c, const rg_track_desc *tracks, size_t n, size_t pcm_bytes);  // argument checks of an enqueue

#define RG_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return rg_set_err((ctx), RG_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e__)); \
    } while (0)
