/* mp3rgain_amd.h -- C ABI of the MI355X (gfx950) ReplayGain 1.0 analysis path.
 *
 * Drop-in boundary for mp3rgain's `replaygain` module (reference v1.5.0,
 * citations are file:line under /root/reference).  The reference has no FFI
 * seam of its own; its per-packet hot loop is
 *     process_audio_buffer(&AudioBufferRef, &mut [EqualLoudnessFilter],
 *                          &mut ReplayGainAnalyzer, &mut f64)   src/replaygain.rs:953-958
 * fed by the decode loop of analyze_track_internal (src/replaygain.rs:866-925)
 * and merged by analyze_album_with_index (src/replaygain.rs:1044-1074).  This
 * library replaces everything from "decoded planar PCM" to "ReplayGainResult /
 * AlbumGainResult": a Rust host binds these symbols with a ~40-line
 * `extern "C"` block and calls them where it used to run the
 * per-sample loop.
 *
 * Plain C types only: pointers, sizes, PODs.  No torch / HIP types appear in a
 * signature; a HIP stream or an RCCL communicator crosses as `void *`.
 *
 * There is NO CPU fallback.  Every compute entry point fails with
 * RG_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef MP3RGAIN_AMD_H
#define MP3RGAIN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RG_ABI_VERSION 5
#define RG_HISTOGRAM_SIZE 12000         /* HISTOGRAM_SIZE        src/replaygain.rs:630 */
#define RG_HISTOGRAM_OFFSET 2000        /* HISTOGRAM_OFFSET      src/replaygain.rs:635 */
#define RG_REPLAYGAIN_REFERENCE_DB 89.0 /* REPLAYGAIN_REFERENCE_DB src/replaygain.rs:37 */
#define RG_PINK_REF 64.82               /* PINK_REF              src/replaygain.rs:44  */
#define RG_GAIN_STEP_DB 1.5             /* GAIN_STEP_DB          src/lib.rs:48         */

/* status codes; the text a Rust caller would put in anyhow::Error comes from rg_last_error() */
typedef enum rg_status {
    RG_OK = 0,
    RG_ERR_INVALID_ARG = -1,
    RG_ERR_UNSUPPORTED_RATE = -2, /* "Unsupported sample rate: {} Hz. Supported rates: ..." src/replaygain.rs:868-873 */
    RG_ERR_DEVICE = -3,           /* a HIP call failed */
    RG_ERR_NO_DEVICE = -4,        /* no usable gfx950 device: there is no CPU path */
    RG_ERR_NOMEM = -5,
    RG_ERR_STATE = -6,            /* e.g. collect with nothing enqueued */
    RG_ERR_COLLECTIVE = -7,       /* RCCL symbol lookup or call failed */
    RG_ERR_IO = -8,               /* "Failed to open: {path}" src/replaygain.rs:804-805 */
    RG_ERR_FORMAT = -9,           /* "Failed to probe format: {path}" src/replaygain.rs:815-822 */
    RG_ERR_REFUSED = -10          /* rg_comm_library / rg_node_create_backend: a test seam (a collective library not called
                                     librccl.so[.N], foreign engines) without MP3RGAIN_AMD_TEST_SEAMS=1 in the environment */
} rg_status;

/* planar sample formats, the three AudioBufferRef arms of src/replaygain.rs:959-1024 */
typedef enum rg_sample_format {
    RG_FMT_F32_PLANAR = 0, /* normalised [-1,1]; scaled by 32768 before filtering (:969) */
    RG_FMT_S16_PLANAR = 1, /* used as-is (:990); peak on x/32768                          */
    RG_FMT_S32_PLANAR = 2  /* x * 32768/2^31 (:1005); peak on that /32768                 */
} rg_sample_format;

/* AudioFileType, src/replaygain.rs:48-53 (carried through, set by the caller's demuxer) */
typedef enum rg_file_type { RG_FILE_MP3 = 0, RG_FILE_AAC = 1 } rg_file_type;

/* One decoded track inside a caller-owned PCM arena.  Channel c of the track starts at
 * pcm_base + offset_bytes + c * frames * bytes_per_sample (planar, the layout
 * `buf.chan(c)[frame]` of src/replaygain.rs:966,972).  Only channels 0 and 1 are read
 * (src/replaygain.rs:971); channels == 1 selects add_mono_sample (:731-740). */
typedef struct rg_track_desc {
    uint64_t offset_bytes;
    uint64_t frames;
    uint32_t sample_rate;
    uint16_t channels;
    uint16_t format; /* rg_sample_format */
} rg_track_desc;

/* ReplayGainResult (src/replaygain.rs:57-68) + gain_steps() (:72-74) */
typedef struct rg_track_result {
    double loudness_db;
    double gain_db;
    double peak;
    uint32_t sample_rate;
    int32_t gain_steps;
    uint32_t windows; /* number of 50 ms windows that landed in the histogram */
    uint32_t file_type;
    uint32_t flags;   /* RG_TRACK_FLAG_* */
    uint32_t reserved;
} rg_track_result;
/* the track held samples that are not finite (NaN / Inf): every window from the first one on is a NaN window */
#define RG_TRACK_FLAG_NONFINITE 1u
/* Variant 2 only: some window's energy is so much smaller than the energies it was assembled from (the high-passed
 * tail of a large DC offset, say) that rounding may have moved it across a bin edge, a few 0.01 dB at most.  The
 * synchronous entry points (rg_analyze_pcm_batch, rg_analyze_album_pcm, rg_analyze_wav_batch and the file-level
 * functions) then repeat the batch, the flagged tracks on the order-faithful kernel, when the variant is 0 (auto) and return exact
 * results with the flag cleared; callers of rg_enqueue_pcm_batch / rg_collect see the flag and decide. */
#define RG_TRACK_FLAG_IMPRECISE 2u

/* AlbumGainResult minus the per-track vector (src/replaygain.rs:79-95) */
typedef struct rg_album_result {
    double album_loudness_db;
    double album_gain_db;
    double album_peak;
    int32_t album_gain_steps;
    uint32_t windows;
} rg_album_result;

/* PeakAmplitudeResult, src/replaygain.rs:1125-1132 */
typedef struct rg_peak_result {
    double peak;
    double peak_pcm;
    uint32_t sample_rate;
    uint32_t reserved;
} rg_peak_result;

/* device-side views for callers that keep results in HBM (pipelines, multi-GPU album) */
typedef struct rg_device_view {
    void *d_track_hist;   /* uint32_t [n_tracks][RG_HISTOGRAM_SIZE]                 */
    void *d_track_result; /* rg_track_result [n_tracks]                             */
    void *d_album_hist;   /* uint32_t [RG_HISTOGRAM_SIZE] (sum over this ctx's tracks) */
    void *d_album_peak;   /* double [1] (max over this ctx's tracks)                */
    uint64_t n_tracks;
} rg_device_view;

typedef struct rg_ctx rg_ctx;

/* ---- pure helpers (host) ---------------------------------------------------------------- */
int rg_abi_version(void);
int rg_is_available(void);                         /* replaygain::is_available  src/replaygain.rs:1119-1121 */
int rg_supported_rate(uint32_t sample_rate);       /* EqualLoudnessFilter::new -> Option  :555-584 */
uint32_t rg_window_samples(uint32_t sample_rate);  /* ReplayGainAnalyzer::new   :702-704 */
double rg_hist_loudness(const uint32_t *hist);     /* LoudnessHistogram::get_loudness :665-682 */
double rg_gain_from_loudness(double loudness_db);  /* PINK_REF - loudness        :911 */
int32_t rg_gain_steps(double gain_db);             /* gain_steps()               :72-74 */
int32_t rg_db_to_steps(double db);                 /* db_to_steps                src/lib.rs:632-634 */
double rg_steps_to_db(int32_t steps);              /* steps_to_db                src/lib.rs:637-639 */
/* -k clip limiting of the CLI, src/main.rs:2033-2058 */
int32_t rg_clip_limit_steps(int32_t steps, double gain_db, double peak, int prevent_clipping, int wrap_gain);

/* diagnostic: what the library derived from one coefficient row (host only, no GPU needed).
 * stable = 0 for the 88.2 kHz row, whose recursion diverges in the reference as written. */
int rg_rate_design_info(uint32_t sample_rate, int *stable, uint32_t *halo_frames, double *decay_ratio);

/* ---- context ------------------------------------------------------------------------------ */
/* One context per GPU (the reference is single-threaded; a ctx is not thread-safe, several
 * ctxs may run concurrently).  device = HIP ordinal.  Returns NULL on failure; the reason is
 * then available from rg_last_error(NULL). */
rg_ctx *rg_create(int device);
void rg_destroy(rg_ctx *ctx);
const char *rg_last_error(const rg_ctx *ctx);
/* Attach (attach = 1) a caller-owned HIP stream, given as void*; NULL then means the HIP default stream,
 * which is what torch.cuda.current_stream().cuda_stream is unless the caller switched streams.
 * attach = 0 detaches (the stream argument is ignored).  The analysis kernels keep
 * running on the context's own pipeline streams; the caller's stream is used for (a) input ordering:
 * the first enqueue after rg_set_stream / rg_wait_user_stream / rg_synth_fill_device waits for
 * everything submitted to it so far, and (b) the album tail: after an album enqueue the caller's
 * stream waits for the batch, and rg_album_allreduce / rg_album_result_enqueue / rg_album_finish run
 * on it, so that a collective the caller issues on that stream (RCCL through torch.distributed, say)
 * sits between them in stream order. */
int rg_set_stream(rg_ctx *ctx, void *hip_stream, int attach);
/* order the next enqueue behind everything submitted to the caller's stream so far (PCM produced there) */
int rg_wait_user_stream(rg_ctx *ctx);
/* The HIP stream (as void*) that the most recent enqueue runs on.  With no caller stream attached the album
 * tail (rg_album_reduce_gathered / rg_album_allreduce / rg_album_result_enqueue) runs on it too, so a
 * collective the caller issues ON THIS STREAM (torch.cuda.ExternalStream(handle), say) sits between the batch
 * and the tail in plain stream order: no cross-stream events per step.  This is the fast multi-GPU form --
 * other batches keep running on the context's other pipeline streams meanwhile. */
void *rg_batch_stream(rg_ctx *ctx);
/* kernel variant: 0 = auto (the transient-moment kernels at every rate whose filter is stable -- all but 88.2 kHz, whose
 * coefficient row diverges in the reference as written and runs on the order-faithful kernel -- with flagged tracks repeated
 * on the order-faithful kernel by the synchronous entry points), 1 = halo-tiled order-faithful kernel everywhere,
 * 2 = transient-moment kernels wherever the rate's filter is stable, no repeat */
int rg_set_kernel(rg_ctx *ctx, int variant);

/* tuning knobs (0 restores the default): key 1 = segment length of variant 2 in frames (must divide
 * the 50 ms window), key 2 = number of segments (lanes) variant 2 aims for when it picks one,
 * key 3 = number of pipeline slots (1..8, default 8): buffer sets that consecutive enqueues rotate through, spread
 * over min(slots, 4) HIP streams so that batches overlap on the GPU; rg_collect / rg_album_finish always refer to
 * the most recent enqueue,
 * key 4 = most windows one lane of variant 2 may run in a row (multi-window segments; 0 = chosen from the batch size, up
 * to 16; 1 = never more than one),
 * key 5 = sub-batch size in KiB of the streamed host ingest (0 = 2 GiB): rg_analyze_pcm_batch / rg_analyze_album_pcm on a
 * HOST arena larger than this are cut at track boundaries into sub-batches; two device arenas of that size take turns,
 * the copy of one sub-batch running under the kernels of the previous, so an arena larger than HBM is fine,
 * key 6 = how MPEG Layer III files of the file-level entry points are decoded (mp3rgain_amd_dec.h; identical PCM, bit for
 * bit, whichever is chosen): 3 (default) = the host only finds the frames and strips headers and side information from
 * the stream; side-information parsing (which frames decode), scalefactors, Huffman, requantisation, joint stereo, IMDCT
 * and the polyphase filterbank run on the GPU, file reads, copies and decode overlap chunk by chunk, and the PCM is
 * written straight into the analysis arena; 2 = the host parses the side information too; 1 = scalefactors + Huffman on
 * the host's cores, the rest on the GPU; 0 = the host decoder.  (An album of 64 three-minute 320 kb/s files:
 * 0.96 s / 0.25 s / 0.04 s / 0.017 s for 0 / 1 / 2 / 3.),
 * key 7 = host threads the file-level entry points load files with (0 = every core this process may use; a node of
 * several contexts gives each its share, mp3rgain_amd_node.h),
 * keys 10-13 = routing of the file-level entry points, per context (their defaults come from the environment as it was when
 * the context was created -- RG_ALBUM_PARTS, RG_PARTS_MIN_BYTES_PER_UNIT, RG_MP3_STAGE_BYTES, RG_TRACKS_GROUP_BYTES,
 * INTEGRATION.md -- and the library never calls getenv after rg_create): key 10 = album parts: 1 =
 * never, 2 = on, 3 = on for copy-bound chunks only (not for chunks the device had to wait for the host's loaders for); key 11 = the copy-bound rule of the parts, compressed bytes per granule-channel from which a chunk
 * becomes a part, PLUS ONE (1 = every chunk); key 12 = bytes of a pinned staging block (>= 4096); key 13 = estimated PCM
 * bytes per group of files of rg_analyze_tracks / rg_analyze_album.
 * (Key 9 of ABI 4 -- windows 2..m in a kernel of their own -- is gone with that kernel: it spilled and was never faster.) */
int rg_set_tuning(rg_ctx *ctx, int key, int64_t value);
/* diagnostic (host only): variant 2's design for one rate and segment length.  T_out: [L][12],
 * gram_last_out: [78]; either may be NULL.  RG_ERR_INVALID_ARG when no design exists. */
int rg_tm_design_info(uint32_t sample_rate, uint32_t L, uint32_t *H10, uint32_t *rounds, uint32_t *rounds_fast,
                      double *decoupling_residual, double *T_out, double *gram_last_out);
/* diagnostic (host only): the affine side of that design.  servo = 1 when the Butterworth stage runs as output minus double
 * integrator with linear lanes (its numerator is exactly g (1, -2, 1)); alpha = 2 + a1, beta = 1 + a1 + a2, g = butter b0;
 * d_inf = the constant every true output carries from the reference's "+1e-10" terms (0 in the classic form, where the
 * lanes inject them); sigma0: [12] track-start state in the coordinates of T.  Any pointer may be NULL. */
int rg_tm_design_affine(uint32_t sample_rate, uint32_t L, int *servo, double *alpha, double *beta, double *g, double *d_inf,
                        double *sigma0_out);

/* ---- synchronous analysis (mirrors analyze_track / analyze_album minus the decoder) -------- */
/* analyze_track_internal from the filters onwards, for n independent tracks (`-r` mode).
 * pcm_on_device: 0 = pcm_base is host memory (copied H2D), 1 = pcm_base is a device pointer.
 * hist_out: NULL or host uint32_t[n][RG_HISTOGRAM_SIZE]. */
int rg_analyze_pcm_batch(rg_ctx *ctx, const rg_track_desc *tracks, size_t n, const void *pcm_base,
                         size_t pcm_bytes, int pcm_on_device, rg_track_result *out, uint32_t *hist_out);
/* analyze_album_with_index (src/replaygain.rs:1044-1074): per-track results in input order
 * plus the merged-histogram album result, on this GPU's tracks only. */
int rg_analyze_album_pcm(rg_ctx *ctx, const rg_track_desc *tracks, size_t n, const void *pcm_base,
                         size_t pcm_bytes, int pcm_on_device, rg_track_result *tracks_out,
                         rg_album_result *album_out, uint32_t *album_hist_out);
/* find_peak_amplitude's scan (src/replaygain.rs:1210-1241): max |x| over ALL channels */
int rg_find_peak_pcm(rg_ctx *ctx, const rg_track_desc *track, const void *pcm_base, size_t pcm_bytes,
                     int pcm_on_device, rg_peak_result *out);

/* ---- asynchronous / device-resident pipeline ------------------------------------------------ */
/* Enqueue the whole analysis for a batch whose PCM is already in HBM; nothing is copied back.
 * album != 0 also produces d_album_hist / d_album_peak for this ctx's tracks. */
int rg_enqueue_pcm_batch(rg_ctx *ctx, const rg_track_desc *tracks, size_t n, const void *d_pcm_base,
                         size_t pcm_bytes, int album);
int rg_device_view_get(rg_ctx *ctx, rg_device_view *view);
/* wait for the stream and copy back; any of the outputs may be NULL */
int rg_collect(rg_ctx *ctx, rg_track_result *tracks_out, uint32_t *hist_out);
/* The same with the synchronous calls' guarantee (track mode): if a collected track carries RG_TRACK_FLAG_IMPRECISE the batch --
 * described again by the caller, its PCM still in place -- is run once more with the flagged tracks on the order-faithful
 * kernel, and the exact results are returned.  tracks / n must be what the last rg_enqueue_pcm_batch was given. */
int rg_collect_exact(rg_ctx *ctx, const rg_track_desc *tracks, size_t n, const void *d_pcm_base, size_t pcm_bytes,
                     rg_track_result *tracks_out, uint32_t *hist_out);
/* album across GPUs (src/replaygain.rs:1056-1066 as a collective): in-place
 * all-reduce(sum) of d_album_hist and all-reduce(max) of d_album_peak over `nccl_comm`
 * (an ncclComm_t; NULL = single GPU, no-op).  The RCCL entry points are resolved from the
 * already-loaded process image first, then from librccl.so. */
int rg_album_allreduce(rg_ctx *ctx, void *nccl_comm);
/* The same exchange over a communicator the context owns, as ONE collective on the stream of the batch
 * (rg_batch_stream): all-gather of every rank's [histogram | peak] pack + device fold.  No cross-stream event
 * per step, which is what makes it the fast form (a torch.distributed collective hops into torch's RCCL stream
 * and back).  Bootstrap: rank 0 calls rg_comm_unique_id, the 128 bytes travel by whatever the host has
 * (torch.distributed broadcast in bench.py), every rank calls rg_comm_init (ncclCommInitRank).
 * rg_comm_library names the librccl.so to resolve from first (e.g. the one PyTorch already loaded); a file that is not
 * called librccl.so[.N] is refused unless MP3RGAIN_AMD_TEST_SEAMS=1 (the tests' stand-in transport: unsupported in production).
 * Without a communicator rg_album_exchange is a no-op (single GPU). */
#define RG_COMM_ID_BYTES 128
int rg_comm_library(const char *librccl_path);
int rg_comm_unique_id(void *id_out /* RG_COMM_ID_BYTES */);
int rg_comm_init(rg_ctx *ctx, const void *id /* RG_COMM_ID_BYTES */, int world, int rank);
int rg_comm_destroy(rg_ctx *ctx);
/* ranks the context's communicator spans (0 = none) and the collective library's version (ncclGetVersion, 0 = unknown):
 * what a multi-GPU benchmark line states about its exchange */
int rg_comm_info(rg_ctx *ctx, int *world_out, int *version_out);
int rg_album_exchange(rg_ctx *ctx);
/* The same exchange as ONE collective: d_album_hist and d_album_peak are contiguous (12000 u32 + one f64 =
 * 12002 words, RG_ALBUM_PACK_WORDS).  All-gather every rank's pack (ncclAllGather / all_gather_into_tensor),
 * then this call folds the `world` gathered packs (sum of bins, max of peaks) into this context's album
 * histogram / peak, on the same stream as the other album-tail calls. */
#define RG_ALBUM_PACK_WORDS (RG_HISTOGRAM_SIZE + 2)
int rg_album_reduce_gathered(rg_ctx *ctx, const void *d_gathered, uint32_t world);
/* percentile scan of d_album_hist on the device, result to host */
int rg_album_finish(rg_ctx *ctx, rg_album_result *album_out, uint32_t *album_hist_out);
/* the same scan, enqueued only: the rg_album_result stays in HBM until rg_album_finish */
int rg_album_result_enqueue(rg_ctx *ctx);

/* ---- measurement hooks ------------------------------------------------------------------------ */
/* When enabled, every enqueue brackets the dominant kernel (IIR+RMS+histogram) with HIP events
 * on the stream it is launched on. */
int rg_timing_enable(rg_ctx *ctx, int on);
/* Since the last reset: sum of the bracketed kernel durations, their count, and the span from the
 * first bracketed start to the last bracketed end (launches of consecutive batches overlap across
 * pipeline slots, so span < sum when the pipeline is deeper than one).  Synchronises. */
int rg_timing_read(rg_ctx *ctx, double *sum_ms, uint64_t *launches, double *span_ms, int reset);

/* ---- synthetic PCM directly in HBM (bench / tests; include/rg_synth.h) --------------------- */
int rg_synth_fill_device(rg_ctx *ctx, void *d_dst_f32, uint64_t seed, uint32_t channel,
                         uint32_t sample_rate, uint64_t first_frame, uint64_t frames);

/* ---- file level (last row): the reference's public functions on files -------------------
 * The reference decodes MP3/AAC with a third-party decoder (symphonia) that is outside this library; here a
 * file is a RIFF/WAVE file (integer PCM 8/16/24/32 bit, IEEE float 32 bit), or anything else run through an
 * external decoder command that writes a WAV stream to stdout.  The interleaved samples are copied to HBM and
 * de-interleaved into the planar arena by a device kernel; the rest is rg_analyze_pcm_batch's path. */
typedef struct rg_wav_info {
    uint32_t sample_rate;
    uint16_t channels;
    uint16_t bits_per_sample;
    uint16_t sample_format; /* 1 = integer PCM, 3 = IEEE float (WAVE_FORMAT_EXTENSIBLE resolved to its sub-format) */
    uint16_t block_align;
    uint32_t reserved;
    uint64_t data_offset;   /* first sample byte */
    uint64_t frames;        /* a data size of 0 / 0xFFFFFFFF (streamed WAV) means "to the end of the buffer" */
} rg_wav_info;
/* pure host helper: RG_OK, or RG_ERR_INVALID_ARG when `data` is not a usable RIFF/WAVE stream */
int rg_wav_parse(const void *data, size_t len, rg_wav_info *out);
/* command template run through /bin/sh for files that are not RIFF/WAVE; "{}" is replaced by the shell-quoted
 * path (appended when absent); it must write a WAV stream to stdout.  NULL or "" = none. */
int rg_set_decoder_command(rg_ctx *ctx, const char *command_template);
/* n WAV streams in host memory -> per-track results (+ album result when album != 0) */
int rg_analyze_wav_batch(rg_ctx *ctx, const void *const *wav, const size_t *wav_len, size_t n, int album,
                         rg_track_result *out, rg_album_result *album_out);
/* analyze_track_with_index (src/replaygain.rs:935-941); track_index < 0 = None */
int rg_analyze_track(rg_ctx *ctx, const char *path, int32_t track_index, rg_track_result *out);
/* `-r` over many files (src/main.rs:1937-2001 runs analyze_track on one file after the other): the files are loaded on
 * all host cores and decoded + analysed as one GPU batch.  status_out[i] = RG_OK or file i's error code (its text:
 * rg_tracks_error(ctx, i)); a failing file does not stop the others.  Results are those of rg_analyze_track per file.
 * A long list is taken in groups whose PCM fits a third of the free device memory (at most 64 GB): a whole library can be
 * passed in one call. */
int rg_analyze_tracks(rg_ctx *ctx, const char *const *paths, size_t n, int32_t track_index, rg_track_result *out,
                      int32_t *status_out);
const char *rg_tracks_error(const rg_ctx *ctx, size_t i);
/* analyze_album_with_index (src/replaygain.rs:1044-1074): results in input order; the first failing file aborts.  An album
 * whose PCM does not fit the device at once is analysed in parts whose histograms and peaks are folded: same result. */
int rg_analyze_album(rg_ctx *ctx, const char *const *paths, size_t n, int32_t track_index,
                     rg_track_result *tracks_out, rg_album_result *album_out);
/* The same up to, not including, the album percentile: per-file results are out, the album's [histogram | peak] pack of
 * THESE files is ready on the device.  rg_album_finish completes it; when other GPUs hold the rest of the album,
 * rg_album_exchange (or a host fold of the packs) comes first -- mp3rgain_amd_node.h does exactly that over all GPUs of a
 * node.  *failed_index (may be NULL): which file failed, (size_t)-1 when the failure is not a file's. */
int rg_analyze_album_begin(rg_ctx *ctx, const char *const *paths, size_t n, int32_t track_index,
                           rg_track_result *tracks_out, size_t *failed_index);
/* find_peak_amplitude (src/replaygain.rs:1140-1249) */
int rg_find_peak_amplitude(rg_ctx *ctx, const char *path, rg_peak_result *out);
/* One MPEG Layer III stream through the split decoder (stage A on the host, stages B-E on the device), PCM copied back
 * to the host: same outputs as rg_mp3_decode_f32 of mp3rgain_amd_dec.h, bit for bit.  `info` is an rg_mp3_stream_info. */
int rg_mp3_decode_device(rg_ctx *ctx, const void *data, size_t len, float *ch0, float *ch1, uint64_t capacity, void *info);

/* Measurement hook: the device decode chain alone on `copies` copies of one stream (one chunk of the default route), each of
 * its three stages bracketed with HIP events on their own stream: ms_out[0..2] = frame parser, Huffman, back half (average of
 * `reps` repetitions), ms_out[3] = the chain, ms_out[4] = per chunk in the file route's own arrangement (the frame parser on the
 * copy stream, beside the kernels of the chunk before); ms_out holds five values.  units = granule-channels, frames = PCM frames per channel decoded
 * per repetition, compressed_bytes = main data + slots the chain reads. */
int rg_mp3_decode_bench(rg_ctx *ctx, const void *data, size_t len, uint32_t copies, uint32_t reps, double *ms_out,
                        uint64_t *units_out, uint64_t *compressed_bytes_out, uint64_t *frames_out);

#ifdef __cplusplus
}
#endif
#endif /* MP3RGAIN_AMD_H */
