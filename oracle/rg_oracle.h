/* rg_oracle.h -- CPU oracle for the ReplayGain 1.0 analysis hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call this.  The shipped
 * library (mp3rgain_amd/csrc) never includes or links anything from oracle/.
 *
 * This is a plain-C restatement, operation for operation, of the reference's
 * sequential f64 algorithm in mp3rgain v1.5.0 src/replaygain.rs (citations on
 * each function are file:line in /root/reference).  The reference is Rust and
 * cannot be compiled in the build image (no cargo/rustc), so there is no
 * oracle/_ref build.
 *
 * PINNING STATUS: the reference's own tests for this path pin only (a) the set
 * of supported sample rates (src/replaygain.rs:1275-1294) and (b) two loudness
 * *ranges* for 1 kHz sines (src/replaygain.rs:1296-1365).  The oracle is
 * checked against both in tests/test_oracle.py.  No reference test pins a dB
 * value or a histogram, so exact-value parity of this oracle with the Rust
 * binary is "parity unpinned" beyond those two checks; see DESIGN.md.
 */
#ifndef RG_ORACLE_H
#define RG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/rg_coeffs.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RGO_HISTOGRAM_SIZE 12000 /* src/replaygain.rs:630 */
#define RGO_HISTOGRAM_OFFSET 2000 /* src/replaygain.rs:635 */

/* sample formats the reference's process_audio_buffer accepts (replaygain.rs:959-1028) */
enum { RGO_FMT_F32 = 0, RGO_FMT_S16 = 1, RGO_FMT_S32 = 2 };

/* EqualLoudnessFilter, src/replaygain.rs:534-551 */
typedef struct rgo_filter {
    const rg_rate_coeffs *c;
    double yule_x[11], yule_y[11], butter_x[3], butter_y[3];
} rgo_filter;

/* ReplayGainAnalyzer + LoudnessHistogram, src/replaygain.rs:644-698 */
typedef struct rgo_analyzer {
    double lsum, rsum;
    size_t totsamp, window_samples;
    uint32_t hist[RGO_HISTOGRAM_SIZE];
} rgo_analyzer;

/* per-track state of analyze_track_internal's decode loop, src/replaygain.rs:866-878 */
typedef struct rgo_track_state {
    rgo_filter filt[2];
    rgo_analyzer an;
    double peak;
    unsigned sample_rate;
    unsigned channels; /* 1 or >=2 (only ch0/ch1 are read) */
} rgo_track_state;

/* ReplayGainResult minus file_type, src/replaygain.rs:57-68 */
typedef struct rgo_result {
    double loudness_db, gain_db, peak;
    uint32_t sample_rate;
    int32_t gain_steps;
} rgo_result;

const rg_rate_coeffs *rgo_rate_coeffs(unsigned sample_rate);
int rgo_filter_init(rgo_filter *f, unsigned sample_rate);
double rgo_filter_process(rgo_filter *f, double sample);

void rgo_analyzer_init(rgo_analyzer *a, unsigned sample_rate);
void rgo_analyzer_add_sample(rgo_analyzer *a, double l, double r);
void rgo_analyzer_add_mono_sample(rgo_analyzer *a, double s);
void rgo_analyzer_finish_window(rgo_analyzer *a);

double rgo_hist_loudness(const uint32_t *hist);
void rgo_hist_accumulate(uint32_t *dst, const uint32_t *src);
uint64_t rgo_percentile_threshold(uint64_t total);

int rgo_track_begin(rgo_track_state *s, unsigned sample_rate, unsigned channels);
void rgo_process_buffer(rgo_track_state *s, const void *ch0, const void *ch1, size_t frames, int fmt);
void rgo_track_finish(rgo_track_state *s, rgo_result *out);

/* whole planar buffer at once: ch1 == NULL means mono */
int rgo_analyze_pcm(const void *ch0, const void *ch1, size_t frames, unsigned sample_rate, int fmt,
                    rgo_result *out, uint32_t *hist_out /* 12000 or NULL */);

double rgo_gain_from_loudness(double loudness_db);
int32_t rgo_gain_steps(double gain_db);

/* find_peak_amplitude's sample loop, src/replaygain.rs:1210-1241 (max over ALL channels) */
double rgo_find_peak(const void *const *chans, unsigned nch, size_t frames, int fmt);

/* CLI clip limiting for -k, src/main.rs:2033-2058: returns the (possibly reduced) step count */
int32_t rgo_clip_limit_steps(int32_t steps, double gain_db, double peak, int prevent_clipping, int wrap);

/* reference unit-test signal fed as f64 straight into filter.process (replaygain.rs:1296-1365) */
double rgo_unit_test_sine(unsigned sample_rate, double frequency, double amplitude_normalized,
                          size_t duration_samples, uint32_t *hist_out);

/* deterministic synthetic PCM (include/rg_synth.h), host side */
void rgo_synth_fill_f32(float *dst, uint64_t seed, unsigned channel, unsigned sample_rate,
                        uint64_t first_frame, size_t frames);

#ifdef __cplusplus
}
#endif
#endif
