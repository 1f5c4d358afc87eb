/* rg_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see rg_oracle.h header comment).
 *
 * Sequential f64 restatement of mp3rgain v1.5.0 src/replaygain.rs.  Build with
 * -ffp-contract=off so that no multiply-add is fused: the reference (rustc) never
 * contracts a*b+c, and the evaluation order below is the reference's.
 */
#include "rg_oracle.h"

#include <math.h>
#include <string.h>

#include "../include/rg_synth.h"

#define PINK_REF 64.82                 /* src/replaygain.rs:44 */
#define DENORMAL_PREVENTION 1e-10      /* src/replaygain.rs:530 */
#define STEPS_PER_DB 100.0             /* src/replaygain.rs:624 */
#define RMS_PERCENTILE 0.95            /* src/replaygain.rs:638 */
#define SAMPLE_SCALE_16BIT 32768.0     /* src/replaygain.rs:949 */
#define GAIN_STEP_DB 1.5               /* src/lib.rs:48 */

/* EqualLoudnessFilter::new's rate match, src/replaygain.rs:558-572 */
const rg_rate_coeffs *rgo_rate_coeffs(unsigned sample_rate) {
    for (int i = 0; i < RG_NUM_RATES; ++i)
        if (RG_RATE_TABLE[i].sample_rate == sample_rate) return &RG_RATE_TABLE[i];
    return NULL;
}

/* src/replaygain.rs:555-584: zero-initialised histories; unsupported rate -> None */
int rgo_filter_init(rgo_filter *f, unsigned sample_rate) {
    memset(f, 0, sizeof *f);
    f->c = rgo_rate_coeffs(sample_rate);
    return f->c ? 0 : -1;
}

/* EqualLoudnessFilter::process, src/replaygain.rs:586-616.
 * copy_within(0..10, 1) shifts the history towards higher indices; the sum is
 * ((1e-10 + b0*x0) + S) with S folded from 0.0 over i = 1..10 of (b_i*x_i - a_i*y_i). */
double rgo_filter_process(rgo_filter *f, double sample) {
    const rg_rate_coeffs *c = f->c;
    memmove(&f->yule_x[1], &f->yule_x[0], 10 * sizeof(double));
    memmove(&f->yule_y[1], &f->yule_y[0], 10 * sizeof(double));
    f->yule_x[0] = sample;

    double acc = 0.0;
    for (int i = 1; i < 11; ++i) {
        double t = c->yule_b[i] * f->yule_x[i] - c->yule_a[i] * f->yule_y[i];
        acc = acc + t;
    }
    double yule_out = (DENORMAL_PREVENTION + c->yule_b[0] * f->yule_x[0]) + acc;
    f->yule_y[0] = yule_out;

    memmove(&f->butter_x[1], &f->butter_x[0], 2 * sizeof(double));
    memmove(&f->butter_y[1], &f->butter_y[0], 2 * sizeof(double));
    f->butter_x[0] = yule_out;

    double acc2 = 0.0;
    for (int i = 1; i < 3; ++i) {
        double t = c->butter_b[i] * f->butter_x[i] - c->butter_a[i] * f->butter_y[i];
        acc2 = acc2 + t;
    }
    double butter_out = (DENORMAL_PREVENTION + c->butter_b[0] * f->butter_x[0]) + acc2;
    f->butter_y[0] = butter_out;
    return butter_out;
}

/* ReplayGainAnalyzer::new, src/replaygain.rs:702-712 */
void rgo_analyzer_init(rgo_analyzer *a, unsigned sample_rate) {
    memset(a, 0, sizeof *a);
    a->window_samples = ((size_t)sample_rate * 50) / 1000;
}

/* Rust `f64 as i32`: truncate toward zero, saturate, NaN -> 0 */
static int32_t rust_f64_as_i32(double v) {
    if (v != v) return 0;
    if (v >= 2147483647.0) return INT32_MAX;
    if (v <= -2147483648.0) return INT32_MIN;
    return (int32_t)v;
}

/* finish_window, src/replaygain.rs:743-765 */
void rgo_analyzer_finish_window(rgo_analyzer *a) {
    if (a->totsamp == 0) return;
    double mean_square = (a->lsum + a->rsum) / (double)a->totsamp * 0.5;
    double val = STEPS_PER_DB * 10.0 * log10(mean_square + 1e-37);
    /* (val as i32 + HISTOGRAM_OFFSET) as usize: i32 add wraps in a release build,
     * then sign-extends to usize, so any negative sum fails idx < HISTOGRAM_SIZE */
    int32_t iv = rust_f64_as_i32(val);
    int32_t sum = (int32_t)((uint32_t)iv + (uint32_t)RGO_HISTOGRAM_OFFSET);
    if (sum >= 0 && sum < RGO_HISTOGRAM_SIZE) a->hist[sum] += 1;
    a->lsum = 0.0;
    a->rsum = 0.0;
    a->totsamp = 0;
}

/* add_sample, src/replaygain.rs:720-728 */
void rgo_analyzer_add_sample(rgo_analyzer *a, double l, double r) {
    a->lsum += l * l;
    a->rsum += r * r;
    a->totsamp += 1;
    if (a->totsamp >= a->window_samples) rgo_analyzer_finish_window(a);
}

/* add_mono_sample, src/replaygain.rs:731-740 */
void rgo_analyzer_add_mono_sample(rgo_analyzer *a, double s) {
    double sq = s * s;
    a->lsum += sq;
    a->rsum += sq;
    a->totsamp += 1;
    if (a->totsamp >= a->window_samples) rgo_analyzer_finish_window(a);
}

/* threshold of get_loudness, src/replaygain.rs:671 */
uint64_t rgo_percentile_threshold(uint64_t total) {
    return (uint64_t)ceil((double)total * (1.0 - RMS_PERCENTILE));
}

/* LoudnessHistogram::get_loudness, src/replaygain.rs:665-682 */
double rgo_hist_loudness(const uint32_t *hist) {
    uint64_t total = 0;
    for (int i = 0; i < RGO_HISTOGRAM_SIZE; ++i) total += hist[i];
    if (total == 0) return -20.0;
    uint64_t threshold = rgo_percentile_threshold(total);
    uint64_t count = 0;
    for (int i = RGO_HISTOGRAM_SIZE - 1; i >= 0; --i) {
        count += hist[i];
        if (count >= threshold) return (double)(i - RGO_HISTOGRAM_OFFSET) / STEPS_PER_DB;
    }
    return -20.0;
}

/* LoudnessHistogram::accumulate, src/replaygain.rs:658-662 */
void rgo_hist_accumulate(uint32_t *dst, const uint32_t *src) {
    for (int i = 0; i < RGO_HISTOGRAM_SIZE; ++i) dst[i] += src[i];
}

double rgo_gain_from_loudness(double loudness_db) { return PINK_REF - loudness_db; } /* :911 */

/* ReplayGainResult::gain_steps, src/replaygain.rs:72-74 (Rust round = half away from zero) */
int32_t rgo_gain_steps(double gain_db) { return rust_f64_as_i32(round(gain_db / GAIN_STEP_DB)); }

/* filters + analyzer + peak creation, src/replaygain.rs:866-878 */
int rgo_track_begin(rgo_track_state *s, unsigned sample_rate, unsigned channels) {
    memset(s, 0, sizeof *s);
    s->sample_rate = sample_rate;
    s->channels = channels;
    if (rgo_filter_init(&s->filt[0], sample_rate) != 0) return -1;
    rgo_filter_init(&s->filt[1], sample_rate);
    rgo_analyzer_init(&s->an, sample_rate);
    s->peak = 0.0;
    return 0;
}

static double peak_max(double p, double v) { return v > p ? v : p; } /* f64::max, no NaNs here */

/* process_audio_buffer, src/replaygain.rs:953-1029 */
void rgo_process_buffer(rgo_track_state *s, const void *ch0, const void *ch1, size_t frames, int fmt) {
    const int stereo = (s->channels >= 2) && ch1 != NULL;
    if (fmt == RGO_FMT_F32) {
        const float *l = (const float *)ch0, *r = (const float *)ch1;
        for (size_t i = 0; i < frames; ++i) {
            double ln = (double)l[i];
            s->peak = peak_max(s->peak, fabs(ln));
            double lf = rgo_filter_process(&s->filt[0], ln * SAMPLE_SCALE_16BIT);
            if (stereo) {
                double rn = (double)r[i];
                s->peak = peak_max(s->peak, fabs(rn));
                double rf = rgo_filter_process(&s->filt[1], rn * SAMPLE_SCALE_16BIT);
                rgo_analyzer_add_sample(&s->an, lf, rf);
            } else {
                rgo_analyzer_add_mono_sample(&s->an, lf);
            }
        }
    } else if (fmt == RGO_FMT_S16) {
        const int16_t *l = (const int16_t *)ch0, *r = (const int16_t *)ch1;
        for (size_t i = 0; i < frames; ++i) {
            double lv = (double)l[i];
            s->peak = peak_max(s->peak, fabs(lv / SAMPLE_SCALE_16BIT));
            double lf = rgo_filter_process(&s->filt[0], lv);
            if (stereo) {
                double rv = (double)r[i];
                s->peak = peak_max(s->peak, fabs(rv / SAMPLE_SCALE_16BIT));
                double rf = rgo_filter_process(&s->filt[1], rv);
                rgo_analyzer_add_sample(&s->an, lf, rf);
            } else {
                rgo_analyzer_add_mono_sample(&s->an, lf);
            }
        }
    } else if (fmt == RGO_FMT_S32) {
        const int32_t *l = (const int32_t *)ch0, *r = (const int32_t *)ch1;
        const double scale = SAMPLE_SCALE_16BIT / 2147483648.0;
        for (size_t i = 0; i < frames; ++i) {
            double lv = (double)l[i] * scale;
            s->peak = peak_max(s->peak, fabs(lv / SAMPLE_SCALE_16BIT));
            double lf = rgo_filter_process(&s->filt[0], lv);
            if (stereo) {
                double rv = (double)r[i] * scale;
                s->peak = peak_max(s->peak, fabs(rv / SAMPLE_SCALE_16BIT));
                double rf = rgo_filter_process(&s->filt[1], rv);
                rgo_analyzer_add_sample(&s->an, lf, rf);
            } else {
                rgo_analyzer_add_mono_sample(&s->an, lf);
            }
        }
    }
    /* any other sample format: packet silently skipped (src/replaygain.rs:1025-1027) */
}

/* tail of analyze_track_internal, src/replaygain.rs:906-925 */
void rgo_track_finish(rgo_track_state *s, rgo_result *out) {
    rgo_analyzer_finish_window(&s->an);
    out->loudness_db = rgo_hist_loudness(s->an.hist);
    out->gain_db = PINK_REF - out->loudness_db;
    out->peak = s->peak;
    out->sample_rate = s->sample_rate;
    out->gain_steps = rgo_gain_steps(out->gain_db);
}

int rgo_analyze_pcm(const void *ch0, const void *ch1, size_t frames, unsigned sample_rate, int fmt,
                    rgo_result *out, uint32_t *hist_out) {
    rgo_track_state s;
    if (rgo_track_begin(&s, sample_rate, ch1 ? 2u : 1u) != 0) return -1;
    rgo_process_buffer(&s, ch0, ch1, frames, fmt);
    rgo_track_finish(&s, out);
    if (hist_out) memcpy(hist_out, s.an.hist, sizeof s.an.hist);
    return 0;
}

/* find_peak_amplitude's loop, src/replaygain.rs:1210-1241 */
double rgo_find_peak(const void *const *chans, unsigned nch, size_t frames, int fmt) {
    double max_peak = 0.0;
    for (size_t i = 0; i < frames; ++i) {
        for (unsigned c = 0; c < nch; ++c) {
            double v;
            if (fmt == RGO_FMT_F32) v = (double)fabsf(((const float *)chans[c])[i]);
            else if (fmt == RGO_FMT_S16) v = fabs((double)((const int16_t *)chans[c])[i]) / SAMPLE_SCALE_16BIT;
            else v = fabs((double)((const int32_t *)chans[c])[i]) / 2147483648.0;
            max_peak = peak_max(max_peak, v);
        }
    }
    return max_peak;
}

/* src/main.rs:2033-2058 with db_to_steps from src/lib.rs:632-634 */
int32_t rgo_clip_limit_steps(int32_t steps, double gain_db, double peak, int prevent_clipping, int wrap) {
    int32_t actual = steps;
    if (steps > 0 && !wrap) {
        double gain_linear = pow(10.0, gain_db / 20.0);
        double new_peak = peak * gain_linear;
        if (new_peak > 1.0 && prevent_clipping) {
            double max_safe_db = -20.0 * log10(peak);
            int32_t max_safe_steps = rust_f64_as_i32(round(max_safe_db / GAIN_STEP_DB));
            actual = max_safe_steps > 0 ? max_safe_steps : 0;
        }
    }
    return actual;
}

/* The signal of the reference's two unit tests, src/replaygain.rs:1296-1365: f64 sine fed
 * directly to filter.process -> add_mono_sample; the tests read get_loudness() without a
 * final flush, which changes nothing here because 1 s is a whole number of windows. */
double rgo_unit_test_sine(unsigned sample_rate, double frequency, double amplitude_normalized,
                          size_t duration_samples, uint32_t *hist_out) {
    rgo_filter f;
    rgo_analyzer a;
    if (rgo_filter_init(&f, sample_rate) != 0) return NAN;
    rgo_analyzer_init(&a, sample_rate);
    const double amplitude = amplitude_normalized * SAMPLE_SCALE_16BIT;
    const double pi = 3.14159265358979323846264338327950288; /* std::f64::consts::PI */
    for (size_t i = 0; i < duration_samples; ++i) {
        double t = (double)i / (double)sample_rate;
        double sample = amplitude * sin(2.0 * pi * frequency * t);
        rgo_analyzer_add_mono_sample(&a, rgo_filter_process(&f, sample));
    }
    if (hist_out) memcpy(hist_out, a.hist, sizeof a.hist);
    return rgo_hist_loudness(a.hist);
}

void rgo_synth_fill_f32(float *dst, uint64_t seed, unsigned channel, unsigned sample_rate,
                        uint64_t first_frame, size_t frames) {
    for (size_t i = 0; i < frames; ++i)
        dst[i] = rg_synth_sample_f32(seed, channel, sample_rate, first_frame + i);
}
