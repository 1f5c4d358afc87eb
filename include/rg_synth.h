/* rg_synth.h -- deterministic synthetic PCM shared by host (oracle, tests) and device (bench).
 *
 * The reference has no synthetic-input generator; this one exists so that BASELINE's
 * configs (10-minute tracks, 1000-track batches: up to 63.5 GB of planar f32) can be
 * produced directly in HBM and, sample for sample identically, on the host for the CPU
 * oracle.  Everything is integer arithmetic followed by one exact int->float scale by
 * 2^-23, so host and gfx950 agree bit for bit without any floating-point contract issues.
 *
 * Signal ("music-like"): approximately Gaussian white noise (sum of four 16-bit uniforms)
 * plus a per-track triangle tone, shaped by a slow triangle^2 envelope (period 3 s) and a
 * per-track level in [0.25, 1.0]; one second of digital silence at t in [10 s, 11 s)
 * exercises the reference's dropped-window rule (src/replaygain.rs:757: idx < HISTOGRAM_SIZE).
 * Seeds with bit 40 set ("hot" tracks) are boosted 8x and hard-clipped to +-1.0 so that
 * peak >= 1.0 occurs for the -k clip-limiting rule (src/main.rs:2033-2058).
 */
#ifndef RG_SYNTH_H
#define RG_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RG_SYNTH_FN static __host__ __device__ __forceinline__
#else
#define RG_SYNTH_FN static inline
#endif

#define RG_SYNTH_HOT_BIT (1ull << 40)

RG_SYNTH_FN uint64_t rg_synth_mix64(uint64_t x) {
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

/* 24-bit signed sample value k, |k| <= 2^23; the PCM sample is k * 2^-23. */
RG_SYNTH_FN int32_t rg_synth_sample_q23(uint64_t seed, unsigned channel, unsigned sample_rate,
                                        uint64_t frame) {
    /* silence gap [10 s, 11 s) */
    const uint64_t gap0 = 10ull * sample_rate, gap1 = 11ull * sample_rate;
    if (frame >= gap0 && frame < gap1) return 0;

    const uint64_t tp = rg_synth_mix64(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull);
    const int64_t level_q15 = 8192 + (int64_t)(tp % 24577ull);            /* 0.25 .. 1.0 */
    const uint64_t tone_step = 400ull + ((tp >> 20) % 3000ull);            /* phase step / 2^16 */

    const uint64_t u =
        rg_synth_mix64((seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(channel + 1u))) + frame * 0x9E3779B97F4A7C15ull);
    const int64_t g = (int64_t)(u & 0xFFFFull) + (int64_t)((u >> 16) & 0xFFFFull) +
                      (int64_t)((u >> 32) & 0xFFFFull) + (int64_t)(u >> 48) - 131070;   /* +-131070 */

    const int64_t ph = (int64_t)((frame * tone_step + (channel ? 16384ull : 0ull)) & 0xFFFFull);
    const int64_t tri = (ph < 32768 ? ph : 65535 - ph) * 2 - 32768;                      /* +-32768 */

    const uint64_t period = 3ull * sample_rate;
    const uint64_t pos = frame % period;
    const uint64_t half = period / 2;
    const uint64_t up = pos < half ? pos : period - pos;                                 /* 0..half */
    const int64_t q = (int64_t)((up << 15) / half);                                      /* 0..32768 */
    const int64_t env_q15 = 33 + ((19661 * q * q) >> 30);                                /* ~0.001..0.6 */

    int64_t v = (3 * g + tri) * env_q15;            /* < 2^34 */
    v *= level_q15;                                 /* < 2^49 */
    int64_t mag = v < 0 ? -v : v;
    mag >>= 25;
    if (seed & RG_SYNTH_HOT_BIT) mag <<= 3;
    if (mag > 8388608) mag = 8388608;
    return (int32_t)(v < 0 ? -mag : mag);
}

RG_SYNTH_FN float rg_synth_sample_f32(uint64_t seed, unsigned channel, unsigned sample_rate,
                                      uint64_t frame) {
    return (float)rg_synth_sample_q23(seed, channel, sample_rate, frame) * (1.0f / 8388608.0f);
}

#endif /* RG_SYNTH_H */
