// rg_mp3dev.hip -- the device half of the split MP3 decoder: stages B-E of rg_mp3dec.cpp on gfx950.
//
//   rg_mp3_hybrid_kernel   one block per granule (both channels): requantisation, joint stereo (mid/side, intensity in
//                          the MPEG-1 and the LSF form), short-block reordering, alias reduction, IMDCT + windowing.
//                          Writes the two halves of every subband's 36 windowed samples: `first` overlaps with the
//                          previous granule's `second`.
//   rg_mp3_synth_kernel    one block per granule and channel: overlap-add + frequency inversion, then the polyphase
//                          synthesis filterbank (matrixing into 64-vectors, 512-tap window over sixteen of them) ->
//                          576 PCM samples, planar f32, straight into the analysis arena.
//
// Nothing here is recursive across granules: the overlap is a read of the previous granule's second half, the
// filterbank's FIFO a read of the previous fifteen time slots' subband samples (recomputed from the previous granule's
// halves), so every granule of every track of a batch is independent work.
//
// Bit-identical to the host decoder by construction: compiled with -ffp-contract=off, every sum in the host's order
// (sequential, from 0.0f), every constant from the host's own tables (rg_mp3_fill_device_tables), the data-dependent
// powers of two from a table indexed by the exact integer exponent.  tests/test_gpu_mp3.py demands equality.
#include <hip/hip_runtime.h>

#include "rg_mp3dev.h"

namespace {

__device__ __forceinline__ uint32_t find_by_granule(const RgMp3DevTrack *__restrict__ tr, uint32_t n, uint32_t g) {
    uint32_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (tr[mid].granule_base <= g) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__device__ __forceinline__ uint32_t find_by_unit(const RgMp3DevTrack *__restrict__ tr, uint32_t n, uint64_t u) {
    uint32_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (tr[mid].unit_base <= u) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// hyb[unit][half][t][sb]
__device__ __forceinline__ size_t hyb_index(uint64_t unit, int half, int t, int sb) {
    return (((size_t)unit * 2 + (size_t)half) * 18 + (size_t)t) * 32 + (size_t)sb;
}

}  // namespace

__global__ void __launch_bounds__(256)
rg_mp3_hybrid_kernel(const RgMp3DevTables *__restrict__ T, const RgMp3DevTrack *__restrict__ tracks, uint32_t n_tracks,
                     const rg_mp3_unit *__restrict__ units, const int16_t *__restrict__ is, float *__restrict__ hyb) {
    __shared__ float xr[2][576];
    __shared__ float tmp[576];
    __shared__ rg_mp3_unit U[2];
    __shared__ float gain_long[2][22], gain_short[2][39];
    __shared__ int band_nz[64];
    __shared__ short band_mode[64];
    const int tid = threadIdx.x;
    const uint32_t ti = find_by_granule(tracks, n_tracks, blockIdx.x);
    const RgMp3DevTrack tr = tracks[ti];
    const uint32_t g = blockIdx.x - tr.granule_base;
    const int nch = (int)tr.channels;
    const int rr = (int)tr.rate_row;
    const uint64_t u0 = tr.unit_base + (uint64_t)g * nch;
    if (tid < nch) U[tid] = units[u0 + tid];
    if (tid < 64) { band_nz[tid] = 0; band_mode[tid] = 0; }
    __syncthreads();

    // ---- stage B: requantisation (rg_mp3dec.cpp: requantize) -----------------------------------------------------
    // gains per band: 2^(e), e = (global_gain - 210)/4 - mult (sf + preflag pretab) [- 2 subblock_gain], a multiple
    // of 1/4 exactly: the table is indexed by 4e
    for (int c = 0; c < nch; ++c) {
        const rg_mp3_unit &u = U[c];
        const int m4 = u.scalefac_scale ? 4 : 2;  // 4 * mult
        const int base4 = (int)u.global_gain - 210;
        if (tid < 22) {
            const int q = base4 - m4 * ((int)u.sf[tid] + (u.preflag ? (int)T->pretab[tid] : 0));
            gain_long[c][tid] = T->gain[q - RG_MP3_GAIN_Q_MIN];
        }
        if (tid >= 64 && tid < 64 + 39) {
            const int k = tid - 64;  // (band - short_start) * 3 + window
            const int band = (int)u.short_start + k / 3, w = k % 3;
            float gv = 0.0f;
            if (band < 13) {
                const int s = band < 12 ? (int)u.sf[(int)u.long_end + k] : 0;
                const int q = base4 - 8 * (int)u.subblock_gain[w] - m4 * s;
                gv = T->gain[q - RG_MP3_GAIN_Q_MIN];
            }
            gain_short[c][k] = gv;
        }
    }
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const rg_mp3_unit &u = U[c];
        const int long_lines = (int)T->sfb_long[rr][u.long_end];  // 0 when long_end == 0
        const int short_off = 3 * (int)T->sfb_short[rr][u.short_start < 13 ? u.short_start : 13];
        const int16_t *__restrict__ src = is + (u0 + c) * 576;
        for (int line = tid; line < 576; line += 256) {
            float gv;
            if (u.block_type != 2 || line < long_lines) {
                gv = gain_long[c][T->long_band_of_line[rr][line]];
            } else {
                const int k = (int)T->short_idx_of_line[rr][line - long_lines + short_off] - 3 * (int)u.short_start;
                gv = gain_short[c][k];
            }
            const int v = src[line];
            const int a = v < 0 ? -v : v;
            const float m = T->pow43[a] * gv;
            xr[c][line] = v < 0 ? -m : m;
        }
    }
    __syncthreads();

    // ---- stage C: joint stereo (rg_mp3dec.cpp: stereo) -----------------------------------------------------------
    if (nch == 2 && U[0].mode_ext != 0) {
        const rg_mp3_unit &u1 = U[1];
        const bool ms = (U[0].mode_ext & 2) != 0, is_on = (U[0].mode_ext & 1) != 0;
        const float isq2 = 0.70710678118654752440f;
        if (!is_on) {
            const int n = U[0].nz > U[1].nz ? U[0].nz : U[1].nz;
            for (int i = tid; i < n; i += 256) {
                const float a = xr[0][i], b = xr[1][i];
                xr[0][i] = (a + b) * isq2;
                xr[1][i] = (a - b) * isq2;
            }
        } else {
            const int long_lines = (int)T->sfb_long[rr][u1.long_end];
            const int short_off = 3 * (int)T->sfb_short[rr][u1.short_start < 13 ? u1.short_start : 13];
            // stereo band of a line: short bands 0..38 = (band - short_start) * 3 + window, long bands 39 + band
            auto band_of = [&](int line) -> int {
                if (u1.block_type != 2 || line < long_lines) return 39 + (int)T->long_band_of_line[rr][line];
                return (int)T->short_idx_of_line[rr][line - long_lines + short_off] - 3 * (int)u1.short_start;
            };
            for (int line = tid; line < 576; line += 256)
                if (xr[1][line] != 0.0f) band_nz[band_of(line)] = 1;
            __syncthreads();
            if (tid == 0) {
                // walk the bands from the top: a band is intensity coded while every band above it (of the same
                // window, for short blocks) has an all-zero right channel and its own position is legal
                const bool lsf = tr.lsf != 0;
                bool found[3] = {false, false, false};
                bool found_long = false;
                if (u1.block_type == 2) {
                    for (int b = 12; b >= (int)u1.short_start; --b) {
                        const int sb = b == 12 ? 11 : b;
                        for (int w = 2; w >= 0; --w) {
                            const int k = (b - (int)u1.short_start) * 3 + w;
                            const int idx = (int)u1.long_end + 3 * (sb - (int)u1.short_start) + w;
                            bool intensity = false;
                            int mode = 0;
                            if (!found[w]) {
                                if (band_nz[k]) {
                                    found[w] = true;
                                } else {
                                    const int p = u1.sf[idx];
                                    intensity = lsf ? !((u1.illegal >> idx) & 1ull) : p < 7;
                                    if (intensity) mode = 2 + p;
                                }
                            }
                            if (!intensity && ms) mode = 1;
                            band_mode[k] = (short)mode;
                        }
                    }
                    found_long = found[0] || found[1] || found[2];
                }
                if (!(u1.block_type == 2 && !u1.mixed)) {
                    for (int b = (int)u1.long_end - 1; b >= 0; --b) {
                        const int sb = b == 21 ? 20 : b;
                        bool intensity = false;
                        int mode = 0;
                        if (!found_long) {
                            if (band_nz[39 + b]) {
                                found_long = true;
                            } else {
                                const int p = u1.sf[sb];
                                intensity = lsf ? !((u1.illegal >> sb) & 1ull) : p < 7;
                                if (intensity) mode = 2 + p;
                            }
                        }
                        if (!intensity && ms) mode = 1;
                        band_mode[39 + b] = (short)mode;
                    }
                }
            }
            __syncthreads();
            const int scale = u1.intensity_scale & 1;
            for (int line = tid; line < 576; line += 256) {
                const int mode = band_mode[band_of(line)];
                if (mode == 1) {
                    const float a = xr[0][line], b = xr[1][line];
                    xr[0][line] = (a + b) * isq2;
                    xr[1][line] = (a - b) * isq2;
                } else if (mode >= 2) {
                    const int pos = mode - 2;
                    float kl, kr;
                    if (!tr.lsf) {
                        kl = T->is_l[pos];
                        kr = T->is_r[pos];
                    } else if (pos == 0) {
                        kl = kr = 1.0f;
                    } else if (pos & 1) {
                        kl = T->lsf_is[scale][(pos + 1) >> 1];
                        kr = 1.0f;
                    } else {
                        kl = 1.0f;
                        kr = T->lsf_is[scale][pos >> 1];
                    }
                    const float v = xr[0][line];
                    xr[0][line] = v * kl;
                    xr[1][line] = v * kr;
                }
            }
        }
        __syncthreads();
    }

    // ---- stage D per channel: reorder, alias reduction, IMDCT + window -------------------------------------------
    for (int c = 0; c < nch; ++c) {
        const rg_mp3_unit &u = U[c];
        float *X = xr[c];
        if (u.block_type == 2) {
            const int long_lines = u.mixed ? (int)T->sfb_long[rr][u.long_end] : 0;
            const int short_off = 3 * (int)T->sfb_short[rr][u.short_start];
            for (int line = tid; line < 576; line += 256)
                tmp[line] = line < long_lines ? X[line]
                                              : X[(int)T->short_reorder_src[rr][line - long_lines + short_off] - short_off + long_lines];
            __syncthreads();
            for (int line = tid; line < 576; line += 256) X[line] = tmp[line];
            __syncthreads();
        }
        const int boundaries = u.block_type == 2 ? (u.mixed ? 1 : 0) : 31;
        if (tid < boundaries * 8) {
            const int sb = 1 + tid / 8, i = tid % 8;
            const float a = X[sb * 18 - 1 - i], b = X[sb * 18 + i];
            X[sb * 18 - 1 - i] = a * T->cs[i] - b * T->ca[i];
            X[sb * 18 + i] = b * T->cs[i] + a * T->ca[i];
        }
        __syncthreads();
        for (int o = tid; o < 32 * 36; o += 256) {
            const int sb = o / 36, i = o % 36;
            const float *Xs = X + sb * 18;
            const int bt = (u.block_type == 2 && u.mixed && sb < 2) ? 0 : (int)u.block_type;
            float raw;
            if (bt != 2) {
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < 18; ++k) s += Xs[k] * T->imdct36[i][k];
                raw = s * T->win[bt][i];
            } else {
                raw = 0.0f;
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const int ii = i - 6 - 6 * w;
                    if (ii >= 0 && ii < 12) {
                        float s = 0.0f;
#pragma unroll
                        for (int k = 0; k < 6; ++k) s += Xs[3 * k + w] * T->imdct12[ii][k];
                        raw += s * T->win[2][ii];
                    }
                }
            }
            hyb[hyb_index(u0 + c, i < 18 ? 0 : 1, i < 18 ? i : i - 18, sb)] = raw;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
rg_mp3_synth_kernel(const RgMp3DevTables *__restrict__ T, const RgMp3DevTrack *__restrict__ tracks, uint32_t n_tracks,
                    const float *__restrict__ hyb, uint64_t first_unit) {
    __shared__ float S[33][32];   // subband samples of time slots -15 .. 17 relative to this granule
    __shared__ float V[33][64];
    const int tid = threadIdx.x;
    const uint64_t unit = first_unit + blockIdx.x;
    const uint32_t ti = find_by_unit(tracks, n_tracks, unit);
    const RgMp3DevTrack tr = tracks[ti];
    const uint64_t local = unit - tr.unit_base;
    const int nch = (int)tr.channels;
    const uint32_t g = (uint32_t)(local / nch);
    const int c = (int)(local % nch);
    // ---- overlap-add + frequency inversion (rg_mp3dec.cpp: hybrid, tail) -----------------------------------------
    for (int e = tid; e < 33 * 32; e += 256) {
        const int r = e / 32, sb = e % 32;
        const int slot = r - 15;
        float v;
        int t;
        if (slot >= 0) {
            t = slot;
            const float ov = g >= 1 ? hyb[hyb_index(unit - nch, 1, t, sb)] : 0.0f;
            v = hyb[hyb_index(unit, 0, t, sb)] + ov;
        } else {
            t = 18 + slot;
            if (g >= 1) {
                const float ov = g >= 2 ? hyb[hyb_index(unit - 2 * nch, 1, t, sb)] : 0.0f;
                v = hyb[hyb_index(unit - nch, 0, t, sb)] + ov;
            } else {
                v = 0.0f;
            }
        }
        if ((sb & 1) && (t & 1)) v = -v;
        S[r][sb] = v;
    }
    __syncthreads();
    // ---- polyphase synthesis: matrixing (rg_mp3dec.cpp: synth) ----------------------------------------------------
    for (int e = tid; e < 33 * 64; e += 256) {
        const int r = e / 64, i = e % 64;
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 32; ++k) s += T->matrix[i][k] * S[r][k];
        V[r][i] = s;
    }
    __syncthreads();
    float *__restrict__ dst = (c == 0 ? tr.ch0 : tr.ch1) + (size_t)g * 576;
    for (int e = tid; e < 576; e += 256) {
        const int t = e / 32, j = e % 32;
        const int r = 15 + t;
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s += V[r - 2 * i][j] * T->D[i * 64 + j];
            s += V[r - 2 * i - 1][32 + j] * T->D[i * 64 + 32 + j];
        }
        dst[e] = s;
    }
}

extern "C" hipError_t rg_launch_mp3_hybrid(const RgMp3DevTables *d_tab, const RgMp3DevTrack *d_tracks, uint32_t n_tracks,
                                           uint32_t n_granules, const rg_mp3_unit *d_units, const int16_t *d_is, float *d_hyb,
                                           hipStream_t s) {
    if (n_granules == 0) return hipSuccess;
    hipLaunchKernelGGL(rg_mp3_hybrid_kernel, dim3(n_granules), dim3(256), 0, s, d_tab, d_tracks, n_tracks, d_units, d_is, d_hyb);
    return hipGetLastError();
}

extern "C" hipError_t rg_launch_mp3_synth(const RgMp3DevTables *d_tab, const RgMp3DevTrack *d_tracks, uint32_t n_tracks,
                                          uint64_t n_units, const float *d_hyb, hipStream_t s) {
    if (n_units == 0) return hipSuccess;
    hipLaunchKernelGGL(rg_mp3_synth_kernel, dim3((uint32_t)n_units), dim3(256), 0, s, d_tab, d_tracks, n_tracks, d_hyb, 0ull);
    return hipGetLastError();
}
