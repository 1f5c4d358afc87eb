// rg_k2_tm.hip -- variant 2: transient-moment (TM) kernels, the fast path.  See rg_tm.h for the math.
//
//   rg_tm_main_kernel   one lane = one segment of L frames, all channels of it.  Streams the PCM once,
//                       runs the cascade (src/replaygain.rs:586-616) from the zero state in transposed
//                       direct form II with FMA, accumulates A = sum zs^2 and B_j = sum zs T_j, tracks the
//                       peak (src/replaygain.rs:967,973) and stores (A, B, E) per segment and channel.
//   rg_tm_fix_kernel    one lane = one segment.  Reconstructs the true start state of every segment from
//                       its predecessors' zero-state end states (doubling scan in LDS over 2^R lanes),
//                       evaluates sum z^2 = A + 2 B.sigma + sigma'G sigma, adds the k segments of each
//                       50 ms window, converts to the 0.01 dB bin (finish_window, src/replaygain.rs:743-765)
//                       and merges equal bins in LDS before one global atomic per distinct bin.
//
// FP64 vector FMA bound (27 FMA + 1 convert per channel-sample, plus 2..12 moment FMAs); MFMA is
// not used.  Compiled with the default -ffp-contract (explicit fma() everywhere anyway).
#include <hip/hip_runtime.h>

#include "rg_device.h"
#include "rg_device_inl.h"
#include "rg_tm.h"

namespace {

// one cascade step, DF2T.  The reference's "+1e-10" per stage (src/replaygain.rs:595,608) is the
// constant K.c0 injected at the deepest state of each stage: it then reaches the stage output as a
// constant once per sample, which is what the reference adds (rg_tm.h).
__device__ __forceinline__ double tm_step(double (&s)[10], double (&t)[2], const double x, const RgTmCoef &K) {
    const double y = fma(K.b[0], x, s[0]);
#pragma unroll
    for (int i = 0; i < 9; ++i) s[i] = fma(-K.a[i + 1], y, fma(K.b[i + 1], x, s[i + 1]));
    s[9] = fma(-K.a[10], y, fma(K.b[10], x, K.c0));
    const double z = fma(K.bb[0], y, t[0]);
    t[0] = fma(-K.ba[1], z, fma(K.bb[1], y, t[1]));
    t[1] = fma(-K.ba[2], z, fma(K.bb[2], y, K.c0));
    return z;
}

template <int NCH>
struct TmLane {
    double s[NCH][10];
    double t[NCH][2];
    double A[NCH];
    double B[NCH][RG_TM_DIM];
};

// NX = number of transient moments still alive: 12 (all) or 2 (Butterworth pair only)
template <int NCH, int NX>
__device__ __forceinline__ void tm_sample(TmLane<NCH> &st, const double (&x)[NCH], const double *__restrict__ Trow,
                                          const RgTmCoef &K) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const double z = tm_step(st.s[c], st.t[c], x[c], K);
        st.A[c] = fma(z, z, st.A[c]);
#pragma unroll
        for (int j = RG_TM_DIM - NX; j < RG_TM_DIM; ++j) st.B[c][j] = fma(z, Trow[j], st.B[c][j]);
    }
}

struct __attribute__((packed, aligned(4))) F32x4 { float v[4]; };

// ---- per-format sample access: raw value as double (the power-of-two scale is folded into K.b) and
// ---- the peak accumulator in the format's own domain
template <int FMT> struct Fmt;
template <> struct Fmt<RG_FMT_F32_PLANAR> {
    typedef float elem;
    typedef float peak_t;
    static __device__ __forceinline__ double cvt(float v, float &pk) { pk = fmaxf(pk, fabsf(v)); return (double)v; }
    static __device__ __forceinline__ double peak_norm(float pk) { return (double)pk; }
};
template <> struct Fmt<RG_FMT_S16_PLANAR> {
    typedef int16_t elem;
    typedef uint32_t peak_t;
    static __device__ __forceinline__ double cvt(int16_t v, uint32_t &pk) {
        const int32_t w = v;
        const uint32_t m = (uint32_t)(w < 0 ? -w : w);
        pk = m > pk ? m : pk;
        return (double)w;
    }
    static __device__ __forceinline__ double peak_norm(uint32_t pk) { return (double)pk / 32768.0; }
};
template <> struct Fmt<RG_FMT_S32_PLANAR> {
    typedef int32_t elem;
    typedef uint32_t peak_t;
    static __device__ __forceinline__ double cvt(int32_t v, uint32_t &pk) {
        const uint32_t m = v < 0 ? (uint32_t)0 - (uint32_t)v : (uint32_t)v;
        pk = m > pk ? m : pk;
        return (double)v;
    }
    static __device__ __forceinline__ double peak_norm(uint32_t pk) { return (double)pk / 2147483648.0; }
};

template <typename T>
__device__ __forceinline__ uint32_t find_track(const RgTmTrack *__restrict__ tracks, uint32_t n_tracks, uint32_t block,
                                               T RgTmTrack::*base) {
    uint32_t lo = 0, hi = n_tracks - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (tracks[mid].*base <= block) lo = mid; else hi = mid - 1;
    }
    return lo;
}

}  // namespace

// =================================================================================================
template <int FMT, int NCH>
__global__ void __launch_bounds__(RG_TM_BLOCK)
rg_tm_main_kernel(const RgTmCoef K, const RgTmGeom G, const RgTmTrack *__restrict__ tracks, uint32_t n_tracks,
                  double *__restrict__ rec, uint32_t total_recs, unsigned long long *__restrict__ peak_bits) {
    typedef Fmt<FMT> F;
    typedef typename F::elem elem;
    const uint32_t t = find_track(tracks, n_tracks, blockIdx.x, &RgTmTrack::main_block_base);
    const RgTmTrack tr = tracks[t];
    const uint32_t seg = (blockIdx.x - tr.main_block_base) * RG_TM_BLOCK + threadIdx.x;
    const uint32_t L = G.L;
    const bool active = seg < tr.nseg;
    const uint64_t start = (uint64_t)seg * L;
    uint32_t len = 0;
    if (active) {
        const uint64_t rem = tr.frames - start;
        len = rem < L ? (uint32_t)rem : L;
    }
    const elem *__restrict__ p[2] = {(const elem *)tr.ch0 + start, (const elem *)tr.ch1 + start};
    const double *__restrict__ T = G.T;

    TmLane<NCH> st;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int i = 0; i < 10; ++i) st.s[c][i] = 0.0;
        st.t[c][0] = st.t[c][1] = 0.0;
        st.A[c] = 0.0;
#pragma unroll
        for (int j = 0; j < RG_TM_DIM; ++j) st.B[c][j] = 0.0;
    }
    typename F::peak_t pk = 0;

    if (__all(len == L)) {
        // ---- fast path: every lane of the wave owns a full segment ------------------------------
        const uint32_t L4 = L & ~3u;
        const uint32_t H = G.H10;  // multiple of 4, <= L4
        uint32_t n = 0;
        if constexpr (FMT == RG_FMT_F32_PLANAR) {
            F32x4 cur[NCH], nxt[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) cur[c] = *reinterpret_cast<const F32x4 *>(p[c]);
            for (; n < H; n += 4) {
                const uint32_t nn = n + 4 < L4 ? n + 4 : n;  // prefetch the next chunk (clamped)
#pragma unroll
                for (int c = 0; c < NCH; ++c) nxt[c] = *reinterpret_cast<const F32x4 *>(p[c] + nn);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    double x[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) x[c] = F::cvt(cur[c].v[u], pk);
                    tm_sample<NCH, 12>(st, x, T + (size_t)(n + u) * RG_TM_DIM, K);
                }
#pragma unroll
                for (int c = 0; c < NCH; ++c) cur[c] = nxt[c];
            }
            for (; n < L4; n += 4) {
                const uint32_t nn = n + 4 < L4 ? n + 4 : n;
#pragma unroll
                for (int c = 0; c < NCH; ++c) nxt[c] = *reinterpret_cast<const F32x4 *>(p[c] + nn);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    double x[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) x[c] = F::cvt(cur[c].v[u], pk);
                    tm_sample<NCH, 2>(st, x, T + (size_t)(n + u) * RG_TM_DIM, K);
                }
#pragma unroll
                for (int c = 0; c < NCH; ++c) cur[c] = nxt[c];
            }
        } else {
            for (; n < H; ++n) {
                double x[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) x[c] = F::cvt(p[c][n], pk);
                tm_sample<NCH, 12>(st, x, T + (size_t)n * RG_TM_DIM, K);
            }
            for (; n < L4; ++n) {
                double x[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) x[c] = F::cvt(p[c][n], pk);
                tm_sample<NCH, 2>(st, x, T + (size_t)n * RG_TM_DIM, K);
            }
        }
        for (; n < L; ++n) {  // L mod 4 trailing frames
            double x[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) x[c] = F::cvt(p[c][n], pk);
            tm_sample<NCH, 12>(st, x, T + (size_t)n * RG_TM_DIM, K);
        }
    } else {
        // ---- tail path: some lane of this wave has a short (or no) segment ------------------------
        // frames past `len` are fed as zeros and their output is excluded from the moments; the end
        // state of such a lane is never used (the track ends inside it).
        for (uint32_t n = 0; n < L; ++n) {
            const bool valid = n < len;
            double x[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                x[c] = 0.0;
                if (valid) x[c] = F::cvt(p[c][n], pk);
            }
            const double *__restrict__ Trow = T + (size_t)n * RG_TM_DIM;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                double z = tm_step(st.s[c], st.t[c], x[c], K);
                z = valid ? z : 0.0;
                st.A[c] = fma(z, z, st.A[c]);
#pragma unroll
                for (int j = 0; j < RG_TM_DIM; ++j) st.B[c][j] = fma(z, Trow[j], st.B[c][j]);
            }
        }
    }

    // ---- segment records, structure-of-arrays: field f of channel c at ((c*25 + f) * total_recs + idx)
    if (active) {
        const size_t idx = (size_t)tr.rec_base + seg;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            double *__restrict__ r = rec + (size_t)c * RG_TM_REC * total_recs + idx;
            r[0] = st.A[c];
#pragma unroll
            for (int j = 0; j < RG_TM_DIM; ++j) r[(size_t)(1 + j) * total_recs] = st.B[c][j];
#pragma unroll
            for (int j = 0; j < 10; ++j) r[(size_t)(13 + j) * total_recs] = st.s[c][j];
            r[(size_t)23 * total_recs] = st.t[c][0];
            r[(size_t)24 * total_recs] = st.t[c][1];
        }
    }

    // ---- peak: wave max, one atomic per wave (f64 bit pattern of a non-negative value is ordered) ----
    unsigned long long pb = (unsigned long long)__double_as_longlong(F::peak_norm(pk));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(pb, off, 64);
        pb = o > pb ? o : pb;
    }
    if ((threadIdx.x & 63) == 0 && pb != 0) atomicMax(&peak_bits[tr.track_index], pb);
}

// =================================================================================================
template <int NCH>
__global__ void __launch_bounds__(RG_TM_BLOCK)
rg_tm_fix_kernel(const RgTmGeom G, const RgTmFixTables FT, const RgTmTrack *__restrict__ tracks, uint32_t n_tracks,
                 const double *__restrict__ rec, uint32_t total_recs, uint32_t *__restrict__ hist) {
    __shared__ double wx[RG_TM_DIM][RG_TM_BLOCK];
    __shared__ double pieces[RG_TM_BLOCK];
    __shared__ int bins[RG_TM_BLOCK];

    const uint32_t t = find_track(tracks, n_tracks, blockIdx.x, &RgTmTrack::fix_block_base);
    const RgTmTrack tr = tracks[t];
    const uint32_t b = blockIdx.x - tr.fix_block_base;
    const int i = threadIdx.x;
    const int warm = (int)G.warm;
    const uint32_t NB = G.fix_windows * G.k;
    const long long seg = (long long)b * NB - warm + i;
    const bool in_block = i < warm + (int)NB;
    const bool seg_valid = in_block && seg >= 0 && seg < (long long)tr.nseg;
    const bool owner = seg_valid && i >= warm;
    const size_t idx = (size_t)tr.rec_base + (size_t)(seg_valid ? seg : 0);

    double S = 0.0;
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        const double *__restrict__ r = rec + (size_t)c * RG_TM_REC * total_recs + idx;
        // zero-state end state of this segment in block-diagonal coordinates: t' = t + X s
        double w[RG_TM_DIM];
        if (seg_valid) {
#pragma unroll
            for (int j = 0; j < RG_TM_DIM; ++j) w[j] = r[(size_t)(13 + j) * total_recs];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                double acc = w[10 + q];
#pragma unroll
                for (int j = 0; j < 10; ++j) acc = fma(FT.X[q * 10 + j], w[j], acc);
                w[10 + q] = acc;
            }
        } else {
            const bool start = in_block && seg == -1;  // virtual segment -1 carries the track-start state
#pragma unroll
            for (int j = 0; j < RG_TM_DIM; ++j) w[j] = start ? FT.sigma0[j] : 0.0;
        }
        // doubling scan: after round r, w_k = sum_{q < 2^(r+1)} Phi^q e_{k-q}
        for (uint32_t rd = 0; rd < G.rounds; ++rd) {
            const int d = 1 << rd;
#pragma unroll
            for (int j = 0; j < RG_TM_DIM; ++j) wx[j][i] = w[j];
            __syncthreads();
            double wn[RG_TM_DIM];
#pragma unroll
            for (int j = 0; j < RG_TM_DIM; ++j) wn[j] = i >= d ? wx[j][i - d] : 0.0;
            __syncthreads();
            if (rd < G.rounds_fast) {
                const double *__restrict__ PY = FT.PhiY + (size_t)rd * 100;
#pragma unroll
                for (int a = 0; a < 10; ++a) {
                    double acc = w[a];
#pragma unroll
                    for (int q = 0; q < 10; ++q) acc = fma(PY[a * 10 + q], wn[q], acc);
                    w[a] = acc;
                }
            }
            const double *__restrict__ PB = FT.PhiB + (size_t)rd * 4;
            w[10] = fma(PB[0], wn[10], fma(PB[1], wn[11], w[10]));
            w[11] = fma(PB[2], wn[10], fma(PB[3], wn[11], w[11]));
        }
        // the true start state of segment k is the scanned end state of segment k-1
#pragma unroll
        for (int j = 0; j < RG_TM_DIM; ++j) wx[j][i] = w[j];
        __syncthreads();
        double sg[RG_TM_DIM];
#pragma unroll
        for (int j = 0; j < RG_TM_DIM; ++j) sg[j] = i >= 1 ? wx[j][i - 1] : 0.0;
        __syncthreads();

        if (owner) {
            const uint64_t start = (uint64_t)seg * G.L;
            const uint64_t rem = tr.frames - start;
            const uint32_t len = rem < G.L ? (uint32_t)rem : G.L;
            const double *__restrict__ Gm = FT.Gp + (size_t)(len - 1) * RG_TM_GRAM;
            double lin = 0.0;
#pragma unroll
            for (int j = 0; j < RG_TM_DIM; ++j) lin = fma(r[(size_t)(1 + j) * total_recs], sg[j], lin);
            double quad = 0.0;
            int p = 0;
#pragma unroll
            for (int j = 0; j < RG_TM_DIM; ++j) {
                double row = 0.5 * Gm[p] * sg[j];
                ++p;
#pragma unroll
                for (int q = j + 1; q < RG_TM_DIM; ++q, ++p) row = fma(Gm[p], sg[q], row);
                quad = fma(row, sg[j], quad);
            }
            S += r[0] + 2.0 * (lin + quad);
        }
    }
    if (NCH == 1) S *= 2.0;  // add_mono_sample feeds both sums (src/replaygain.rs:731-740)
    pieces[i] = owner ? S : 0.0;
    __syncthreads();

    // ---- 50 ms windows: k consecutive segments each (finish_window, src/replaygain.rs:743-765) ----
    int bin = -1;
    if ((uint32_t)i < G.fix_windows) {
        const uint64_t widx = (uint64_t)b * G.fix_windows + i;
        if (widx < tr.n_windows) {
            double total = 0.0;
            for (uint32_t q = 0; q < G.k; ++q) total += pieces[warm + i * G.k + q];
            const uint64_t rem = tr.frames - widx * G.W;
            const uint32_t n = rem < G.W ? (uint32_t)rem : G.W;
            bin = rg_window_bin(total, 0.0, n);
        }
    }
    bins[i] = bin;
    __syncthreads();
    // ---- LDS-side merge of equal bins, one global atomic per distinct bin of this block ----------
    if (bin >= 0) {
        bool leader = true;
        for (int q = 0; q < i; ++q)
            if (bins[q] == bin) { leader = false; break; }
        if (leader) {
            uint32_t count = 1;
            for (uint32_t q = i + 1; q < G.fix_windows; ++q) count += bins[q] == bin ? 1u : 0u;
            atomicAdd(&hist[(size_t)tr.track_index * RG_HISTOGRAM_SIZE + bin], count);
        }
    }
}

// =================================================================================================
template <int FMT>
static hipError_t launch_main_fmt(int nch, const RgTmCoef &K, const RgTmGeom &G, const RgTmTrack *d_tracks,
                                  uint32_t n_tracks, uint32_t grid, double *d_rec, uint32_t total_recs,
                                  unsigned long long *d_peak_bits, hipStream_t s) {
    if (nch == 1)
        hipLaunchKernelGGL((rg_tm_main_kernel<FMT, 1>), dim3(grid), dim3(RG_TM_BLOCK), 0, s, K, G, d_tracks, n_tracks,
                           d_rec, total_recs, d_peak_bits);
    else
        hipLaunchKernelGGL((rg_tm_main_kernel<FMT, 2>), dim3(grid), dim3(RG_TM_BLOCK), 0, s, K, G, d_tracks, n_tracks,
                           d_rec, total_recs, d_peak_bits);
    return hipGetLastError();
}

extern "C" hipError_t rg_launch_tm_main(int fmt, int nch, const RgTmCoef *K, const RgTmGeom *G,
                                        const RgTmTrack *d_tracks, uint32_t n_tracks, uint32_t grid, double *d_rec,
                                        uint32_t total_recs, unsigned long long *d_peak_bits, hipStream_t s) {
    if (grid == 0) return hipSuccess;
    switch (fmt) {
        case RG_FMT_F32_PLANAR: return launch_main_fmt<RG_FMT_F32_PLANAR>(nch, *K, *G, d_tracks, n_tracks, grid, d_rec, total_recs, d_peak_bits, s);
        case RG_FMT_S16_PLANAR: return launch_main_fmt<RG_FMT_S16_PLANAR>(nch, *K, *G, d_tracks, n_tracks, grid, d_rec, total_recs, d_peak_bits, s);
        default: return launch_main_fmt<RG_FMT_S32_PLANAR>(nch, *K, *G, d_tracks, n_tracks, grid, d_rec, total_recs, d_peak_bits, s);
    }
}

extern "C" hipError_t rg_launch_tm_fix(int nch, const RgTmGeom *G, const RgTmFixTables *FT, const RgTmTrack *d_tracks,
                                       uint32_t n_tracks, uint32_t grid, const double *d_rec, uint32_t total_recs,
                                       uint32_t *d_hist, hipStream_t s) {
    if (grid == 0) return hipSuccess;
    if (nch == 1)
        hipLaunchKernelGGL((rg_tm_fix_kernel<1>), dim3(grid), dim3(RG_TM_BLOCK), 0, s, *G, *FT, d_tracks, n_tracks, d_rec,
                           total_recs, d_hist);
    else
        hipLaunchKernelGGL((rg_tm_fix_kernel<2>), dim3(grid), dim3(RG_TM_BLOCK), 0, s, *G, *FT, d_tracks, n_tracks, d_rec,
                           total_recs, d_hist);
    return hipGetLastError();
}
