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
// FP64 vector pipe: 26 operations + 1 convert per channel-sample with the Butterworth stage in servo form (rg_tm.h; 27
// multiply-adds in the classic form kept for four rates), plus 2..12 moment FMAs in the window the moments cover; MFMA is
// not used.  Compiled with the default -ffp-contract (explicit fma() everywhere anyway).
#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>

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

// the same step with the Butterworth stage in servo form (rg_tm.h: RgTmCoef): t = (v1, v2), linear (no offsets)
__device__ __forceinline__ double tm_step_servo(double (&s)[10], double (&t)[2], const double x, const RgTmCoef &K) {
    const double y = fma(K.b[0], x, s[0]);
#pragma unroll
    for (int i = 0; i < 9; ++i) s[i] = fma(-K.a[i + 1], y, fma(K.b[i + 1], x, s[i + 1]));
    s[9] = fma(-K.a[10], y, K.b[10] * x);
    const double z = y - t[0];
    const double p = t[0] + t[1];
    t[1] = fma(K.beta, z, t[1]);
    t[0] = fma(K.alpha, z, p);
    return z;
}

template <int NCH>
struct TmLane {
    double s[NCH][10];
    double t[NCH][2];
    double A[NCH];
    double B[NCH][RG_TM_DIM];
};

// The T table is read through the constant address space: its rows are wave-uniform, and only a
// constant-address-space load is guaranteed to become a scalar s_load (the asm memory clobbers around
// the LDS staging would otherwise demote these loads to per-lane vector loads).
typedef __attribute__((address_space(4))) const double rg_cdouble;

// NX = number of transient moments still alive: 12 (all) or 2 (Butterworth pair only)
template <int NCH, int NX>
__device__ __forceinline__ void tm_sample(TmLane<NCH> &st, const double (&x)[NCH], rg_cdouble *__restrict__ Trow,
                                          const RgTmCoef &K) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const double z = tm_step(st.s[c], st.t[c], x[c], K);
        st.A[c] = fma(z, z, st.A[c]);
#pragma unroll
        for (int j = RG_TM_DIM - NX; j < RG_TM_DIM; ++j) st.B[c][j] = fma(z, Trow[j], st.B[c][j]);
    }
}

// ---- per-format sample access: raw value as double (the power-of-two scale is folded into K.b) and
// ---- the peak accumulator in the format's own domain
template <int FMT> struct Fmt;
template <> struct Fmt<RG_FMT_F32_PLANAR> {
    typedef float elem;
    typedef float peak_t;
    static __device__ __forceinline__ double cvt(float v, float &pk) { pk = fmaxf(pk, fabsf(v)); return (double)v; }
    static __device__ __forceinline__ double peak_norm(float pk) { return (double)pk; }
    static __device__ __forceinline__ float peak_denorm(double v) { return (float)v; }  // exact: v came from a float
    // 32-bit LDS word <-> sample
    static __device__ __forceinline__ uint32_t word(float v) { return __float_as_uint(v); }
    static __device__ __forceinline__ double cvt_word(uint32_t w, float &pk) { return cvt(__uint_as_float(w), pk); }
    // the peak over a 4-frame piece in two v_max3_f32 instead of four v_max_f32 (fmaxf drops NaNs either way)
    static __device__ __forceinline__ void peak4(const uint32_t (&f)[4], float &pk) {
        pk = fmaxf(fmaxf(pk, fabsf(__uint_as_float(f[0]))), fabsf(__uint_as_float(f[1])));
        pk = fmaxf(fmaxf(pk, fabsf(__uint_as_float(f[2]))), fabsf(__uint_as_float(f[3])));
    }
    static __device__ __forceinline__ double word_value(uint32_t w) { return (double)__uint_as_float(w); }
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
    static __device__ __forceinline__ uint32_t peak_denorm(double v) { return (uint32_t)(v * 32768.0); }  // exact: a power of two
    static __device__ __forceinline__ uint32_t word(int16_t v) { return (uint32_t)(int32_t)v; }
    static __device__ __forceinline__ double cvt_word(uint32_t w, uint32_t &pk) { return cvt((int16_t)(int32_t)w, pk); }
    static __device__ __forceinline__ void peak4(const uint32_t (&f)[4], uint32_t &pk) {
#pragma unroll
        for (int u = 0; u < 4; ++u) (void)cvt_word(f[u], pk);
    }
    static __device__ __forceinline__ double word_value(uint32_t w) { return (double)(int32_t)w; }
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
    static __device__ __forceinline__ uint32_t peak_denorm(double v) { return (uint32_t)(v * 2147483648.0); }
    static __device__ __forceinline__ uint32_t word(int32_t v) { return (uint32_t)v; }
    static __device__ __forceinline__ double cvt_word(uint32_t w, uint32_t &pk) { return cvt((int32_t)w, pk); }
    static __device__ __forceinline__ void peak4(const uint32_t (&f)[4], uint32_t &pk) {
#pragma unroll
        for (int u = 0; u < 4; ++u) (void)cvt_word(f[u], pk);
    }
    static __device__ __forceinline__ double word_value(uint32_t w) { return (double)(int32_t)w; }
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
// Main kernel.  One lane = one (channel, segment); blockIdx.y is the channel.
//
// F32 fast path, everything the inner loop touches lives in LDS:
//  * PCM.  One wave owns 64 consecutive segments (rows) of one channel: 64 strided runs, L frames
//    apart.  Per tile of RG_TM_TILE = 16 frames every lane issues 4 global_load_dwordx4: instruction
//    q, lane j fetches one 16-byte piece of row 16q + (j >> 2), so an instruction reads 16 contiguous
//    64-byte runs (each 128-byte line of the stream is fetched once or twice in total, instead of
//    eight times by per-lane strided loads).  The pieces go through registers into the wave's private
//    4 KiB tile (ds_write_b128, row-major) one tile ahead of their use; the piece a lane fetches is
//    XOR-swizzled with (row >> 2) & 3 so that the consumer's row-per-lane ds_read_b128 is bank
//    conflict free (the 16 lanes of a read group hit 16 distinct 16-byte slots of the 256-byte bank
//    row).  The tile is wave-private: no block barrier in the loop.
//  * The transient-response table.  T rows are wave-uniform, but as scalar loads they cost one
//    exposed s_waitcnt lgkmcnt(0) per row (SMEM returns out of order), which left a lone wave at a
//    ninth of the FP64 issue rate.  The block copies the rows it needs into LDS once (12 doubles per
//    frame for n < H10, then only the Butterworth pair) and reads them as broadcast ds_read_b128,
//    which the compiler can keep in flight with counted lgkmcnt waits.
#define RG_TM_TILE 16

typedef uint32_t __attribute__((ext_vector_type(4))) rg_u32x4;
typedef uint32_t __attribute__((ext_vector_type(4), aligned(4))) rg_u32x4u;  // 16-byte load, 4-byte aligned
typedef short __attribute__((ext_vector_type(4), aligned(2))) rg_s16x4u;     // 8-byte load, 2-byte aligned

// One frame of one lane: cascade step + moments.  NX = live transient moments (12, or only the slow
// pair); MASK: frames at or past `len` do not count (tail of a track).
//
// The FP64 FMA pipe takes a new wave instruction every 4 cycles but a dependent one only after ~8, and
// a lone wave (short batches leave a SIMD one or two waves) has nothing else to issue.  The frame is
// therefore written as four groups inside which every instruction is independent, fenced with
// sched_barrier so that hipcc keeps them apart (left alone it pairs each u_i with the s_i that
// consumes it, which is a dependent chain of 2 x 10 instructions):
//   G1  y,  u_i = b_{i+1} x + s_{i+1}                      (10 + 1, need x and the old state)
//   G2  z,  w_1 = bb_1 y + t_1,  w_2 = bb_2 y + c          (need y, issued 10 slots earlier)
//   G3  s_i = u_i - a_{i+1} y                              (10, need y and u_i)
//   G4  t_0, t_1, A, B_j                                   (need z, issued 10 slots earlier)
template <int FMT, int NX, bool MASK, bool PEAK = true, bool SERVO = false>
__device__ __forceinline__ void tm_frame(TmLane<1> &st, const uint32_t wbits, typename Fmt<FMT>::peak_t &pk,
                                         const double *__restrict__ tr /* NX values; unused when NX == 0 */, const RgTmCoef &K,
                                         const uint32_t n, const uint32_t len) {
    double (&s)[10] = st.s[0];
    double (&t)[2] = st.t[0];
    // frames past the end were staged as zeros; the peak is tracked here or, for whole pieces, by the caller (peak4)
    const double x = PEAK ? Fmt<FMT>::cvt_word(wbits, pk) : Fmt<FMT>::word_value(wbits);
#ifdef RG_TM_LOADER_ONLY
    // Measurement build (tools/build_variant.sh loader -DRG_TM_LOADER_ONLY; never shipped, wrong results): the kernel's own
    // loads, LDS staging and reads with ONE operation per sample instead of the cascade -- the memory-side floor of this access
    // pattern, its FETCH_SIZE against the known byte count, its power without the FP64 pipe.
    st.A[0] += x;
    return;
#endif
    if constexpr (SERVO) {
        // 26 operations: the Butterworth stage is the Yule output (butter b0 folded into K.b) minus a double integrator,
        // t = (v1, v2).  Three groups of independent instructions:
        //   G1  y, u_i = b_{i+1} x + s_{i+1}, p = v1 + v2      (12; p last: v1, v2 are the previous frame's last writes)
        //   G2  z = y - v1,  s_i = u_i - a_{i+1} y               (11)
        //   G3  v2, v1, A, B_j                                    (3 + NX, need z, issued 10 slots earlier)
        const double y = fma(K.b[0], x, s[0]);
        double u[10];
#pragma unroll
        for (int i = 0; i < 9; ++i) u[i] = fma(K.b[i + 1], x, s[i + 1]);
        u[9] = K.b[10] * x;
        const double p = t[0] + t[1];
        __builtin_amdgcn_sched_barrier(0);
        double z = y - t[0];
#pragma unroll
        for (int i = 0; i < 10; ++i) s[i] = fma(-K.a[i + 1], y, u[i]);
        __builtin_amdgcn_sched_barrier(0);
        // a masked frame (past the end of the track) leaves v2 where it is: v2 / beta is the sum of the outputs so far
        if (MASK) z = n < len ? z : 0.0;
        t[1] = fma(K.beta, z, t[1]);
        t[0] = fma(K.alpha, z, p);
        st.A[0] = fma(z, z, st.A[0]);
#pragma unroll
        for (int j = 0; j < NX; ++j) st.B[0][RG_TM_DIM - NX + j] = fma(z, tr[j], st.B[0][RG_TM_DIM - NX + j]);
        __builtin_amdgcn_sched_barrier(0);
        return;
    }
    const double y = fma(K.b[0], x, s[0]);
    double u[10];
#pragma unroll
    for (int i = 0; i < 9; ++i) u[i] = fma(K.b[i + 1], x, s[i + 1]);
    u[9] = fma(K.b[10], x, K.c0);
    __builtin_amdgcn_sched_barrier(0);
    double z = fma(K.bb[0], y, t[0]);
    const double w1 = fma(K.bb[1], y, t[1]);
    const double w2 = fma(K.bb[2], y, K.c0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 10; ++i) s[i] = fma(-K.a[i + 1], y, u[i]);
    __builtin_amdgcn_sched_barrier(0);
    t[0] = fma(-K.ba[1], z, w1);
    t[1] = fma(-K.ba[2], z, w2);
    if (MASK) z = n < len ? z : 0.0;
    st.A[0] = fma(z, z, st.A[0]);
#pragma unroll
    for (int j = 0; j < NX; ++j) st.B[0][RG_TM_DIM - NX + j] = fma(z, tr[j], st.B[0][RG_TM_DIM - NX + j]);
    __builtin_amdgcn_sched_barrier(0);
}

template <int NX>
__device__ __forceinline__ void tm_load_row(double (&dst)[NX], const double *__restrict__ src) {
#pragma unroll
    for (int j = 0; j < NX; j += 2) {
        const double2 v = *reinterpret_cast<const double2 *>(src + j);  // 16-byte aligned broadcast read
        dst[j] = v.x;
        dst[j + 1] = v.y;
    }
}

// The F32 fast path of one wave.  TAIL = some row of this wave is shorter than L (the end of the
// track): short rows are staged zero-filled, element by element where a 16-byte piece would cross the
// end of the channel, and their moments are masked.
//
// Per wave, no block barrier: the global loads of tile t+1 are issued when tile t has just been
// written to the wave's LDS tile and stay in flight during tile t's arithmetic; inside a tile the
// next 4-frame piece is read from LDS while the current one computes.
//
// MOM = the transient moments B are accumulated (the first window of a segment); without it the tile loop is the
// cascade and the energy alone (windows 2..m of a multi-window segment: no table reads at all).
// A row is up to L frames: lane r of the wave owns row r, which starts at `rowp` (this lane's) and has `len` valid
// frames.  The rows of a wave need not belong to one track or channel: the loader lanes fetch the bases by shuffle.
template <int FMT, bool TAIL, bool MOM, bool SERVO>
__device__ __forceinline__ void tm_fast_path(TmLane<1> &st, typename Fmt<FMT>::peak_t &pk, const RgTmCoef &K, const uint32_t L,
                                             const uint32_t H,
                                             const __attribute__((address_space(1))) typename Fmt<FMT>::elem *rowp,
                                             const uint32_t len,
                                             const double *__restrict__ T12, const double *__restrict__ T2,
                                             char *const wtile /* RG_TM_WAVE_TILE_BYTES, twice that when !MOM */) {
    typedef Fmt<FMT> F;
    typedef const __attribute__((address_space(1))) typename F::elem gelem;
    const int lane = threadIdx.x & 63;
    // Experiment RG_TM_LINEWIDE: in the moment-free pair loop one load instruction covers 8 rows x 128
    // bytes (a pair of tiles) instead of 16 rows x 64, so that the two 64-byte halves of a 128-byte line are asked for by ONE
    // instruction; LDS slot 1 then holds its rows swapped in pairs (row r at 64 (r ^ 1)): a row's eight pieces, four to each
    // slot, fall on all 32 banks.
#ifdef RG_TM_LINEWIDE
    constexpr bool kLineWide = !MOM && !TAIL && FMT == RG_FMT_F32_PLANAR;
#else
    constexpr bool kLineWide = false;
#endif
    // loader role: instruction q covers rows 16q .. 16q+15; this lane fetches for row 16q + (lane >> 2)
    const int lrow = lane >> 2, lslot = lane & 3;
    gelem *lfirst[4];    // the piece this lane fetches in tile 0
    uint32_t llen[4];    // valid frames of that row
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 16 * q + lrow;
        const int piece = lslot ^ ((row >> 2) & 3);
        const unsigned long long base = __shfl((unsigned long long)(uintptr_t)rowp, row, 64);
        lfirst[q] = (gelem *)(uintptr_t)base + 4u * piece;
        llen[q] = L;
        if (TAIL) llen[q] = __shfl(len, row, 64);
    }
    // consumer role: row == lane
    const char *const rrow = wtile + lane * 64;
    const char *const rrow1 = wtile + RG_TM_WAVE_TILE_BYTES + (kLineWide ? lane ^ 1 : lane) * 64;
    const int rswz = (lane >> 2) & 3;
    const uint32_t ntiles = (L + RG_TM_TILE - 1) / RG_TM_TILE;

    // Staging registers: four samples of one row as 32-bit LDS words (float bits, or sign-extended integers) per load.
    // Windows without moments (MOM == false) have TWO sets and two LDS slots: the twelve moments' registers are dead there.
    // A pair of tiles goes to LDS together, the loads of the next pair are issued at once and have TWO tile periods to land,
    // not one -- a wave stalls on the slowest of its outstanding 64-byte requests, and with one period (about 3 us at three
    // waves per SIMD) the tail of the HBM latency distribution showed: the same launch with every descriptor pointing at one
    // track's PCM (Infinity Cache resident) ran 17 % faster (tools/ubench/alias_tracks.py).
    constexpr int NSTAGE = MOM ? 1 : 2;
    rg_u32x4 stage[NSTAGE][4];
    auto load4 = [&](gelem *src) -> rg_u32x4 {
        if constexpr (FMT == RG_FMT_S16_PLANAR) {
            const rg_s16x4u v = *(const __attribute__((address_space(1))) rg_s16x4u *)src;
            return rg_u32x4{(uint32_t)(int32_t)v.x, (uint32_t)(int32_t)v.y, (uint32_t)(int32_t)v.z, (uint32_t)(int32_t)v.w};
        } else {
            return *(const __attribute__((address_space(1))) rg_u32x4u *)src;
        }
    };
    auto load_tile = [&](uint32_t tile, auto buf, auto whole) {
        constexpr int SB = decltype(buf)::value;
        constexpr bool WHOLE = decltype(whole)::value;  // the tile is known to be a whole one: no guards
        const uint32_t n0 = tile * RG_TM_TILE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 16 * q + lrow;
            const int piece = lslot ^ ((row >> 2) & 3);
            const uint32_t pn = n0 + 4u * piece;  // frame index of the piece within its row
            if (!TAIL) {
                // pieces starting past the row's end are never read; only the last tile can have such pieces
                if (WHOLE || n0 + RG_TM_TILE <= L || pn < L) stage[SB][q] = load4(lfirst[q] + n0);
            } else {
                gelem *src = lfirst[q] + n0;
                if (pn + 4u <= llen[q]) {
                    stage[SB][q] = load4(src);
                } else {
                    stage[SB][q].x = pn + 0u < llen[q] ? F::word(src[0]) : 0u;
                    stage[SB][q].y = pn + 1u < llen[q] ? F::word(src[1]) : 0u;
                    stage[SB][q].z = pn + 2u < llen[q] ? F::word(src[2]) : 0u;
                    stage[SB][q].w = pn + 3u < llen[q] ? F::word(src[3]) : 0u;
                }
            }
        }
    };
    // staging set SB -> LDS slot SB of the wave
    auto store_tile = [&](auto buf) {
        constexpr int SB = decltype(buf)::value;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<uint4 *>(wtile + SB * RG_TM_WAVE_TILE_BYTES + q * 1024 + (lane ^ (kLineWide && SB ? 4 : 0)) * 16) =
                make_uint4(stage[SB][q].x, stage[SB][q].y, stage[SB][q].z, stage[SB][q].w);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto read_piece = [&](int p, int slot) -> uint4 { return *reinterpret_cast<const uint4 *>((slot ? rrow1 : rrow) + 16 * (p ^ rswz)); };

    // One tile.  MODE 1: every piece of the tile lies below H (all 12 moments live); MODE 2: every piece lies at or
    // past H (slow pair only); MODE 0: decided per piece (the one tile H falls into, and the last tile of a row);
    // MODE 3: no moments, whole tile; MODE 4: no moments, the ragged last tile.
    // Whole-tile modes keep the piece loop on a single path: with both frame bodies behind a branch inside one
    // loop the register allocator reconciles the rotated filter state with 15 v_mov_b64 per piece on one of them.
    auto compute_tile = [&](const uint32_t tile, auto mode, auto buf) {
        constexpr int MODE = decltype(mode)::value;
        constexpr int SLOT = decltype(buf)::value;
        const uint32_t n0 = tile * RG_TM_TILE;
        const int np = (MODE != 0 && MODE != 4) ? 4 : (L - n0 >= RG_TM_TILE ? 4 : (int)((L - n0) >> 2));  // full 4-frame pieces in this tile
        auto piece = [&](const int p, const uint4 v) {
            const uint32_t f[4] = {v.x, v.y, v.z, v.w};
            const uint32_t n = n0 + 4u * p;
            F::peak4(f, pk);
            if (MODE == 3 || MODE == 4) {  // cascade and energy only
#pragma unroll
                for (int u = 0; u < 4; ++u) tm_frame<FMT, 0, TAIL, false, SERVO>(st, f[u], pk, nullptr, K, n + u, len);
            } else if (MODE == 1 || (MODE == 0 && n < H)) {  // all 12 transient moments live (H is a multiple of 4, or the whole segment)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    double row[12];
                    tm_load_row<12>(row, T12 + (size_t)(n + u) * 12);
                    tm_frame<FMT, 12, TAIL, false, SERVO>(st, f[u], pk, row, K, n + u, len);
                }
            } else {  // only the slow (Butterworth) pair
                double rows[8];
                tm_load_row<4>(reinterpret_cast<double (&)[4]>(rows[0]), T2 + (size_t)(n - H) * 2);
                tm_load_row<4>(reinterpret_cast<double (&)[4]>(rows[4]), T2 + (size_t)(n - H) * 2 + 4);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double tr[2] = {rows[2 * u], rows[2 * u + 1]};
                    tm_frame<FMT, 2, TAIL, false, SERVO>(st, f[u], pk, tr, K, n + u, len);
                }
            }
        };
        if constexpr (MODE != 0 && MODE != 4) {
            // whole-tile modes: the four pieces are unrolled (no loop counter, no copy of the prefetched piece)
            uint4 v[4];
            v[0] = read_piece(0, SLOT);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (p + 1 < 4) v[p + 1] = read_piece(p + 1, SLOT);
                piece(p, v[p]);
            }
        } else {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (np) v = read_piece(0, SLOT);
#pragma unroll 1
            for (int p = 0; p < np; ++p) {
                const uint4 vn = p + 1 < np ? read_piece(p + 1, SLOT) : v;
                piece(p, v);
                v = vn;
            }
        }
        if ((MODE == 0 || MODE == 4) && tile + 1 == ntiles) {
            // L & 3 trailing frames, in this (the last) tile
            for (uint32_t n = L & ~3u; n < L; ++n) {
                const uint32_t o = n - n0;
                const uint32_t f = *reinterpret_cast<const uint32_t *>((SLOT ? rrow1 : rrow) + 16 * ((int)(o >> 2) ^ rswz) + 4 * (o & 3));
                if (MODE == 4) {
                    tm_frame<FMT, 0, TAIL, true, SERVO>(st, f, pk, nullptr, K, n, len);
                } else if (n < H) {
                    double row[12];
                    tm_load_row<12>(row, T12 + (size_t)n * 12);
                    tm_frame<FMT, 12, TAIL, true, SERVO>(st, f, pk, row, K, n, len);
                } else {
                    const double tr[2] = {T2[(size_t)(n - H) * 2], T2[(size_t)(n - H) * 2 + 1]};
                    tm_frame<FMT, 2, TAIL, true, SERVO>(st, f, pk, tr, K, n, len);
                }
            }
        }
    };
    typedef std::integral_constant<int, 0> Mixed;
    typedef std::integral_constant<int, 1> All12;
    typedef std::integral_constant<int, 2> All2;
    typedef std::integral_constant<int, 3> None;
    typedef std::integral_constant<int, 4> NoneRagged;

    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, NSTAGE - 1> S1;
    typedef std::false_type Guarded;
    typedef std::true_type Whole;
    // one tile: LDS <- its staging set (every read of what the slot held is behind us), the next load into that set, the arithmetic
    auto run_tile = [&](const uint32_t tile, auto mode, auto buf) {
        store_tile(buf);
        if (tile + NSTAGE < ntiles) load_tile(tile + NSTAGE, buf, Guarded{});  // in flight during the next NSTAGE tiles' arithmetic
        compute_tile(tile, mode, buf);
    };
    const uint32_t full_tiles = L / RG_TM_TILE;                         // tiles with four whole pieces
    uint32_t tile = 0;
    if constexpr (MOM) load_tile(0, S0{}, Guarded{});
    if constexpr (!MOM) {
        if constexpr (kLineWide) {
            // instruction q, lane j: row 8 q + (j >> 3), piece j & 7 of the pair's 32 frames (pieces 0-3 = first tile -> slot 0,
            // 4-7 = second tile -> slot 1, swizzled as always).  The rows' addresses are 32-bit offsets from the lowest row of the
            // wave (one scalar base: eight registers, as many as the plain loader's four pointers); a wave whose rows lie more
            // than 2 GB apart (several tracks of a huge arena) takes the plain loader.
            const int wrow = lane >> 3, wslot = (lane >> 2) & 1, wpos = lane & 3;
            unsigned long long lo64 = (unsigned long long)(uintptr_t)rowp;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const unsigned long long o = __shfl_xor(lo64, d, 64);
                lo64 = o < lo64 ? o : lo64;
            }
            const unsigned long long wbase = __builtin_amdgcn_readfirstlane((uint32_t)lo64) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(lo64 >> 32)) << 32);
            const unsigned long long delta = (unsigned long long)(uintptr_t)rowp - wbase;
            const bool near = __all(delta < (1ull << 31)) && full_tiles >= 4;
            if (near) {
                uint32_t woff[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int row = 8 * q + wrow;
                    woff[q] = __shfl((uint32_t)delta, row, 64) + 64u * wslot + 16u * (wpos ^ ((row >> 2) & 3));
                }
                const __attribute__((address_space(1))) char *const gb = (const __attribute__((address_space(1))) char *)(uintptr_t)wbase;
                auto pair_load = [&](const uint32_t t0) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        stage[q >> 2][q & 3] = *(const __attribute__((address_space(1))) rg_u32x4u *)(gb + (size_t)t0 * (RG_TM_TILE * 4) + woff[q]);
                };
                auto pair_store = [&]() {
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        *reinterpret_cast<uint4 *>(wtile + wslot * RG_TM_WAVE_TILE_BYTES + ((8 * q + wrow) ^ wslot) * 64 + wpos * 16) =
                            make_uint4(stage[q >> 2][q & 3].x, stage[q >> 2][q & 3].y, stage[q >> 2][q & 3].z, stage[q >> 2][q & 3].w);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                };
                pair_load(0);
                for (; tile + 3 < full_tiles; tile += 2) {
                    pair_store();
                    if (tile + 5 < full_tiles) {
                        pair_load(tile + 2);
                    } else {  // the tiles behind the last pair go the plain way
                        load_tile(tile + 2, S0{}, Guarded{});
                        if (tile + 3 < ntiles) load_tile(tile + 3, S1{}, Guarded{});
                    }
                    compute_tile(tile, None{}, S0{});
                    compute_tile(tile + 1, None{}, S1{});
                }
            } else {
                load_tile(0, S0{}, Guarded{});
                if (ntiles > 1) load_tile(1, S1{}, Guarded{});
                for (; tile + 3 < full_tiles; tile += 2) {
                    store_tile(S0{});
                    store_tile(S1{});
                    load_tile(tile + 2, S0{}, Whole{});
                    load_tile(tile + 3, S1{}, Whole{});
                    compute_tile(tile, None{}, S0{});
                    compute_tile(tile + 1, None{}, S1{});
                }
            }
        } else {
        load_tile(0, S0{}, Guarded{});
        if (ntiles > 1) load_tile(1, S1{}, Guarded{});
        // pairs of whole tiles (the staging set is a compile-time choice: registers cannot be indexed): both to LDS, the
        // next pair's eight loads issued together, two tiles of arithmetic under them
        for (; tile + 3 < full_tiles; tile += 2) {
            store_tile(S0{});
            store_tile(S1{});
            load_tile(tile + 2, S0{}, Whole{});
            load_tile(tile + 3, S1{}, Whole{});
            compute_tile(tile, None{}, S0{});
            compute_tile(tile + 1, None{}, S1{});
        }
        }
        // what is left, tile by tile: up to three whole tiles and the ragged last one; `tile` is even at every turn
        while (tile < ntiles) {
            if (tile < full_tiles) run_tile(tile, None{}, S0{}); else run_tile(tile, NoneRagged{}, S0{});
            if (++tile >= ntiles) break;
            if (tile < full_tiles) run_tile(tile, None{}, S1{}); else run_tile(tile, NoneRagged{}, S1{});
            ++tile;
        }
        return;
    }
    const uint32_t t12 = (H / RG_TM_TILE) < full_tiles ? H / RG_TM_TILE : full_tiles;  // tiles entirely below H
    for (; tile < t12; ++tile) run_tile(tile, All12{}, S0{});
    if (tile < ntiles && (tile * RG_TM_TILE < H || tile >= full_tiles)) { run_tile(tile, Mixed{}, S0{}); ++tile; }  // the tile H falls into
    for (; tile < full_tiles; ++tile) run_tile(tile, All2{}, S0{});
    for (; tile < ntiles; ++tile) run_tile(tile, Mixed{}, S0{});              // the ragged last tile
}

// MULTI = multi-window segments (G.m > 1); a separate instantiation, so that the one-window kernel's register
// allocation is not disturbed by the window loop
template <int FMT, bool MULTI, bool SERVO>
__global__ void __launch_bounds__(MULTI ? RG_TM_BLOCK_WIDE_MULTI : RG_TM_BLOCK_WIDE)
rg_tm_main_kernel(const RgTmCoef K, const RgTmGeom G, const RgTmTrack *__restrict__ tracks, uint32_t n_tracks,
                  double *__restrict__ rec, uint32_t total_recs, double *__restrict__ win_energy /* [channels][total_windows], m > 1 */,
                  uint32_t total_windows, uint32_t *__restrict__ nonfinite, uint32_t lds_tables,
                  uint32_t *__restrict__ zero_words, uint64_t zero_count /* batch accumulators to clear, or nullptr */,
                  unsigned long long *__restrict__ dbg /* nullptr, or 6 words per wave: start, end, hw id, path, cycle counter start, end */) {
    typedef Fmt<FMT> F;
    const unsigned long long dbg_t0 = dbg ? wall_clock64() : 0ull;
    const unsigned long long dbg_c0 = dbg ? __builtin_readcyclecounter() : 0ull;
    typedef typename F::elem elem;
    typedef __attribute__((address_space(1))) const elem gelem;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // the batch's histograms, peaks and arrival counters are cleared here instead of by a memset of
    // their own: nothing in this kernel reads them, and everything that does is behind it in the stream
    if (zero_words) {
        const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
        for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < zero_count; w += stride) zero_words[w] = 0u;
    }
    // Lanes are numbered through the whole launch group: track t owns lanes [lane_base, lane_base + channels * nseg),
    // channel 0's segments first.  Blocks and waves are not padded per track (a 3-minute track in 5-window segments is
    // 720 lanes: padded to whole 768-lane blocks per channel that was 6 % idle lanes, at other lengths up to 60 %).
    const uint32_t gl = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = find_track(tracks, n_tracks, gl, &RgTmTrack::lane_base);
    // Per-lane now (a wave can straddle tracks), so only what the frame loops need stays in registers: the channel
    // base, the track length and the segment's place; everything else is read again from tracks[t] where it is used.
    struct { const void *ch0, *ch1; uint64_t frames; uint32_t nseg; } tr;
    {
        const RgTmTrack &q = tracks[t];
        tr.ch0 = q.ch0; tr.ch1 = q.ch1; tr.frames = q.frames; tr.nseg = q.nseg;
    }
    const uint32_t rel = gl - tracks[t].lane_base;
    const int chan = (tr.ch1 != nullptr && rel >= tr.nseg) ? 1 : 0;
    const uint32_t seg = rel - (chan ? tr.nseg : 0u);
    const uint32_t L = G.L;
    const uint32_t H = G.H10;
    const uint32_t m = MULTI ? G.m : 1u;            // windows per segment (L == W when m > 1)
    const uint64_t seg_stride = (uint64_t)L * m;
    const bool active = seg < tr.nseg;  // false only for the lanes past the group's last track
    const uint64_t start = (uint64_t)seg * seg_stride;
    uint32_t len = 0;                               // valid frames of the first window (the moment zone)
    if (active) {
        const uint64_t rem = tr.frames - start;
        len = rem < L ? (uint32_t)rem : L;
    }
    gelem *const chp = (gelem *)(chan == 0 ? tr.ch0 : tr.ch1);
    rg_cdouble *__restrict__ T = (rg_cdouble *)G.T;

    TmLane<1> st;
#pragma unroll
    for (int i = 0; i < 10; ++i) st.s[0][i] = 0.0;
    st.t[0][0] = st.t[0][1] = 0.0;
    st.A[0] = 0.0;
#pragma unroll
    for (int j = 0; j < RG_TM_DIM; ++j) st.B[0][j] = 0.0;
    typename F::peak_t pk = 0;
    // segment record, structure-of-arrays: field f of channel c at ((c*RG_TM_REC + f) * total_recs + idx)
    // (the address is formed where it is used: a pointer held across the frame loops costs two VGPRs there)
    auto rec_ptr = [&]() -> double * { return rec + (size_t)chan * RG_TM_REC * total_recs + ((size_t)tracks[t].rec_base + (active ? seg : 0)); };
    // A sample that is not finite (NaN / Inf in float PCM) leaves the reference's filter state NaN for the rest of
    // the track (src/replaygain.rs:586-616 has no reset): remember the first unit (segment, or window of a multi-window
    // segment) it happens in; every window from there on becomes a NaN window (bin 2000, as `NaN as i32` = 0 does)
    auto note_nonfinite = [&](const double energy, const uint32_t unit, const bool counts) {
        const bool bad = counts && !(fabs(energy) <= 1.7976931348623157e308);
        if (__any(bad) && bad) atomicMax(&nonfinite[tracks[t].track_index], 0xFFFFFFFFu - unit);
    };

    // servo: the lanes are linear and every true output is the linear one + d_inf (rg_tm.h).  The part of that which does not
    // depend on the segment's start state goes into the stored energy right here, while v2 still is the sum of beta z over
    // the window the moments cover: 2 d_inf v2 / beta + len d_inf^2.  (The fix-up kernel adds the start state's share.)
    auto first_energy = [&]() -> double {
        // (an energy that is not finite keeps its class: an Inf sample in the last frame leaves +Inf here and -Inf in v2)
        if constexpr (SERVO) return fabs(st.A[0]) <= 1.7976931348623157e308 ? st.A[0] + fma(K.aff_lin, st.t[0][1], K.aff_n * (double)len) : st.A[0];
        return st.A[0];
    };
    bool done = false;
    {
        if (lds_tables) {
            done = true;
            // ---- block-shared tables, one packed image: T12[n][12] for n < H, then T2[n - H][2] -------------
            const uint32_t tbl_doubles = H * 12 + (L - H) * 2;  // even
            {
                const double2 *__restrict__ src = reinterpret_cast<const double2 *>(G.Tlds);
                double2 *dst = reinterpret_cast<double2 *>(smem);
                const uint32_t n16 = tbl_doubles / 2;
                const uint32_t BS = blockDim.x;
                uint32_t i = threadIdx.x;
                for (; i + 3 * BS < n16; i += 4 * BS) {  // four loads in flight per thread
                    const double2 a = src[i], b = src[i + BS], c = src[i + 2 * BS], d = src[i + 3 * BS];
                    dst[i] = a;
                    dst[i + BS] = b;
                    dst[i + 2 * BS] = c;
                    dst[i + 3 * BS] = d;
                }
                for (; i < n16; i += BS) dst[i] = src[i];
                __syncthreads();
            }
            const double *const T12 = reinterpret_cast<const double *>(smem);
            const double *const T2 = T12 + (size_t)H * 12;
            char *const wtile = smem + (size_t)tbl_doubles * sizeof(double) + (threadIdx.x >> 6) * (RG_TM_WAVE_TILE_BYTES * (MULTI ? 2 : 1));
            // a 16-byte piece may reach up to 3 frames past its row: full rows that end exactly at the end
            // of the channel go through the element-wise staging of the TAIL variant too
            const bool plain = len == L && start + ((L + 3u) & ~3u) <= tr.frames;
            if (__all(plain))
                tm_fast_path<FMT, false, true, SERVO>(st, pk, K, L, H, chp + start, len, T12, T2, wtile);
            else if (__any(len != 0))
                tm_fast_path<FMT, true, true, SERVO>(st, pk, K, L, H, chp + start, len, T12, T2, wtile);
            if constexpr (MULTI) {
                // ---- windows 2..m of the segment: the start state's transient is gone (|Phi| < 1e-15 per window,
                // rg_design.cpp), what is left is the running cascade and one energy per window.  The moments of the
                // first window are final: they leave the registers now.
                note_nonfinite(st.A[0], seg * m, active);
                if (active) {
                    double *__restrict__ const r = rec_ptr();
                    r[0] = first_energy();
#pragma unroll
                    for (int j = 0; j < RG_TM_DIM; ++j) r[(size_t)(1 + j) * total_recs] = st.B[0][j];
                }
#pragma unroll 1
                for (uint32_t w = 1; w < m; ++w) {
                    const uint64_t wstart = start + (uint64_t)w * L;
                    uint32_t lenw = 0;
                    if (active && wstart < tr.frames) {
                        const uint64_t rem = tr.frames - wstart;
                        lenw = rem < L ? (uint32_t)rem : L;
                    }
                    if (!__any(lenw != 0)) break;  // the track ended in an earlier window for every row of this wave
                    st.A[0] = 0.0;
                    const double v2_start = st.t[0][1];
                    // A row whose track ended in an earlier window of this lane must not send the whole wave down the masked
                    // path (with 98 segments of 37 windows per three-minute track two waves in three hold such a row, for 26
                    // of their 37 windows): it re-reads its track's FIRST window instead -- same track, same channel, so the
                    // peak does not change; its energy is not stored and its state is never used again.
                    const bool idle = lenw == 0 && tr.frames >= (uint64_t)((L + 3u) & ~3u);
                    const bool plainw = idle || (lenw == L && wstart + ((L + 3u) & ~3u) <= tr.frames);
                    gelem *const rowp = idle ? chp : chp + wstart;
                    if (__all(plainw))
                        tm_fast_path<FMT, false, false, SERVO>(st, pk, K, L, H, rowp, lenw, T12, T2, wtile);
                    else
                        tm_fast_path<FMT, true, false, SERVO>(st, pk, K, L, H, rowp, lenw, T12, T2, wtile);
                    note_nonfinite(st.A[0], seg * m + w, lenw != 0);
                    // servo: the lanes are linear; the reference's constant offsets add d_inf to every output, and
                    // sum (z + d_inf)^2 = sum z^2 + 2 d_inf (v2_end - v2_start) / beta + n d_inf^2 (rg_tm.h)
                    double energy = st.A[0];
                    if constexpr (SERVO) energy = fabs(energy) <= 1.7976931348623157e308 ? energy + fma(K.aff_lin, st.t[0][1] - v2_start, K.aff_n * (double)lenw) : energy;
                    if (lenw != 0) win_energy[(size_t)chan * total_windows + tracks[t].win_base + (size_t)seg * m + w] = energy;
                }
            }
        }
    }
    if (!done) {
        // ---- generic path: tables too large for LDS (m == 1 only; rg_enqueue.hip never picks multi-window segments
        // then).  Frames past `len` are fed as zeros and their output is excluded from the moments;
        // the end state of such a lane is never used (the track ends inside it).
        gelem *const p0 = chp + start;
        for (uint32_t n = 0; n < L; ++n) {
            const bool valid = n < len;
            double x = 0.0;
            if (valid) x = F::cvt(p0[n], pk);
            rg_cdouble *__restrict__ Trow = T + (size_t)n * RG_TM_DIM;
            double z;
            if constexpr (SERVO) {
                // a frame past the end must leave v2 alone (v2 / beta = sum of the outputs): the state after the end is never used
                const double v2 = st.t[0][1];
                z = tm_step_servo(st.s[0], st.t[0], x, K);
                if (!valid) st.t[0][1] = v2;
            } else {
                z = tm_step(st.s[0], st.t[0], x, K);
            }
            z = valid ? z : 0.0;
            st.A[0] = fma(z, z, st.A[0]);
#pragma unroll
            for (int j = 0; j < RG_TM_DIM; ++j) st.B[0][j] = fma(z, Trow[j], st.B[0][j]);
        }
    }

    if (dbg && (threadIdx.x & 63) == 0) {
        const size_t w = ((size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * 6;
        dbg[w + 4] = dbg_c0;
        dbg[w + 5] = __builtin_readcyclecounter();
        dbg[w + 0] = dbg_t0;
        dbg[w + 1] = wall_clock64();
        dbg[w + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |  // XCC_ID
                     __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));                                  // HW_ID
        dbg[w + 3] = done ? 1ull : 0ull;
    }

    if constexpr (!MULTI) {
        note_nonfinite(st.A[0], seg, active);
        if (active) {
            double *__restrict__ const r = rec_ptr();
            r[0] = first_energy();
#pragma unroll
            for (int j = 0; j < RG_TM_DIM; ++j) r[(size_t)(1 + j) * total_recs] = st.B[0][j];
        }
    }
    if (active) {
        double *__restrict__ const r = rec_ptr();
#pragma unroll
        for (int j = 0; j < 10; ++j) r[(size_t)(13 + j) * total_recs] = st.s[0][j];
        r[(size_t)23 * total_recs] = st.t[0][0];
        r[(size_t)24 * total_recs] = st.t[0][1];
        r[(size_t)25 * total_recs] = F::peak_norm(pk);  // max |x| of this segment, normalised (replaygain.rs:967)
    }
}

// =================================================================================================
// Fix-up kernel.  One lane = one segment (all channels).  Lanes [0, warm) of a block repeat the last
// `warm` segments of the previous block so that every owner lane finds its 2^R predecessors in LDS.

// Device-scope atomics are performed at the memory side; waiting for the returned value is waiting for that.
template <typename V>
__device__ __forceinline__ void tm_performed(V old) {
    asm volatile("" ::"v"(old));
}

// sigma' G sigma for the packed upper triangle G (row major), evaluated as sum_j s_j (G_jj s_j / 2 + sum_{q>j} G_jq s_q) * 2
template <typename GP>
__device__ __forceinline__ double tm_quad_half(GP Gm, const double (&sg)[RG_TM_DIM]) {
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
    return quad;
}

template <int NCH>
__global__ void __launch_bounds__(RG_TM_BLOCK)
rg_tm_fix_kernel(const RgTmGeom G, const RgTmFixTables FT, const RgTmTrack *__restrict__ tracks, uint32_t n_tracks,
                 const double *__restrict__ rec, uint32_t total_recs,
                 const double *__restrict__ win_energy /* [NCH][total_windows], m > 1 */, uint32_t total_windows,
                 uint32_t *__restrict__ nonfinite,
                 uint32_t *__restrict__ imprecise, uint32_t *__restrict__ hist,
                 unsigned long long *__restrict__ peak_bits, uint32_t *__restrict__ done_count,
                 rg_track_result *__restrict__ results,
                 unsigned long long *__restrict__ dbg /* nullptr, or 8 stage timestamps per block */) {
#define TM_FIX_STAMP(k)                                                            \
    do {                                                                           \
        if (dbg && threadIdx.x == 0) dbg[(size_t)blockIdx.x * 8 + (k)] = wall_clock64(); \
    } while (0)
    // A fix-up block that finds room beside resident main-kernel waves is short, latency-bound work on which its stream's next
    // main kernel waits: it goes first wherever it shares a SIMD (configs[1]: 68.3 -> 67.1 us per step; configs[2] unchanged).
    __builtin_amdgcn_s_setprio(3);
    TM_FIX_STAMP(0);
    // LDS (about 17 KiB, so that fix-up blocks fit next to resident main-kernel blocks of other pipeline
    // slots): lanes exchange scan values with wave shuffles; only the last RG_TM_EDGE lanes of each wave go
    // through LDS for the lanes of the next wave
    __shared__ double edge[RG_TM_BLOCK / 64][RG_TM_EDGE][NCH * RG_TM_DIM];
    __shared__ uint64_t pct_scan[RG_PCT_THREADS];
    __shared__ double pieces[RG_TM_BLOCK];
    __shared__ double pieces_m[RG_TM_BLOCK];  // A + sigma'G sigma of the same segments: what the sum was assembled from
    __shared__ double pieces_e[RG_TM_BLOCK];  // bound on what the cut of the fast moments at H10 left out of the segment's sum
    __shared__ int is_last;
    // the (at most one) segment of this block that the track ends in: its start state and length, for the
    // cooperative evaluation of its quadratic term below
    __shared__ double part_sg[NCH][RG_TM_DIM];
    __shared__ uint32_t part_len;
    __shared__ int part_lane;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) part_len = 0;
    // value of `x` held by the lane d (<= RG_TM_EDGE) positions below in the block, 0.0 before the block's first lane;
    // `slot` = index of x among the values exchanged in this step (all lanes call with the same sequence)
    auto publish_edge = [&](const double x, const int slot) {
        if (lane >= 64 - RG_TM_EDGE) edge[wave][lane - (64 - RG_TM_EDGE)][slot] = x;
    };
    auto from_below = [&](const double x, const int d, const int slot) -> double {
        double v = __shfl_up(x, d, 64);
        if (lane < d) v = wave > 0 ? edge[wave - 1][RG_TM_EDGE - d + lane][slot] : 0.0;
        return v;
    };

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
    // ---- the wave-uniform tables, one contiguous image in the design blob (rg_enqueue.hip: the last prefix
    // Gram matrix is followed by PhiY, PhiB, X, sigma0), copied to LDS under the record loads: as scalar
    // loads they missed the constant cache in every wave and sat on the critical path of each scan round
    __shared__ double ftab[RG_TM_GRAM + RG_TM_MAX_ROUNDS * 104 + 36 + 100 + 12];
    {
        const double *__restrict__ src = FT.Gp + (size_t)(G.L - 1) * RG_TM_GRAM;
        const int n = RG_TM_GRAM + (int)G.rounds * 104 + 36 + (G.servo ? 112 : (G.whiten ? 100 : 0));
        for (int q = i; q < n; q += RG_TM_BLOCK) ftab[q] = src[q];
    }
    const double *const Gfull = ftab;
    const double *const PhiY = ftab + RG_TM_GRAM;
    const double *const PhiB = PhiY + G.rounds * 100;
    const double *const X = PhiB + G.rounds * 4;
    const double *const S0 = X + 24;
    const double *const Wf = S0 + 12;
    const double *const STf = Wf + 100;  // servo: sum over a full segment of the responses (affine cross term)

    // ---- zero-state end states of this segment, all channels (the moments are fetched where they are used:
    // the kernel is latency bound and lives on occupancy, so the live register set is kept small) ----------
    double w[NCH][RG_TM_DIM], pk = 0.0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const double *__restrict__ r = rec + (size_t)c * RG_TM_REC * total_recs + idx;
#pragma unroll
        for (int j = 0; j < RG_TM_DIM; ++j) w[c][j] = seg_valid ? r[(size_t)(13 + j) * total_recs] : 0.0;
    }
    // zero-state end state in block-diagonal coordinates: t' = t + X s; virtual segment -1 carries the
    // track-start state
    __syncthreads();  // ftab
    TM_FIX_STAMP(1);
    const bool vstart = in_block && seg == -1;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        double tq[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            // below 64 kHz Xs = [X | I]: t' = t + X s, summed as it always was; above, the pair's own 2 x 2 comes with it
            double acc = G.whiten ? fma(X[q * 12 + 10], w[c][10], X[q * 12 + 11] * w[c][11]) : w[c][10 + q];
#pragma unroll
            for (int j = 0; j < 10; ++j) acc = fma(X[q * 12 + j], w[c][j], acc);
            tq[q] = acc;
        }
        w[c][10] = tq[0];
        w[c][11] = tq[1];
        if (G.whiten) {  // the fast block into the coordinates it is carried in: Wf is upper triangular, row i needs w[i..9]
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < 10; ++j)
                    if (j >= i) acc = fma(Wf[i * 10 + j], w[c][j], acc);
                w[c][i] = acc;
            }
        }
        if (vstart) {
#pragma unroll
            for (int j = 0; j < RG_TM_DIM; ++j) w[c][j] = S0[j];
        }
    }
    // ---- doubling scan: after round r, w_k = sum_{q < 2^(r+1)} Phi^q e_{k-q} ------------------------------
    for (uint32_t rd = 0; rd < G.rounds; ++rd) {
        const int d = 1 << rd;
        const bool fast = rd < G.rounds_fast;  // the fast (Yule) block has usually decayed within one segment
        double wn[NCH][RG_TM_DIM];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int j = 0; j < RG_TM_DIM; ++j)
                if (j >= 10 || fast) publish_edge(w[c][j], c * RG_TM_DIM + j);
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int j = 0; j < RG_TM_DIM; ++j) wn[c][j] = (j >= 10 || fast) ? from_below(w[c][j], d, c * RG_TM_DIM + j) : 0.0;
        __syncthreads();
        if (fast) {
            const double *PY = PhiY + (size_t)rd * 100;
#pragma unroll
            for (int a = 0; a < 10; ++a) {
                double py[10];
#pragma unroll
                for (int q = 0; q < 10; ++q) py[q] = PY[a * 10 + q];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    double acc = w[c][a];
#pragma unroll
                    for (int q = 0; q < 10; ++q) acc = fma(py[q], wn[c][q], acc);
                    w[c][a] = acc;
                }
                if (a & 1) __builtin_amdgcn_sched_barrier(0);  // two rows of the table in flight, not all ten
            }
        }
        const double *PB = PhiB + (size_t)rd * 4;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            w[c][10] = fma(PB[0], wn[c][10], fma(PB[1], wn[c][11], w[c][10]));
            w[c][11] = fma(PB[2], wn[c][10], fma(PB[3], wn[c][11], w[c][11]));
        }
    }
    // the true start state of segment k is the scanned end state of segment k-1
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < RG_TM_DIM; ++j) publish_edge(w[c][j], c * RG_TM_DIM + j);
    __syncthreads();
    double sgm[NCH][RG_TM_DIM];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < RG_TM_DIM; ++j) sgm[c][j] = from_below(w[c][j], 1, c * RG_TM_DIM + j);

    TM_FIX_STAMP(2);
    double S = 0.0, Mseg = 0.0, Eseg = 0.0;
    if (owner) {
        const uint64_t start = (uint64_t)seg * G.L * G.m;  // the moments cover the segment's first window (all of it when m == 1)
        const uint64_t rem = tr.frames - start;
        const uint32_t len = rem < G.L ? (uint32_t)rem : G.L;
        const bool full = len == G.L;
        double lin[NCH], quad[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const double *__restrict__ r = rec + (size_t)c * RG_TM_REC * total_recs + idx;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < RG_TM_DIM; ++j) acc = fma(r[(size_t)(1 + j) * total_recs], sgm[c][j], acc);
            lin[c] = acc;
            quad[c] = 0.0;
            pk = fmax(pk, r[(size_t)25 * total_recs]);
            S += r[0];
            Mseg += r[0];
            // the fast moments stop at H10: |sum_{n >= H10} z T_j| <= sqrt(sum z^2) tau_j (Cauchy-Schwarz; r[0] is the zero-state
            // energy, in servo form with its tiny affine terms), times 2 |sigma_j| in the window's sum
            double st = 0.0;
#pragma unroll
            for (int j = 0; j < 10; ++j) st = fma(fabs(sgm[c][j]), G.tau10[j], st);
            Eseg = fma(2.0 * st, sqrt(fabs(r[0])) * 1.0001, Eseg);
        }
        // full segments share one Gram matrix (LDS broadcast reads, one row at a time for all channels: the
        // scheduling fences keep the compiler from hoisting all 78 reads into registers).  The segment a
        // track ends in needs the prefix matrix of its own length: 78 per-lane loads that would cost every
        // lane of the kernel 150 VGPRs, so that one lane parks its state in LDS and wave 0 evaluates it below.
        if (full) {
            // constant trip counts on both loops (the triangle is a compile-time predicate), so that the
            // unroller resolves every index before the register promotion of sgm runs
#pragma unroll
            for (int j = 0; j < RG_TM_DIM; ++j) {
                const int p = j * RG_TM_DIM - (j * (j - 1)) / 2 - j;  // packed index of G[j][q] is p + q
                double g[RG_TM_DIM];
#pragma unroll
                for (int q = 0; q < RG_TM_DIM; ++q)
                    if (q >= j) g[q] = Gfull[p + q];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    double row = 0.5 * g[j] * sgm[c][j];
#pragma unroll
                    for (int q = 0; q < RG_TM_DIM; ++q)
                        if (q > j) row = fma(g[q], sgm[c][q], row);
                    quad[c] = fma(row, sgm[c][j], quad[c]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int j = 0; j < RG_TM_DIM; ++j) part_sg[c][j] = sgm[c][j];
            part_len = len;
            part_lane = i;
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            S += 2.0 * (lin[c] + quad[c]);
            Mseg += 2.0 * quad[c];
        }
        // servo: every true output is the linear one + d_inf (rg_tm.h).  The main kernel has put 2 d_inf sum zs + len d_inf^2
        // into A; the start state's share 2 d_inf sigma . ST (ST = sum of the responses over the segment) is added here (a
        // segment the track ends in gets it with its own prefix sum from wave 0 below)
        if (G.servo && full) {
            double aff = 0.0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                double sx = 0.0;
#pragma unroll
                for (int j = 0; j < RG_TM_DIM; ++j) sx = fma(STf[j], sgm[c][j], sx);
                aff = fma(G.aff_sig, sx, aff);
            }
            S += aff;
        }
    }
    // windows after the first non-finite unit of the track are NaN windows (see the main kernel).  The unit itself has
    // the class of its zero-state energy (the true output differs from the zero-state one by finite terms): NaN, or
    // +Inf when an Inf sample is the unit's very last frame and nothing after it has turned into NaN yet -- if that
    // is also the window's last frame the reference's sum is +Inf and the window is DROPPED (`val as i32` saturates,
    // the index wraps out of range, src/replaygain.rs:749-759), not counted in bin 2000.  A + 2 B.sigma + ... itself
    // would be Inf - Inf there.
    const uint32_t nf = nonfinite[tr.track_index];
    if (nf != 0 && owner) {
        const uint64_t unit = (uint64_t)seg * G.m, bad = 0xFFFFFFFFu - nf;
        if (unit > bad) S = __longlong_as_double(0x7FF8000000000000ll);
        else if (unit == bad) {
            double cls = 0.0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const double a = rec[(size_t)c * RG_TM_REC * total_recs + idx];
                if (!(fabs(a) <= 1.7976931348623157e308)) cls += a;  // +Inf stays +Inf, anything + NaN is NaN
            }
            S = cls;
        }
    }
    pieces[i] = owner ? S : 0.0;
    pieces_m[i] = owner ? Mseg : 0.0;
    pieces_e[i] = owner ? Eseg : 0.0;
    __syncthreads();
    if (wave == 0 && part_len != 0) {
        // term p of the packed upper triangle is G[p] s_j s_q (halved on the diagonal); two terms per lane
        const double *__restrict__ Gm = FT.Gp + (size_t)(part_len - 1) * RG_TM_GRAM;
        double quad = 0.0, affp = 0.0;
        if (G.servo && lane < RG_TM_DIM) {  // the affine cross term with the prefix sum of this length
            const double sj = FT.ST[(size_t)(part_len - 1) * RG_TM_DIM + lane];
#pragma unroll
            for (int c = 0; c < NCH; ++c) affp = fma(G.aff_sig * sj, part_sg[c][lane], affp);
        }
        for (int p = lane; p < RG_TM_GRAM; p += 64) {
            int j = 0, base = 0;
            while (p >= base + RG_TM_DIM - j) { base += RG_TM_DIM - j; ++j; }
            const int q = j + (p - base);
            const double g = Gm[p] * (q == j ? 0.5 : 1.0);
#pragma unroll
            for (int c = 0; c < NCH; ++c) quad = fma(g * part_sg[c][j], part_sg[c][q], quad);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            quad += __shfl_xor(quad, off, 64);
            affp += __shfl_xor(affp, off, 64);
        }
        if (lane == 0) {
            pieces[part_lane] += 2.0 * quad + affp;
            pieces_m[part_lane] += 2.0 * quad;
        }
    }
    // peak of the block's segments: wave max, then one atomic per wave (the bit pattern of a
    // non-negative double is ordered like the value)
    {
        unsigned long long pb = (unsigned long long)__double_as_longlong(pk);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(pb, off, 64);
            pb = o > pb ? o : pb;
        }
        if ((i & 63) == 0 && pb != 0) tm_performed(atomicMax(&peak_bits[tr.track_index], pb));
    }
    __syncthreads();

    TM_FIX_STAMP(3);
    // ---- 50 ms windows: k consecutive segments each (finish_window, src/replaygain.rs:743-765) ----
    int bin = -1;
    bool cancelled = false;
    if ((uint32_t)i < G.fix_windows) {
        const uint64_t widx = ((uint64_t)b * G.fix_windows + i) * G.m;  // m > 1: k == 1, the segment's first window
        if (widx < tr.n_windows) {
            double total = 0.0, mtot = 0.0, etot = 0.0;
            for (uint32_t q = 0; q < G.k; ++q) {
                total += pieces[warm + i * G.k + q];
                mtot += pieces_m[warm + i * G.k + q];
                etot += pieces_e[warm + i * G.k + q];
            }
            if (NCH == 1) { total *= 2.0; etot *= 2.0; }  // add_mono_sample feeds both sums (src/replaygain.rs:731-740)
            // a sum of squares is never negative; A + 2 B.sigma + sigma'G sigma of a window whose true energy is
            // far below the energy of the filter state (the high-passed tail of a DC offset, say) can come out a
            // rounding error below zero, and log10 of that would be a NaN window.  (A NaN stays a NaN.)
            if (total < 0.0) total = 0.0;
            const uint64_t rem = tr.frames - widx * G.W;
            const uint32_t n = rem < G.W ? (uint32_t)rem : G.W;
            bin = rg_window_bin(total, 0.0, n);
            // could rounding have put this window into another bin?  (NaN compares false: a NaN window is exact)
            // ... or the tail of the fast moments behind H10?  (etot is far below the rounding of `total` at the design's default
            // cut, 1e-13: the second test is then never taken)
            if (mtot > 1.0e3 * total || etot > 1.0e-13 * total) {
                const double e = (mtot > 1.0e3 * total ? RG_TM_CEPS * mtot : 0.0) + etot;
                const double lo = total - e;
                cancelled = rg_window_bin(lo < 0.0 ? 0.0 : lo, 0.0, n) != rg_window_bin(total + e, 0.0, n);
            }
        }
    }
    if (__any(cancelled) && lane == 0) tm_performed(atomicOr(&imprecise[tr.track_index], 1u));
    // ---- one count per distinct bin of a wave (neighbouring windows mostly share a few bins, and same-address atomics
    // serialise at the memory side): ballots instead of the quadratic search through LDS that this stage used to be (7 of the
    // block's 22 us on a 10-minute track); no return value -- every wave waits once for its own before the arrival barrier
    {
        bool pending = bin >= 0;
        while (true) {
            const unsigned long long todo = __ballot(pending);
            if (todo == 0ull) break;
            const int leader = __ffsll((long long)todo) - 1;
            const int lb = __shfl(bin, leader, 64);
            const bool same = pending && bin == lb;
            const uint32_t count = (uint32_t)__popcll(__ballot(same));
            if (lane == leader) (void)atomicAdd(&hist[(size_t)tr.track_index * RG_HISTOGRAM_SIZE + bin], count);
            pending = pending && !same;
        }
    }

    // ---- multi-window segments: windows 2..m of this lane's segment arrive as plain energies from the main kernel.
    // The start state's transient in them is sigma' T[n >= W]: below 1e-15 of sigma at the start of window 2 and
    // falling; it is bounded here with the lane's own sigma (slow pair; the fast block is gone after H10 frames) and a
    // window whose bin that bound could change marks the track imprecise, like a cancelling moment window above.
    if (G.m > 1) {
        bool any_flag = false;
        if (owner) {
            double dlt[NCH];
            const double t10 = fabs(G.T[(size_t)(G.L - 1) * RG_TM_DIM + 10]), t11 = fabs(G.T[(size_t)(G.L - 1) * RG_TM_DIM + 11]);
#pragma unroll
            for (int c = 0; c < NCH; ++c) dlt[c] = fabs(sgm[c][10]) * t10 + fabs(sgm[c][11]) * t11;
            for (uint32_t w = 1; w < G.m; ++w) {
                const uint64_t widx = (uint64_t)seg * G.m + w;
                if (widx >= tr.n_windows) break;
                const uint64_t rem = tr.frames - widx * G.W;
                const uint32_t n = rem < G.W ? (uint32_t)rem : G.W;
                double total = 0.0, e = 0.0;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const double a = win_energy[(size_t)c * total_windows + tr.win_base + widx];
                    total += a;
                    if (w == 1) e += 2.0 * dlt[c] * sqrt((double)n * fabs(a)) + (double)n * dlt[c] * dlt[c];
                }
                if (NCH == 1) { total *= 2.0; e *= 2.0; }
                if (nf != 0 && widx > 0xFFFFFFFFu - nf) total = __longlong_as_double(0x7FF8000000000000ll);  // == : its own class
                if (total < 0.0) total = 0.0;  // servo form: the signed affine terms can leave a rounding error below zero (as above)
                const int wb = rg_window_bin(total, 0.0, n);
                if (w == 1 && e > 1.0e-13 * total) {
                    const double lo = total - e;
                    any_flag = any_flag || rg_window_bin(lo < 0.0 ? 0.0 : lo, 0.0, n) != rg_window_bin(total + e, 0.0, n);
                }
                if (wb >= 0) (void)atomicAdd(&hist[(size_t)tr.track_index * RG_HISTOGRAM_SIZE + wb], 1u);  // no return value: waited for once, below
            }
        }
        if (__any(any_flag) && lane == 0) tm_performed(atomicOr(&imprecise[tr.track_index], 1u));
    }

    TM_FIX_STAMP(4);
    // ---- the last block of a track to get here finishes the track: percentile, gain, peak -> result
    // (tail of analyze_track_internal, src/replaygain.rs:910-918).  Everything a block publishes goes through
    // device-scope atomics whose results have come back (tm_performed) before the barrier, so the arrival
    // counter moves after them without a release fence: an agent-scope release on this multi-XCD part writes
    // the whole L2 back, and one per wave made that the most expensive thing in the kernel.  The finisher
    // drops its XCD's possibly stale histogram lines before reading.  (The per-window counts of multi-window segments
    // are atomics without a return value -- up to m - 1 per lane, not worth a round trip each: every wave waits here
    // until the memory side has acknowledged its own.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (i == 0) is_last = atomicAdd(&done_count[tr.track_index], 1u) + 1u == tr.fix_blocks ? 1 : 0;
    __syncthreads();
    TM_FIX_STAMP(5);
    if (is_last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        TM_FIX_STAMP(6);
        const RgLoudness l = rg_block_loudness(hist + (size_t)tr.track_index * RG_HISTOGRAM_SIZE, pct_scan);
        if (i == 0) {
            if (nf != 0) nonfinite[tr.track_index] = 0;  // every block of the track has read it: clean for the next batch
            const uint32_t imp = atomicExch(&imprecise[tr.track_index], 0u);  // coherent read, and clean again
            const uint32_t flags = (nf != 0 ? RG_TRACK_FLAG_NONFINITE : 0u) | (imp != 0 ? RG_TRACK_FLAG_IMPRECISE : 0u);
            const unsigned long long pb = atomicMax(&peak_bits[tr.track_index], 0ull);  // coherent read
            rg_store_track_result(results + tr.track_index, l, __longlong_as_double((long long)pb), tr.sample_rate,
                                  tr.file_type, flags);
        }
        TM_FIX_STAMP(7);
    }
#undef TM_FIX_STAMP
}

// =================================================================================================
// diagnostic timeline buffer (tools/ubench/timeline.py); nullptr in normal operation
static unsigned long long *g_tm_debug = nullptr;
extern "C" void rg_tm_set_debug_buffer(unsigned long long *d_buf) { g_tm_debug = d_buf; }
static unsigned long long *g_tm_fix_debug = nullptr;
extern "C" void rg_tm_set_fix_debug_buffer(unsigned long long *d_buf) { g_tm_fix_debug = d_buf; }

template <int FMT, bool SERVO>
static hipError_t launch_main_fmt(int nch, const RgTmCoef &K, const RgTmGeom &G, const RgTmTrack *d_tracks,
                                  uint32_t n_tracks, uint32_t grid, double *d_rec, uint32_t total_recs, double *d_win,
                                  uint32_t total_windows, uint32_t *d_nonfinite, uint32_t *d_zero, uint64_t zero_count,
                                  hipStream_t s) {
    // LDS: T12 (H10 x 12 doubles) + T2 ((L - H10) x 2 doubles) + one 4 KiB PCM tile per wave
    size_t lds = rg_tm_lds_bytes(G.L, G.H10, G.block, G.m);
    uint32_t lds_tables = lds <= RG_TM_LDS_BYTES ? 1u : 0u;
    if (!lds_tables) lds = 0;
#ifdef RG_TM_LDS_PAD_EXPERIMENT  // experiment: blocks of 256 lanes ask for more LDS than they use -> two of them per CU instead of three
    {
        static const long pad_kib = getenv("RG_TM_LDS_MIN_KIB") ? atol(getenv("RG_TM_LDS_MIN_KIB")) : 0;
        if (lds_tables && G.block == RG_TM_BLOCK && G.m == 1 && (long)lds < pad_kib * 1024) lds = (size_t)pad_kib * 1024;
    }
#endif
    // the attribute belongs to the function ON THE CURRENT DEVICE: a node drives several devices from one process
    // (rg_node.hip), each from its own host thread
    static std::atomic<unsigned long long> attr_set{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long dev_bit = 1ull << (dev & 63);
    if (lds > 48 * 1024 && !(attr_set.load(std::memory_order_acquire) & dev_bit)) {
        (void)hipFuncSetAttribute((const void *)rg_tm_main_kernel<FMT, false, SERVO>, hipFuncAttributeMaxDynamicSharedMemorySize, RG_TM_LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)rg_tm_main_kernel<FMT, true, SERVO>, hipFuncAttributeMaxDynamicSharedMemorySize, RG_TM_LDS_BYTES);
        attr_set.fetch_or(dev_bit, std::memory_order_release);
    }
    if (G.m > 1) {
        if (!lds_tables) return hipErrorInvalidValue;  // multi-window segments exist on the LDS path only
        hipLaunchKernelGGL((rg_tm_main_kernel<FMT, true, SERVO>), dim3(grid), dim3(G.block), lds, s, K, G, d_tracks, n_tracks,
                           d_rec, total_recs, d_win, total_windows, d_nonfinite, lds_tables, d_zero, zero_count, g_tm_debug);
    } else {
        hipLaunchKernelGGL((rg_tm_main_kernel<FMT, false, SERVO>), dim3(grid), dim3(G.block), lds, s, K, G, d_tracks, n_tracks,
                           d_rec, total_recs, d_win, total_windows, d_nonfinite, lds_tables, d_zero, zero_count, g_tm_debug);
    }
    return hipGetLastError();
}

extern "C" hipError_t rg_launch_tm_main(int fmt, int nch, const RgTmCoef *K, const RgTmGeom *G,
                                        const RgTmTrack *d_tracks, uint32_t n_tracks, uint32_t grid, double *d_rec,
                                        uint32_t total_recs, double *d_win, uint32_t total_windows, uint32_t *d_nonfinite,
                                        uint32_t *d_zero, uint64_t zero_count, hipStream_t s) {
    if (grid == 0) return hipSuccess;
#define RG_TM_LAUNCH(F, SV) launch_main_fmt<F, SV>(nch, *K, *G, d_tracks, n_tracks, grid, d_rec, total_recs, d_win, total_windows, d_nonfinite, d_zero, zero_count, s)
    if (G->servo) {
        switch (fmt) {
            case RG_FMT_F32_PLANAR: return RG_TM_LAUNCH(RG_FMT_F32_PLANAR, true);
            case RG_FMT_S16_PLANAR: return RG_TM_LAUNCH(RG_FMT_S16_PLANAR, true);
            default: return RG_TM_LAUNCH(RG_FMT_S32_PLANAR, true);
        }
    }
    switch (fmt) {
        case RG_FMT_F32_PLANAR: return RG_TM_LAUNCH(RG_FMT_F32_PLANAR, false);
        case RG_FMT_S16_PLANAR: return RG_TM_LAUNCH(RG_FMT_S16_PLANAR, false);
        default: return RG_TM_LAUNCH(RG_FMT_S32_PLANAR, false);
    }
#undef RG_TM_LAUNCH
}

extern "C" hipError_t rg_launch_tm_fix(int nch, const RgTmGeom *G, const RgTmFixTables *FT, const RgTmTrack *d_tracks,
                                       uint32_t n_tracks, uint32_t grid, const double *d_rec, uint32_t total_recs,
                                       const double *d_win, uint32_t total_windows, uint32_t *d_nonfinite, uint32_t *d_imprecise, uint32_t *d_hist,
                                       unsigned long long *d_peak_bits, uint32_t *d_done, rg_track_result *d_results,
                                       hipStream_t s) {
    if (grid == 0) return hipSuccess;
    if (nch == 1)
        hipLaunchKernelGGL((rg_tm_fix_kernel<1>), dim3(grid), dim3(RG_TM_BLOCK), 0, s, *G, *FT, d_tracks, n_tracks, d_rec,
                           total_recs, d_win, total_windows, d_nonfinite, d_imprecise, d_hist, d_peak_bits, d_done, d_results, g_tm_fix_debug);
    else
        hipLaunchKernelGGL((rg_tm_fix_kernel<2>), dim3(grid), dim3(RG_TM_BLOCK), 0, s, *G, *FT, d_tracks, n_tracks, d_rec,
                           total_recs, d_win, total_windows, d_nonfinite, d_imprecise, d_hist, d_peak_bits, d_done, d_results, g_tm_fix_debug);
    return hipGetLastError();
}
