// rg_tm.h -- data layout of variant 2, the transient-moment (TM) formulation.
//
// The reference runs one sequential recursion per channel over the whole track
// (src/replaygain.rs:866-904: the filters and the analyzer live across all packets).  Variant 2
// cuts every channel into segments of L frames (L divides the 50 ms window W) and gives each
// segment to one lane.  The cascade (10th-order Yule-Walker -> 2nd-order Butterworth,
// src/replaygain.rs:586-616, here in transposed direct form II with fused multiply-add) is an
// affine system with a 12-dimensional state  q = (s1..s10 | t1 t2):
//       q[n+1] = F q[n] + g x[n] + c,      z[n] = h.q[n] + d x[n]
// so the true output of a segment that starts in state sigma is
//       z[n] = zs[n] + sum_j sigma_j T_j[n]
// with zs the response from the ZERO state (what the lane computes) and T_j the response of the
// homogeneous system to unit state j.  The window energy then needs only three moments per
// segment and channel, accumulated in the same pass:
//       A = sum zs^2,   B_j = sum zs T_j,   G_jk = sum T_j T_k  (input independent, tabulated)
//       sum z^2 = A + 2 B.sigma + sigma' G sigma
// and the true start state follows from the zero-state end states E of the preceding segments:
//       sigma_k = E_{k-1} + Phi sigma_{k-1},   Phi = F^L,  |Phi^m| -> 0 geometrically,
// evaluated by the fix-up kernel as a truncated doubling scan over 2^R predecessors.
// Nothing is approximated beyond f64 rounding: the truncation is chosen so that the dropped
// terms are below 1e-17 relative (rg_design.cpp).
#pragma once

#include <stdint.h>

#define RG_TM_DIM 12          // state dimension: 10 (Yule) + 2 (Butter)
#define RG_TM_GRAM 78         // upper triangle of a symmetric 12x12
#define RG_TM_REC 26          // doubles per (segment, channel): A, B[12], E[12], peak
#define RG_TM_BLOCK 256
// Main kernel only: when the LDS image of the response tables is so large that fewer than three 256-thread
// blocks fit a CU (160 KiB), one 768-thread block shares a single image and still puts 3 waves on each SIMD.
#define RG_TM_BLOCK_WIDE 768
#ifndef RG_TM_BLOCK_WIDE_MULTI
#define RG_TM_BLOCK_WIDE_MULTI 768   // multi-window segments (the m > 1 instantiation of the main kernel).  1024 = 128 VGPRs = four
                                     // waves per SIMD was tried in round 3: 204 bytes per lane spill into the frame loop, 20.8 ms per
                                     // step against 18.7; the loop without the moments alone needs 139 registers
#endif
#define RG_TM_WAVE_TILE_BYTES 4096  // PCM staging tile of one wave: 64 rows x 16 frames x 4 B
#define RG_TM_LDS_BYTES (160u * 1024u)
// Self-check of the fix-up kernel.  A window's energy S = A + 2 B.sigma + sigma'G sigma is assembled from
// M = A + sigma'G sigma; rounding leaves an absolute error of about c * 2^-53 * M in S.  A window is flagged
// imprecise when S - e and S + e, e = RG_TM_CEPS * M, do not fall into the same histogram bin (or one of them is
// dropped and the other is not).  Measured over 9000 random and pathological tracks: the displaced windows sit at
// M / S of 1e6 .. 1e8 and need e / M of 1e-14 .. 1e-12 to be explained; 1e-11 leaves an order of magnitude.
#ifndef RG_TM_CEPS
#define RG_TM_CEPS 1.0e-11
#endif
#define RG_TM_MAX_ROUNDS 4    // the doubling scan reaches 2^4 = 16 predecessors
#define RG_TM_EDGE 16         // lanes of a wave whose scan values are visible to the next wave (>= 2^(MAX_ROUNDS-1), and 1 for the final shift)

// filter constants of one launch group, passed by value (wave-uniform -> SGPRs)
//
// Two realisations of the Butterworth high-pass stage (src/replaygain.rs:604-615):
//   classic  transposed direct form II, 5 multiply-adds per sample, the reference's "+1e-10" per stage injected at the
//            deepest state of each stage (rounds 1-3; still used for the rates whose numerator is not exactly g (1, -2, 1))
//   servo    (round 4) for numerators that ARE exactly g (1 - z^-1)^2 -- 96, 48, 44.1, 32, 16, 12 and 8 kHz: g is folded
//            into the Yule stage's feed-forward taps and the stage is the output minus a double integrator,
//                z = y - v1;   v1 += v2 + alpha z;   v2 += beta z;      alpha = 2 + a1,  beta = 1 + a1 + a2 = A(1)
//            which has the reference's denominator to 1e-19 (alpha, beta are small numbers computed in long double),
//            costs 4 operations instead of 5 (26 per sample with the energy instead of 27) and is better conditioned
//            than either direct form: v1, v2 follow the low-frequency content instead of holding differences of large
//            numbers (30 Hz full-scale sine, window energies against a long double run of the reference's recursion:
//            reference order 1.1e-12, classic 8.7e-13, servo 8e-15).  The lanes are LINEAR in this mode; the "+1e-10"
//            offsets are an affine term that does not depend on the signal and is added analytically: the true
//            system is the linear one started at -q_inf plus a constant output offset d_inf = 1e-10 / beta, and
//            sum over a window of (z + d_inf)^2 = sum z^2 + 2 d_inf (v2_end - v2_start) / beta + n d_inf^2 because
//            v2 IS the running sum of beta z.
struct RgTmCoef {
    double b[11];   // yule b, pre-multiplied by the (power of two) input scale of the sample format (servo: and by butter b0)
    double a[11];   // yule a (a[0] unused)
    double bb[3];   // butter b (classic)
    double ba[3];   // butter a (ba[0] unused; classic)
    double c0;      // classic: the reference's per-stage +1e-10 (src/replaygain.rs:530), injected at the deepest state
    double alpha;   // servo: 2 + butter a1
    double beta;    // servo: 1 + butter a1 + butter a2
    double aff_lin; // servo: 2 d_inf / beta   -- window energy += aff_lin * (v2_end - v2_start) + aff_n * frames
    double aff_n;   // servo: d_inf^2
};

struct RgTmTrack {
    const void *ch0;
    const void *ch1;           // nullptr for mono
    uint64_t frames;
    uint32_t nseg;             // ceil(frames / L)
    uint32_t n_windows;        // ceil(frames / W)
    uint32_t rec_base;         // first segment record of this track
    uint32_t lane_base;        // first lane of this track in the main kernel's grid (channel 0's segments, then channel 1's)
    uint32_t fix_block_base;   // first block of this track in the fix-up kernel's grid
    uint32_t track_index;      // row in the histogram / peak arrays
    uint32_t fix_blocks;       // number of fix-up blocks of this track (the last one to finish writes the result)
    uint32_t sample_rate;
    uint32_t file_type;
    uint32_t win_base;         // first entry of this track in the per-window energy array (multi-window segments)
};

// launch-group geometry (passed by value)
struct RgTmGeom {
    uint32_t L;            // segment length in frames
    uint32_t W;            // window length in frames
    uint32_t k;            // segments per window (W / L)
    uint32_t H10;          // frames after which the fast (Yule) part of a transient is negligible
    uint32_t rounds;       // doubling rounds R; 2^R predecessors are summed
    uint32_t rounds_fast;  // rounds in which the fast block still contributes
    uint32_t warm;         // warm-up lanes per fix-up block (2^R)
    uint32_t fix_windows;  // whole windows per fix-up block
    uint32_t block;        // threads per main-kernel block: RG_TM_BLOCK, or RG_TM_BLOCK_WIDE when the LDS image is large
    uint32_t m;            // windows per segment (1 = a segment is L <= W frames; > 1 needs L == W: a lane then runs m
                           // whole windows, the transient moments only over the first -- by the second window the
                           // start state has decayed below 1e-30 of itself -- and hands windows 2..m over as plain
                           // energies, which rg_tm_direct_kernel bins)
    uint32_t whiten;       // 64 / 96 kHz: the fix-up kernel takes a DF2T end state's fast block through Wf (rg_design.cpp)
    uint32_t servo;        // the Butterworth stage runs in servo form and the lanes are linear (RgTmCoef)
    double aff_lin, aff_n; // servo: the affine term of a window, see RgTmCoef (0 otherwise)
    double aff_sig;        // servo: 2 d_inf -- times (start state . sum of the responses) for the window the moments cover
    double tau10[10];      // tail norms of the fast responses behind H10 (rg_design.h): the self-check's bound on what the cut leaves out
    const double *T;       // [L][12] homogeneous responses, block-diagonal coordinates
    const double *Tlds;    // the same packed for LDS: [H10][12] then [L - H10][2] (only the slow pair)
};

// device tables of the fix-up kernel (passed by value; all wave-uniform reads)
struct RgTmFixTables {
    const double *X;       // [2][12]  slow pair of a DF2T end state (s, t) in the carried coordinates: Xs [s; t] (= t + X s below 64 kHz)
    const double *sigma0;  // [12]     state at the start of a track
    const double *PhiY;    // [rounds][10][10]  (F_y^L)^(2^r)
    const double *PhiB;    // [rounds][2][2]    (F_b^L)^(2^r)
    const double *Gp;      // [L][78]  prefix Gram matrices: Gp[len-1] = sum_{n<len} T T'
    const double *ST;      // [L][12]  servo: prefix sums of the responses, ST[len-1] = sum_{n<len} T[n] (the affine cross term)
};

// LDS bytes of a main-kernel block of `block` threads, and the block size for an (L, H10) design.  A wave of a
// multi-window launch (m > 1) has TWO tile slots: the windows without moments keep two tiles in LDS and two in flight
// (rg_k2_tm.hip: tm_fast_path).
static inline size_t rg_tm_lds_bytes(uint32_t L, uint32_t H10, uint32_t block, uint32_t m = 1) {
    return ((size_t)H10 * 12 + (size_t)(L - H10) * 2) * sizeof(double) + (size_t)(block / 64) * RG_TM_WAVE_TILE_BYTES * (m > 1 ? 2u : 1u);
}
static inline uint32_t rg_tm_choose_block(uint32_t L, uint32_t H10, uint32_t m = 1) {
    if (3 * rg_tm_lds_bytes(L, H10, RG_TM_BLOCK, m) <= RG_TM_LDS_BYTES) return RG_TM_BLOCK;
    const uint32_t wide = m > 1 ? RG_TM_BLOCK_WIDE_MULTI : RG_TM_BLOCK_WIDE;
    if (rg_tm_lds_bytes(L, H10, wide, m) <= RG_TM_LDS_BYTES) return wide;
    return RG_TM_BLOCK;
}
