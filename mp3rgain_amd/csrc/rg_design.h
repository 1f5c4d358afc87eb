// rg_design.h -- host-side analysis of one sample rate's equal-loudness cascade
// (Yule-Walker 10th order -> Butterworth high-pass 2nd order, src/replaygain.rs:586-616).
// Everything here is derived from the coefficient rows alone; it is evaluated once per
// context in extended precision.
#pragma once

#include <stdint.h>

#include "../../include/rg_coeffs.h"

struct RgRateDesign {
    bool stable;           // false for the 88.2 kHz row as the reference has it (poles outside |z|=1)
    uint32_t halo_frames;  // frames after which the cascade's memory of earlier input is < 1e-16 (relative, L1)
    double pole_radius;    // estimated decay ratio of the impulse-response tail
};

void rg_design_rate(const rg_rate_coeffs &rc, RgRateDesign *out);

// ---- variant 2 (transient-moment) tables, see rg_tm.h -------------------------------------------
#include <vector>

struct RgTmDesign {
    uint32_t L = 0, W = 0, m = 1;
    uint32_t H10 = 0;         // multiple of 4, <= L rounded down to a multiple of 4 (or == that bound)
    double tau10[10] = {0};   // what the cut at H10 leaves out: tau10[j] = sqrt(sum_{n >= H10} T[n][j]^2), so that the neglected part of
                              // a moment, |sum_{n >= H10} z[n] T[n][j]|, is at most sqrt(sum z^2) tau10[j] (the fix-up kernel's self-check)
    uint32_t rounds = 0;      // doubling rounds for the slow (Butter) block
    uint32_t rounds_fast = 0; // rounds after which the fast (Yule) block's power is below 1e-18
    std::vector<double> T;       // [L][12]  responses in block-diagonal coordinates
    std::vector<double> Gp;      // [L][78]  prefix Gram matrices (upper triangle, row major)
    std::vector<double> PhiY;    // [rounds][10][10]  (F_y^L)^(2^r)
    std::vector<double> PhiB;    // [rounds][2][2]    (F_b^L)^(2^r)
    double X[2][10];             // coordinate change: t' = t + X s
    bool whiten = false;         // 64 / 96 kHz: each block is carried in coordinates in which its Gram matrix is the identity
    double Wf[100];              // [10][10] upper triangular: fast block of a DF2T end state -> carried coordinates (identity if !whiten)
    double Xs[2][12];            // slow pair of a DF2T end state (s, t) -> carried coordinates: Rs (t + X s) = Xs [s; t]
    double sigma0[12];           // track-start state in block-diagonal coordinates
    bool servo = false;          // Butterworth stage in servo form, linear lanes, analytic affine term (rg_tm.h: RgTmCoef)
    double alpha = 0, beta = 0;  // servo constants as the kernel holds them (rounded once, from long double)
    double g = 1;                // servo: butter b0, folded into the Yule stage's feed-forward taps
    double dinf = 0;             // servo: constant output offset of the reference's "+1e-10" terms (= 1e-10 / beta)
    std::vector<double> ST;      // servo: [L][12] prefix sums of T (affine cross term 2 d_inf sigma . ST)
    double resid;                // Sylvester residual (diagnostic)
    bool ok = false;
};

// L must divide W = rate*50/1000; m = windows per segment (m > 1 needs L == W): the segment stride is L * m frames,
// which is what the state transition Phi spans.  Returns ok=false for an unstable row or when no truncation
// with <= RG_TM_MAX_ROUNDS doubling rounds reaches 1e-18.
void rg_tm_design(const rg_rate_coeffs &rc, uint32_t L, RgTmDesign *out, uint32_t m = 1);
// true when the row's Butterworth numerator is exactly g (1, -2, 1): the servo form then realises the same filter
bool rg_tm_servo_ok(const rg_rate_coeffs &rc);
