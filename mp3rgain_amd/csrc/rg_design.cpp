// rg_design.cpp -- see rg_design.h
#ifndef RG_TM_H10_CUT_DEFAULT
#define RG_TM_H10_CUT_DEFAULT 1e-10L
#endif
#include <stdlib.h>
#include "rg_design.h"

#include <math.h>

#include <vector>

namespace {

// impulse response of the cascade, direct form I in long double, without the 1e-10 offsets
// (they are part of the particular solution, not of the system's memory)
std::vector<long double> impulse_response(const rg_rate_coeffs &rc, size_t n, bool *finite) {
    std::vector<long double> h(n);
    long double yx[11] = {0}, yy[11] = {0}, bx[3] = {0}, by[3] = {0};
    *finite = true;
    for (size_t k = 0; k < n; ++k) {
        for (int i = 10; i >= 1; --i) { yx[i] = yx[i - 1]; yy[i] = yy[i - 1]; }
        yx[0] = k == 0 ? 1.0L : 0.0L;
        long double y = (long double)rc.yule_b[0] * yx[0];
        for (int i = 1; i < 11; ++i) y += (long double)rc.yule_b[i] * yx[i] - (long double)rc.yule_a[i] * yy[i];
        yy[0] = y;
        bx[2] = bx[1]; bx[1] = bx[0]; bx[0] = y;
        by[2] = by[1]; by[1] = by[0];
        long double z = (long double)rc.butter_b[0] * bx[0];
        for (int i = 1; i < 3; ++i) z += (long double)rc.butter_b[i] * bx[i] - (long double)rc.butter_a[i] * by[i];
        by[0] = z;
        h[k] = z;
        if (!(fabsl(z) < 1e30L)) { *finite = false; h.resize(k + 1); break; }
    }
    return h;
}

}  // namespace

void rg_design_rate(const rg_rate_coeffs &rc, RgRateDesign *out) {
    const size_t N = 1u << 16;
    bool finite = true;
    std::vector<long double> h = impulse_response(rc, N, &finite);
    out->stable = false;
    out->halo_frames = 0xFFFFFFFFu;
    out->pole_radius = 1.0;
    if (!finite) return;
    long double total = 0.0L;
    for (long double v : h) total += fabsl(v);
    // tail mass must have decayed to nothing by the end of the simulated span
    long double late = 0.0L;
    for (size_t k = N - 1024; k < N; ++k) late += fabsl(h[k]);
    if (!(late <= 1e-25L * total)) return;  // unstable or marginally stable: sequential order only
    long double tail = 0.0L;
    size_t H = N;
    for (size_t k = N; k-- > 0;) {
        tail += fabsl(h[k]);
        if (tail > 1e-16L * total) { H = k + 1; break; }
    }
    out->stable = true;
    out->halo_frames = (uint32_t)((H + 63) / 64 * 64);
    // decay ratio from two points of the envelope (diagnostic only)
    long double e1 = 0.0L, e2 = 0.0L;
    const size_t a = H / 4, b = H / 2, w = 64;
    for (size_t k = a; k < a + w && k < h.size(); ++k) e1 += fabsl(h[k]);
    for (size_t k = b; k < b + w && k < h.size(); ++k) e2 += fabsl(h[k]);
    out->pole_radius = (e1 > 0 && e2 > 0 && b > a) ? (double)powl(e2 / e1, 1.0L / (long double)(b - a)) : 0.0;
}

// =================================================================================================
// variant 2: transient-moment tables
// =================================================================================================
#include <string.h>

#include "rg_tm.h"

namespace {

typedef long double ld;

// servo constants exactly as the kernel holds them
struct ServoK {
    bool on;
    ld g, alpha, beta;
};
ServoK servo_constants(const rg_rate_coeffs &rc) {
    ServoK k;
    k.on = rg_tm_servo_ok(rc);
    k.g = (ld)rc.butter_b[0];
    k.alpha = (ld)(double)(2.0L + (ld)rc.butter_a[1]);
    k.beta = (ld)(double)(1.0L + (ld)rc.butter_a[1] + (ld)rc.butter_a[2]);
    return k;
}

// one step of the cascade exactly as the kernel runs it (rg_k2_tm.hip: tm_step / tm_frame), homogeneous part only when
// x = 0 and c = 0.
//   classic: both stages in transposed direct form II, the reference's +c per stage injected at the deepest state of each
//   servo:   Yule stage in DF2T with butter b0 folded into its feed-forward taps, Butterworth stage as output minus double
//            integrator.  With c != 0 this is the EXACT affine system of the reference (used for its fixed point only; the
//            kernel's lanes are linear): A_b Z = (1 - z^-1)^2 Y' + c U needs z = y' - v1 + c, v1 += v2 + alpha z - 2c,
//            v2 += beta z - c, and the Yule stage's +c enters its output (times g).
void tm_step_ld(const rg_rate_coeffs &rc, const ServoK &sv, ld q[12], ld x, ld c, ld *z_out) {
    ld n[12];
    if (sv.on) {
        const ld y = (ld)(double)(sv.g * (ld)rc.yule_b[0]) * x + q[0] + sv.g * c;
        for (int i = 0; i < 9; ++i) n[i] = q[i + 1] + (ld)(double)(sv.g * (ld)rc.yule_b[i + 1]) * x - (ld)rc.yule_a[i + 1] * y;
        n[9] = (ld)(double)(sv.g * (ld)rc.yule_b[10]) * x - (ld)rc.yule_a[10] * y;
        const ld z = y - q[10] + c;
        n[10] = q[10] + q[11] + sv.alpha * z - 2.0L * c;
        n[11] = q[11] + sv.beta * z - c;
        memcpy(q, n, sizeof n);
        *z_out = z;
        return;
    }
    const ld y = (ld)rc.yule_b[0] * x + q[0];
    for (int i = 0; i < 9; ++i) n[i] = q[i + 1] + (ld)rc.yule_b[i + 1] * x - (ld)rc.yule_a[i + 1] * y;
    n[9] = (ld)rc.yule_b[10] * x - (ld)rc.yule_a[10] * y + c;
    const ld z = (ld)rc.butter_b[0] * y + q[10];
    n[10] = q[11] + (ld)rc.butter_b[1] * y - (ld)rc.butter_a[1] * z;
    n[11] = (ld)rc.butter_b[2] * y - (ld)rc.butter_a[2] * z + c;
    memcpy(q, n, sizeof n);
    *z_out = z;
}

// dense solve with partial pivoting, in place; returns false when singular
bool solve_ld(int n, std::vector<ld> &A, std::vector<ld> &b) {
    for (int col = 0; col < n; ++col) {
        int piv = col;
        for (int r = col + 1; r < n; ++r)
            if (fabsl(A[r * n + col]) > fabsl(A[piv * n + col])) piv = r;
        if (fabsl(A[piv * n + col]) < 1e-300L) return false;
        if (piv != col) {
            for (int k = 0; k < n; ++k) std::swap(A[col * n + k], A[piv * n + k]);
            std::swap(b[col], b[piv]);
        }
        for (int r = col + 1; r < n; ++r) {
            const ld f = A[r * n + col] / A[col * n + col];
            if (f == 0.0L) continue;
            for (int k = col; k < n; ++k) A[r * n + k] -= f * A[col * n + k];
            b[r] -= f * b[col];
        }
    }
    for (int r = n - 1; r >= 0; --r) {
        ld s = b[r];
        for (int k = r + 1; k < n; ++k) s -= A[r * n + k] * b[k];
        b[r] = s / A[r * n + r];
    }
    return true;
}

void matmul_ld(int n, const ld *A, const ld *B, ld *C) {
    std::vector<ld> t((size_t)n * n, 0.0L);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < n; ++k) {
            const ld a = A[i * n + k];
            for (int j = 0; j < n; ++j) t[i * n + j] += a * B[k * n + j];
        }
    memcpy(C, t.data(), sizeof(ld) * n * n);
}

// G = R'R, R upper triangular; false when G is not (numerically) positive definite
bool cholesky_ld(int n, const ld *G, ld *R) {
    for (int i = 0; i < n * n; ++i) R[i] = 0.0L;
    for (int i = 0; i < n; ++i) {
        ld d = G[i * n + i];
        for (int k = 0; k < i; ++k) d -= R[k * n + i] * R[k * n + i];
        if (!(d > 0.0L)) return false;
        R[i * n + i] = sqrtl(d);
        for (int j = i + 1; j < n; ++j) {
            ld v = G[i * n + j];
            for (int k = 0; k < i; ++k) v -= R[k * n + i] * R[k * n + j];
            R[i * n + j] = v / R[i * n + i];
        }
    }
    return true;
}
// x := x R^-1 for a row vector x (R upper triangular)
void row_times_inverse_ld(int n, ld *x, const ld *R) {
    for (int j = 0; j < n; ++j) {
        ld v = x[j];
        for (int k = 0; k < j; ++k) v -= x[k] * R[k * n + j];
        x[j] = v / R[j * n + j];
    }
}

ld maxabs(const ld *A, int n) {
    ld m = 0;
    for (int i = 0; i < n; ++i) m = fabsl(A[i]) > m ? fabsl(A[i]) : m;
    return m;
}

}  // namespace

bool rg_tm_servo_ok(const rg_rate_coeffs &rc) {
    return rc.butter_b[1] == -2.0 * rc.butter_b[0] && rc.butter_b[2] == rc.butter_b[0];
}

void rg_tm_design(const rg_rate_coeffs &rc, uint32_t L, RgTmDesign *out, uint32_t m) {
    *out = RgTmDesign();
    const uint32_t W = (uint32_t)(((uint64_t)rc.sample_rate * 50u) / 1000u);
    if (L == 0 || W % L != 0 || m == 0 || (m > 1 && L != W)) return;
    out->m = m;
    RgRateDesign rd;
    rg_design_rate(rc, &rd);
    if (!rd.stable) return;
    out->L = L;
    out->W = W;
    const ServoK sv = servo_constants(rc);
    out->servo = sv.on;
    out->alpha = (double)sv.alpha;
    out->beta = (double)sv.beta;
    out->g = sv.on ? (double)sv.g : 1.0;

    // state matrix F (column j = one homogeneous step from unit state j) and output row h
    ld F[12][12], h[12];
    for (int j = 0; j < 12; ++j) {
        ld q[12] = {0};
        q[j] = 1.0L;
        ld z;
        tm_step_ld(rc, sv, q, 0.0L, 0.0L, &z);
        for (int i = 0; i < 12; ++i) F[i][j] = q[i];
        h[j] = z;
    }
    // F = [[Fy 0][Cb Fb]]; find X (2x10) with X Fy - Fb X = -Cb so that t' = t + X s decouples
    std::vector<ld> M(400, 0.0L), rhs(20, 0.0L);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 10; ++j) {
            const int r = i * 10 + j;
            for (int k = 0; k < 10; ++k) M[r * 20 + i * 10 + k] += F[k][j];
            for (int k = 0; k < 2; ++k) M[r * 20 + k * 10 + j] -= F[10 + i][10 + k];
            rhs[r] = -F[10 + i][j];
        }
    std::vector<ld> M0 = M, x = rhs;
    if (!solve_ld(20, M, x)) return;
    {   // one step of iterative refinement
        std::vector<ld> res(20);
        for (int r = 0; r < 20; ++r) {
            ld s = rhs[r];
            for (int k = 0; k < 20; ++k) s -= M0[r * 20 + k] * x[k];
            res[r] = s;
        }
        std::vector<ld> M1 = M0;
        if (solve_ld(20, M1, res))
            for (int k = 0; k < 20; ++k) x[k] += res[k];
    }
    ld X[2][10];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 10; ++j) { X[i][j] = x[i * 10 + j]; out->X[i][j] = (double)X[i][j]; }
    // residual of the decoupling (diagnostic)
    ld resid = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 10; ++j) {
            ld s = F[10 + i][j];
            for (int k = 0; k < 10; ++k) s += X[i][k] * F[k][j];
            for (int k = 0; k < 2; ++k) s -= F[10 + i][10 + k] * X[k][j];
            resid = fabsl(s) > resid ? fabsl(s) : resid;
        }
    out->resid = (double)resid;

    // block-diagonal dynamics: Fy (10x10), Fb (2x2); output row h' = h P^-1, P^-1 = [[I 0][-X I]]
    ld Fy[100], Fb[4], hp[12];
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 10; ++j) Fy[i * 10 + j] = F[i][j];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) Fb[i * 2 + j] = F[10 + i][10 + j];
    for (int j = 0; j < 10; ++j) hp[j] = h[j] - (h[10] * X[0][j] + h[11] * X[1][j]);
    hp[10] = h[10];
    hp[11] = h[11];

    // T'[n] = h' F'^n  (row vector iteration)
    std::vector<ld> Tl((size_t)L * 12);
    {
        ld row[12];
        memcpy(row, hp, sizeof row);
        for (uint32_t n = 0; n < L; ++n) {
            for (int j = 0; j < 12; ++j) Tl[(size_t)n * 12 + j] = row[j];
            ld nr[12];
            for (int j = 0; j < 10; ++j) {
                ld s = 0;
                for (int i = 0; i < 10; ++i) s += row[i] * Fy[i * 10 + j];
                nr[j] = s;
            }
            for (int j = 0; j < 2; ++j) nr[10 + j] = row[10] * Fb[0 * 2 + j] + row[11] * Fb[1 * 2 + j];
            memcpy(row, nr, sizeof row);
        }
    }
    // ---- whitening (64 / 96 kHz).  There the Yule-Walker poles crowd z = 1: in DF2T coordinates the responses to the ten
    // fast states are nearly parallel, a unit state reaches the output with gain 71 / 478, and sigma'G sigma is the small
    // difference of terms 10^4 - 10^5 times its size.  Each block is therefore carried in coordinates in which its own Gram
    // matrix over the segment's first window is the identity: G_ff = R'R (Cholesky), sigma_f := R sigma_f, T_f := T_f R^-1,
    // Phi_f := R Phi_f R^-1, likewise the slow pair.  The main kernel is unchanged (it multiplies by whatever table it is
    // given and hands over its DF2T end state); the fix-up kernel applies R once per segment (Wf / Xs below).
    ld Rf[100], Rs[4];
    for (int i = 0; i < 100; ++i) Rf[i] = (i / 10 == i % 10) ? 1.0L : 0.0L;
    Rs[0] = Rs[3] = 1.0L; Rs[1] = Rs[2] = 0.0L;
    out->whiten = false;
    if (rc.sample_rate > 48000) {
        ld Gf[100] = {0}, Gs[4] = {0};
        for (uint32_t n = 0; n < L; ++n) {
            const ld *r = &Tl[(size_t)n * 12];
            for (int i = 0; i < 10; ++i)
                for (int j = 0; j < 10; ++j) Gf[i * 10 + j] += r[i] * r[j];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) Gs[i * 2 + j] += r[10 + i] * r[10 + j];
        }
        ld Rf2[100], Rs2[4];
        if (!(cholesky_ld(10, Gf, Rf2) && cholesky_ld(2, Gs, Rs2))) {
            // without the whitening these rates bring back the cancellation it was added to remove (rg_enqueue.hip sends
            // every stable rate to variant 2): the candidate is refused, choose_tm_tables tries the next one
            out->ok = false;
            return;
        }
        {
            memcpy(Rf, Rf2, sizeof Rf);
            memcpy(Rs, Rs2, sizeof Rs);
            out->whiten = true;
            for (uint32_t n = 0; n < L; ++n) {
                row_times_inverse_ld(10, &Tl[(size_t)n * 12], Rf);
                row_times_inverse_ld(2, &Tl[(size_t)n * 12 + 10], Rs);
            }
            // Phi := R Phi R^-1 (one step; its powers follow)
            ld A[100];
            matmul_ld(10, Rf, Fy, A);
            for (int i = 0; i < 10; ++i) row_times_inverse_ld(10, &A[i * 10], Rf);
            memcpy(Fy, A, sizeof A);
            ld B2[4];
            matmul_ld(2, Rs, Fb, B2);
            for (int i = 0; i < 2; ++i) row_times_inverse_ld(2, &B2[i * 2], Rs);
            memcpy(Fb, B2, sizeof B2);
        }
    }
    for (int i = 0; i < 100; ++i) out->Wf[i] = (double)Rf[i];
    // Xs = [Rs X | Rs]: slow pair of a DF2T end state (s, t) in the carried coordinates = Rs (t + X s)
    for (int q = 0; q < 2; ++q) {
        for (int j = 0; j < 10; ++j) out->Xs[q][j] = (double)(Rs[q * 2 + 0] * X[0][j] + Rs[q * 2 + 1] * X[1][j]);
        out->Xs[q][10] = (double)Rs[q * 2 + 0];
        out->Xs[q][11] = (double)Rs[q * 2 + 1];
    }
    // the rounded table and the prefix Gram matrices (the kernel multiplies by the rounded table, so the Gram matrices use
    // the rounded values too)
    out->T.resize((size_t)L * 12);
    out->Gp.resize((size_t)L * RG_TM_GRAM);
    if (sv.on) out->ST.resize((size_t)L * 12);
    std::vector<ld> gram(RG_TM_GRAM, 0.0L);
    ld tsum[12] = {0};
    ld tmax = 0;
    for (uint32_t n = 0; n < L; ++n) {
        for (int j = 0; j < 12; ++j) {
            out->T[(size_t)n * 12 + j] = (double)Tl[(size_t)n * 12 + j];
            if (j < 10 && fabsl(Tl[(size_t)n * 12 + j]) > tmax) tmax = fabsl(Tl[(size_t)n * 12 + j]);
        }
        int p = 0;
        for (int j = 0; j < 12; ++j)
            for (int k = j; k < 12; ++k, ++p)
                gram[p] += (ld)out->T[(size_t)n * 12 + j] * (ld)out->T[(size_t)n * 12 + k];
        for (int q = 0; q < RG_TM_GRAM; ++q) out->Gp[(size_t)n * RG_TM_GRAM + q] = (double)gram[q];
        if (sv.on)
            for (int j = 0; j < 12; ++j) {
                tsum[j] += (ld)out->T[(size_t)n * 12 + j];
                out->ST[(size_t)n * 12 + j] = (double)tsum[j];
            }
    }
    // H10: first multiple of 4 after which every fast response stays below `cut` of its maximum.  What the cut leaves out of a
    // moment is bounded per window in the fix-up kernel (tau10, below) and goes into the self-check's margin: a window whose
    // bin the neglected tail could change marks its track RG_TRACK_FLAG_IMPRECISE like a cancelling one.  1e-13 (rounds 2-5) was
    // below the rounding of the moments themselves; 1e-10 (round 6: H10 248 -> 184 frames at 44.1 kHz, 3-4 % fewer vector
    // instructions at L = 735) is what the bound made safe -- DESIGN.md section 7 has the flag rates per cut; RG_TM_H10_CUT (read
    // once per process: measurement) moves it.
    static const ld cut = [] {
        const char *e = getenv("RG_TM_H10_CUT");
        const ld v = e ? strtold(e, nullptr) : 0.0L;
        return (v > 0.0L && v < 1e-3L) ? v : (ld)RG_TM_H10_CUT_DEFAULT;
    }();
    uint32_t H10 = 0;
    for (uint32_t n = L; n-- > 0;) {
        ld m = 0;
        for (int j = 0; j < 10; ++j) m = fmaxl(m, fabsl((ld)out->T[(size_t)n * 12 + j]));
        if (m > cut * tmax) { H10 = n + 1; break; }
    }
    // A multiple of 4 (the kernel consumes 4-frame pieces and a piece must not straddle H10) -- or the whole
    // segment when the fast block outlives it: the L & 3 trailing frames then keep all 12 moments too.  (Cutting
    // at L & ~3 dropped the fast moments of up to three frames whose responses were nowhere near decayed:
    // 0.2 % errors in quiet windows after loud ones for L = 150 at 24 kHz, L = 245 at 44.1 kHz.)
    H10 = (H10 + 3u) & ~3u;
    out->H10 = H10 >= L ? L : H10;
    for (int j = 0; j < 10; ++j) {
        ld t2 = 0;
        for (uint32_t n = out->H10; n < L; ++n) t2 += (ld)out->T[(size_t)n * 12 + j] * (ld)out->T[(size_t)n * 12 + j];
        out->tau10[j] = (double)(sqrtl(t2) * 1.000001L);
    }

    // Phi blocks: F^L by repeated multiplication, then squarings for the doubling rounds
    ld PY[100], PB[4];
    for (int i = 0; i < 100; ++i) PY[i] = (i / 10 == i % 10) ? 1.0L : 0.0L;
    PB[0] = PB[3] = 1.0L; PB[1] = PB[2] = 0.0L;
    for (uint64_t n = 0; n < (uint64_t)L * m; ++n) {  // one segment = L * m frames
        matmul_ld(10, PY, Fy, PY);
        matmul_ld(2, PB, Fb, PB);
        if (maxabs(PY, 100) < 1e-40L && maxabs(PB, 4) < 1e-40L) break;  // nothing left to multiply (and no denormal crawl)
    }
    uint32_t rounds = 0, rounds_fast = 0;
    bool fast_done = false;
    for (int r = 0; r <= RG_TM_MAX_ROUNDS; ++r) {
        // entering round r the blocks hold Phi^(2^r): if both are negligible, r rounds suffice
        const bool y_small = maxabs(PY, 100) < 1e-18L, b_small = maxabs(PB, 4) < 1e-18L;
        if (y_small && !fast_done) { rounds_fast = r; fast_done = true; }
        if (y_small && b_small) { rounds = r; out->ok = true; break; }
        if (r == RG_TM_MAX_ROUNDS) break;
        for (int i = 0; i < 100; ++i) out->PhiY.push_back((double)PY[i]);
        for (int i = 0; i < 4; ++i) out->PhiB.push_back((double)PB[i]);
        matmul_ld(10, PY, PY, PY);
        matmul_ld(2, PB, PB, PB);
    }
    if (!out->ok) return;
    out->rounds = rounds;  // 0 is legal: sigma_k = E_{k-1} alone
    out->rounds_fast = fast_done ? rounds_fast : rounds;
    out->PhiY.resize((size_t)rounds * 100);
    out->PhiB.resize((size_t)rounds * 4);

    // track-start state in the kernel's own coordinates
    const ld c = 1e-10L;
    ld q0[12];
    if (!sv.on) {
        // classic: every DF2T state holds the +1e-10 offset once (see rg_tm.h)
        for (int j = 0; j < 12; ++j) q0[j] = c;
    } else {
        // servo: the lanes are linear.  The reference's affine system (tm_step_ld with c) has the fixed point
        // q_inf = (I - F)^-1 c_vec and settles at the constant output d_inf; in coordinates q - q_inf it IS the linear
        // system, so a track starts at -q_inf and every output carries + d_inf (added per window by the kernels).
        ld cvec[12] = {0}, zc;
        tm_step_ld(rc, sv, cvec, 0.0L, c, &zc);  // Phi(0)
        std::vector<ld> A(144), b(12);
        for (int i = 0; i < 12; ++i) {
            for (int j = 0; j < 12; ++j) A[i * 12 + j] = (i == j ? 1.0L : 0.0L) - F[i][j];
            b[i] = cvec[i];
        }
        std::vector<ld> A0 = A, x = b;
        if (!solve_ld(12, A, x)) { out->ok = false; return; }
        for (int it = 0; it < 2; ++it) {  // iterative refinement
            std::vector<ld> res(12);
            for (int i = 0; i < 12; ++i) {
                ld sres = b[i];
                for (int j = 0; j < 12; ++j) sres -= A0[i * 12 + j] * x[j];
                res[i] = sres;
            }
            std::vector<ld> A1 = A0;
            if (solve_ld(12, A1, res))
                for (int j = 0; j < 12; ++j) x[j] += res[j];
        }
        ld qinf[12], zinf;
        for (int j = 0; j < 12; ++j) qinf[j] = x[j];
        tm_step_ld(rc, sv, qinf, 0.0L, c, &zinf);  // one more step from the fixed point: its output is d_inf
        out->dinf = (double)zinf;
        for (int j = 0; j < 12; ++j) q0[j] = -x[j];
    }
    // ... then t' = t + X s, then the carried coordinates (the identity below 64 kHz)
    ld s0[12];
    for (int j = 0; j < 10; ++j) s0[j] = q0[j];
    for (int i = 0; i < 2; ++i) {
        ld s = q0[10 + i];
        for (int j = 0; j < 10; ++j) s += X[i][j] * q0[j];
        s0[10 + i] = s;
    }
    for (int i = 0; i < 10; ++i) {
        ld v = 0;
        for (int j = 0; j < 10; ++j) v += Rf[i * 10 + j] * s0[j];
        out->sigma0[i] = (double)v;
    }
    for (int i = 0; i < 2; ++i) out->sigma0[10 + i] = (double)(Rs[i * 2 + 0] * s0[10] + Rs[i * 2 + 1] * s0[11]);
}
