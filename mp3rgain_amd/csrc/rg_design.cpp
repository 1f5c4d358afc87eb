// rg_design.cpp -- see rg_design.h
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
