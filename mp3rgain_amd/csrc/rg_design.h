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
