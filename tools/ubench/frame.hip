// micro-benchmark 3: the TM frame (cascade step + moments) in isolation (tools only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../mp3rgain_amd/csrc/rg_tm.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int NX, bool BARRIERS>
__global__ void __launch_bounds__(256) k(const RgTmCoef K, double *out, int frames) {
    double s[10], t[2], A = 0, B[12];
    for (int i = 0; i < 10; ++i) s[i] = 1e-3 * i + threadIdx.x * 1e-6;
    t[0] = t[1] = 0;
    for (int j = 0; j < 12; ++j) B[j] = 0;
    double tr[12];
    for (int j = 0; j < 12; ++j) tr[j] = 0.5 + j * 1e-3;
    float f = threadIdx.x * 1e-4f;
    for (int n = 0; n < frames; n += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double x = (double)f;
            f += 1e-6f;
            const double y = fma(K.b[0], x, s[0]);
            double uu[10];
#pragma unroll
            for (int i = 0; i < 9; ++i) uu[i] = fma(K.b[i + 1], x, s[i + 1]);
            uu[9] = fma(K.b[10], x, K.c0);
            if (BARRIERS) __builtin_amdgcn_sched_barrier(0);
            double z = fma(K.bb[0], y, t[0]);
            const double w1 = fma(K.bb[1], y, t[1]);
            const double w2 = fma(K.bb[2], y, K.c0);
            if (BARRIERS) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 10; ++i) s[i] = fma(-K.a[i + 1], y, uu[i]);
            if (BARRIERS) __builtin_amdgcn_sched_barrier(0);
            t[0] = fma(-K.ba[1], z, w1);
            t[1] = fma(-K.ba[2], z, w2);
            A = fma(z, z, A);
#pragma unroll
            for (int j = 0; j < NX; ++j) B[12 - NX + j] = fma(z, tr[j], B[12 - NX + j]);
            if (BARRIERS) __builtin_amdgcn_sched_barrier(0);
        }
    }
    double r = A + t[0] + t[1];
    for (int i = 0; i < 10; ++i) r += s[i];
    for (int j = 0; j < 12; ++j) r += B[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NX, bool BARRIERS>
int bench(const RgTmCoef &K, double *d, int wps) {
    const int frames = 8000, blocks = 256 * wps;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<NX, BARRIERS>), dim3(blocks), dim3(256), 0, 0, K, d, frames);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<NX, BARRIERS>), dim3(blocks), dim3(256), 0, 0, K, d, frames);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("NX %2d barriers %d waves/SIMD %d: %.1f cycles/frame/wave-slot (@2.4GHz)  [%d DP ops/frame]\n", NX, (int)BARRIERS, wps,
           ms * 1e-3 * 2.4e9 / frames / wps, 27 + 1 + NX + 1);
    return 0;
}

int main() {
    double *d; CK(hipMalloc(&d, 1 << 26));
    RgTmCoef K;
    for (int i = 0; i < 11; ++i) { K.b[i] = 0.01 * (i + 1); K.a[i] = 0.02 * (i + 1) - 0.1; }
    for (int i = 0; i < 3; ++i) { K.bb[i] = 0.3 * (i + 1); K.ba[i] = 0.1 * i - 0.05; }
    K.c0 = 1e-10;
    for (int wps : {1, 2, 3, 4}) {
        bench<2, true>(K, d, wps); bench<2, false>(K, d, wps); bench<12, true>(K, d, wps);
    }
    return 0;
}
