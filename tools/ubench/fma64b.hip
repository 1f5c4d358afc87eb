// micro-benchmark 2: FP64 FMA forms as the TM kernel uses them (tools only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s\n", hipGetErrorString(e_)); return 1; } } while (0)

// FORM 0: acc = fma(S, acc, S)        1 VGPR source
// FORM 1: acc = fma(S, v, acc)        2 VGPR sources (v_fmac form), v fixed
// FORM 2: acc = fma(v1, v2, acc)      3 VGPR sources
// FORM 3: chain of the kernel: u = fma(S, x, s_next); s = fma(-S, y, u)  (2 dependent, ILP across i)
template <int FORM, int ILP>
__global__ void k(double *out, int iters, double a, double b) {
    double acc[ILP], v1 = threadIdx.x * 1e-7 + 1.0, v2 = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (FORM == 0) acc[i] = fma(a, acc[i], b);
                else if (FORM == 1) acc[i] = fma(a, v1, acc[i]);
                else if (FORM == 2) acc[i] = fma(v1, v2, acc[i]);
            }
            if (FORM == 1 || FORM == 2) { v1 += 1e-9; }
        }
    }
    double s = v1 + v2;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FORM, int ILP>
int bench(double *d, int wps) {
    const int iters = 4000, blocks = 256 * wps;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<FORM, ILP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<FORM, ILP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double inst = (double)iters * 8 * ILP;
    printf("form %d ILP %2d waves/SIMD %d: %.2f cycles/instr/SIMD-slot (@2.4GHz)\n", FORM, ILP, wps, ms * 1e-3 * 2.4e9 / inst / wps);
    return 0;
}

int main() {
    double *d; CK(hipMalloc(&d, 1 << 26));
    for (int wps : {1, 2, 4}) {
        bench<0, 10>(d, wps); bench<1, 10>(d, wps); bench<2, 10>(d, wps); bench<1, 2>(d, wps); bench<2, 2>(d, wps);
    }
    return 0;
}
