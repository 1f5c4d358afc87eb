// micro-benchmark: FP64 FMA issue rate / dependent latency on gfx950 (tools only, not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int ILP, bool SGPR>
__global__ void k(double *out, int iters, double a, double b) {
    double acc[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    const double va = SGPR ? a : a + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) acc[i] = fma(acc[i], va, b);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP>
__global__ void k32(float *out, int iters, float a, float b) {
    float acc[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) acc[i] = fmaf(acc[i], a, b);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float run(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    double *d; hipMalloc(&d, 1 << 26);
    const int iters = 2000;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    int wps_list[] = {1, 2, 4, 8};
    for (int wps : wps_list) {
        const int blocks = 256 * wps;  // 256-thread blocks: 4 waves -> 1 wave per SIMD per block
#define RUN64(ILP, SG) { float ms = run([&] { hipLaunchKernelGGL((k<ILP, SG>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9); }); \
        double inst = (double)iters * 16 * ILP; double cyc = ms * 1e-3 * 2.4e9; \
        printf("f64 waves/SIMD %d ILP %2d sgpr %d: %.3f ms  -> %.2f cycles/instr/wave(@2.4GHz), %.1f TFLOP/s\n", wps, ILP, (int)SG, ms, cyc / inst / wps * 1.0, 2.0 * inst * 64 * 4 * blocks / (ms * 1e-3) / 1e12); }
        RUN64(1, true) RUN64(2, true) RUN64(4, true) RUN64(8, true) RUN64(16, true) RUN64(8, false)
#define RUN32(ILP) { float ms = run([&] { hipLaunchKernelGGL((k32<ILP>), dim3(blocks), dim3(256), 0, 0, (float *)d, iters, 1.0000001f, 1e-9f); }); \
        double inst = (double)iters * 16 * ILP; double cyc = ms * 1e-3 * 2.4e9; \
        printf("f32 waves/SIMD %d ILP %2d: %.3f ms -> %.2f cycles/instr/wave, %.1f TFLOP/s\n", wps, ILP, ms, cyc / inst / wps, 2.0 * inst * 64 * 4 * blocks / (ms * 1e-3) / 1e12); }
        RUN32(1) RUN32(8)
    }
    return 0;
}
