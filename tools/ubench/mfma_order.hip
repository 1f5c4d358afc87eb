// tools/ubench/mfma_order.hip -- in which order, and with which roundings, does v_mfma_f32_16x16x4_f32 add its four products
// to the accumulator?  (The device MP3 decoder's matrixing is held to the host decoder bit for bit: a matrix-core
// formulation needs the host to sum in the hardware's order.)
//
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/mfma_order.hip -o build_ab/mfma_order && build_ab/mfma_order gpurun_out/mfma_order.bin
//   python tools/ubench/mfma_order_check.py gpurun_out/mfma_order.bin     (here or there: pure numpy)
//
// Every trial is one 16 x 16 x 4 product D = A B + C with its operands dumped as the lanes held them; trial classes:
//   0  random values, exponents spread over 2^-20 .. 2^20 (cancellation makes the order visible)
//   1  the same, C = 0
//   2  operands whose products and sums are subnormal (are they flushed?)
//   3  two instructions chained through the accumulator (k = 0..7)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave per trial; a[t][lane], b[t][lane] (second pair for class 3), c[t][lane][4] -> d[t][lane][4]
__global__ void __launch_bounds__(64) mfma_trials(const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ a2,
                                                   const float *__restrict__ b2, const f32x4 *__restrict__ c, f32x4 *__restrict__ d,
                                                   const int *__restrict__ cls) {
    const int t = blockIdx.x, l = threadIdx.x;
    f32x4 acc = c[t * 64 + l];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t * 64 + l], b[t * 64 + l], acc, 0, 0, 0);
    if (cls[t] == 3) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[t * 64 + l], b2[t * 64 + l], acc, 0, 0, 0);
    d[t * 64 + l] = acc;
}

static uint64_t s_rng = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
    s_rng ^= s_rng << 13; s_rng ^= s_rng >> 7; s_rng ^= s_rng << 17;
    return (uint32_t)(s_rng >> 32);
}
static float rnd_float(int emin, int emax) {
    const uint32_t mant = rnd() & 0x7FFFFFu, sign = rnd() & 1u;
    const int e = emin + (int)(rnd() % (uint32_t)(emax - emin + 1));
    const uint32_t bits = (sign << 31) | ((uint32_t)(e + 127) << 23) | mant;
    float f; memcpy(&f, &bits, 4); return f;
}

int main(int argc, char **argv) {
    const int T = 4096;
    std::vector<float> a(T * 64), b(T * 64), a2(T * 64), b2(T * 64), c(T * 256), d(T * 256);
    std::vector<int> cls(T);
    for (int t = 0; t < T; ++t) {
        cls[t] = t & 3;
        for (int l = 0; l < 64; ++l) {
            if (cls[t] == 2) {
                a[t * 64 + l] = rnd_float(-70, -60); b[t * 64 + l] = rnd_float(-75, -62);
                a2[t * 64 + l] = 0.0f; b2[t * 64 + l] = 0.0f;
                for (int v = 0; v < 4; ++v) c[(t * 64 + l) * 4 + v] = (rnd() & 1) ? rnd_float(-126, -120) * 0.0009765625f : 0.0f;
            } else {
                a[t * 64 + l] = rnd_float(-10, 10); b[t * 64 + l] = rnd_float(-10, 10);
                a2[t * 64 + l] = rnd_float(-10, 10); b2[t * 64 + l] = rnd_float(-10, 10);
                for (int v = 0; v < 4; ++v) c[(t * 64 + l) * 4 + v] = cls[t] == 1 ? 0.0f : rnd_float(-20, 20);
            }
        }
    }
    float *da, *db, *da2, *db2; f32x4 *dc, *dd; int *dcls;
    hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&da2, a.size() * 4); hipMalloc(&db2, b.size() * 4);
    hipMalloc(&dc, c.size() * 4); hipMalloc(&dd, d.size() * 4); hipMalloc(&dcls, T * 4);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(da2, a2.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db2, b2.data(), b.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dcls, cls.data(), T * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_trials, dim3(T), dim3(64), 0, 0, da, db, da2, db2, dc, dd, dcls);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost);
    FILE *f = fopen(argc > 1 ? argv[1] : "mfma_order.bin", "wb");
    if (!f) { perror("open"); return 1; }
    fwrite(&T, 4, 1, f);
    fwrite(cls.data(), 4, T, f);
    fwrite(a.data(), 4, a.size(), f); fwrite(b.data(), 4, b.size(), f);
    fwrite(a2.data(), 4, a2.size(), f); fwrite(b2.data(), 4, b2.size(), f);
    fwrite(c.data(), 4, c.size(), f); fwrite(d.data(), 4, d.size(), f);
    fclose(f);
    printf("wrote %d trials\n", T);
    return 0;
}
