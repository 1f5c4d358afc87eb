// rg_mp3_math.h -- arithmetic shared VERBATIM by the host decoder (rg_mp3dec.cpp) and the device decoder (rg_mp3dev.hip).
//
// The two decoders are held to bit-identical PCM (tests/test_gpu_mp3.py), so wherever the order of floating-point
// operations is not the obvious left-to-right sum, both sides compile the same source: the 32-point DCT behind the
// polyphase matrixing lives here, and so does the one fused operation both sides use (rg_mp3_mac).
#pragma once

#if defined(__HIP__) || defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RG_MP3_HD __host__ __device__ __forceinline__
#else
#define RG_MP3_HD inline
#endif

// Multiply-accumulate as ONE rounding, on both sides: the dot products of the IMDCT and of the synthesis window are chains
// of these (v_fma_f32 on the device, vfmadd on the host: rg_mp3dec.cpp is compiled with -mfma, and both with
// -ffp-contract=off so that nothing else is fused behind the source's back).
RG_MP3_HD float rg_mp3_mac(float a, float b, float acc) { return __builtin_fmaf(a, b, acc); }

// 32-point DCT-II, unnormalised:  A[m] = sum_k x[k] cos(pi m (2k+1) / 64),  by Lee's recursive even/odd split
//   u[k] = x[k] + x[N-1-k],  v[k] = (x[k] - x[N-1-k]) * sec_N[k],   sec_N[k] = 1 / (2 cos(pi (2k+1) / (2N)))
//   A[2m] = DCT_{N/2}(u)[m],  A[2m+1] = DCT_{N/2}(v)[m] + DCT_{N/2}(v)[m+1]
// 80 multiplications and 209 additions instead of 1024 + 992.  `sec` holds the secants of the levels N = 32, 16, 8, 4,
// 2 back to back (16 + 8 + 4 + 2 + 1 = 31 values).
template <int N>
struct RgMp3Dct {
    static RG_MP3_HD void run(const float *x, float *X, const float *sec) {
        constexpr int H = N / 2;
        float u[H], v[H], U[H], W[H];
#pragma unroll
        for (int k = 0; k < H; ++k) {
            u[k] = x[k] + x[N - 1 - k];
            v[k] = (x[k] - x[N - 1 - k]) * sec[k];
        }
        RgMp3Dct<H>::run(u, U, sec + H);
        RgMp3Dct<H>::run(v, W, sec + H);
#pragma unroll
        for (int m = 0; m < H; ++m) {
            X[2 * m] = U[m];
            X[2 * m + 1] = m + 1 < H ? W[m] + W[m + 1] : W[m];
        }
    }
};
template <>
struct RgMp3Dct<1> {
    static RG_MP3_HD void run(const float *x, float *X, const float *) { X[0] = x[0]; }
};

// The polyphase matrixing V[i] = sum_k S[k] cos((16 + i)(2k+1) pi / 64), i = 0..63, from the DCT above:
//   V[i] = A[16+i] (i < 16),  V[16] = 0,  V[i] = -A[48-i] (17 <= i <= 47),  V[48] = -A[0],  V[i] = -A[i-48] (i >= 49)
// `V` may be any random-access target with operator[] (a plain array, or the host's ring buffer view).
template <typename Out>
RG_MP3_HD void rg_mp3_matrixing(const float *S /* 32 */, Out &&V, const float *sec) {
    float A[32];
    RgMp3Dct<32>::run(S, A, sec);
#pragma unroll
    for (int i = 0; i < 16; ++i) V[i] = A[16 + i];
    V[16] = 0.0f;
#pragma unroll
    for (int i = 17; i < 48; ++i) V[i] = -A[48 - i];
    V[48] = -A[0];
#pragma unroll
    for (int i = 49; i < 64; ++i) V[i] = -A[i - 48];
}
