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

// The same DCT as a dense product behind ONE even/odd split -- the form the device runs on its matrix cores
// (rg_mp3dev.hip, wave 2: v_mfma_f32_16x16x4_f32), and therefore the form the host runs too:
//   u[k] = x[k] + x[31-k],  v[k] = x[k] - x[31-k]                                      (k = 0..15)
//   A[2i]   = sum_k u[k] cos(pi (2i)   (2k+1) / 64),   A[2i+1] = sum_k v[k] cos(pi (2i+1) (2k+1) / 64)   (i = 0..15)
// every sum a chain of fused multiply-adds from +0.0f in ascending k -- the order in which the matrix instruction adds its
// four products to the accumulator and in which the device chains four of them (tools/ubench/mfma_order.hip established
// it on gfx950).  512 fused operations and 32 additions per time slot instead of Lee's 289 -- on the host's vector units
// (sixteen independent chains per parity, the inner loop runs over i) no slower, and closer to the exact sum: no secants
// up to 10.2 in front of the additions.  `c`: [parity][k][i], (float) of the cosines above (rg_mp3dec.cpp: tables()).
RG_MP3_HD void rg_mp3_dct32_split(const float *x /* 32 */, float *A /* 32 */, const float (*c)[16][16]) {
    float u[16], v[16], e[16], o[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        u[k] = x[k] + x[31 - k];
        v[k] = x[k] - x[31 - k];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = o[i] = 0.0f;
    for (int k = 0; k < 16; ++k) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            e[i] = rg_mp3_mac(u[k], c[0][k][i], e[i]);
            o[i] = rg_mp3_mac(v[k], c[1][k][i], o[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        A[2 * i] = e[i];
        A[2 * i + 1] = o[i];
    }
}

// The polyphase matrixing V[i] = sum_k S[k] cos((16 + i)(2k+1) pi / 64), i = 0..63, from the DCT above:
//   V[i] = A[16+i] (i < 16),  V[16] = 0,  V[i] = -A[48-i] (17 <= i <= 47),  V[48] = -A[0],  V[i] = -A[i-48] (i >= 49)
// `V` may be any random-access target with operator[] (a plain array, or the host's ring buffer view).
template <typename Out>
RG_MP3_HD void rg_mp3_matrixing(const float *S /* 32 */, Out &&V, const float *sec, const float (*dct16)[16][16]) {
    float A[32];
#ifdef RG_MP3_DCT_LEE
    (void)dct16;
    RgMp3Dct<32>::run(S, A, sec);
#else
    (void)sec;
    rg_mp3_dct32_split(S, A, dct16);
#endif
#pragma unroll
    for (int i = 0; i < 16; ++i) V[i] = A[16 + i];
    V[16] = 0.0f;
#pragma unroll
    for (int i = 17; i < 48; ++i) V[i] = -A[48 - i];
    V[48] = -A[0];
#pragma unroll
    for (int i = 49; i < 64; ++i) V[i] = -A[i - 48];
}

// ---- 36-point IMDCT of a long block, fast -------------------------------------------------------------------------
// x[i] = sum_k X[k] cos(pi/72 (2i + 1 + 18)(2k + 1)), i = 0..35, has x[17 - i] = -x[i] and x[35 - j] = x[18 + j]; the
// eighteen independent samples are an 18-point DCT-IV,  t[n] = sum_k X[k] cos(pi/72 (2n + 1)(2k + 1)):
//     x[i] = t[9 + i] (i = 0..8),   x[18 + j] = -t[8 - j] (j = 0..8).
// The DCT-IV goes through a 9-point complex DFT (ISO 11172-3 leaves the method open; this is the textbook route):
//     z[k] = (X[2k] + i X[17 - 2k]) e^{-i pi (4k + 1) / 72},   Z = DFT_9(z) as 3 x 3 radix-3 butterflies,
//     t[2k] = Re(Z[k] e^{-i pi k / 18}),   t[17 - 2k] = -Im(Z[k] e^{-i pi k / 18})
// -- 156 multiply / add / fused operations instead of 324 multiply-adds.  Every fused operation is an explicit rg_mp3_mac,
// so host and device produce the same bits.
struct RgMp3Cx { float r, i; };
RG_MP3_HD RgMp3Cx rg_mp3_cmul(const RgMp3Cx a, const float wr, const float wi) {
    return RgMp3Cx{rg_mp3_mac(a.r, wr, -(a.i * wi)), rg_mp3_mac(a.r, wi, a.i * wr)};
}
// 3-point DFT, W = e^{-2 pi i / 3}
RG_MP3_HD void rg_mp3_dft3(const RgMp3Cx x0, const RgMp3Cx x1, const RgMp3Cx x2, RgMp3Cx &y0, RgMp3Cx &y1, RgMp3Cx &y2) {
    const float s3 = 0.866025388f;  // sqrt(3) / 2
    const float sr = x1.r + x2.r, si = x1.i + x2.i, dr = x1.r - x2.r, di = x1.i - x2.i;
    const float mr = rg_mp3_mac(-0.5f, sr, x0.r), mi = rg_mp3_mac(-0.5f, si, x0.i);
    y0 = RgMp3Cx{x0.r + sr, x0.i + si};
    y1 = RgMp3Cx{rg_mp3_mac(s3, di, mr), rg_mp3_mac(-s3, dr, mi)};
    y2 = RgMp3Cx{rg_mp3_mac(-s3, di, mr), rg_mp3_mac(s3, dr, mi)};
}
RG_MP3_HD void rg_mp3_dct4_18(const float *X /* 18 */, float *t /* 18 */) {
    RgMp3Cx z[9];
    z[0] = rg_mp3_cmul(RgMp3Cx{X[0], X[17]}, 0.999048233f, -0.0436193869f);
    z[1] = rg_mp3_cmul(RgMp3Cx{X[2], X[15]}, 0.976296008f, -0.21643962f);
    z[2] = rg_mp3_cmul(RgMp3Cx{X[4], X[13]}, 0.923879504f, -0.382683426f);
    z[3] = rg_mp3_cmul(RgMp3Cx{X[6], X[11]}, 0.843391418f, -0.537299633f);
    z[4] = rg_mp3_cmul(RgMp3Cx{X[8], X[9]}, 0.737277329f, -0.675590217f);
    z[5] = rg_mp3_cmul(RgMp3Cx{X[10], X[7]}, 0.60876143f, -0.793353319f);
    z[6] = rg_mp3_cmul(RgMp3Cx{X[12], X[5]}, 0.4617486f, -0.887010813f);
    z[7] = rg_mp3_cmul(RgMp3Cx{X[14], X[3]}, 0.300705791f, -0.953716934f);
    z[8] = rg_mp3_cmul(RgMp3Cx{X[16], X[1]}, 0.130526185f, -0.991444886f);
    RgMp3Cx F[3][3], Z[9];
    rg_mp3_dft3(z[0], z[3], z[6], F[0][0], F[0][1], F[0][2]);
    rg_mp3_dft3(z[1], z[4], z[7], F[1][0], F[1][1], F[1][2]);
    rg_mp3_dft3(z[2], z[5], z[8], F[2][0], F[2][1], F[2][2]);
    rg_mp3_dft3(F[0][0], F[1][0], F[2][0], Z[0], Z[3], Z[6]);
    rg_mp3_dft3(F[0][1], rg_mp3_cmul(F[1][1], 0.766044438f, -0.642787635f), rg_mp3_cmul(F[2][1], 0.173648179f, -0.98480773f), Z[1], Z[4], Z[7]);
    rg_mp3_dft3(F[0][2], rg_mp3_cmul(F[1][2], 0.173648179f, -0.98480773f), rg_mp3_cmul(F[2][2], -0.939692616f, -0.342020154f), Z[2], Z[5], Z[8]);
    t[0] = Z[0].r;
    t[17] = -Z[0].i;
    { const RgMp3Cx y = rg_mp3_cmul(Z[1], 0.98480773f, -0.173648179f); t[2] = y.r; t[15] = -y.i; }
    { const RgMp3Cx y = rg_mp3_cmul(Z[2], 0.939692616f, -0.342020154f); t[4] = y.r; t[13] = -y.i; }
    { const RgMp3Cx y = rg_mp3_cmul(Z[3], 0.866025388f, -0.5f); t[6] = y.r; t[11] = -y.i; }
    { const RgMp3Cx y = rg_mp3_cmul(Z[4], 0.766044438f, -0.642787635f); t[8] = y.r; t[9] = -y.i; }
    { const RgMp3Cx y = rg_mp3_cmul(Z[5], 0.642787635f, -0.766044438f); t[10] = y.r; t[7] = -y.i; }
    { const RgMp3Cx y = rg_mp3_cmul(Z[6], 0.5f, -0.866025388f); t[12] = y.r; t[5] = -y.i; }
    { const RgMp3Cx y = rg_mp3_cmul(Z[7], 0.342020154f, -0.939692616f); t[14] = y.r; t[3] = -y.i; }
    { const RgMp3Cx y = rg_mp3_cmul(Z[8], 0.173648179f, -0.98480773f); t[16] = y.r; t[1] = -y.i; }
}

// The 36 windowed samples of a long block (block types 0, 1, 3) from its 18 lines:  raw[i] = x[i] * win[i].
template <typename Out>
RG_MP3_HD void rg_mp3_imdct36_windowed(const float *X /* 18 */, const float *win /* 36 */, Out &&raw /* 36 */) {
    float t[18];
    rg_mp3_dct4_18(X, t);
#pragma unroll
    for (int m = 0; m < 9; ++m) {
        const float a = t[9 + m], b = t[8 - m];
        raw[m] = a * win[m];
        raw[17 - m] = -a * win[17 - m];
        raw[18 + m] = -b * win[18 + m];
        raw[35 - m] = -b * win[35 - m];
    }
}
