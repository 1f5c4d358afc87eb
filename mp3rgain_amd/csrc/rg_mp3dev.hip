// rg_mp3dev.hip -- the device half of the MP3 decoder: everything of rg_mp3dec.cpp after the frame walk, on gfx950.
//
//   rg_mp3_frames_kernel   (tuning key 6 = 3) headers + side information -> the records the Huffman stage works from:
//                          which frames decode, where each granule's bits are (the code of rg_mp3_frame.h, shared with
//                          the host)
//   rg_mp3_huffman_kernel  scalefactors + Huffman-coded spectrum, one thread per granule and channel
//   rg_mp3_backhalf_kernel everything after that, one block per run of 32 granules of a track (both channels), a pipeline
//                          of four waves: requantisation + joint stereo (mid/side, intensity in the MPEG-1 and the LSF
//                          form) + short-block reordering | alias reduction + IMDCT + windowing + overlap-add + frequency
//                          inversion | matrixing of the polyphase filterbank | its 512-tap window -> 576 PCM samples per
//                          granule and channel, planar f32, straight into the analysis arena
//
// Nothing here is recursive across granules: the overlap is the previous granule's second IMDCT half, the filterbank's
// FIFO the previous fifteen time slots' subband samples; a run that starts inside a track computes the two granules before
// its own once more, so every run of every track of a batch is independent work.
//
// Bit-identical to the host decoder by construction: compiled with -ffp-contract=off, every sum in the host's order
// (sequential, from 0.0f), every constant from the host's own tables (rg_mp3_fill_device_tables), the data-dependent
// powers of two from a table indexed by the exact integer exponent.  tests/test_gpu_mp3.py demands equality.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "rg_mp3dev.h"
#include "rg_mp3_math.h"
#include "rg_mp3_frame.h"

namespace {

__device__ __forceinline__ uint32_t find_by_granule(const RgMp3DevTrack *__restrict__ tr, uint32_t n, uint32_t g) {
    uint32_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (tr[mid].granule_base <= g) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__device__ __forceinline__ uint32_t find_by_unit(const RgMp3DevTrack *__restrict__ tr, uint32_t n, uint64_t u) {
    uint32_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (tr[mid].unit_base <= u) lo = mid; else hi = mid - 1;
    }
    return lo;
}


// The quantised spectra between the Huffman stage and the back half: 72 pieces of 16 bytes (8 lines) per unit.  With the
// Huffman stage on the device, groups of G = 8 units are interleaved piece by piece -- the Huffman kernel's lanes are
// consecutive units at the same piece, so eight lanes fill a 128-byte line with one store, where rows of their own made
// every 16-byte store a partial line (read for ownership + a masked write: 3.5 KB of traffic per unit for 1.2 KB of
// spectrum).  G = 1: plain rows (what the host's rg_mp3_parse_units writes).  Index in 16-byte pieces; G = 2^group_log2.
typedef float rg_f32x2 __attribute__((ext_vector_type(2)));
typedef short rg_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short rg_u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint64_t rg_mp3_is_index(uint64_t unit, int chunk, uint32_t group_log2) {
    return ((((unit >> group_log2) * 72 + (uint64_t)chunk) << group_log2) | (unit & ((1u << group_log2) - 1u)));
}

// Bit reader over the batch's main-data buffer; bits at or past `limit` (the end of the frame's own main data) read
// as zero, which is what the host decoder's private copy of the frame's data does.
struct DevBits {
    const uint8_t *__restrict__ p;
    uint64_t pos, end, limit;
    __device__ __forceinline__ uint32_t window() const {  // 32 bits starting at the byte that holds `pos`
        const uint64_t byte = pos >> 3;
        const uint8_t *q = p + byte;
        uint32_t w = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
        const int64_t valid = (int64_t)limit - (int64_t)(byte << 3);
        if (valid < 32) w = valid <= 0 ? 0u : (w & (0xFFFFFFFFu << (32 - (int)valid)));
        return w;
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return (window() << (pos & 7)) >> (32 - n); }  // 1 <= n <= 25
    __device__ __forceinline__ uint32_t get(int n) {
        if (n == 0) return 0;
        const uint32_t v = peek(n);
        pos += n;
        return v;
    }
    __device__ __forceinline__ uint32_t get1() { return get(1); }
};

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Stages B-E, one kernel.  One block = a run of RG_MP3_RUN consecutive granules of one track (both channels), four waves,
// a four-stage pipeline over the run's granules with ONE barrier per granule; every hand-over goes through LDS:
//   wave 1  granule k      units and quantised spectra in (fetched one granule ahead); requantisation -- a line's gain
//                          2^((global_gain - 210)/4 - mult (sf + preflag pretab) [- 2 subblock_gain]) depends on its band
//                          (and window) only, so the wave first writes the granule's 22 long-band and 39 (short band,
//                          window) gains, from a table indexed by the exact integer exponent, and a line then costs one
//                          look-up; what is the same for the whole granule sits in scalar registers --, mid/side,
//                          intensity stereo, short-block reordering (wave-local steps, no block barrier inside)
//   wave 0  granule k - 1  lane (channel, subband): alias butterflies folded into the load of its eighteen lines, the
//                          fast 36-point IMDCT of rg_mp3_math.h (the code the host runs) or three 12-point ones, window,
//                          overlap-add with the second half it kept from the granule before (an LDS column of its own),
//                          frequency inversion -> 18 x 32 subband samples
//   wave 2  granule k - 2  matrixing: lane (channel, time slot) runs the 32-point DCT of rg_mp3_math.h (one source
//                          compiled into both decoders); the 32 DCT outputs are kept, the 64 matrixing values follow
//                          from them by symmetry
//   wave 3  granule k - 3  the 512-tap window: lane (channel, j) owns PCM sample j of the granule's eighteen time slots;
//                          the two DCT columns it needs of the fifteen slots of history stay in its registers from one
//                          granule to the next (2 LDS reads per output instead of 32), the symmetry's signs are folded
//                          into its sixteen window coefficients; 576 PCM samples per granule straight into the arena
// A run that starts inside the track takes the two granules before it through the first stages (the second one's subband
// samples need the first one's overlap, and its DCT rows are the filterbank's history).  Each wave runs its own loop, so
// the register file is sized for the largest stage, not for their sum: 128 VGPRs, 40 KB of LDS, four blocks per CU.
// (Rounds 2 and 3 had two kernels with the subband samples in memory between them -- 4.6 KB of traffic per granule and
// channel, the fifteen slots of history transformed again by every block, 0.60 ms per 256 K units against 0.46 for the first version of this kernel and 0.35 now (what the per-wave
// clock stamps of RG_BH_TIMING showed); a
// first fused kernel in round 2, six waves stepping through barrier-separated phases together, had lost to them.)

// The block's barrier between pipeline steps.  What the waves hand each other is in LDS, so only LDS traffic has to be
// complete at the barrier: __syncthreads() would also wait for every global load and store in flight -- the next
// granule's spectra and units, the PCM on its way out -- and make a step as long as a round trip to memory.
__device__ __forceinline__ void rg_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

#ifdef RG_BH_TIMING
// Instrument (tools/build_variant.sh NAME -DRG_BH_TIMING=<block>; tools/bh_timing.py): shader-clock stamps of one block's
// waves around their work of every pipeline step, and inside the requantisation wave
__device__ unsigned long long rg_bh_dbg[4][40][2];
__device__ unsigned long long rg_bh_dbg2[40][4];
extern "C" int rg_bh_dbg_read(void *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rg_bh_dbg), sizeof(unsigned long long) * 4 * 40 * 2); }
extern "C" int rg_bh_dbg2_read(void *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rg_bh_dbg2), sizeof(unsigned long long) * 40 * 4); }
#define RG_BH_STAMP(w, k, e) do { if (blockIdx.x == RG_BH_TIMING && (threadIdx.x & 63) == 0 && (k) < 40) rg_bh_dbg[w][k][e] = __builtin_amdgcn_s_memtime(); } while (0)
#define RG_BH_STAMP2(k, e) do { if (blockIdx.x == RG_BH_TIMING && (threadIdx.x & 63) == 0 && (k) < 40) rg_bh_dbg2[k][e] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RG_BH_STAMP(w, k, e) do { } while (0)
#define RG_BH_STAMP2(k, e) do { } while (0)
#endif
#define RG_MP3_BH_THREADS 256
__global__ void __launch_bounds__(RG_MP3_BH_THREADS) __attribute__((amdgpu_waves_per_eu(4)))
rg_mp3_backhalf_kernel(const RgMp3DevTables *__restrict__ T, const RgMp3DevTrack *__restrict__ tracks, uint32_t n_tracks,
                       const rg_mp3_unit *__restrict__ units, const int16_t *__restrict__ is, const uint32_t is_group_log2) {
    constexpr int R = RG_MP3_RUN;
    __shared__ float xrb[2][2][576];   // [buffer][channel][line]
    __shared__ __attribute__((aligned(16))) rg_mp3_unit Ub[3][2];
    __shared__ int band_nz[64];
    __shared__ short band_mode[64];
    __shared__ float c12[12][6], wn[4][36], cs_l[8], ca_l[8];
    __shared__ uint8_t ptab[24];
    __shared__ float ovl[18][64];
    __shared__ float gain_l[RG_MP3_GAIN_Q_MAX - RG_MP3_GAIN_Q_MIN + 1];
    constexpr int kPowLds = 704;       // x^(4/3) for the values that occur (what four blocks per CU leave room for); larger ones
                                       // go to the table in memory: a wait the wave cannot hide, once per round that has one
    __shared__ float pow_l[kPowLds];
    __shared__ uint16_t sfbl_l[24], sfbs_l[16];
    __shared__ float gtab[2][64];
    __shared__ __attribute__((aligned(4))) uint8_t sidx_l[576];
    __shared__ float Sin[2][2][18][33];  // [granule parity][channel][time slot][subband]: wave 0 -> wave 2
    __shared__ float Ar[2][2][18][33];   // the DCT outputs of those slots (word 32 = 0.0f = V[16]): wave 2 -> wave 3
    constexpr int NT = RG_MP3_BH_THREADS;
    const int tid = threadIdx.x;
    uint32_t ti = 0;
    {
        uint32_t lo = 0, hi = n_tracks - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (tracks[mid].run_base <= blockIdx.x) lo = mid; else hi = mid - 1;
        }
        ti = lo;
    }
    const RgMp3DevTrack tr = tracks[ti];
    const uint32_t g0 = (blockIdx.x - tr.run_base) * R;
    if (g0 >= tr.n_granules) return;  // past what the device-side frame parser found decodable (block-uniform)
    const int ng = (int)(tr.n_granules - g0 < (uint32_t)R ? tr.n_granules - g0 : (uint32_t)R);
    const int nch = (int)tr.channels;
    const int rr = (int)tr.rate_row;
    const int gi0 = g0 > 0 ? -2 : 0;   // first granule of the pipeline, relative to g0 (g0 is a multiple of R >= 2)
    const int pd0 = g0 > 0 ? 1 : 0;    // first pipeline granule whose subband samples are right (the one before has no overlap)
    const uint64_t ubase = tr.unit_base + (uint64_t)((long long)g0 + gi0) * nch;
    const int nsteps = ng - gi0;       // granules through the pipeline
    for (int e = tid; e < 144; e += NT) (&wn[0][0])[e] = (&T->win[0][0])[e];
    if (tid < 72) (&c12[0][0])[tid] = (&T->imdct12[0][0])[tid];
    if (tid < 8) { cs_l[tid] = T->cs[tid]; ca_l[tid] = T->ca[tid]; }
    if (tid < 24) ptab[tid] = T->pretab[tid];
    for (int e = tid; e < RG_MP3_GAIN_Q_MAX - RG_MP3_GAIN_Q_MIN + 1; e += NT) gain_l[e] = T->gain[e];
    for (int e = tid; e < kPowLds; e += NT) pow_l[e] = T->pow43[e];
    if (tid >= 32 && tid < 56) sfbl_l[tid - 32] = T->sfb_long[rr][tid - 32];
    if (tid >= 64 && tid < 80) sfbs_l[tid - 64] = T->sfb_short[rr][tid - 64];
    for (int e = tid; e < 144; e += NT) reinterpret_cast<uint32_t *>(sidx_l)[e] = reinterpret_cast<const uint32_t *>(T->short_idx_of_line[rr])[e];
    if (tid < 64) {
#pragma unroll
        for (int i = 0; i < 18; ++i) ovl[i][tid] = 0.0f;  // a granule without a predecessor adds 0.0f, as the host does
    }
    static_assert(sizeof(rg_mp3_unit) == 64, "the unit prefetch assumes 64-byte units");
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // a scalar: each wave takes one branch whole
    const int lane = tid & 63;

    if (wave == 1) {
        // ================= wave 1: units and spectra in, requantised (and stereo-processed, reordered) spectrum out ===
        // This wave is the pipeline's longest stage, and a wave issues at most one instruction in four cycles: its length is
        // its instruction count.  The body is compiled once per channel count, so that nothing in it asks how many channels
        // there are.
        // the block's step is as long as this wave's: where it shares a SIMD with the other blocks' lighter waves it goes first
        // (priorities for the other stages as well were slower)
        __builtin_amdgcn_s_setprio(3);
        auto requant_wave = [&](auto nch_c) {
        constexpr int nch = decltype(nch_c)::value;
        constexpr int kRounds = 3;
        uint32_t rq_lb[kRounds] = {0u, 0u, 0u};  // long-block band numbers of a piece's four lines
#pragma unroll
        for (int r = 0; r < kRounds; ++r)
            if (lane + 64 * r < 144) rq_lb[r] = *reinterpret_cast<const uint32_t *>(&T->long_band_of_line[rr][4 * (lane + 64 * r)]);
        const bool uq = lane >= 56 && lane < 56 + 4 * nch;  // lanes that carry the units: four 16-byte words each
        const int uq_t = lane - 56;
        uint4 u_reg = make_uint4(0u, 0u, 0u, 0u), u_first = make_uint4(0u, 0u, 0u, 0u);
        if (uq) {
            u_first = reinterpret_cast<const uint4 *>(units + ubase)[uq_t];
            reinterpret_cast<uint4 *>(&Ub[0][0])[uq_t] = u_first;
            if (nsteps > 1) u_reg = reinterpret_cast<const uint4 *>(units + ubase + nch)[uq_t];
        }
        // A unit's last sixteen bytes -- nz, global_gain, block_type | mixed, subblock_gain | scalefac_scale, preflag,
        // long_end, short_start | mode_ext, ... -- are the same for the whole wave and sit in lane 59 + 4 c of the units' way
        // in: they go to scalar registers from there (v_readlane), one step ahead, so that the gains of a granule cost one
        // look-up of the scalefactor and one of the gain instead of a chain of byte reads from the unit in LDS.
        static_assert(offsetof(rg_mp3_unit, nz) == 48 && offsetof(rg_mp3_unit, mixed) == 52 && offsetof(rg_mp3_unit, scalefac_scale) == 56 &&
                      offsetof(rg_mp3_unit, long_end) == 58 && offsetof(rg_mp3_unit, mode_ext) == 60, "unit header layout");
        uint32_t h_next[2][4];
        auto header_of = [&](const uint4 &reg, uint32_t (&h)[2][4]) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                h[c][0] = (uint32_t)__builtin_amdgcn_readlane((int)reg.x, 59 + 4 * c);
                h[c][1] = (uint32_t)__builtin_amdgcn_readlane((int)reg.y, 59 + 4 * c);
                h[c][2] = (uint32_t)__builtin_amdgcn_readlane((int)reg.z, 59 + 4 * c);
                h[c][3] = (uint32_t)__builtin_amdgcn_readlane((int)reg.w, 59 + 4 * c);
            }
        };
        header_of(u_first, h_next);
        // what a lane's gain depends on besides the unit: its pretab entry, its (band, window), its window's byte of the
        // subblock gains; the band limits long_end / short_start can take (0, 6, 8, 22 / 0, 3, 13) as line numbers
        const int gq_kk = lane - 22, gq_band = gq_kk / 3, gq_win = gq_kk - 3 * gq_band;
        const int gq_pt = lane < 22 ? (int)T->pretab[lane] : 0;
        const int gq_sh = 8 * (1 + (gq_win < 0 ? 0 : gq_win));
        const int ll6 = __builtin_amdgcn_readfirstlane((int)T->sfb_long[rr][6]), ll8 = __builtin_amdgcn_readfirstlane((int)T->sfb_long[rr][8]);
        const int ll22 = __builtin_amdgcn_readfirstlane((int)T->sfb_long[rr][22]);
        const int so3 = __builtin_amdgcn_readfirstlane(3 * (int)T->sfb_short[rr][3]), so13 = __builtin_amdgcn_readfirstlane(3 * (int)T->sfb_short[rr][13]);
        uint2 rq_next[kRounds][2];
#pragma unroll
        for (int r = 0; r < kRounds; ++r) rq_next[r][0] = rq_next[r][1] = make_uint2(0u, 0u);
        // 16-byte piece `chunk` of unit U sits at ((U / G) 72 + chunk) G + U % G (rg_mp3_is_index): a lane's share of that
        // is fixed, the unit's share is a scalar
        uint32_t rq_off[kRounds];  // bytes from the unit's first piece to this lane's four lines
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {  // the third round's idle lanes read its last piece again: no branch around a load
            const uint32_t piece = (uint32_t)(lane + 64 * r < 144 ? lane + 64 * r : 143);
            rq_off[r] = ((piece >> 1) << (is_group_log2 + 4)) + 8u * (piece & 1u);
        }
        // the block's spectra start at the group its first unit lies in; from there 32-bit offsets do (a run is 68 units)
        const uint32_t group_mask = (1u << is_group_log2) - 1u;
        const uint32_t u_in_group0 = (uint32_t)(ubase & group_mask);
        const uint8_t *const is_block = reinterpret_cast<const uint8_t *>(is) + (ubase >> is_group_log2) * ((uint64_t)(72 * 16) << is_group_log2);
        auto fetch_spectra = [&](const int step) {  // every load of it is issued whatever the step: the compiler can count them
#pragma unroll
            for (int c = 0; c < nch; ++c) {
                const uint32_t ul = u_in_group0 + (uint32_t)step * nch + c;
                const uint8_t *const row = is_block + ((((ul >> is_group_log2) * 72u) << is_group_log2) + (ul & group_mask)) * 16u;
#pragma unroll
                for (int r = 0; r < kRounds; ++r) rq_next[r][c] = *reinterpret_cast<const uint2 *>(row + rq_off[r]);
            }
        };
        auto wave_sync = [] {  // LDS hand-over between lanes of this wave
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        };
        fetch_spectra(0);
        rg_lds_barrier();
        for (int k = 0; k <= nsteps + 2; ++k) {
            RG_BH_STAMP(1, k, 0);
            if (k < nsteps) {
                float (*const XP)[576] = xrb[k & 1];
                const rg_mp3_unit *const UP = Ub[k % 3];
                // the units of step k + 1 become visible at this step's barrier; those of step k + 2 start travelling
                uint32_t h[2][4];
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) h[c][i] = h_next[c][i];
                // (past the run's last granule the same units and spectra are asked for again and never used)
                if (uq) reinterpret_cast<uint4 *>(&Ub[(k + 1) % 3][0])[uq_t] = u_reg;
                header_of(u_reg, h_next);
                // ---- stage B: requantisation (rg_mp3dec.cpp: requantize)
                uint2 raw[kRounds][2];
#pragma unroll
                for (int r = 0; r < kRounds; ++r) { raw[r][0] = rq_next[r][0]; raw[r][1] = rq_next[r][1]; }
                // The lines from a unit's nz on are zero and the Huffman stage does not write them (it completes the
                // 16-byte piece the last value falls into): what was fetched from there is replaced by zeros.
                int nz8_max = 0;  // lines from here on are zero in every channel
#pragma unroll
                for (int c = 0; c < nch; ++c) {
                    const int nz8 = ((int)(h[c][0] & 0xFFFFu) + 7) & ~7;
                    nz8_max = nz8 > nz8_max ? nz8 : nz8_max;
#pragma unroll
                    for (int r = 0; r < kRounds; ++r)
                        if (nz8 < 256 * (r + 1) && 4 * (lane + 64 * r) >= nz8) raw[r][c] = make_uint2(0u, 0u);  // the first test is the wave's
                }
                RG_BH_STAMP2(k, 2);
                // Order matters from here to the end of the step.  Everything this step needs from memory has arrived; the
                // loads for the steps to come are issued below and nothing later in the step may wait for memory: the compiler
                // cannot count loads across a branch and waits for ALL of them wherever a conditional load meets the code
                // after it, i.e. it would sit out a round trip of the prefetch in every step.  The one conditional read of the
                // step -- a quantised value beyond the LDS part of the x^(4/3) table takes its power from the full table in
                // memory -- therefore happens HERE, before the prefetch, and the power waits in the line's own place in the
                // output buffer (which nobody else touches before this step's barrier).
                bool big_r[kRounds];  // the wave's: some lane of round r holds such a value
                {
                    bool any_big = false;
#pragma unroll
                    for (int r = 0; r < kRounds; ++r) {
                        rg_u16x2 amax = {0, 0};
                        if (nz8_max > 256 * r) {
#pragma unroll
                            for (int c = 0; c < nch; ++c) {
                                amax = __builtin_elementwise_max(amax, __builtin_bit_cast(rg_u16x2, __builtin_elementwise_abs(__builtin_bit_cast(rg_s16x2, raw[r][c].x))));
                                amax = __builtin_elementwise_max(amax, __builtin_bit_cast(rg_u16x2, __builtin_elementwise_abs(__builtin_bit_cast(rg_s16x2, raw[r][c].y))));
                            }
                        }
                        big_r[r] = __builtin_amdgcn_ballot_w64((amax.x > amax.y ? amax.x : amax.y) >= kPowLds) != 0;
                        any_big = any_big || big_r[r];
                    }
                    if (any_big) {
#pragma unroll
                        for (int r = 0; r < kRounds; ++r) {
                            if (!big_r[r]) continue;  // a round without such a value (the usual case even here) costs no trip to memory
                            float big[2][4];
#pragma unroll
                            for (int c = 0; c < nch; ++c) {
                                const uint32_t w[2] = {raw[r][c].x, raw[r][c].y};
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const int v = (int)(int16_t)(w[j >> 1] >> (16 * (j & 1)));
                                    const int a = v < 0 ? -v : v;
                                    big[c][j] = a >= kPowLds ? T->pow43[a] : 0.0f;
                                }
                            }
                            // every lane consumes what it asked for, on every path: a load whose value is only looked at
                            // under a condition stays "in flight" in the compiler's books on the other path, and the next
                            // write to its register -- the prefetch's address arithmetic below -- then waits for everything
#pragma unroll
                            for (int c = 0; c < nch; ++c)
#pragma unroll
                                for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(big[c][j]));
                            if (lane + 64 * r < 144) {
#pragma unroll
                                for (int c = 0; c < nch; ++c)
#pragma unroll
                                    for (int j = 0; j < 4; ++j)
                                        if (big[c][j] != 0.0f) XP[c][4 * (lane + 64 * r) + j] = big[c][j];
                            }
                        }
                    }
                }
                RG_BH_STAMP2(k, 3);
                if (uq) u_reg = reinterpret_cast<const uint4 *>(units + ubase + (uint64_t)(k + 2 < nsteps ? k + 2 : nsteps - 1) * nch)[uq_t];
                fetch_spectra(k + 1 < nsteps ? k + 1 : nsteps - 1);
                int bt_s[2] = {0, 0}, ll_s[2] = {0, 0}, so_s[2] = {0, 0};
                int gq_idx[2] = {0, 0};
                uint32_t gq_sf[2] = {0u, 0u};
#pragma unroll
                for (int c = 0; c < 2; ++c) {  // the scalefactor each lane's gain needs: one byte read per channel, asked for together
                    if (c >= nch) continue;
                    const int long_end = (int)((h[c][2] >> 16) & 0xFFu), short_start = (int)(h[c][2] >> 24);
                    const int rel = gq_kk - 3 * short_start;
                    const bool short_sf = lane >= 22 && gq_band < 12 && rel >= 0;
                    gq_idx[c] = lane < 22 ? lane : (short_sf ? long_end + rel : -1);
                    gq_sf[c] = UP[c].sf[gq_idx[c] < 0 ? 0 : gq_idx[c]];
                }
                float gq_g[2] = {0.0f, 0.0f};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (c >= nch) continue;
                    const int scalefac_scale = (int)(h[c][2] & 0xFFu), preflag = (int)((h[c][2] >> 8) & 0xFFu);
                    const int long_end = (int)((h[c][2] >> 16) & 0xFFu), short_start = (int)(h[c][2] >> 24);
                    const int m4 = scalefac_scale ? 4 : 2;  // 4 * mult
                    const int base4 = (int)((h[c][0] >> 16) & 0xFFu) - 210;
                    const int sv = gq_idx[c] < 0 ? 0 : (int)gq_sf[c];
                    int q;
                    if (lane < 22) q = base4 - m4 * (sv + (preflag ? gq_pt : 0));
                    else q = base4 - 8 * (int)((h[c][1] >> gq_sh) & 0xFFu) - m4 * sv;
                    q = q < RG_MP3_GAIN_Q_MIN ? RG_MP3_GAIN_Q_MIN : (q > RG_MP3_GAIN_Q_MAX ? RG_MP3_GAIN_Q_MAX : q);
                    gq_g[c] = gain_l[q - RG_MP3_GAIN_Q_MIN];
                    bt_s[c] = (int)(h[c][0] >> 24);
                    ll_s[c] = long_end == 22 ? ll22 : (long_end == 8 ? ll8 : (long_end == 6 ? ll6 : (long_end == 0 ? 0 : (int)sfbl_l[long_end])));
                    so_s[c] = short_start >= 13 ? so13 : (short_start == 3 ? so3 : (short_start == 0 ? 0 : 3 * (int)sfbs_l[short_start]));
                }
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    if (c < nch) gtab[c][lane] = lane < 61 ? gq_g[c] : 0.0f;
                const int nz0 = (int)(h[0][0] & 0xFFFFu), nz1 = (int)(h[1][0] & 0xFFFFu);
                const int ms_n = (nch == 2 && (h[0][3] & 3u) == 2u) ? (nz0 > nz1 ? nz0 : nz1) : 0;
                wave_sync();
                RG_BH_STAMP2(k, 0);
                // This wave is the pipeline's longest stage and it runs alone on its data, so its length is the sum of its
                // LDS round trips: the look-ups of a round (8 gains, 8 powers per lane, both channels) are all asked for
                // before the first one is used, the rare value beyond the LDS part of the power table is dealt with once per
                // round, and the third round's idle lanes run along on clamped indices instead of branching.
#pragma unroll
                for (int r = 0; r < kRounds; ++r) {
                    const int piece = lane + 64 * r;
                    const int rq_l0 = 4 * (piece < 144 ? piece : 143);
                    if (nz8_max <= 256 * r) {  // the whole round lies behind the last value of every channel: zeros (mid/side of zeros included)
                        if (piece < 144) {
#pragma unroll
                            for (int c = 0; c < nch; ++c) *reinterpret_cast<float4 *>(&XP[c][4 * piece]) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        }
                        continue;
                    }
                    float gv[2][4], mg[2][4];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        if (c >= nch) continue;
                        if (bt_s[c] != 2) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) gv[c][j] = gtab[c][(rq_lb[r] >> (8 * j)) & 0xFFu];
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int line = rq_l0 + j;
                                const int idx = line < ll_s[c] ? (int)((rq_lb[r] >> (8 * j)) & 0xFFu) : 22 + (int)sidx_l[line - ll_s[c] + so_s[c]];
                                gv[c][j] = gtab[c][idx];
                            }
                        }
                        // two lines per 32-bit word: |v|, the largest of them and the index into the LDS part of the power
                        // table as packed 16-bit operations
                        const uint32_t w[2] = {raw[r][c].x, raw[r][c].y};
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const rg_u16x2 ua = __builtin_bit_cast(rg_u16x2, __builtin_elementwise_abs(__builtin_bit_cast(rg_s16x2, w[h])));
                            const rg_u16x2 top = {(unsigned short)(kPowLds - 1), (unsigned short)(kPowLds - 1)};
                            const rg_u16x2 ci = __builtin_elementwise_min(ua, top);
                            mg[c][2 * h] = pow_l[ci.x];
                            mg[c][2 * h + 1] = pow_l[ci.y];
                        }
                    }
                    if (big_r[r]) {  // rare: a value beyond the LDS part of the table
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            if (c >= nch) continue;
                            const uint32_t w[2] = {raw[r][c].x, raw[r][c].y};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int v = (int)(int16_t)(w[j >> 1] >> (16 * (j & 1)));
                                const int a = v < 0 ? -v : v;
                                if (a >= kPowLds) mg[c][j] = XP[c][rq_l0 + j];  // put there at the top of the step
                            }
                        }
                    }
                    float val[2][4];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        if (c >= nch) continue;
                        const uint32_t w[2] = {raw[r][c].x, raw[r][c].y};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            // the product is >= +0; a negative value takes its sign (-t, also of a product that underflowed):
                            // bit 31 of the word, shifted up for the low half
                            const float t = mg[c][j] * gv[c][j];
                            val[c][j] = __builtin_copysignf(t, __uint_as_float((j & 1) ? w[j >> 1] : w[j >> 1] << 16));
                        }
                    }
                    // ---- stage C, the plain case: mid/side on the lines below the longer channel's end; the end is the same
                    // for all lanes, so at most one round has lanes on both sides of it
                    if (ms_n > 256 * r) {
                        const float isq2 = 0.70710678118654752440f;
                        if (ms_n >= 256 * r + 256) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float a = val[0][j], b = val[1][j];
                                val[0][j] = (a + b) * isq2;
                                val[1][j] = (a - b) * isq2;
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (4 * piece + j < ms_n) {
                                    const float a = val[0][j], b = val[1][j];
                                    val[0][j] = (a + b) * isq2;
                                    val[1][j] = (a - b) * isq2;
                                }
                        }
                    }
                    if (piece < 144) {
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            if (c < nch) *reinterpret_cast<float4 *>(&XP[c][4 * piece]) = make_float4(val[c][0], val[c][1], val[c][2], val[c][3]);
                    }
                }
                RG_BH_STAMP2(k, 1);
                const bool special = (nch == 2 && (h[0][3] & 1u)) || (h[0][0] >> 24) == 2u || (nch == 2 && (h[1][0] >> 24) == 2u);
                if (special) {
                    wave_sync();
                    if (nch == 2 && (UP[0].mode_ext & 1)) {
                        // ---- intensity stereo, with mid/side on the bands it leaves (rg_mp3dec.cpp: stereo)
                        const rg_mp3_unit &u1 = UP[1];
                        const bool ms = (UP[0].mode_ext & 2) != 0;
                        const float isq2 = 0.70710678118654752440f;
                        band_nz[lane] = 0;
                        band_mode[lane] = 0;
                        wave_sync();
                        const int long_lines = (int)sfbl_l[u1.long_end];
                        const int short_off = 3 * (int)sfbs_l[u1.short_start < 13 ? u1.short_start : 13];
                        auto band_of = [&](int line) -> int {
                            if (u1.block_type != 2 || line < long_lines) return 39 + (int)T->long_band_of_line[rr][line];
                            return (int)sidx_l[line - long_lines + short_off] - 3 * (int)u1.short_start;
                        };
                        for (int line = lane; line < 576; line += 64)
                            if (XP[1][line] != 0.0f) band_nz[band_of(line)] = 1;
                        wave_sync();
                        if (lane == 0) {
                            const bool lsf = tr.lsf != 0;
                            bool found[3] = {false, false, false};
                            bool found_long = false;
                            if (u1.block_type == 2) {
                                for (int b = 12; b >= (int)u1.short_start; --b) {
                                    const int sb = b == 12 ? 11 : b;
                                    for (int w = 2; w >= 0; --w) {
                                        const int kk = (b - (int)u1.short_start) * 3 + w;
                                        const int idx = (int)u1.long_end + 3 * (sb - (int)u1.short_start) + w;
                                        bool intensity = false;
                                        int mode = 0;
                                        if (!found[w]) {
                                            if (band_nz[kk]) {
                                                found[w] = true;
                                            } else {
                                                const int p = u1.sf[idx];
                                                intensity = lsf ? !((u1.illegal >> idx) & 1ull) : p < 7;
                                                if (intensity) mode = 2 + p;
                                            }
                                        }
                                        if (!intensity && ms) mode = 1;
                                        band_mode[kk] = (short)mode;
                                    }
                                }
                                found_long = found[0] || found[1] || found[2];
                            }
                            if (!(u1.block_type == 2 && !u1.mixed)) {
                                for (int b = (int)u1.long_end - 1; b >= 0; --b) {
                                    const int sb = b == 21 ? 20 : b;
                                    bool intensity = false;
                                    int mode = 0;
                                    if (!found_long) {
                                        if (band_nz[39 + b]) {
                                            found_long = true;
                                        } else {
                                            const int p = u1.sf[sb];
                                            intensity = lsf ? !((u1.illegal >> sb) & 1ull) : p < 7;
                                            if (intensity) mode = 2 + p;
                                        }
                                    }
                                    if (!intensity && ms) mode = 1;
                                    band_mode[39 + b] = (short)mode;
                                }
                            }
                        }
                        wave_sync();
                        const int scale = u1.intensity_scale & 1;
                        for (int line = lane; line < 576; line += 64) {
                            const int mode = band_mode[band_of(line)];
                            if (mode == 1) {
                                const float a = XP[0][line], b = XP[1][line];
                                XP[0][line] = (a + b) * isq2;
                                XP[1][line] = (a - b) * isq2;
                            } else if (mode >= 2) {
                                const int pos = mode - 2;
                                float kl, kr;
                                if (!tr.lsf) {
                                    kl = T->is_l[pos];
                                    kr = T->is_r[pos];
                                } else if (pos == 0) {
                                    kl = kr = 1.0f;
                                } else if (pos & 1) {
                                    kl = T->lsf_is[scale][(pos + 1) >> 1];
                                    kr = 1.0f;
                                } else {
                                    kl = 1.0f;
                                    kr = T->lsf_is[scale][pos >> 1];
                                }
                                const float v = XP[0][line];
                                XP[0][line] = v * kl;
                                XP[1][line] = v * kr;
                            }
                        }
                        wave_sync();
                    }
                    // ---- short blocks: bitstream order [band][window][line] -> [line][window] (rg_mp3dec.cpp: reorder),
                    // in place: a lane's nine lines wait in registers while the wave's reads finish
                    for (int c = 0; c < nch; ++c) {
                        const rg_mp3_unit &u = UP[c];
                        if (u.block_type != 2) continue;
                        float *X = XP[c];
                        const int long_lines = u.mixed ? (int)sfbl_l[u.long_end] : 0;
                        const int short_off = 3 * (int)sfbs_l[u.short_start];
                        float keep9[9];
#pragma unroll
                        for (int i = 0; i < 9; ++i) {
                            const int line = lane + 64 * i;
                            keep9[i] = line < long_lines ? X[line] : X[(int)T->short_reorder_src[rr][line - long_lines + short_off] - short_off + long_lines];
                        }
                        wave_sync();
#pragma unroll
                        for (int i = 0; i < 9; ++i) X[lane + 64 * i] = keep9[i];
                        wave_sync();
                    }
                }
            }
            RG_BH_STAMP(1, k, 1);
            rg_lds_barrier();
        }
        };
        if (nch == 2) requant_wave(std::integral_constant<int, 2>{});
        else requant_wave(std::integral_constant<int, 1>{});
        return;
    }

    if (wave == 0) {
        // ================= wave 0: lane (channel, subband): spectrum -> subband samples of eighteen time slots =========
        const bool active = lane < 32 * nch;
        const int my_c = lane >> 5, my_sb = lane & 31;
        rg_lds_barrier();
        for (int k = 0; k <= nsteps + 2; ++k) {
            RG_BH_STAMP(0, k, 0);
            if (k >= 1 && k <= nsteps && active) {
                // ---- stage D of granule k - 1 (rg_mp3dec.cpp: antialias, hybrid).  The butterflies between subbands
                // sb - 1 | sb, sb = 1 .. nb: this lane evaluates its own half of the two it touches.  Windowed sample i:
                // the first eighteen are added to the overlap and leave (frequency inversion: odd samples of odd subbands
                // change sign), the second eighteen are the next granule's overlap.
                const int q = k - 1;
                const rg_mp3_unit &u = Ub[q % 3][my_c];
                const float *const X = xrb[q & 1][my_c];
                const int bt = (u.block_type == 2 && u.mixed && my_sb < 2) ? 0 : (int)u.block_type;
                const int nb = u.block_type == 2 ? (u.mixed ? 1 : 0) : 31;
                float xs[18];
                const float *Xr = X + my_sb * 18;
#pragma unroll
                for (int i = 0; i < 18; ++i) xs[i] = Xr[i];
                if (my_sb >= 1 && my_sb <= nb) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float a = Xr[-1 - i], b = xs[i];
                        xs[i] = b * cs_l[i] + a * ca_l[i];
                    }
                }
                if (my_sb < nb) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float a = xs[17 - i], b = Xr[18 + i];
                        xs[17 - i] = a * cs_l[i] - b * ca_l[i];
                    }
                }
                float *const dst = &Sin[q & 1][my_c][0][my_sb];
                const bool flip = (my_sb & 1) != 0;
                auto emit = [&](const int i, const float val) {
                    if (i < 18) {
                        float v = val + ovl[i][lane];
                        if (flip && (i & 1)) v = -v;
                        dst[i * 33] = v;
                    } else {
                        ovl[i - 18][lane] = val;
                    }
                };
                if (bt != 2) {
                    struct Sink {
                        decltype(emit) &f;
                        struct Ref {
                            decltype(emit) &f;
                            int i;
                            __device__ __forceinline__ void operator=(float v) { f(i, v); }
                        };
                        __device__ __forceinline__ Ref operator[](int i) { return Ref{f, i}; }
                    } sink{emit};
                    rg_mp3_imdct36_windowed(xs, wn[bt], sink);
                } else {
#pragma unroll 1
                    for (int i = 0; i < 36; ++i) {
                        float raw = 0.0f;
#pragma unroll
                        for (int w = 0; w < 3; ++w) {
                            const int ii = i - 6 - 6 * w;
                            if (ii >= 0 && ii < 12) {
                                float s2 = 0.0f;
#pragma unroll
                                for (int kk = 0; kk < 6; ++kk) s2 = rg_mp3_mac(xs[3 * kk + w], c12[ii][kk], s2);
                                raw = rg_mp3_mac(s2, wn[2][ii], raw);
                            }
                        }
                        if (i < 18) {
                            float v = raw + ovl[i][lane];
                            if (flip && (i & 1)) v = -v;
                            dst[i * 33] = v;
                        } else {
                            ovl[i - 18][lane] = raw;
                        }
                    }
                }
            }
            RG_BH_STAMP(0, k, 1);
            rg_lds_barrier();
        }
        return;
    }

    if (wave == 2) {
        // ================= wave 2: matrixing, lane (channel, time slot): the 32-point DCT (rg_mp3dec.cpp: synth) =======
        const int c = lane >> 5, t = lane & 31;
        const bool active = t < 18 && c < nch;
        rg_lds_barrier();
        for (int k = 0; k <= nsteps + 2; ++k) {
            RG_BH_STAMP(2, k, 0);
            const int p = k - 2;
            if (p >= pd0 && p < nsteps && active) {
                float x[32], A[32];
                const float *src = &Sin[p & 1][c][t][0];
#pragma unroll
                for (int i = 0; i < 32; ++i) x[i] = src[i];
                RgMp3Dct<32>::run(x, A, T->sec);
                float *dstA = &Ar[p & 1][c][t][0];
#pragma unroll
                for (int i = 0; i < 32; ++i) dstA[i] = A[i];
                dstA[32] = 0.0f;
            }
            RG_BH_STAMP(2, k, 1);
            rg_lds_barrier();
        }
        return;
    }

    {
        // ================= wave 3: the 512-tap window, lane (channel, j) -> PCM sample j of every time slot ============
        // PCM sample j of time slot r is  sum_{i<8} V[r-2i][j] D[64i+j] + V[r-2i-1][32+j] D[64i+32+j]  in that order (the
        // host's); V[.][j] and V[.][32+j] are two fixed columns of the DCT rows with the signs folded into the window
        // coefficients: V[.][j] = A[.][16+j] (j < 16), 0 (j = 16), -A[.][48-j] (j > 16); V[.][32+j] = -A[.][16-j] (j < 16),
        // -A[.][0] (j = 16), -A[.][j-16] (j > 16) (fma(-a, d, s) and fma(a, -d, s) are the same bits).  cA / cB [q]: those
        // columns of time slot q - 15 relative to the granule.
        const int c = lane >> 5, wj = lane & 31;
        const bool active = c < nch;
        float D1[8], D2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float da = T->D[i * 64 + wj], db = T->D[i * 64 + 32 + wj];
            D1[i] = wj > 16 ? -da : da;
            D2[i] = -db;
        }
        const int col1 = wj < 16 ? 16 + wj : (wj == 16 ? 32 : 48 - wj);
        const int col2 = wj < 16 ? 16 - wj : wj - 16;
        float cA[33], cB[33];
#pragma unroll
        for (int q = 0; q < 33; ++q) cA[q] = cB[q] = 0.0f;  // the slots before the track are silence
        float *const plane = c == 0 ? tr.ch0 : tr.ch1;
        rg_lds_barrier();
        for (int k = 0; k <= nsteps + 2; ++k) {
            RG_BH_STAMP(3, k, 0);
            const int p = k - 3;
            if (p >= pd0 && p < nsteps && active) {
                const float *rows = &Ar[p & 1][c][0][0];
#pragma unroll
                for (int q = 0; q < 18; ++q) {
                    cA[15 + q] = rows[q * 33 + col1];
                    cB[15 + q] = rows[q * 33 + col2];
                }
                if (p + gi0 >= 0) {
                    float *__restrict__ dst = plane + ((size_t)g0 + (size_t)(p + gi0)) * 576 + wj;
                    // two time slots per instruction: the same sixteen fused multiply-adds per output, in the same order, as
                    // packed operations (v_pk_fma_f32: both halves are the IEEE fma the host's rg_mp3_mac is)
#pragma unroll
                    for (int q = 0; q < 18; q += 2) {
                        rg_f32x2 s2 = {0.0f, 0.0f};
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            s2 = __builtin_elementwise_fma(rg_f32x2{cA[q + 15 - 2 * i], cA[q + 16 - 2 * i]}, rg_f32x2{D1[i], D1[i]}, s2);
                            s2 = __builtin_elementwise_fma(rg_f32x2{cB[q + 14 - 2 * i], cB[q + 15 - 2 * i]}, rg_f32x2{D2[i], D2[i]}, s2);
                        }
                        dst[(size_t)q * 32] = s2.x;
                        dst[(size_t)(q + 1) * 32] = s2.y;
                    }
                }
#pragma unroll
                for (int q = 0; q < 15; ++q) { cA[q] = cA[q + 18]; cB[q] = cB[q + 18]; }
            }
            RG_BH_STAMP(3, k, 1);
            rg_lds_barrier();
        }
    }
}

// =====================================================================================================================
// Stage A's heavy part on the device: scalefactors + Huffman-coded spectrum (rg_mp3dec.cpp: read_scalefactors_v1 /
// read_scalefactors_lsf / decode_spectrum), one thread per granule and channel.  Integer work throughout: the outputs
// (576 int16 per unit + one rg_mp3_unit) are the values rg_mp3_parse_units writes on the host; the spectra are laid out for
// this kernel's stores (rg_mp3_is_index).
//
// The decode is a chain of dependent look-ups per symbol, so everything a symbol touches is kept close: the code tables
// of all 32 table numbers sit in LDS (31 KB, copied once per block of 512 units), the bit stream is read through three
// registers of 32 bits with the next word always in flight, scalefactors are parsed into an LDS column per thread, and
// the spectrum leaves in 16-byte stores that the eight lanes of a group put side by side.  Granule 1 of an MPEG-1 frame may reuse granule 0's scalefactors (scfsi): its
// thread then parses granule 0's scalefactors first, which is cheaper than chaining the two granules in one thread.
namespace {

constexpr int kHuffThreads = 512;  // 1024 (eight waves per SIMD instead of six) is slower: 0.44 against 0.38 ms on the dense stream

// 96 bits of the track's main data around the read position, big-endian words; bits at or past `limit` (the end of the
// frame's own main data) read as zero, which is what the host decoder's private copy of the frame's data does.  All
// positions are 32-bit bit counts from the 32-bit word the granule starts in (a granule is at most 4095 bits long, its
// frame's data ends at most a few thousand bits later).
struct BitCache {
    const uint32_t *__restrict__ w;  // the word the granule starts in (the track's main data is 4-byte aligned)
    uint32_t pos, end;               // read position, end of the granule's bits
    int32_t limit;                   // end of the frame's own data; may lie before `pos` in damaged streams
    uint32_t idx;                    // word index of w0
    uint32_t w0, w1, w2, w3;         // words idx .. idx + 2 (and idx + 3 while idx is even): fetched in aligned pairs
    __device__ __forceinline__ uint32_t mask_word(uint32_t raw, uint32_t i) const {
        const int32_t valid = limit - (int32_t)(i << 5);
        if (valid <= 0) return 0u;
        uint32_t x = __builtin_bswap32(raw);
        if (valid < 32) x &= 0xFFFFFFFFu << (32 - valid);
        return x;
    }
    // words i and i + 1 (i even, relative to an 8-byte aligned base): one 8-byte load -- the stream is read in half as
    // many requests as word by word, and every request of a thread costs a line fetch when 400 000 threads stream at once
    __device__ __forceinline__ void load_pair(uint32_t i, uint32_t *a, uint32_t *b) const {
        if (limit - (int32_t)(i << 5) <= 0) { *a = 0u; *b = 0u; return; }
        const uint2 raw = *reinterpret_cast<const uint2 *>(w + i);
        *a = mask_word(raw.x, i);
        *b = mask_word(raw.y, i + 1);
    }
    // granule at absolute bit `bit_off` of the stream `base`, `length` bits long, frame data ending at `frame_end_bit`
    __device__ __forceinline__ void open(const uint8_t *base, uint64_t bit_off, uint32_t length, uint64_t frame_end_bit) {
        const uint64_t word0 = (bit_off >> 6) << 1;  // even: the pairs are 8-byte aligned (the stream is)
        w = reinterpret_cast<const uint32_t *>(base) + word0;
        pos = (uint32_t)(bit_off - (word0 << 5));
        end = pos + length;
        const int64_t lim = (int64_t)frame_end_bit - (int64_t)(word0 << 5);
        limit = lim < -(1 << 30) ? -(1 << 30) : (lim > (1 << 30) ? (1 << 30) : (int32_t)lim);
        idx = pos >> 5;  // 0 or 1
        uint32_t a, b, c, d;
        load_pair(0, &a, &b);
        load_pair(2, &c, &d);
        if (idx == 0) { w0 = a; w1 = b; w2 = c; w3 = d; }
        else { w0 = b; w1 = c; w2 = d; w3 = 0u; }
    }
    // the 32 bits at the read position
    __device__ __forceinline__ uint32_t window() const {
        const uint64_t two = ((uint64_t)w0 << 32) | (uint64_t)w1;
        return (uint32_t)((two << (pos & 31)) >> 32);
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return window() >> (32 - n); }  // 1 <= n <= 32
    __device__ __forceinline__ void skip(int n) {  // n <= 32
        pos += (uint32_t)n;
        if ((pos >> 5) != idx) {
            ++idx;
            w0 = w1;
            w1 = w2;
            w2 = w3;
            if ((idx & 1u) == 0) load_pair(idx + 2, &w2, &w3);  // idx even again: the next aligned pair
        }
    }
    __device__ __forceinline__ uint32_t get(int n) {
        if (n == 0) return 0;
        const uint32_t v = peek(n);
        skip(n);
        return v;
    }
    __device__ __forceinline__ uint32_t get1() { return get(1); }
};

// Scalefactors of one granule into the thread's LDS column sf[i * kHuffThreads] (rg_mp3dec.cpp: read_scalefactors_v1 /
// read_scalefactors_lsf).  `reuse` = granule 1 of an MPEG-1 long block whose column already holds granule 0's values:
// the groups flagged in scfsi keep them.
__device__ __forceinline__ void huff_scalefactors(BitCache &b, const RgMp3HuffRec &r, bool lsf, bool reuse, uint8_t *__restrict__ sf,
                                                  uint64_t *illegal, int *preflag) {
    constexpr int S = kHuffThreads;
    *illegal = 0;
    *preflag = r.preflag;
    if (!lsf) {
        const int sc = r.scalefac_compress & 15;
        const int s1 = (int)((0x4433322211130000ull >> (4 * sc)) & 15);   // {0,0,0,0,3,1,1,1,2,2,2,3,3,3,4,4}
        const int s2 = (int)((0x3232132132103210ull >> (4 * sc)) & 15);   // {0,1,2,3,0,1,2,3,1,2,3,1,2,3,2,3}
        if (r.block_type == 2) {
            int i = 0;
            if (r.mixed) {
                for (; i < 17; ++i) sf[i * S] = (uint8_t)b.get(s1);
                for (int k = 0; k < 18; ++k) sf[(i++) * S] = (uint8_t)b.get(s2);
            } else {
                for (int k = 0; k < 18; ++k) sf[(i++) * S] = (uint8_t)b.get(s1);
                for (int k = 0; k < 18; ++k) sf[(i++) * S] = (uint8_t)b.get(s2);
            }
            for (; i < 40; ++i) sf[i * S] = 0;
        } else {
            for (int k = 0; k < 4; ++k) {
                const int lo = k == 0 ? 0 : 1 + 5 * k, hi = 6 + 5 * k;  // bands 0-5, 6-10, 11-15, 16-20
                const int bits = k < 2 ? s1 : s2;
                if (reuse && ((r.scfsi >> k) & 1)) continue;
                for (int band = lo; band < hi; ++band) sf[band * S] = (uint8_t)b.get(bits);
            }
            for (int i = 21; i < 40; ++i) sf[i * S] = 0;
        }
    } else {
        int slen[4], set;
        int sfc = r.scalefac_compress;
        *preflag = 0;
        if (!r.intensity_right) {
            if (sfc < 400) { slen[0] = (sfc >> 4) / 5; slen[1] = (sfc >> 4) % 5; slen[2] = (sfc & 15) >> 2; slen[3] = sfc & 3; set = 0; }
            else if (sfc < 500) { sfc -= 400; slen[0] = (sfc >> 2) / 5; slen[1] = (sfc >> 2) % 5; slen[2] = sfc & 3; slen[3] = 0; set = 1; }
            else { sfc -= 500; slen[0] = sfc / 3; slen[1] = sfc % 3; slen[2] = 0; slen[3] = 0; set = 2; *preflag = 1; }
        } else {
            sfc >>= 1;
            if (sfc < 180) { slen[0] = sfc / 36; slen[1] = (sfc % 36) / 6; slen[2] = (sfc % 36) % 6; slen[3] = 0; set = 3; }
            else if (sfc < 244) { sfc -= 180; slen[0] = (sfc & 0x3F) >> 4; slen[1] = (sfc & 0xF) >> 2; slen[2] = sfc & 3; slen[3] = 0; set = 4; }
            else { sfc -= 244; slen[0] = sfc / 3; slen[1] = sfc % 3; slen[2] = 0; slen[3] = 0; set = 5; }
        }
        static const uint8_t kPart[6][3][4] = {
            {{6, 5, 5, 5}, {9, 9, 9, 9}, {6, 9, 9, 9}},   {{6, 5, 7, 3}, {9, 9, 12, 6}, {6, 9, 12, 6}},
            {{11, 10, 0, 0}, {18, 18, 0, 0}, {15, 18, 0, 0}}, {{7, 7, 7, 0}, {12, 12, 12, 0}, {6, 15, 12, 0}},
            {{6, 6, 6, 3}, {12, 9, 9, 6}, {6, 12, 9, 6}},  {{8, 8, 5, 0}, {15, 12, 9, 0}, {6, 18, 9, 0}}};
        const int kind = r.block_type == 2 ? (r.mixed ? 2 : 1) : 0;
        int i = 0;
        for (int k = 0; k < 4; ++k) {
            const int n = kPart[set][kind][k];
            for (int q = 0; q < n; ++q, ++i) {
                const int v = (int)b.get(slen[k]);
                sf[i * S] = (uint8_t)v;
                if (r.intensity_right && slen[k] > 0 && v == (1 << slen[k]) - 1) *illegal |= 1ull << i;
            }
        }
        for (; i < 40; ++i) sf[i * S] = 0;
    }
}

// the spectrum leaves four words (eight lines) at a time, into the interleaved layout of rg_mp3_is_index
struct RowOut {
    uint4 *__restrict__ row;  // the unit's first piece; its 72 pieces are 2^RG_MP3_IS_GROUP_LOG2 pieces apart
    uint32_t a, b, c;
    // `line` even; words arrive in order, one per pair of lines, from line 0 on.  The last three wait in a shift register: no
    // choice of a slot per word (three branches per call in a loop whose lanes sit at different lines), the fourth word of a
    // piece finds the other three in place.
    __device__ __forceinline__ void put(int line, uint32_t word) {
        if (((line >> 1) & 3) == 3) row[(line >> 3) << RG_MP3_IS_GROUP_LOG2] = make_uint4(a, b, c, word);
        a = b;
        b = c;
        c = word;
    }
    // zeros from `line` (even) to the end of its 16-byte piece; the pieces behind it stay unwritten: the back half does not
    // read past the unit's nz (rounded up to a piece)
    __device__ __forceinline__ void finish(int line) {
        for (; (line & 7) != 0 && line < 576; line += 2) put(line, 0u);
    }
};

}  // namespace

#ifdef RG_HF_TIMING
// Instrument (tools/bh_timing.py --huffman): shader-clock stamps of the eight waves of block RG_HF_TIMING at the kernel's phases
__device__ unsigned long long rg_hf_dbg[8][8];
extern "C" int rg_hf_dbg_read(void *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rg_hf_dbg), sizeof(unsigned long long) * 64); }
#define RG_HF_STAMP(e) do { if (blockIdx.x == RG_HF_TIMING && (threadIdx.x & 63) == 0) rg_hf_dbg[threadIdx.x >> 6][e] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RG_HF_STAMP(e) do { } while (0)
#endif
__global__ void __launch_bounds__(kHuffThreads)
rg_mp3_huffman_kernel(const RgMp3DevTables *__restrict__ T, const RgMp3DevHuff *__restrict__ H,
                      const RgMp3DevTrack *__restrict__ tracks, uint32_t n_tracks, const RgMp3HuffRec *__restrict__ recs,
                      const uint8_t *__restrict__ main, rg_mp3_unit *__restrict__ units, int16_t *__restrict__ is, uint64_t total_units) {
    __shared__ uint32_t E_all[RG_MP3_HUFF_LDS_ENTRIES];
    __shared__ uint32_t t_base[32];
    __shared__ uint8_t t_pbits[32], t_linbits[32], quadA[64];
    __shared__ uint8_t sf_all[40 * kHuffThreads];
    const int tid = threadIdx.x;
    RG_HF_STAMP(0);
    const uint32_t n_e = H->n_entries;  // <= RG_MP3_HUFF_LDS_ENTRIES: checked when the tables are uploaded
    for (uint32_t i = tid; i < n_e; i += kHuffThreads) E_all[i] = H->e[i];
    if (tid < 32) { t_base[tid] = H->base[tid]; t_pbits[tid] = H->primary_bits[tid]; t_linbits[tid] = H->linbits[tid]; }
    if (tid < 64) quadA[tid] = H->quadA[tid];
    __syncthreads();
    RG_HF_STAMP(1);
    const uint64_t u = (uint64_t)blockIdx.x * kHuffThreads + (uint64_t)tid;
    if (u >= total_units) return;
    const uint32_t ti = find_by_unit(tracks, n_tracks, u);
    const RgMp3DevTrack tr = tracks[ti];
    const int nch = (int)tr.channels, rr = (int)tr.rate_row;
    const uint64_t local = u - tr.unit_base;
    if (local >= (uint64_t)tr.n_granules * nch) return;  // past what the frame parser found decodable
    const RgMp3HuffRec r = recs[u];
    uint8_t *__restrict__ sf = sf_all + tid;
    BitCache b;  // the records' bit offsets are relative to the track's stream
    uint64_t illegal = 0;
    int preflag = 0;
    const bool reuse = !tr.lsf && r.gr == 1 && r.block_type != 2 && r.scfsi != 0;
    if (reuse) {  // granule 0 of the same frame and channel sits nch units back
        const RgMp3HuffRec r0 = recs[u - nch];
        b.open(main + tr.main_base, r0.bit_off, r0.part2_3_length, r0.frame_end_bit);
        huff_scalefactors(b, r0, false, false, sf, &illegal, &preflag);
    }
    b.open(main + tr.main_base, r.bit_off, r.part2_3_length, r.frame_end_bit);
    huff_scalefactors(b, r, tr.lsf != 0, reuse, sf, &illegal, &preflag);
    RG_HF_STAMP(2);
    // ---- band layout and big_values regions (rg_mp3dec.cpp: parse_side_info, derived part) ----
    int long_end, short_start;
    if (r.block_type == 2) {
        if (r.mixed) { long_end = rr <= 2 ? 8 : 6; short_start = 3; }
        else { long_end = 0; short_start = 0; }
    } else { long_end = 22; short_start = 13; }
    const int bv2 = (int)r.big_values * 2;
    int r0e, r1e;
    if (r.block_type != 0) {
        r0e = r.block_type == 2 ? 3 * (int)T->sfb_short[rr][3] : (int)T->sfb_long[rr][8];
        r1e = 576;
    } else {
        const int i0 = r.region0_count + 1, i1 = r.region0_count + r.region1_count + 2;
        r0e = T->sfb_long[rr][i0 > 22 ? 22 : i0];
        r1e = T->sfb_long[rr][i1 > 22 ? 22 : i1];
    }
    // ---- Huffman-coded spectrum ----
    RowOut out{reinterpret_cast<uint4 *>(is) + rg_mp3_is_index(u, 0, RG_MP3_IS_GROUP_LOG2), 0u, 0u, 0u};
    int line = 0;
    // One loop over the big_values pairs of all three regions: which table a pair uses is a per-lane choice made with
    // selects, so the lanes of a wave -- which sit in different regions of different granules -- run the same instructions
    // instead of taking turns through three region loops.
    {
        const int e0 = r0e < bv2 ? r0e : bv2, e1 = r1e < bv2 ? r1e : bv2;
        const int t0 = r.table_select[0], t1 = r.table_select[1], t2 = r.table_select[2];
        const uint32_t tb0 = t_base[t0], tb1 = t_base[t1], tb2 = t_base[t2];
        const int P0 = t_pbits[t0], P1 = t_pbits[t1], P2 = t_pbits[t2];
        const int lb0 = t_linbits[t0], lb1 = t_linbits[t1], lb2 = t_linbits[t2];
        while (line < bv2) {
            const bool in0 = line < e0, in1 = line < e1;
            const int P = in0 ? P0 : (in1 ? P1 : P2);
            uint32_t word = 0u;
            if (P != 0 && b.pos < b.end) {  // table 0 codes nothing; a granule whose bits have run out is zeros from here on
                const uint32_t *__restrict__ E = E_all + (in0 ? tb0 : (in1 ? tb1 : tb2));
                const int linbits = in0 ? lb0 : (in1 ? lb1 : lb2);
                // code, escapes and signs come out of one 32-bit window whenever they fit (a code is at most 19 bits long)
                const uint32_t win = b.window();
                uint32_t e = E[win >> (32 - P)];
                int used = 0;
                if (e & 0x80000000u) {
                    e = E[((e >> 8) & 0x7FFFFF) + ((win << P) >> (32 - (int)(e & 0xFF)))];
                    used = P;
                }
                used += (int)(e & 0xFF);
                int x = (int)((e >> 12) & 15), y = (int)((e >> 8) & 15);
                if (linbits == 0 || (x != 15 && y != 15)) {
                    // a sign bit follows each non-zero value: selects, no branches (a code is at most 19 bits long)
                    const int nx = x != 0;
                    x = (((win << used) >> 31) & (uint32_t)nx) ? -x : x;
                    used += nx;
                    const int ny = y != 0;
                    y = (((win << used) >> 31) & (uint32_t)ny) ? -y : y;
                    used += ny;
                    b.skip(used);
                } else {
                    b.skip(used);
                    if (x) {
                        if (x == 15) x += (int)b.get(linbits);
                        if (b.get1()) x = -x;
                    }
                    if (y) {
                        if (y == 15) y += (int)b.get(linbits);
                        if (b.get1()) y = -y;
                    }
                }
                word = ((uint32_t)x & 0xFFFFu) | ((uint32_t)y << 16);
            }
            out.put(line, word);
            line += 2;
        }
    }
    RG_HF_STAMP(3);
    while (line <= 572 && b.pos < b.end) {
        const uint32_t win = b.window();
        int v, used;
        if (r.count1table) {
            v = (int)(~(win >> 28)) & 15;
            used = 4;
        } else {
            const uint8_t q = quadA[win >> 26];
            used = q >> 4;
            v = q & 15;
        }
        int q4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // 0, or +-1 with the sign in the next bit of the stream: no branches
            const int bit = (v >> (3 - k)) & 1;
            q4[k] = bit - 2 * (bit & (int)((win << used) >> 31));
            used += bit;
        }
        b.skip(used);
        if (b.pos > b.end) break;  // the quadruple ran past the granule's bits: stuffing, not data
        out.put(line, ((uint32_t)q4[0] & 0xFFFFu) | ((uint32_t)q4[1] << 16));
        out.put(line + 2, ((uint32_t)q4[2] & 0xFFFFu) | ((uint32_t)q4[3] << 16));
        line += 4;
    }
    RG_HF_STAMP(4);
    const int nz = line;
    out.finish(line);
    // ---- the unit ----
    rg_mp3_unit o;
#pragma unroll
    for (int i = 0; i < 40; ++i) o.sf[i] = sf[i * kHuffThreads];
    o.illegal = illegal;
    o.nz = (uint16_t)nz;
    o.global_gain = r.global_gain;
    o.block_type = r.block_type;
    o.mixed = r.mixed;
    o.subblock_gain[0] = r.subblock_gain[0]; o.subblock_gain[1] = r.subblock_gain[1]; o.subblock_gain[2] = r.subblock_gain[2];
    o.scalefac_scale = r.scalefac_scale;
    o.preflag = (uint8_t)preflag;
    o.long_end = (uint8_t)long_end;
    o.short_start = (uint8_t)short_start;
    o.mode_ext = r.mode_ext;
    o.intensity_scale = r.intensity_scale;
    o.reserved[0] = o.reserved[1] = 0;
    units[u] = o;
    RG_HF_STAMP(5);
}

// Tuning key 6 = 3: the frame parser.  The host's walk leaves one slot per frame (header + side information,
// rg_mp3_frame.h) and, per tile of 256 frames, the main-data bytes that precede it.  Three small launches:
//   count  one block per tile: a prefix sum of the frames' main-data sizes gives each frame its place in the bit
//          reservoir, rg_mp3_frame_records -- the very code the host route runs -- decides whether it decodes; the tile's
//          number of decodable granule-channels goes to tile_units
//   scan   one block per track: prefix sum over the track's tiles; the track's decoded length replaces the upper bound
//          in its descriptor, the second channel's plane moves up behind the first, the host reads the count from `results`
//   write  only for tracks that lost a frame (the count pass writes every record where it belongs when none is dropped,
//          the scan pass tells): the count pass again, now writing the records compacted (a dropped frame leaves no gap
//          in the PCM, exactly as on the host)
namespace {
__device__ __forceinline__ uint32_t find_by_tile(const RgMp3DevTrack *__restrict__ tr, uint32_t n, uint32_t tile) {
    uint32_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (tr[mid].tile_base <= tile) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// exclusive prefix of `a` over a block of 256 threads; total in *ta
__device__ __forceinline__ uint32_t block_scan256(uint32_t a, uint32_t *ta, uint32_t *wave_sum /* LDS, 4 */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t ia = a;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t xa = __shfl_up(ia, d);
        if (lane >= d) ia += xa;
    }
    __syncthreads();  // an earlier scan's totals have been read
    if (lane == 63) wave_sum[wave] = ia;
    __syncthreads();
    uint32_t off = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wave) off += wave_sum[w];
        sum += wave_sum[w];
    }
    *ta = sum;
    return off + ia - a;
}
}  // namespace

template <bool WRITE>
__global__ void __launch_bounds__(RG_MP3_FRAME_TILE)
rg_mp3_frames_kernel(const RgMp3DevTrack *__restrict__ tracks, uint32_t n_tracks, const uint8_t *__restrict__ chunk,
                     uint32_t *__restrict__ tile_units, const uint32_t *__restrict__ tile_unit_base, RgMp3HuffRec *__restrict__ recs) {
    __shared__ uint32_t wave_sum[4];
    const uint32_t ti = find_by_tile(tracks, n_tracks, blockIdx.x);
    const RgMp3DevTrack tr = tracks[ti];
    if (WRITE && tr.all_frames_decode) return;  // the count pass has put the records in place (block-uniform)
    const uint32_t tile = blockIdx.x - tr.tile_base;
    const uint32_t f = tile * RG_MP3_FRAME_TILE + threadIdx.x;
    const bool live = f < tr.n_frames;
    uint64_t raw[RG_MP3_SLOT_BYTES / 8];
    uint32_t main_bytes = 0;
    if (live) {
        const uint64_t *src = reinterpret_cast<const uint64_t *>(chunk + tr.slots_base + (size_t)f * RG_MP3_SLOT_BYTES);  // 8-byte aligned
#pragma unroll
        for (int k = 0; k < RG_MP3_SLOT_BYTES / 8; ++k) raw[k] = src[k];
        RgMp3FrameHdr h;
        if (rg_mp3_frame_header(reinterpret_cast<const uint8_t *>(raw), &h)) main_bytes = rg_mp3_frame_main_bytes(h);
    }
    uint32_t total;
    const uint32_t have_excl = block_scan256(main_bytes, &total, wave_sum);
    const uint64_t have = reinterpret_cast<const uint64_t *>(chunk + tr.tiles_base)[tile] + have_excl;
    RgMp3HuffRec r[4];
    uint32_t n = 0;
    if (live) {
        uint32_t mb;
        n = (uint32_t)rg_mp3_frame_records(reinterpret_cast<const uint8_t *>(raw), have, (int)tr.channels, r, &mb);
    }
    const uint32_t unit_excl = block_scan256(n, &total, wave_sum);
    if (!WRITE) {
        if (threadIdx.x == 0) tile_units[blockIdx.x] = total;
        // the records already go where they belong if no frame of the track is dropped (the usual case: the scan pass finds
        // out and the write pass has nothing left to do for the track)
        const uint32_t upf = (tr.lsf ? 1u : 2u) * tr.channels;
        RgMp3HuffRec *__restrict__ dst = recs + tr.unit_base + (uint64_t)tile * RG_MP3_FRAME_TILE * upf + unit_excl;
        for (uint32_t i = 0; i < n; ++i) dst[i] = r[i];
    } else {
        RgMp3HuffRec *__restrict__ dst = recs + tr.unit_base + tile_unit_base[blockIdx.x] + unit_excl;
        for (uint32_t i = 0; i < n; ++i) dst[i] = r[i];
    }
}

__global__ void __launch_bounds__(256)
rg_mp3_frames_scan_kernel(RgMp3DevTrack *__restrict__ tracks, const uint32_t *__restrict__ tile_units, uint32_t *__restrict__ tile_unit_base,
                          uint32_t *__restrict__ results) {
    __shared__ uint32_t wave_sum[4];
    const RgMp3DevTrack tr = tracks[blockIdx.x];
    const uint32_t n_tiles = (tr.n_frames + RG_MP3_FRAME_TILE - 1) / RG_MP3_FRAME_TILE;
    uint32_t run = 0;
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += 256) {
        const uint32_t t = t0 + threadIdx.x;
        const uint32_t v = t < n_tiles ? tile_units[tr.tile_base + t] : 0u;
        uint32_t total;
        const uint32_t ex = block_scan256(v, &total, wave_sum);
        if (t < n_tiles) tile_unit_base[tr.tile_base + t] = run + ex;
        run += total;
    }
    if (threadIdx.x == 0) {
        const uint32_t granules = run / tr.channels;
        tracks[blockIdx.x].n_granules = granules;
        tracks[blockIdx.x].all_frames_decode = run == tr.n_frames * (tr.lsf ? 1u : 2u) * tr.channels ? 1u : 0u;
        if (tr.channels == 2) tracks[blockIdx.x].ch1 = tr.ch0 + (size_t)granules * 576;
        results[tr.result_index] = granules;
    }
}

extern "C" hipError_t rg_launch_mp3_huffman(const RgMp3DevTables *d_tab, const RgMp3DevHuff *d_huff, const RgMp3DevTrack *d_tracks,
                                            uint32_t n_tracks, const RgMp3HuffRec *d_recs, const uint8_t *d_main, rg_mp3_unit *d_units,
                                            int16_t *d_is, uint64_t total_units, hipStream_t s) {
    if (total_units == 0) return hipSuccess;
    hipLaunchKernelGGL(rg_mp3_huffman_kernel, dim3((uint32_t)((total_units + kHuffThreads - 1) / kHuffThreads)), dim3(kHuffThreads), 0, s, d_tab,
                       d_huff, d_tracks, n_tracks, d_recs, d_main, d_units, d_is, total_units);
    return hipGetLastError();
}

// n_runs: blocks of the grid, one per run of RG_MP3_RUN granules (RgMp3DevTrack::run_base); is_group_log2: layout of d_is
// (rg_mp3_is_index): RG_MP3_IS_GROUP_LOG2 behind the device Huffman stage, 0 for spectra parsed on the host
extern "C" hipError_t rg_launch_mp3_backhalf(const RgMp3DevTables *d_tab, const RgMp3DevTrack *d_tracks, uint32_t n_tracks,
                                             uint32_t n_runs, const rg_mp3_unit *d_units, const int16_t *d_is, uint32_t is_group_log2,
                                             hipStream_t s) {
    if (n_runs == 0) return hipSuccess;
    hipLaunchKernelGGL(rg_mp3_backhalf_kernel, dim3(n_runs), dim3(RG_MP3_BH_THREADS), 0, s, d_tab, d_tracks, n_tracks, d_units, d_is, is_group_log2);
    return hipGetLastError();
}

// tile_scratch: 2 x n_tiles words
extern "C" hipError_t rg_launch_mp3_frames(RgMp3DevTrack *d_tracks, uint32_t n_tracks, uint32_t n_tiles, const uint8_t *d_chunk, uint32_t *tile_scratch,
                                           RgMp3HuffRec *d_recs, uint32_t *d_results, hipStream_t s) {
    if (n_tracks == 0 || n_tiles == 0) return hipSuccess;
    uint32_t *units = tile_scratch, *base = tile_scratch + n_tiles;
    hipLaunchKernelGGL((rg_mp3_frames_kernel<false>), dim3(n_tiles), dim3(RG_MP3_FRAME_TILE), 0, s, d_tracks, n_tracks, d_chunk, units, base, d_recs);
    hipLaunchKernelGGL(rg_mp3_frames_scan_kernel, dim3(n_tracks), dim3(256), 0, s, d_tracks, units, base, d_results);
    hipLaunchKernelGGL((rg_mp3_frames_kernel<true>), dim3(n_tiles), dim3(RG_MP3_FRAME_TILE), 0, s, d_tracks, n_tracks, d_chunk, units, base, d_recs);
    return hipGetLastError();
}
