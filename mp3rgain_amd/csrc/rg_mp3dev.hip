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


typedef float rg_f32x2 __attribute__((ext_vector_type(2)));
typedef short rg_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short rg_u16x2 __attribute__((ext_vector_type(2)));

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Stages B-E, one kernel.  One block = a run of RG_MP3_RUN consecutive granules of one track (both channels), four waves,
// a pipeline over the run's granules with ONE barrier per step; every hand-over goes through LDS.  Granule k of the run:
//   steps k, k + 1  wave 1 (k even) or wave 2 (k odd): requantisation, in two halves so that one wave's first half runs
//                   beside the other's second.  First half: the granule's units and quantised spectra (asked for a granule
//                   of this wave -- two steps -- ago) are taken in, the next ones asked for; a line's gain
//                   2^((global_gain - 210)/4 - mult (sf + preflag pretab) [- 2 subblock_gain]) depends on its band (and
//                   window) only, so the wave writes the granule's 22 long-band and 39 (short band, window) gains, from a
//                   table indexed by the exact integer exponent.  Second half: three rounds of four lines per lane -- a
//                   line costs one look-up of its gain and one of x^(4/3) --, mid/side, then the special cases: intensity
//                   stereo, short-block reordering (wave-local steps, no block barrier inside)
//   step k + 2      wave 0, lane (channel, subband): alias butterflies folded into the load of its eighteen lines, the
//                   fast 36-point IMDCT of rg_mp3_math.h (the code the host runs) or three 12-point ones, window,
//                   overlap-add with the second half it kept from the granule before (in its registers), frequency
//                   inversion -> 18 x 32 subband samples
//   step k + 3      wave 3: matrixing, the 32-point DCT of the granule's 36 columns (channel, time slot) as 24
//                   v_mfma_f32_16x16x4_f32 (rg_mp3_math.h: rg_mp3_dct32_split is the same arithmetic on the host)
//   step k + 4      wave 3 again, in the same instruction stream as the matrixing of granule k + 1: the 512-tap window,
//                   lane (channel, j) owns PCM sample j of the granule's eighteen time slots; the two DCT columns it
//                   needs of the fifteen slots of history stay in its registers from one granule to the next (2 LDS reads
//                   per output instead of 32), the symmetry's signs are folded into its sixteen window coefficients; 576
//                   PCM samples per granule and channel straight into the arena
// A run that starts inside the track takes the two granules before it through the first stages (the second one's subband
// samples need the first one's overlap, and its DCT rows are the filterbank's history).  Each wave runs its own loop, so
// the register file is sized for the largest stage, not for their sum: 128 VGPRs, 38 KB of LDS, four blocks per CU.
// History: rounds 2 and 3 had two kernels with the subband samples in memory between them (0.60 ms per 256 K units); rounds
// 3-5 this kernel with ONE requantisation wave, a matrixing wave (Lee's DCT on the vector pipe) and a window wave: 0.46,
// then 0.35 ms; round 6 the arrangement above: 0.35-0.36 ms again -- DESIGN section 10 has the measurements that say why
// (a block alone on its CU steps in 4000 cycles instead of 5000, four blocks together in 5400 either way: the CU's vector
// pipe, which the f32 matrix instructions share, is three quarters busy).
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
__device__ unsigned long long rg_bh_dbg2[40][6];
extern "C" int rg_bh_dbg_read(void *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rg_bh_dbg), sizeof(unsigned long long) * 4 * 40 * 2); }
extern "C" int rg_bh_dbg2_read(void *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rg_bh_dbg2), sizeof(unsigned long long) * 40 * 6); }
#define RG_BH_STAMP(w, k, e) do { if (blockIdx.x == RG_BH_TIMING && (threadIdx.x & 63) == 0 && (k) < 40) rg_bh_dbg[w][k][e] = __builtin_amdgcn_s_memtime(); } while (0)
#define RG_BH_STAMP2(k, e) do { if (blockIdx.x == RG_BH_TIMING && (threadIdx.x & 63) == 0 && (k) < 40) rg_bh_dbg2[k][e] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long rg_bh_dbg3[40][6];  // inside wave 3: [step][matrixing done | rows in registers | sums done | stored]
extern "C" int rg_bh_dbg3_read(void *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rg_bh_dbg3), sizeof(unsigned long long) * 40 * 6); }
#define RG_BH_STAMP3(k, e) do { if (blockIdx.x == RG_BH_TIMING && (threadIdx.x & 63) == 0 && (k) < 40) rg_bh_dbg3[k][e] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RG_BH_STAMP3(k, e) do { } while (0)
#define RG_BH_STAMP(w, k, e) do { } while (0)
#define RG_BH_STAMP2(k, e) do { } while (0)
#endif
#define RG_MP3_BH_THREADS 256
__global__ void __launch_bounds__(RG_MP3_BH_THREADS) __attribute__((amdgpu_waves_per_eu(4)))
rg_mp3_backhalf_kernel(const RgMp3DevTables *__restrict__ T, const RgMp3DevTrack *__restrict__ tracks, uint32_t n_tracks,
                       const rg_mp3_unit *__restrict__ units, const int16_t *__restrict__ is, const uint32_t planes) {
    constexpr int R = RG_MP3_RUN;
    __shared__ float xrb[2][2][576];   // [granule parity][channel][line]: written in the second step of its requantisation, read in the next
    __shared__ __attribute__((aligned(16))) rg_mp3_unit Ub[4][2];  // [granule % 4]: in two steps before its requantisation starts, read until its IMDCT
    __shared__ uint8_t band_nz[64];
    __shared__ short band_mode[64];
    __shared__ float c12[12][6], wn[4][36], cs_l[8], ca_l[8];
    __shared__ uint8_t ptab[24];
    __shared__ float gain_l[RG_MP3_GAIN_Q_MAX - RG_MP3_GAIN_Q_MIN + 1];
    constexpr int kPowLds = 1728;      // x^(4/3) for the values that occur (what four blocks per CU leave room for); larger ones
                                       // go to the table in memory: a wait the wave cannot hide, once per round that has one
    __shared__ float pow_l[kPowLds];
    __shared__ uint16_t sfbl_l[24], sfbs_l[16];
    __shared__ float gtab[2][2][64];   // [requantisation wave][channel][band | 22 + 3 * short band + window]
    __shared__ __attribute__((aligned(4))) uint8_t sidx_l[576];
    // rows of 34 words: a matrix-core operand access is (column j = lane & 15, k or row group g = lane >> 4) -> word 34 j + g + ..
    // = bank 2 j + g of 32: the sixteen columns of a half-wave's two groups fall on different banks (33 made them collide)
    constexpr int kRow = 34;
    __shared__ float Sin[2][2][18][kRow];  // [granule parity][channel][time slot][subband]: wave 0 -> wave 3
    __shared__ float cos_l[2][16][16];     // [parity][k][i]: the matrix cores' A operands (T->dct16), read by wave 3 as it goes
    __shared__ float Ar[2][18][kRow];    // the DCT outputs of those slots (word 32 = 0.0f = V[16]): wave 3 to itself (matrix cores -> window)
    constexpr int NT = RG_MP3_BH_THREADS;
    const int tid = threadIdx.x;
#ifdef RG_BH_LDS_PAD  // experiment: fewer blocks per CU (what the step costs with less company)
    __shared__ volatile float lds_pad[RG_BH_LDS_PAD / 4];
    lds_pad[tid] = 1.0f;
#endif
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
    constexpr int kTail = 4;           // steps a granule spends in the pipeline behind its first: requantisation x 2, IMDCT, matrixing, window
    for (int e = tid; e < 144; e += NT) (&wn[0][0])[e] = (&T->win[0][0])[e];
    if (tid < 72) (&c12[0][0])[tid] = (&T->imdct12[0][0])[tid];
    if (tid < 8) { cs_l[tid] = T->cs[tid]; ca_l[tid] = T->ca[tid]; }
    if (tid < 24) ptab[tid] = T->pretab[tid];
    for (int e = tid; e < RG_MP3_GAIN_Q_MAX - RG_MP3_GAIN_Q_MIN + 1; e += NT) gain_l[e] = T->gain[e];
    for (int e = tid; e < kPowLds; e += NT) pow_l[e] = T->pow43[e];
    for (int e = tid; e < 512; e += NT) (&cos_l[0][0][0])[e] = (&T->dct16[0][0][0])[e];
    if (tid >= 32 && tid < 56) sfbl_l[tid - 32] = T->sfb_long[rr][tid - 32];
    if (tid >= 64 && tid < 80) sfbs_l[tid - 64] = T->sfb_short[rr][tid - 64];
    for (int e = tid; e < 144; e += NT) reinterpret_cast<uint32_t *>(sidx_l)[e] = reinterpret_cast<const uint32_t *>(T->short_idx_of_line[rr])[e];
    static_assert(sizeof(rg_mp3_unit) == 64, "the unit prefetch assumes 64-byte units");
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // a scalar: each wave takes one branch whole
    const int lane = tid & 63;

    if (wave == 1 || wave == 2) {
        // ================= waves 1, 2: units and spectra in, requantised (and stereo-processed, reordered) spectrum out =
        // Requantisation is the pipeline's longest stage by far (its length is a chain of LDS round trips and memory
        // instructions, not arithmetic), so two waves share it: wave 1 takes the even granules of the pipeline, wave 2 the odd
        // ones, each over TWO steps -- first half: units, clean-up of what arrived, large values, prefetch, gains; second half:
        // the three rounds and the special cases -- so that one wave's first half runs beside the other's second.  The body
        // is compiled once per channel count, so that nothing in it asks how many channels there are.
        // (No wave of the block has a priority of its own any more: with the stages this even -- 2300 cycles a step each, wave
        // 3 3800, for a block that has the CU to itself -- raising any one of them measured 1-3 % slower)
#ifdef RG_BH_PRIO_RQ
        __builtin_amdgcn_s_setprio(RG_BH_PRIO_RQ);
#endif
        const int par = wave - 1;  // the wave's granules: par, par + 2, ... (a scalar)
        // PLANES: the spectra come from the device Huffman stage, a byte per line in two planes (rg_mp3dev.h); else rows of int16
        auto requant_wave = [&](auto nch_c, auto planes_c) {
        constexpr int nch = decltype(nch_c)::value;
        constexpr bool PLANES = decltype(planes_c)::value;
        constexpr int kRounds = 3;
        uint32_t rq_lb[kRounds] = {0u, 0u, 0u};  // long-block band numbers of a piece's four lines
#pragma unroll
        for (int r = 0; r < kRounds; ++r)
            if (lane + 64 * r < 144) rq_lb[r] = *reinterpret_cast<const uint32_t *>(&T->long_band_of_line[rr][4 * (lane + 64 * r)]);
        const bool uq = lane >= 56 && lane < 56 + 4 * nch;  // lanes that carry the units: four 16-byte words each
        const int uq_t = lane - 56;
        // the units of this wave's first granule go to LDS here; those of its second are on their way
        const int q_first = par < nsteps ? par : nsteps - 1;
        uint4 u_reg = make_uint4(0u, 0u, 0u, 0u), u_first = make_uint4(0u, 0u, 0u, 0u), u_keep = make_uint4(0u, 0u, 0u, 0u);
        if (uq) {
            u_first = reinterpret_cast<const uint4 *>(units + ubase + (uint64_t)q_first * nch)[uq_t];
            if (par < nsteps) reinterpret_cast<uint4 *>(&Ub[par][0])[uq_t] = u_first;
            u_reg = reinterpret_cast<const uint4 *>(units + ubase + (uint64_t)(par + 2 < nsteps ? par + 2 : nsteps - 1) * nch)[uq_t];
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
        // a lane's four lines of a round: 8 bytes of an int16 row, or 4 bytes of each plane
        uint32_t rq_off[kRounds];
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {  // the third round's idle lanes read its last piece again: no branch around a load
            const uint32_t piece = (uint32_t)(lane + 64 * r < 144 ? lane + 64 * r : 143);
            rq_off[r] = PLANES ? 4u * piece : 8u * piece;
        }
        const uint8_t *const is_block = reinterpret_cast<const uint8_t *>(is) + ubase * (uint64_t)RG_MP3_ROW_BYTES;  // 32-bit offsets from here (a run is 68 units)
        // Every load is issued whatever the step (the compiler can count them: a load under a condition makes the next wait
        // a wait for everything in flight, and this very prefetch with it -- tried: 0.36 -> 0.45 ms), but not every load has to
        // move bytes: lines from the unit's nz on were never written and are replaced by zeros below -- such lanes ask for the
        // last word that holds lines once more -- and the second plane is asked for in the first round only (a memory
        // instruction costs this wave, the pipeline's longest stage, about a hundred cycles to issue whatever it moves), by the
        // lanes below hi_q, the others asking for a word next to their first-plane word; a second plane longer than 64 words (large
        // values above line 256: rare) is read at the top of the step that needs it.  `hd`: the unit headers of the step asked
        // for (its nz and hi_q are the wave's).
        auto fetch_spectra = [&](const int step, const uint32_t (&hd)[2][4]) {
#ifdef RG_BH_NOFETCH  // experiment (wrong results): the spectra are asked for once per run -- what issuing the loads costs the step
            if (step > 0) return;
#endif
#pragma unroll
            for (int c = 0; c < nch; ++c) {
#ifdef RG_BH_SAMEROW  // experiment (wrong results): every step reads the run's first rows again -- what memory latency costs the step
                const uint8_t *const row = is_block + (uint32_t)c * (uint32_t)RG_MP3_ROW_BYTES + 0u * (uint32_t)step;
#else
                const uint8_t *const row = is_block + ((uint32_t)step * nch + c) * (uint32_t)RG_MP3_ROW_BYTES;
#endif
                const uint32_t nz = hd[c][0] & 0xFFFFu;
                if (PLANES) {
                    const uint32_t last = nz ? ((nz + 3u) & ~3u) - 4u : 0u;  // byte offset of the last word with lines in it
                    const uint32_t hi_q = (hd[c][3] >> 16) & 0xFFu;
#pragma unroll
                    for (int r = 0; r < kRounds; ++r) {
                        const uint32_t lo_at = rq_off[r] < last ? rq_off[r] : last;
                        rq_next[r][c].x = *reinterpret_cast<const uint32_t *>(row + lo_at);
                        // (the neighbour of the first-plane word, not the word itself: the compiler would reuse the first load's
                        // result for those lanes, i.e. wait for it right here)
                        if (r == 0) rq_next[r][c].y = *reinterpret_cast<const uint32_t *>(row + ((uint32_t)lane < hi_q ? (uint32_t)(RG_MP3_ROW_BYTES / 2) + rq_off[r] : (lo_at ^ 4u)));
                    }
                } else {
                    const uint32_t last = nz ? 2u * (((nz + 3u) & ~3u) - 4u) : 0u;
#pragma unroll
                    for (int r = 0; r < kRounds; ++r) rq_next[r][c] = *reinterpret_cast<const uint2 *>(row + (rq_off[r] < last ? rq_off[r] : last));
                }
            }
        };
        // magnitudes of a word's two lines: the word is two int16, or (PLANES, rounds with a second plane) sign << 15 | magnitude
        auto mag2 = [](const uint32_t w) -> rg_u16x2 {
            if (PLANES) return __builtin_bit_cast(rg_u16x2, w & 0x7FFF7FFFu);
            return __builtin_bit_cast(rg_u16x2, __builtin_elementwise_abs(__builtin_bit_cast(rg_s16x2, w)));
        };
        auto mag1 = [](const uint32_t w, const int half) -> int {
            if (PLANES) return (int)((w >> (16 * half)) & 0x7FFFu);
            const int v = (int)(int16_t)(w >> (16 * half));
            return v < 0 ? -v : v;
        };
        auto wave_sync = [] {  // LDS hand-over between lanes of this wave
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        };
        fetch_spectra(q_first, h_next);
        float (*const GT)[64] = gtab[par];
        rg_lds_barrier();
        int steps_done = 0;  // the block's barriers are counted: every wave passes kTail + nsteps of them
        if (par == 1) {      // step 0 belongs to the even wave's first half alone
            rg_lds_barrier();
            steps_done = 1;
        }
        for (int k = par; k < nsteps; k += 2) {
            RG_BH_STAMP(wave, k, 0);
            {
                // ---------------- first half: step k ----------------
                float (*const XP)[576] = xrb[k & 1];
                const rg_mp3_unit *const UP = Ub[k & 3];
                uint32_t h[2][4];
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) h[c][i] = h_next[c][i];
                RG_BH_STAMP2(k, 4);
                // ---- stage B: requantisation (rg_mp3dec.cpp: requantize)
                uint2 raw[kRounds][2];
#pragma unroll
                for (int r = 0; r < kRounds; ++r) { raw[r][0] = rq_next[r][0]; raw[r][1] = rq_next[r][1]; }
                // The lines from a unit's nz on are zero and were neither written by the Huffman stage (it completes the sector
                // the last value falls into) nor asked for (fetch_spectra): what arrived in their place is replaced by zeros.
                int nz8_max = 0;  // lines from here on are zero in every channel
#pragma unroll
                for (int c = 0; c < nch; ++c) {
                    const int nz8 = ((int)(h[c][0] & 0xFFFFu) + 3) & ~3;
                    nz8_max = nz8 > nz8_max ? nz8 : nz8_max;
#pragma unroll
                    for (int r = 0; r < kRounds; ++r)
                        if (nz8 < 256 * (r + 1) && 4 * (lane + 64 * r) >= nz8) raw[r][c] = make_uint2(0u, 0u);  // the first test is the wave's
                }
                RG_BH_STAMP2(k, 5);
                // Two planes of a byte per line.  Past the longest second plane of the step's units (hi_q words of four lines: as far
                // as a magnitude above 127 was actually found, usually nowhere, else in the first few dozen lines) a byte is its
                // line's index into the LDS part of the x^(4/3) table -- no absolute value, no clamp, no large-value pass.  A round
                // with lines below that mark is put into sign << 15 | magnitude, two lines per word, and goes the int16 form's
                // way from there.
                uint32_t hi_q_max = 0u;
                if (PLANES) {
#pragma unroll
                    for (int c = 0; c < nch; ++c) {
                        const uint32_t hq = (h[c][3] >> 16) & 0xFFu;
                        hi_q_max = hq > hi_q_max ? hq : hi_q_max;
                    }
                    if (hi_q_max > 64u) {  // rare: the second plane's words of rounds 1 and 2, read here and now (nothing is in flight)
#pragma unroll
                        for (int c = 0; c < nch; ++c) {
                            const uint8_t *const row = is_block + ((uint32_t)k * nch + c) * (uint32_t)RG_MP3_ROW_BYTES + (uint32_t)(RG_MP3_ROW_BYTES / 2);
#pragma unroll
                            for (int r = 1; r < kRounds; ++r) raw[r][c].y = *reinterpret_cast<const uint32_t *>(row + rq_off[r]);
                        }
#pragma unroll
                        for (int c = 0; c < nch; ++c)
#pragma unroll
                            for (int r = 1; r < kRounds; ++r) asm volatile("" ::"v"(raw[r][c].y));  // consumed on this path (see below)
                    }
                }
                bool bytes_r[kRounds];  // the wave's
#pragma unroll
                for (int r = 0; r < kRounds; ++r) {
                    bytes_r[r] = PLANES && (uint32_t)(64 * r) >= hi_q_max;
                    if (PLANES && !bytes_r[r]) {
#pragma unroll
                        for (int c = 0; c < nch; ++c) {
                            const uint32_t lo = raw[r][c].x, hi = (uint32_t)(lane + 64 * r) < ((h[c][3] >> 16) & 0xFFu) ? raw[r][c].y : 0u;
                            uint32_t sm[2];
#pragma unroll
                            for (int q = 0; q < 2; ++q) {  // bytes lo[2q], hi[2q], lo[2q + 1], hi[2q + 1]
                                const uint32_t t = __builtin_amdgcn_perm(hi, lo, q ? 0x07030602u : 0x05010400u);
                                sm[q] = ((t & 0xFF00FF00u) >> 1) | (t & 0x007F007Fu) | ((t & 0x00800080u) << 8);
                            }
                            raw[r][c] = make_uint2(sm[0], sm[1]);
                        }
                    }
                }
                RG_BH_STAMP2(k, 2);
                RG_BH_STAMP2(k, 3);
                // (asked for at the top of the step instead -- 2000 cycles earlier -- the units change nothing: 0.358 against 0.354 ms)
                // The units of this wave's next granule (k + 2; asked for a granule ago) are here: their headers say where its
                // spectra end, and they go to LDS in the second half (their place still holds the units of granule k - 2 until
                // this step's IMDCT is done with them).  Those of granule k + 4 start travelling.  (Past the run's last granule
                // the same units and spectra are asked for again and never used.)
                u_keep = u_reg;
                header_of(u_keep, h_next);
#ifdef RG_BH_SAMEUNIT  // experiment (wrong results): every step reads the run's first unit again -- what the units' latency costs the step
                if (uq) u_reg = reinterpret_cast<const uint4 *>(units + ubase + (uint64_t)(k & 0) * nch)[uq_t];
#else
                if (uq) u_reg = reinterpret_cast<const uint4 *>(units + ubase + (uint64_t)(k + 4 < nsteps ? k + 4 : nsteps - 1) * nch)[uq_t];
#endif
                fetch_spectra(k + 2 < nsteps ? k + 2 : nsteps - 1, h_next);
                int bt_s[2] = {0, 0}, ll_s[2] = {0, 0}, so_s[2] = {0, 0};
                int gq_idx[2] = {0, 0};
                uint32_t gq_sf[2] = {0u, 0u};
#pragma unroll
                for (int c = 0; c < 2; ++c) {  // the scalefactor each lane's gain needs: one byte read per channel, asked for together
                    if (c >= nch) continue;
                    const int long_end = (int)((h[c][2] >> 16) & 0xFFu), short_start = (int)(h[c][2] >> 24);
                    // (no multiplication: for gq_kk - 3 short_start the compiler takes v_mad_u64_u32, whose 64-bit addend's unused upper half
                    // landed in a register of the prefetch just issued -- and the wave waited for memory right here, every step)
                    const int rel = gq_kk - (short_start == 0 ? 0 : (short_start == 3 ? 9 : 39));  // 3 short_start: it is 0, 3 or 13
                    const bool short_sf = lane >= 22 && gq_band < 12 && rel >= 0;
                    gq_idx[c] = lane < 22 ? lane : (short_sf ? long_end + rel : -1);
#ifdef RG_BH_NOGAINS  // experiment (wrong results): no scalefactor, no gain look-up -- what the gains cost the step
                    gq_sf[c] = 0;
#else
                    gq_sf[c] = UP[c].sf[gq_idx[c] < 0 ? 0 : gq_idx[c]];
#endif
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
#ifdef RG_BH_NOGAINS
                    gq_g[c] = __int_as_float(0x3f800000 + (q << 10));
#else
                    gq_g[c] = gain_l[q - RG_MP3_GAIN_Q_MIN];
#endif
                    bt_s[c] = (int)(h[c][0] >> 24);
                    ll_s[c] = long_end == 22 ? ll22 : (long_end == 8 ? ll8 : (long_end == 6 ? ll6 : (long_end == 0 ? 0 : (int)sfbl_l[long_end])));
                    so_s[c] = short_start >= 13 ? so13 : (short_start == 3 ? so3 : (short_start == 0 ? 0 : 3 * (int)sfbs_l[short_start]));
                }
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    if (c < nch) GT[c][lane] = lane < 61 ? gq_g[c] : 0.0f;
                const bool ms_all = nch == 2 && (h[0][3] & 3u) == 2u;  // plain mid/side (no intensity stereo in the frame)
                RG_BH_STAMP2(k, 0);
                RG_BH_STAMP(wave, k, 1);
                // ---------------- second half: step k + 1 (the gains are visible behind the barrier) ----------------
                rg_lds_barrier();
                RG_BH_STAMP(wave, k + 1, 0);
                if (uq && k + 2 < nsteps) reinterpret_cast<uint4 *>(&Ub[(k + 2) & 3][0])[uq_t] = u_keep;
                // The one conditional read of a granule: a quantised value beyond the LDS part of the x^(4/3) table takes its
                // power from the full table in memory and leaves it in the line's own place in the output buffer (free since
                // the last step's IMDCT).  With 1728 entries in LDS that is one unit in a thousand of encoder-made music and
                // none of the reference's VBR fixture; it happens here, at the top of the second half, where the only loads in
                // flight are the prefetch of a step ago (rounds 3-5 had it in front of the prefetch, with 704 entries).
                bool big_r[kRounds];  // the wave's: some lane of round r holds such a value
                {
                    bool any_big = false;
#pragma unroll
                    for (int r = 0; r < kRounds; ++r) {
                        rg_u16x2 amax = {0, 0};
                        if (!bytes_r[r] && nz8_max > 256 * r) {
#pragma unroll
                            for (int c = 0; c < nch; ++c) {
                                amax = __builtin_elementwise_max(amax, mag2(raw[r][c].x));
                                amax = __builtin_elementwise_max(amax, mag2(raw[r][c].y));
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
                                    const int a = mag1(w[j >> 1], j & 1);
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
                            for (int j = 0; j < 4; ++j) gv[c][j] = GT[c][(rq_lb[r] >> (8 * j)) & 0xFFu];
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int line = rq_l0 + j;
                                const int idx = line < ll_s[c] ? (int)((rq_lb[r] >> (8 * j)) & 0xFFu) : 22 + (int)sidx_l[line - ll_s[c] + so_s[c]];
                                gv[c][j] = GT[c][idx];
                            }
                        }
                        if (bytes_r[r]) {  // the byte is the index
#pragma unroll
                            for (int j = 0; j < 4; ++j) mg[c][j] = pow_l[(raw[r][c].x >> (8 * j)) & 127u];
                        } else {
                            // two lines per 32-bit word: |v|, the largest of them and the index into the LDS part of the power
                            // table as packed 16-bit operations
                            const uint32_t w[2] = {raw[r][c].x, raw[r][c].y};
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const rg_u16x2 ua = mag2(w[h]);
                                const rg_u16x2 top = {(unsigned short)(kPowLds - 1), (unsigned short)(kPowLds - 1)};
                                const rg_u16x2 ci = __builtin_elementwise_min(ua, top);
                                mg[c][2 * h] = pow_l[ci.x];
                                mg[c][2 * h + 1] = pow_l[ci.y];
                            }
                        }
                    }
                    if (big_r[r]) {  // rare: a value beyond the LDS part of the table
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            if (c >= nch) continue;
                            const uint32_t w[2] = {raw[r][c].x, raw[r][c].y};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int a = mag1(w[j >> 1], j & 1);
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
                            const uint32_t sign_at_31 = bytes_r[r] ? w[0] << (24 - 8 * j) : ((j & 1) ? w[j >> 1] : w[j >> 1] << 16);
                            val[c][j] = __builtin_copysignf(t, __uint_as_float(sign_at_31));
                        }
                    }
                    // ---- stage C, the plain case: mid/side.  On every line: the host stops at the longer channel's end, and
                    // behind it both values are +0, whose sum and difference times a constant are +0 again
                    if (ms_all) {
                        const float isq2 = 0.70710678118654752440f;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float a = val[0][j], b = val[1][j];
                            val[0][j] = (a + b) * isq2;
                            val[1][j] = (a - b) * isq2;
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
            RG_BH_STAMP(wave, k + 1, 1);
            rg_lds_barrier();
            steps_done += 2;
        }
        for (; steps_done < nsteps + kTail; ++steps_done) rg_lds_barrier();
        };
        if (planes) {
            if (nch == 2) requant_wave(std::integral_constant<int, 2>{}, std::true_type{});
            else requant_wave(std::integral_constant<int, 1>{}, std::true_type{});
        } else {
            if (nch == 2) requant_wave(std::integral_constant<int, 2>{}, std::false_type{});
            else requant_wave(std::integral_constant<int, 1>{}, std::false_type{});
        }
        return;
    }

    if (wave == 0) {
        // ================= wave 0: lane (channel, subband): spectrum -> subband samples of eighteen time slots =========
        const bool active = lane < 32 * nch;
        const int my_c = lane >> 5, my_sb = lane & 31;
        float ovl[18];  // the second half of the lane's subband from the granule before (rounds 3-5: an LDS column, 36 accesses a step)
#pragma unroll
        for (int i = 0; i < 18; ++i) ovl[i] = 0.0f;  // a granule without a predecessor adds 0.0f, as the host does
#ifdef RG_BH_PRIO_IMDCT
        __builtin_amdgcn_s_setprio(RG_BH_PRIO_IMDCT);
#endif
        rg_lds_barrier();
        for (int k = 0; k < nsteps + kTail; ++k) {
            RG_BH_STAMP(0, k, 0);
            if (k >= 2 && k <= nsteps + 1 && active) {
                // ---- stage D of granule k - 1 (rg_mp3dec.cpp: antialias, hybrid).  The butterflies between subbands
                // sb - 1 | sb, sb = 1 .. nb: this lane evaluates its own half of the two it touches.  Windowed sample i:
                // the first eighteen are added to the overlap and leave (frequency inversion: odd samples of odd subbands
                // change sign), the second eighteen are the next granule's overlap.
                const int q = k - 2;
                const rg_mp3_unit &u = Ub[q & 3][my_c];
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
                        float v = val + ovl[i];
                        if (flip && (i & 1)) v = -v;
                        dst[i * kRow] = v;
                    } else {
                        ovl[i - 18] = val;
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
#pragma unroll  // (unrolled: the overlap is in registers, and of the three windows at most two reach a sample)
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
                        emit(i, raw);
                    }
                }
            }
            RG_BH_STAMP(0, k, 1);
            rg_lds_barrier();
        }
        return;
    }

    {
        // ================= wave 3: matrixing on the matrix cores, then the 512-tap window -> PCM ========================
        // Matrixing (rg_mp3dec.cpp: synth): the 36 (18) columns (channel, time slot) of a granule are three (two) tiles of
        // sixteen; the 32-point DCT behind one even/odd split (rg_mp3_math.h: rg_mp3_dct32_split -- what the host runs) is
        // two 16 x 16 matrices with K = 16, i.e. four v_mfma_f32_16x16x4_f32 per parity and tile chained through the
        // accumulator: 24 matrix instructions (768 cycles of a pipe nothing else in this kernel uses), 24 additions and
        // 24 + 24 LDS accesses per granule where Lee's form took 289 vector instructions on a wave of its own (rounds 3-5).
        // The instruction adds its four products to the accumulator one after the other, each a fused multiply-add, in
        // ascending k, subnormals kept (tools/ubench/mfma_order.hip; profiles/r06_mfma_order.txt): a chain of sixteen
        // rg_mp3_mac from +0.0f, which is what the host's loop is.
        // A operand (resident): lane (i = lane & 15, g = lane >> 4) holds cos[parity][k = 4 ks + g][i]; B operand: column
        // j = lane & 15 of the tile, k = 4 ks + g: u or v of subbands k and 31 - k; D: lane holds rows 4 g .. 4 g + 3 of
        // both parities = DCT outputs 8 g .. 8 g + 7 of its column.
        // The window: lane (channel, j) owns PCM sample j of every time slot:
        // PCM sample j of time slot r is  sum_{i<8} V[r-2i][j] D[64i+j] + V[r-2i-1][32+j] D[64i+32+j]  in that order (the
        // host's); V[.][j] and V[.][32+j] are two fixed columns of the DCT rows with the signs folded into the window
        // coefficients: V[.][j] = A[.][16+j] (j < 16), 0 (j = 16), -A[.][48-j] (j > 16); V[.][32+j] = -A[.][16-j] (j < 16),
        // -A[.][0] (j = 16), -A[.][j-16] (j > 16) (fma(-a, d, s) and fma(a, -d, s) are the same bits).  cA / cB [q]: those
        // columns of time slot q - 15 relative to the granule; the fifteen slots of history stay in registers from one
        // granule to the next.
        typedef float rg_f32x4 __attribute__((ext_vector_type(4)));
#ifdef RG_BH_PRIO_W3
        __builtin_amdgcn_s_setprio(RG_BH_PRIO_W3);
#endif
        const int mj = lane & 15, mg = lane >> 4;
        const float *const cos_at = &cos_l[0][mg][mj];  // + 256 parity + 64 ks: the lane's element of an A operand (word = lane + ..: no bank twice)
        const int ncols = 18 * nch;
        if (lane < 36) (&Ar[0][0][0])[lane * kRow + 32] = 0.0f;  // word 32 of every row = V[16] = 0.0f, written once
        // one channel: the upper half-wave repeats the lower one (same values to the same addresses) instead of sitting
        // out under a mask -- the step's body is one straight line
        const int c = nch == 2 ? lane >> 5 : 0, wj = lane & 31;
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
        // where a lane's PCM goes: a scalar base (the granule in the first channel's plane) + 32 bits of the lane's own (its
        // channel's plane, its sample) -- a 64-bit address per lane cost two registers the sums need (and, spilled, a wait
        // for every store in flight at the top of each step); the planes of a track lie less than 2^32 bytes apart (the host
        // refuses a longer track for this route: rg_mp3dev_host.hip)
        const uint32_t lane_off = (c == 0 ? 0u : (uint32_t)(tr.ch1 - tr.ch0)) + (uint32_t)wj;
        // One step of this wave: the matrixing of granule pd = k - 3 on the matrix cores and, BESIDE it, the window sums
        // of granule pw = k - 4 on the vector pipe -- the 24 matrix instructions (32 cycles each in their own pipe) are
        // issued one every twelve fused multiply-adds (48 cycles), so the step is as long as the sums alone (one after the
        // other the two took 1900 + 1600 cycles of a block that has the CU to itself, the longest stage of the pipeline).
        // Order: the rows of pw (written by the step before) into registers, then pd's operands tile by tile, its results
        // into the rows behind.  DV / WM (compile time): is there a granule to transform; 0 = none to window, 1 = its rows
        // only become history (the granules before the run's first: nothing leaves), 2 = sums and PCM.
        auto w3_step = [&](auto dv_c, auto wm_c, const int k) {
#ifdef RG_BH_NO_MFMA  // experiments (wrong results): the step without its matrix instructions / without its sums
            constexpr bool DV = false;
#else
            constexpr bool DV = decltype(dv_c)::value;
#endif
#ifdef RG_BH_NO_SUMS
            constexpr int WM = decltype(wm_c)::value == 2 ? 1 : decltype(wm_c)::value;
#else
            constexpr int WM = decltype(wm_c)::value;
#endif
            const int pd = k - 3, pw = k - 4;
            const float *const S = &Sin[pd & 1][0][0][0];  // [column = channel * 18 + time slot][kRow]
            float *const wr = &Ar[0][0][0] + mj * kRow + 8 * mg;
            if (WM >= 1) {
                const float *rows = &Ar[c][0][0];
#pragma unroll
                for (int q = 0; q < 18; ++q) {
                    cA[15 + q] = rows[q * kRow + col1];
                    cB[15 + q] = rows[q * kRow + col2];
                }
            }
            // a tile's operands: subbands k = 4 ks + g and 31 - k of the lane's column; the pair of a k-step is asked for
            // again, for the next tile, as soon as its two instructions have taken it (eight chunks before it is needed)
            float xa[4], xb[4];
            // (one base address, every tile and k-step a constant behind it; the last tile's idle columns read what lies
            // behind the granule's rows -- whatever it is, their results are dropped)
            const float *const rd_lo = S + mj * kRow + mg, *const rd_hi = S + mj * kRow + 31 - mg;
            auto load_pair = [&](const int nt, const int ks) {
                xa[ks] = rd_lo[16 * nt * kRow + 4 * ks];
                xb[ks] = rd_hi[16 * nt * kRow - 4 * ks];
            };
            if (DV) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) load_pair(0, ks);
            }
            uint32_t lo = lane_off;
            asm volatile("" : "+v"(lo));  // (or the sum is made once, outside the loop, and its two registers are spilled: a reload and a wait for every store in flight at the top of each step)
            // (said to be global memory: behind the asm the compiler no longer knows, and a flat store also counts as an LDS access)
            typedef __attribute__((address_space(1))) float rg_gf32;
            rg_gf32 *__restrict__ const dst = (rg_gf32 *)(tr.ch0 + ((size_t)g0 + (size_t)(pw + gi0)) * 576) + lo;  // (the wave's) + the lane's
            float sum[3] = {0.0f, 0.0f, 0.0f};
            rg_f32x4 de = {0.0f, 0.0f, 0.0f, 0.0f}, dd = {0.0f, 0.0f, 0.0f, 0.0f};
            // the A operands come from LDS two instructions ahead of their use (resident they were eight registers the sums need)
            float ca[2] = {0.0f, 0.0f};
#ifdef RG_BH_MFMA_BATCH  // experiment: the 24 matrix instructions tile by tile, eight back to back, BEFORE the sums
            if (DV) {
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    float cw[2][4], bu[4], bv[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        if (nt > 0) load_pair(nt, ks);
                        cw[0][ks] = cos_at[64 * ks];
                        cw[1][ks] = cos_at[256 + 64 * ks];
                    }
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) { bu[ks] = xa[ks] + xb[ks]; bv[ks] = xa[ks] - xb[ks]; }
                    __builtin_amdgcn_sched_barrier(0);
                    const rg_f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        de = __builtin_amdgcn_mfma_f32_16x16x4f32(cw[0][ks], bu[ks], ks == 0 ? zero : de, 0, 0, 0);
                        dd = __builtin_amdgcn_mfma_f32_16x16x4f32(cw[1][ks], bv[ks], ks == 0 ? zero : dd, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (16 * nt + mj < ncols) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) *reinterpret_cast<float2 *>(wr + 16 * nt * kRow + 2 * v) = make_float2(de[v], dd[v]);
                    }
                }
            }
            constexpr bool DVL = false;
#else
            constexpr bool DVL = DV;
#endif
            if (DVL) {
                ca[0] = cos_at[0];
                ca[1] = cos_at[256];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                    for (int par = 0; par < 2; ++par) {
                        if (DVL) {
                            const rg_f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
                            if (par == 0) {
                                de = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[0], xa[ks] + xb[ks], ks == 0 ? zero : de, 0, 0, 0);
                                ca[0] = cos_at[64 * ((ks + 1) & 3)];
                            } else {
                                dd = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[1], xa[ks] - xb[ks], ks == 0 ? zero : dd, 0, 0, 0);
                                ca[1] = cos_at[256 + 64 * ((ks + 1) & 3)];
                                if (nt < 2) load_pair(nt + 1, ks);
                            }
                        }
                        if (WM == 2) {
                            // chunk m of 24: time slots 3 (m / 4) .. + 2, terms i = 2 (m % 4), 2 (m % 4) + 1 of their sixteen
                            const int m = nt * 8 + ks * 2 + par, q0 = 3 * (m / 4), pp = m % 4;
#pragma unroll
                            for (int ii = 0; ii < 2; ++ii) {
                                const int i = 2 * pp + ii;
#pragma unroll
                                for (int j = 0; j < 3; ++j) {
                                    if (i == 0) sum[j] = 0.0f;
                                    sum[j] = rg_mp3_mac(cA[q0 + j + 15 - 2 * i], D1[i], sum[j]);
                                    sum[j] = rg_mp3_mac(cB[q0 + j + 14 - 2 * i], D2[i], sum[j]);
                                }
                            }
                            if (pp == 3) {
#pragma unroll
                                for (int j = 0; j < 3; ++j) dst[(q0 + j) * 32] = sum[j];
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (DVL) {  // the tile's results into the rows (its last instruction has had the twelve sums behind it to finish)
                    if (16 * nt + mj < ncols) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) *reinterpret_cast<float2 *>(wr + 16 * nt * kRow + 2 * v) = make_float2(de[v], dd[v]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (WM >= 1) {
#pragma unroll
                for (int q = 0; q < 15; ++q) { cA[q] = cA[q + 18]; cB[q] = cB[q + 18]; }
            }
        };
        rg_lds_barrier();
        for (int k = 0; k < nsteps + kTail; ++k) {
            RG_BH_STAMP(3, k, 0);
            const int pd = k - 3, pw = k - 4;
            const bool dv = pd >= pd0 && pd < nsteps;
            const int wm = (pw >= pd0 && pw < nsteps) ? (pw + gi0 >= 0 ? 2 : 1) : 0;
            using std::integral_constant;
            if (dv) {
                if (wm == 2) w3_step(std::true_type{}, integral_constant<int, 2>{}, k);
                else if (wm == 1) w3_step(std::true_type{}, integral_constant<int, 1>{}, k);
                else w3_step(std::true_type{}, integral_constant<int, 0>{}, k);
            } else if (wm == 2) {
                w3_step(std::false_type{}, integral_constant<int, 2>{}, k);
            } else if (wm == 1) {
                w3_step(std::false_type{}, integral_constant<int, 1>{}, k);
            }
            RG_BH_STAMP(3, k, 1);
            rg_lds_barrier();
        }
    }
}

// =====================================================================================================================
// Stage A's heavy part on the device: scalefactors + Huffman-coded spectrum (rg_mp3dec.cpp: read_scalefactors_v1 /
// read_scalefactors_lsf / decode_spectrum), one thread per granule and channel.  Integer work throughout: the outputs
// (576 quantised values per unit + one rg_mp3_unit) are the values rg_mp3_parse_units writes on the host; the spectra leave as
// sign and magnitude, a byte per line in two planes (rg_mp3dev.h: RG_MP3_ROW_BYTES).
//
// The decode is a chain of dependent look-ups per symbol, so everything a symbol touches is kept close: the code tables
// of all 32 table numbers sit in LDS (31 KB, copied once per block of 512 units), the bit stream is read through three
// registers of 32 bits with the next word always in flight, scalefactors are parsed into an LDS column per thread, and
// the spectrum leaves in 16-byte stores that the eight lanes of a group put side by side.  Granule 1 of an MPEG-1 frame may reuse granule 0's scalefactors (scfsi): its
// thread then parses granule 0's scalefactors first, which is cheaper than chaining the two granules in one thread.
namespace {

#ifndef RG_HF_THREADS
#define RG_HF_THREADS 512
#endif
constexpr int kHuffThreads = RG_HF_THREADS;

// The bit reader.  Under SIMT whatever ONE lane of a wave has to do now and then -- take the next word of its stream, ask
// for the next bytes, follow a long code into a second table -- the wave does every time, so the reader has no state to keep
// in step with the position: a lane's stream passes through a ring of eight or sixteen words in LDS (column `tid` of ring[words][threads],
// big-endian words with everything at or past `limit` -- the end of the frame's own main data -- already zero, which is what
// the host decoder's private copy of the frame's data reads there), and a look at the stream is three words from the ring and
// two 64-bit shifts by the position's low five bits, whatever the position.  The ring is fed sixteen bytes at a time -- one
// such group in flight per lane, asked for when the one before goes into the ring: the lanes of a wave are units of equal
// length from anywhere in the chunk and every request is a DRAM page of its own -- at points of the loops where the whole
// wave feeds together.  Positions are 32-bit bit counts from the 16-byte group the granule starts in (a granule is at most
// 4095 bits long).
#ifndef RG_HF_RING
#define RG_HF_RING 16  // words of a lane's ring: 8 (fed 16 bytes at a time) or 16 (fed 32 bytes at a time)
#endif
struct BitRing {
    static constexpr uint32_t kWords = RG_HF_RING, kMask = RG_HF_RING - 1, kFeed = RG_HF_RING / 2;  // words per feed
    static_assert(4 * (kWords + kFeed) + 32 <= RG_MP3_READ_AHEAD_BYTES, "the buffers are reserved with RG_MP3_READ_AHEAD_BYTES behind their payload");
    uint32_t *__restrict__ ring;     // the thread's column: word i of the stream at ring[(i & kMask) * kHuffThreads]
    const uint4 *__restrict__ g;     // the next group to ask for (the track's main data is 16-byte aligned)
    uint4 c[kFeed / 4];              // the groups in flight: words filled .. filled + kFeed - 1
    uint32_t filled;                 // words in the ring (a multiple of kFeed)
    uint32_t pos, end;               // read position, end of the granule's bits
    int32_t limit;                   // end of the frame's own data; may lie before `pos` in damaged streams
    __device__ __forceinline__ uint32_t masked(uint32_t raw, uint32_t i) const {
        int v = limit - (int32_t)(i << 5);
        v = v < 0 ? 0 : (v > 32 ? 32 : v);
        return __builtin_bswap32(raw) & (uint32_t)(0xFFFFFFFF00000000ull >> v);
    }
    __device__ __forceinline__ void commit(const uint4 &q) {
        uint32_t *const at = ring + (filled & kMask) * kHuffThreads;  // filled is a multiple of 4: no wrap inside a group
        if ((int32_t)((filled + 4u) << 5) <= limit) {  // far from the end of the frame's data: nothing to mask
            at[0 * kHuffThreads] = __builtin_bswap32(q.x);
            at[1 * kHuffThreads] = __builtin_bswap32(q.y);
            at[2 * kHuffThreads] = __builtin_bswap32(q.z);
            at[3 * kHuffThreads] = __builtin_bswap32(q.w);
        } else {
            at[0 * kHuffThreads] = masked(q.x, filled);
            at[1 * kHuffThreads] = masked(q.y, filled + 1u);
            at[2 * kHuffThreads] = masked(q.z, filled + 2u);
            at[3 * kHuffThreads] = masked(q.w, filled + 3u);
        }
        filled += 4u;
    }
    // The groups in flight go into the ring if half of the ring is free, and the next ones are asked for.  Called once per
    // step of at most three words (two big_values pairs: 94 bits; four quadruples: 40 bits), this keeps at least five words
    // ahead of the position at the top of every step: a look needs three, and the step's second pair may be two words on.
    // (Half a ring per feed: with sixteen words a lane asks for 32 bytes of a line at a time -- the chunk's lanes hold more
    // lines than the L2 does, and a line is often gone before the lane comes back for more of it.)
    __device__ __forceinline__ void feed() {
        if (filled - (pos >> 5) <= kWords - kFeed) {
#pragma unroll
            for (uint32_t k = 0; k < kFeed / 4; ++k) commit(c[k]);
#pragma unroll
            for (uint32_t k = 0; k < kFeed / 4; ++k) c[k] = g[k];
            g += kFeed / 4;
        }
    }
    // granule at absolute bit `bit_off` of the stream `base`, `length` bits long, frame data ending at `frame_end_bit`
    __device__ __forceinline__ void open(const uint8_t *base, uint64_t bit_off, uint32_t length, uint64_t frame_end_bit) {
        const uint64_t group0 = (bit_off >> 7) & ~(uint64_t)(kFeed / 4 - 1);  // feeds are aligned to their own size
        g = reinterpret_cast<const uint4 *>(base) + group0;
        pos = (uint32_t)(bit_off - (group0 << 7));
        end = pos + length;
        const int64_t lim = (int64_t)frame_end_bit - (int64_t)(group0 << 7);
        limit = lim < -(1 << 30) ? -(1 << 30) : (lim > (1 << 30) ? (1 << 30) : (int32_t)lim);
        uint4 p[kWords / 4];
#pragma unroll
        for (uint32_t k = 0; k < kWords / 4; ++k) p[k] = g[k];
#pragma unroll
        for (uint32_t k = 0; k < kFeed / 4; ++k) c[k] = g[kWords / 4 + k];
        g += kWords / 4 + kFeed / 4;
        filled = 0u;
#pragma unroll
        for (uint32_t k = 0; k < kWords / 4; ++k) commit(p[k]);
    }
    // the 64 bits at the read position
    __device__ __forceinline__ void window64(uint32_t &hi, uint32_t &lo) const {
        const uint32_t wi = pos >> 5, sh = pos & 31u;
        const uint32_t w0 = ring[(wi & kMask) * kHuffThreads], w1 = ring[((wi + 1u) & kMask) * kHuffThreads], w2 = ring[((wi + 2u) & kMask) * kHuffThreads];
        hi = (uint32_t)(((((uint64_t)w0 << 32) | w1) << sh) >> 32);
        lo = (uint32_t)(((((uint64_t)w1 << 32) | w2) << sh) >> 32);
    }
    __device__ __forceinline__ uint32_t window() const {  // the 32 bits at the read position
        const uint32_t wi = pos >> 5, sh = pos & 31u;
        const uint32_t w0 = ring[(wi & kMask) * kHuffThreads], w1 = ring[((wi + 1u) & kMask) * kHuffThreads];
        return (uint32_t)(((((uint64_t)w0 << 32) | w1) << sh) >> 32);
    }
    // scalefactors (at most five bits at a time; lanes on different paths): a lane feeds when it has to
    __device__ __forceinline__ uint32_t get(int n) {
        if (n == 0) return 0;
        if (filled < (pos >> 5) + 2u) feed();
        const uint32_t v = window() >> (32 - n);
        pos += (uint32_t)n;
        return v;
    }
};

// Scalefactors of one granule into the thread's LDS column sf[i * kHuffThreads] (rg_mp3dec.cpp: read_scalefactors_v1 /
// read_scalefactors_lsf).  `reuse` = granule 1 of an MPEG-1 long block whose column already holds granule 0's values:
// the groups flagged in scfsi keep them.
__device__ __forceinline__ void huff_scalefactors(BitRing &b, const RgMp3HuffRec &r, bool lsf, bool reuse, uint8_t *__restrict__ sf,
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
            const int n1 = r.mixed ? 17 : 18;  // one loop, one copy of the bit reader's code
            _Pragma("nounroll") for (; i < n1 + 18; ++i) sf[i * S] = (uint8_t)b.get(i < n1 ? s1 : s2);
            _Pragma("nounroll") for (; i < 40; ++i) sf[i * S] = 0;
        } else {
            _Pragma("nounroll") for (int k = 0; k < 4; ++k) {
                const int lo = k == 0 ? 0 : 1 + 5 * k, hi = 6 + 5 * k;  // bands 0-5, 6-10, 11-15, 16-20
                const int bits = k < 2 ? s1 : s2;
                if (reuse && ((r.scfsi >> k) & 1)) continue;
                _Pragma("nounroll") for (int band = lo; band < hi; ++band) sf[band * S] = (uint8_t)b.get(bits);
            }
            _Pragma("nounroll") for (int i = 21; i < 40; ++i) sf[i * S] = 0;
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
        _Pragma("nounroll") for (int k = 0; k < 4; ++k) {
            const int n = kPart[set][kind][k];
            _Pragma("nounroll") for (int q = 0; q < n; ++q, ++i) {
                const int v = (int)b.get(slen[k]);
                sf[i * S] = (uint8_t)v;
                if (r.intensity_right && slen[k] > 0 && v == (1 << slen[k]) - 1) *illegal |= 1ull << i;
            }
        }
        _Pragma("nounroll") for (; i < 40; ++i) sf[i * S] = 0;
    }
}

}  // namespace

#ifdef RG_HF_TIMING
// Instrument (tools/bh_timing.py --huffman): shader-clock stamps of the eight waves of block RG_HF_TIMING at the kernel's phases
__device__ unsigned long long rg_hf_dbg[8][8];
extern "C" int rg_hf_dbg_read(void *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rg_hf_dbg), sizeof(unsigned long long) * 64); }
#define RG_HF_STAMP(e) do { if (blockIdx.x == RG_HF_TIMING && (threadIdx.x & 63) == 0) rg_hf_dbg[threadIdx.x >> 6][e] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RG_HF_STAMP(e) do { } while (0)
#endif
#ifndef RG_HF_WAVES
#define RG_HF_WAVES (RG_HF_RING == 8 ? 6 : 4)  // per SIMD: three blocks per CU with the 8-word ring (51.6 KB of LDS), two with the 16-word one (67.6 KB)
#endif
__global__ void __launch_bounds__(kHuffThreads) __attribute__((amdgpu_waves_per_eu(RG_HF_WAVES, RG_HF_WAVES)))
rg_mp3_huffman_kernel(const RgMp3DevTables *__restrict__ T, const RgMp3DevHuff *__restrict__ H,
                      const RgMp3DevTrack *__restrict__ tracks, uint32_t n_tracks, const RgMp3HuffRec *__restrict__ recs,
                      const uint8_t *__restrict__ main, rg_mp3_unit *__restrict__ units, int16_t *__restrict__ is,
                      const uint32_t *__restrict__ perm, const uint32_t *__restrict__ sortw, uint64_t total_units) {
    __shared__ uint16_t E_all[RG_MP3_HUFF_LDS_ENTRIES];
    __shared__ uint32_t t_rp[32];
    __shared__ uint8_t quadA[64];
    // the scalefactors' columns (40 bytes per thread) and, once every thread of the block has taken its scalefactors into
    // registers, the columns in which the spectrum's sectors are put together (8 words per thread): 20 KB instead of 36, and with
    // the tables and the bit ring 51.6 KB -- three blocks per CU
    __shared__ __attribute__((aligned(16))) uint8_t sf_all[40 * kHuffThreads];
    uint32_t *const stg_all = reinterpret_cast<uint32_t *>(sf_all);
    static_assert(8 * 4 <= 40, "the sector columns fit where the scalefactor columns were");
    __shared__ uint32_t ring_all[RG_HF_RING * kHuffThreads];
    const int tid = threadIdx.x;
    RG_HF_STAMP(0);
    const uint32_t n_e = H->n_entries;  // + 2 <= RG_MP3_HUFF_LDS_ENTRIES: checked when the tables are uploaded; e16[n_e], [n_e + 1] = 0
    for (uint32_t i = tid; 2 * i < n_e + 2; i += kHuffThreads) reinterpret_cast<uint32_t *>(E_all)[i] = reinterpret_cast<const uint32_t *>(H->e16)[i];
    // what the big_values loop needs of a table in one word: first entry | (32 - primary bits) << 16 | linbits << 24.  A table
    // that codes nothing (0, 4, 14) is the two zero entries behind the last table looked at through one bit: a code of no bits
    // for the pair (0, 0) -- the loop has no case for it
    if (tid < 32) {
        const uint32_t P = H->primary_bits[tid];
        t_rp[tid] = P ? (H->base[tid] | ((32u - P) << 16) | ((uint32_t)H->linbits[tid] << 24)) : (n_e | (31u << 16));
    }
    if (tid < 64) quadA[tid] = H->quadA[tid];
    __syncthreads();
    RG_HF_STAMP(1);
    // Lane i takes the i-th unit of the chunk's units ordered by big_values (rg_mp3_sort_*): a wave lasts as long as its
    // longest lane, and in stream order a wave holds mid and side channels, loud and quiet granules side by side -- 2 to 3
    // times the iterations its units need on average.
#ifndef RG_HF_STRIPE
    // Blocks in the order of the sort, heaviest first: a chunk is several generations of blocks (rg_files.hip: 768 K units
    // against 262 144 resident lanes), the long blocks start first and the short ones fill in behind them.
    const uint32_t slot = blockIdx.x * kHuffThreads + (uint32_t)tid;
#else
    // Tried, and better only when a chunk is a single generation of blocks (0.158 / 0.142 / 0.060 ms per 256 K units without,
    // 0.171 / 0.154 / 0.082 with, in 768 K-unit chunks): the sorted order dealt to the blocks in stripes -- wave w of a block takes
    // the block's share of the w-th eighth of the order (waves w and w + 4, which share a SIMD, the eighths w and 7 - w) -- so
    // that every SIMD gets the same mix of long and short waves.  But then every block lasts as long as the longest waves, and
    // its slots are free only when it has ended.
    const uint32_t wv = (uint32_t)tid >> 6;
    const uint32_t slot = (((wv < 4u ? wv : 11u - wv) * gridDim.x + blockIdx.x) << 6) + ((uint32_t)tid & 63u);
#endif
    const uint32_t n_valid = sortw[RG_MP3_SORT_NVALID];
    if ((slot & ~63u) >= n_valid) return;  // the whole wave lies behind the last unit that decodes
    // the lanes behind it in the one wave that holds it decode that last unit once more (the same bytes to the same places):
    // the loops below are the wave's, and nothing in them has to ask whether a lane is there
    const uint64_t u = perm[slot < n_valid ? slot : n_valid - 1u];
    const uint32_t ti = find_by_unit(tracks, n_tracks, u);
    const RgMp3DevTrack tr = tracks[ti];
    const int nch = (int)tr.channels, rr = (int)tr.rate_row;
    const RgMp3HuffRec r = recs[u];
    uint8_t *__restrict__ sf = sf_all + tid;
    BitRing b;  // the records' bit offsets are relative to the track's stream
    b.ring = ring_all + tid;
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
    // the unit's scalefactors, out of LDS for good
    uint32_t sfw[10];
#pragma unroll
    for (int i = 0; i < 10; ++i)
    {
        sfw[i] = (uint32_t)sf[(4 * i) * kHuffThreads] | ((uint32_t)sf[(4 * i + 1) * kHuffThreads] << 8) | ((uint32_t)sf[(4 * i + 2) * kHuffThreads] << 16) |
                 ((uint32_t)sf[(4 * i + 3) * kHuffThreads] << 24);
        if (i & 1) __builtin_amdgcn_sched_barrier(0);  // eight reads in flight, not forty: their registers set the kernel's count
    }
    // ... and into the unit at once (its first 48 bytes; the last 16 -- nz is among them -- follow when the spectrum is done): ten
    // registers less through the loops below
    {
        uint4 *const up = reinterpret_cast<uint4 *>(units + u);
        up[0] = make_uint4(sfw[0], sfw[1], sfw[2], sfw[3]);
        up[1] = make_uint4(sfw[4], sfw[5], sfw[6], sfw[7]);
        up[2] = make_uint4(sfw[8], sfw[9], (uint32_t)illegal, (uint32_t)(illegal >> 32));
    }
    // (waves that lie wholly behind the last unit have ended above: the hardware's barrier counts the waves still there)
    __syncthreads();
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
    // Out: the unit's row of RG_MP3_ROW_BYTES, sign and magnitude (rg_mp3dev.h): line i's byte `sign << 7 | |v| & 127` at i,
    // and |v| >> 7 at 576 + i for the lines below hi_lines -- the end of the last region whose table has seven or more
    // linbits; no other line can hold a value above 127 -- which for most units is none.  Both loops run over LINE NUMBERS
    // that are the same for the whole wave (a lane whose run is shorter sits out): so the word a lane has just finished is
    // the same word of its row for every lane, it goes to the lane's LDS column, and when the eighth word of a 32-byte sector
    // is in, every lane stores its sector from there at once -- one branch of the wave, not 64 lanes taking the detour in
    // turns; a 16-byte store per lane, half a sector, would be a read-modify-write at the memory, since no two lanes of a wave
    // share a line (they are units of equal length from anywhere in the chunk).
    const int e0 = r0e < bv2 ? r0e : bv2, e1 = r1e < bv2 ? r1e : bv2;
    const uint32_t rp0 = t_rp[r.table_select[0]], rp1 = t_rp[r.table_select[1]], rp2 = t_rp[r.table_select[2]];
    const int hi_lines = ((rp2 >> 24) >= 7u && e1 < bv2) ? bv2 : (((rp1 >> 24) >= 7u && e0 < e1) ? e1 : (((rp0 >> 24) >= 7u && e0 > 0) ? e0 : 0));
    const int hi_q = (hi_lines + 3) >> 2;
    uint32_t *__restrict__ const row = reinterpret_cast<uint32_t *>(is) + u * (RG_MP3_ROW_BYTES / 4);
    uint32_t *__restrict__ const stg = stg_all + tid;
    auto store_sector = [&](const int m) {  // the eight words of sector m from the lane's column
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = stg[i * kHuffThreads];
#ifdef RG_HF_NOSTORE  // experiment (wrong results): the spectrum is not written
        if (w[0] == 0x12345678u)
#endif
        {
            uint4 *const dst = reinterpret_cast<uint4 *>(row + 8 * m);
            dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
            dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
    };
    auto wave_max = [](int v) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(v, d); v = o > v ? o : v; }
        return __builtin_amdgcn_readfirstlane(v);
    };
    uint32_t pend = 0u, pend_hi = 0u;  // the first pair of the word under way
    int hi_found = 0;                  // words of the second plane up to the last one that holds anything
    {
        // One loop over the big_values pairs of all three regions, and one path through it: which table a pair uses, whether
        // its code runs on into a second table, whether its values have linbits or signs are per-lane SELECTS -- the lanes of a
        // wave sit in different regions of different granules, and a branch that one lane takes costs the wave its whole body.
        const int wave_bv2 = wave_max(bv2);
        for (int L = 0; L < wave_bv2; L += 4) {  // a word of the row per step
            b.feed();
            uint32_t lo16[2], hi16[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int Lh = L + 2 * h;
                const bool live = Lh < bv2 && b.pos < b.end;  // a granule whose bits have run out is zeros from here on
                const uint32_t rp = Lh < e0 ? rp0 : (Lh < e1 ? rp1 : rp2);
                const uint16_t *__restrict__ E = E_all + (rp & 0xFFFFu);
                const uint32_t sP = (rp >> 16) & 0xFFu, linbits = rp >> 24;
                uint32_t W0, W1;
                b.window64(W0, W1);
                const uint32_t e1st = E[W0 >> sP];
                // a code longer than the primary bits: the entry names a second table and how many more bits index it; for a
                // leaf the same arithmetic reads some entry of the table that is then not used
                const uint32_t sub = e1st & 15u;
                const uint32_t e2nd = E[((e1st >> 4) & 0x7FFu) + __builtin_amdgcn_ubfe(W0 << (32u - sP), 32u - sub, sub)];
                const bool link = (e1st & 0x8000u) != 0u;
                const uint32_t e = link ? e2nd : e1st;
                uint32_t used = (link ? 32u - sP : 0u) + (e & 15u);
                uint32_t x = (e >> 8) & 15u, y = (e >> 4) & 15u;
                // behind the code: [linbits of x] [sign of x] [linbits of y] [sign of y], each only if its value calls for it
                uint32_t T = (uint32_t)(((((uint64_t)W0 << 32) | W1) << used) >> 32);
                const uint32_t lx = x == 15u ? linbits : 0u;
                x += __builtin_amdgcn_ubfe(T, 32u - lx, lx);
                T <<= lx;
                const uint32_t nx = x != 0u ? 1u : 0u, sx = (T >> 31) & nx;
                T <<= nx;
                const uint32_t ly = y == 15u ? linbits : 0u;
                y += __builtin_amdgcn_ubfe(T, 32u - ly, ly);
                T <<= ly;
                const uint32_t ny = y != 0u ? 1u : 0u, sy = (T >> 31) & ny;
                used += lx + nx + ly + ny;
                b.pos += live ? used : 0u;
                lo16[h] = live ? ((x & 127u) | (sx << 7) | ((y & 127u) << 8) | (sy << 15)) : 0u;
                hi16[h] = live ? ((x >> 7) | ((y >> 7) << 8)) : 0u;
            }
            const uint32_t lo0 = lo16[0], lo1 = lo16[1], hi0 = hi16[0], hi1 = hi16[1];
            const int w = L >> 2;
            if (L + 2 < bv2) {  // both pairs are the lane's
                stg[(w & 7) * kHuffThreads] = lo0 | (lo1 << 16);
                if (w < hi_q) {
                    const uint32_t hw = hi0 | (hi1 << 16);
                    row[RG_MP3_ROW_BYTES / 8 + w] = hw;
                    if (hw) hi_found = w + 1;
                }
                if ((w & 7) == 7) store_sector(w >> 3);  // the test of w is the wave's
            } else if (L < bv2) {  // the run ends inside this word: count1 goes on in it
                pend = lo0;
                pend_hi = hi0;
            }
        }
        if ((bv2 & 2) && (bv2 >> 2) < hi_q) {  // the count1 lines beside it are 0 or +-1
            row[RG_MP3_ROW_BYTES / 8 + (bv2 >> 2)] = pend_hi;
            if (pend_hi) hi_found = (bv2 >> 2) + 1;
        }
    }
    RG_HF_STAMP(3);
    // count1: a quadruple is four bytes.  Where the big_values run ends in the middle of a word (bv2 = 2 mod 4) the lane's
    // quadruples straddle the words: it carries two bytes from word to word.
    int line = bv2, nz = bv2;
    {
        const int w_first = bv2 >> 2, sh = (bv2 & 2) ? 16 : 0;
        uint32_t carry = pend;
        bool done = false;
        int zero_until = -1;  // a lane that is through fills its last sector with zeros
        b.feed();
        for (int w = 143 - wave_max(143 - w_first); w < 144; ++w) {
            if ((w & 3) == 3) b.feed();  // a quadruple is ten bits at most
            const bool mine = !done && w >= w_first;  // then line == 4 w (+ 2)
            const bool take = mine && line <= 572 && b.pos < b.end;
            const uint32_t win = b.window();
            const uint32_t qa = quadA[win >> 26];
            const uint32_t v = r.count1table ? (~(win >> 28)) & 15u : qa & 15u;
            uint32_t used = r.count1table ? 4u : qa >> 4;
            uint32_t q = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // 0, or 1 with its sign in the next bit of the stream
                const uint32_t bit = (v >> (3 - k)) & 1u;
                q |= (bit | ((bit & ((win << used) >> 31)) << 7)) << (8 * k);
                used += bit;
            }
            const bool got = take && b.pos + used <= b.end;  // else the quadruple ran past the granule's bits: stuffing, not data
            if (take) b.pos += used;
            if (got) {
                const uint64_t full = (uint64_t)carry | ((uint64_t)q << sh);
                stg[(w & 7) * kHuffThreads] = (uint32_t)full;
                carry = (uint32_t)(full >> 32);
                line += 4;
            } else if (mine) {
                done = true;
                nz = line;
                stg[(w & 7) * kHuffThreads] = carry;  // lines 4 w, 4 w + 1 if the lane carries, else nothing: zeros
                zero_until = w | 7;
            } else if (w <= zero_until) {
                stg[(w & 7) * kHuffThreads] = 0u;
            }
            if ((w & 7) == 7) {
                // lanes with lines in this sector that the big_values loop has not stored already
                if (w_first <= w && (!done || ((nz + 3) >> 2) > (w & ~7))) store_sector(w >> 3);
                if (__builtin_amdgcn_ballot_w64(!done && w_first < 144) == 0) break;
            }
        }
        if (!done) nz = line > 576 ? 576 : line;
    }
    RG_HF_STAMP(4);
    // ---- the unit's last 16 bytes ----
    static_assert(offsetof(rg_mp3_unit, nz) == 48 && sizeof(rg_mp3_unit) == 64, "unit layout");
    {
        const uint32_t w0 = (uint32_t)nz | ((uint32_t)r.global_gain << 16) | ((uint32_t)r.block_type << 24);
        const uint32_t w1 = (uint32_t)r.mixed | ((uint32_t)r.subblock_gain[0] << 8) | ((uint32_t)r.subblock_gain[1] << 16) | ((uint32_t)r.subblock_gain[2] << 24);
        const uint32_t w2 = (uint32_t)r.scalefac_scale | ((uint32_t)(preflag & 0xFF) << 8) | ((uint32_t)long_end << 16) | ((uint32_t)short_start << 24);
        const uint32_t w3 = (uint32_t)r.mode_ext | ((uint32_t)r.intensity_scale << 8) | ((uint32_t)(hi_found & 0xFF) << 16);  // reserved[0]: words of the second plane worth reading
        reinterpret_cast<uint4 *>(units + u)[3] = make_uint4(w0, w1, w2, w3);
    }
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
    // (the frame's forty bytes too: the side-information reader walks them with a running bit position)
    __shared__ uint64_t raw_all[RG_MP3_FRAME_TILE][RG_MP3_SLOT_BYTES / 8];
    uint64_t *const raw = raw_all[threadIdx.x];
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
    // A frame's (up to four) records wait in LDS, a slice per thread: rg_mp3_frame_records -- the host's code -- fills them
    // through a running index, which as a private array is 208 bytes of scratch memory per thread (rounds 4-5).
    __shared__ RgMp3HuffRec r_all[RG_MP3_FRAME_TILE][4];
    RgMp3HuffRec *const r = r_all[threadIdx.x];
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

// ---------------------------------------------------------------------------------------------------------------------
// The lane sort of the Huffman stage.  The Huffman kernel runs one thread per unit and a wave lasts as long as its longest
// lane: in stream order a wave holds both channels of sixteen frames -- the mid channel of a loud granule beside the side
// channel of a quiet one -- and executes 2 to 3 times the iterations its units need on average (the dense 128 kb/s
// joint-stereo stream: sum over waves of (longest big_values run + longest count1 run) = 2.9 x the mean; ordered by
// big_values 1.3 x).  So the units of a chunk are dealt to the lanes by a counting sort on big_values >> 2, heaviest first:
// a histogram, a scan of its 73 buckets, a scatter with one returning atomic per bucket and block.  The order inside a bucket
// is whatever the atomics make it -- every unit is decoded to its own row whichever lane takes it.
namespace {
constexpr int kSortThreads = 1024, kSortPer = 4;
// bucket of unit u, -1 for a unit past what the frame parser found decodable
__device__ __forceinline__ int rg_mp3_sort_key(const RgMp3DevTrack *__restrict__ tracks, uint32_t n_tracks, const RgMp3HuffRec *__restrict__ recs,
                                               uint64_t u, uint64_t total_units) {
    if (u >= total_units) return -1;
    const uint32_t ti = find_by_unit(tracks, n_tracks, u);
    const uint64_t local = u - tracks[ti].unit_base;
    if (local >= (uint64_t)tracks[ti].n_granules * tracks[ti].channels) return -1;
    const uint32_t bv = recs[u].big_values;
    return 72 - (int)((bv > 288u ? 288u : bv) >> 2);
}
}  // namespace

__global__ void __launch_bounds__(kSortThreads)
rg_mp3_sort_hist_kernel(const RgMp3DevTrack *__restrict__ tracks, uint32_t n_tracks, const RgMp3HuffRec *__restrict__ recs, uint64_t total_units,
                        uint32_t *__restrict__ sortw) {
    __shared__ uint32_t h[RG_MP3_SORT_BUCKETS];
    const int tid = threadIdx.x;
    if (tid < RG_MP3_SORT_BUCKETS) h[tid] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortPer; ++k) {
        const int key = rg_mp3_sort_key(tracks, n_tracks, recs, ((uint64_t)blockIdx.x * kSortPer + k) * kSortThreads + tid, total_units);
        if (key >= 0) atomicAdd(&h[key], 1u);
    }
    __syncthreads();
    if (tid < RG_MP3_SORT_BUCKETS && h[tid]) atomicAdd(&sortw[tid], h[tid]);
}

// one block of 128 threads: exclusive prefix of the histogram -> the buckets' cursors, the total -> sortw[NVALID]; the
// histogram is zero again afterwards (the next chunk of this staging set adds into it)
__global__ void __launch_bounds__(128) rg_mp3_sort_scan_kernel(uint32_t *__restrict__ sortw) {
    __shared__ uint32_t first_wave_total;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t v = tid < RG_MP3_SORT_BUCKETS ? sortw[tid] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t x = __shfl_up(inc, d);
        if (lane >= d) inc += x;
    }
    if (tid == 63) first_wave_total = inc;
    __syncthreads();
    const uint32_t excl = inc - v + (tid >= 64 ? first_wave_total : 0u);
    if (tid < RG_MP3_SORT_BUCKETS) {
        sortw[RG_MP3_SORT_BUCKETS + tid] = excl;
        sortw[tid] = 0u;
    }
    if (tid == 127) sortw[RG_MP3_SORT_NVALID] = excl + v;
}

__global__ void __launch_bounds__(kSortThreads)
rg_mp3_sort_scatter_kernel(const RgMp3DevTrack *__restrict__ tracks, uint32_t n_tracks, const RgMp3HuffRec *__restrict__ recs, uint64_t total_units,
                           uint32_t *__restrict__ sortw, uint32_t *__restrict__ perm) {
    __shared__ uint32_t h[RG_MP3_SORT_BUCKETS], base[RG_MP3_SORT_BUCKETS];
    const int tid = threadIdx.x;
    if (tid < RG_MP3_SORT_BUCKETS) h[tid] = 0u;
    __syncthreads();
    int key[kSortPer];
    uint32_t rank[kSortPer];
#pragma unroll
    for (int k = 0; k < kSortPer; ++k) {
        key[k] = rg_mp3_sort_key(tracks, n_tracks, recs, ((uint64_t)blockIdx.x * kSortPer + k) * kSortThreads + tid, total_units);
        rank[k] = key[k] >= 0 ? atomicAdd(&h[key[k]], 1u) : 0u;
    }
    __syncthreads();
    if (tid < RG_MP3_SORT_BUCKETS && h[tid]) base[tid] = atomicAdd(&sortw[RG_MP3_SORT_BUCKETS + tid], h[tid]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortPer; ++k)
        if (key[k] >= 0) perm[base[key[k]] + rank[k]] = (uint32_t)(((uint64_t)blockIdx.x * kSortPer + k) * kSortThreads + tid);
}

// sortw: RG_MP3_SORT_WORDS words, zero before the first use; perm: total_units words
extern "C" hipError_t rg_launch_mp3_sort(const RgMp3DevTrack *d_tracks, uint32_t n_tracks, const RgMp3HuffRec *d_recs, uint64_t total_units,
                                         uint32_t *d_sortw, uint32_t *d_perm, hipStream_t s) {
    if (total_units == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)((total_units + kSortThreads * kSortPer - 1) / (kSortThreads * kSortPer));
    hipLaunchKernelGGL(rg_mp3_sort_hist_kernel, dim3(blocks), dim3(kSortThreads), 0, s, d_tracks, n_tracks, d_recs, total_units, d_sortw);
    hipLaunchKernelGGL(rg_mp3_sort_scan_kernel, dim3(1), dim3(128), 0, s, d_sortw);
    hipLaunchKernelGGL(rg_mp3_sort_scatter_kernel, dim3(blocks), dim3(kSortThreads), 0, s, d_tracks, n_tracks, d_recs, total_units, d_sortw, d_perm);
    return hipGetLastError();
}

extern "C" hipError_t rg_launch_mp3_huffman(const RgMp3DevTables *d_tab, const RgMp3DevHuff *d_huff, const RgMp3DevTrack *d_tracks,
                                            uint32_t n_tracks, const RgMp3HuffRec *d_recs, const uint8_t *d_main, rg_mp3_unit *d_units,
                                            int16_t *d_is, uint64_t total_units, const uint32_t *d_perm, const uint32_t *d_sortw, hipStream_t s) {
    if (total_units == 0) return hipSuccess;
    hipLaunchKernelGGL(rg_mp3_huffman_kernel, dim3((uint32_t)((total_units + kHuffThreads - 1) / kHuffThreads)), dim3(kHuffThreads), 0, s, d_tab,
                       d_huff, d_tracks, n_tracks, d_recs, d_main, d_units, d_is, d_perm, d_sortw, total_units);
    return hipGetLastError();
}

// n_runs: blocks of the grid, one per run of RG_MP3_RUN granules (RgMp3DevTrack::run_base); planes: the form of d_is (rg_mp3dev.h):
// 1 = two planes of a byte per line (behind the device Huffman stage), 0 = rows of int16 (spectra parsed on the host)
extern "C" hipError_t rg_launch_mp3_backhalf(const RgMp3DevTables *d_tab, const RgMp3DevTrack *d_tracks, uint32_t n_tracks,
                                             uint32_t n_runs, const rg_mp3_unit *d_units, const int16_t *d_is, uint32_t planes,
                                             hipStream_t s) {
    if (n_runs == 0) return hipSuccess;
    hipLaunchKernelGGL(rg_mp3_backhalf_kernel, dim3(n_runs), dim3(RG_MP3_BH_THREADS), 0, s, d_tab, d_tracks, n_tracks, d_units, d_is, planes);
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
