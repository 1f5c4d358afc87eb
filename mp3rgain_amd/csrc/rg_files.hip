// rg_files.hip -- the file-level entry points of the path (last row): analyze_track /
// analyze_album / find_peak_amplitude on files, as the reference's public functions take them
// (src/replaygain.rs:929-941, 1033-1074, 1140-1249).
//
// The reference gets PCM from symphonia (src/replaygain.rs:807-904).  Here a file is, by content:
//   * an MPEG-1/2/2.5 Layer III stream (optionally behind an ID3v2 tag): decoded by the library's own decoder
//     (rg_mp3dec.cpp, include/mp3rgain_amd_dec.h) into planar f32 that goes to HBM as it is -- the files of an
//     album are decoded on all host cores at once;
//   * a RIFF/WAVE file (integer PCM 8/16/24/32 bit, IEEE float 32 bit, plain or WAVE_FORMAT_EXTENSIBLE): the
//     interleaved bytes are copied to HBM and turned into the planar arena by a device kernel;
//   * anything else (M4A/AAC: no AAC decoder is built yet) through an external decoder command that writes a WAV
//     stream to stdout (rg_set_decoder_command, e.g. "ffmpeg -v error -i {} -f wav -c:a pcm_f32le -").
// Everything after the arena is the same path as rg_analyze_pcm_batch.
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <vector>

#include <new>
#include <stdexcept>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "../../include/mp3rgain_amd.h"
#include "../../include/mp3rgain_amd_dec.h"
#include "../../include/mp3rgain_amd_mp4.h"
#include "../../include/mp3rgain_amd_demux.h"
#include "rg_ctx.h"
#include "rg_mp3dev.h"
#include "rg_mp3dev_host.h"
#include "rg_mp3_frame.h"

// =================================================================================================
// WAV container (host)
namespace {

uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

enum WavKind { WAV_U8 = 0, WAV_S16 = 1, WAV_S24 = 2, WAV_S32 = 3, WAV_F32 = 4 };

int wav_kind(const rg_wav_info &w) {
    if (w.sample_format == 3) return w.bits_per_sample == 32 ? WAV_F32 : -1;
    if (w.sample_format != 1) return -1;
    switch (w.bits_per_sample) {
        case 8: return WAV_U8;
        case 16: return WAV_S16;
        case 24: return WAV_S24;
        case 32: return WAV_S32;
        default: return -1;
    }
}

// the sample format of the planar arena a WAV kind is converted to: 8/16-bit -> S16, 24/32-bit -> S32
// (same normalised amplitude: x/2^15 resp. x/2^31, src/replaygain.rs:984-1018), float -> F32
uint16_t planar_format(int kind) { return kind == WAV_F32 ? RG_FMT_F32_PLANAR : (kind <= WAV_S16 ? RG_FMT_S16_PLANAR : RG_FMT_S32_PLANAR); }

}  // namespace

extern "C" int rg_wav_parse(const void *data, size_t len, rg_wav_info *out) {
    if (!data || !out) return RG_ERR_INVALID_ARG;
    const uint8_t *d = (const uint8_t *)data;
    memset(out, 0, sizeof *out);
    if (len < 12 || memcmp(d, "RIFF", 4) != 0 || memcmp(d + 8, "WAVE", 4) != 0) return RG_ERR_INVALID_ARG;
    bool have_fmt = false;
    size_t pos = 12;
    while (pos + 8 <= len) {
        const uint32_t size = le32(d + pos + 4);
        const size_t body = pos + 8;
        if (memcmp(d + pos, "fmt ", 4) == 0) {
            if (size < 16 || body + 16 > len) return RG_ERR_INVALID_ARG;
            uint16_t tag = le16(d + body);
            out->channels = le16(d + body + 2);
            out->sample_rate = le32(d + body + 4);
            out->block_align = le16(d + body + 12);
            out->bits_per_sample = le16(d + body + 14);
            if (tag == 0xFFFE && size >= 40 && body + 40 <= len) tag = le16(d + body + 24);  // SubFormat GUID, first field
            out->sample_format = tag;
            have_fmt = true;
        } else if (memcmp(d + pos, "data", 4) == 0) {
            if (!have_fmt) return RG_ERR_INVALID_ARG;
            const uint32_t bytes_per_frame = (uint32_t)out->channels * (out->bits_per_sample / 8u);
            if (out->channels == 0 || bytes_per_frame == 0 || out->block_align != bytes_per_frame) return RG_ERR_INVALID_ARG;
            // a streamed WAV (decoder pipe) cannot know its length: 0 or 0xFFFFFFFF mean "to the end"
            uint64_t avail = len - body;
            uint64_t n = (size == 0 || size == 0xFFFFFFFFu || size > avail) ? avail : size;
            out->data_offset = body;
            out->frames = n / bytes_per_frame;
            return RG_OK;
        }
        const uint64_t next = (uint64_t)body + size + (size & 1u);  // chunks are word aligned
        if (next > len) break;
        pos = (size_t)next;
    }
    return RG_ERR_INVALID_ARG;
}

// =================================================================================================
// interleaved bytes -> planar arena (device).  One thread per frame in the general kernel; stereo f32 and
// stereo s16 (what decoders emit) move 16 bytes per lane per access when the planes are 16-byte aligned.
namespace {

template <int KIND> struct WavIn;
template <> struct WavIn<WAV_U8> { typedef int16_t out_t; static constexpr int bytes = 1;
    static __device__ out_t load(const uint8_t *p) { return (int16_t)(((int)p[0] - 128) * 256); } };
template <> struct WavIn<WAV_S16> { typedef int16_t out_t; static constexpr int bytes = 2;
    static __device__ out_t load(const uint8_t *p) { return (int16_t)(p[0] | (p[1] << 8)); } };
template <> struct WavIn<WAV_S24> { typedef int32_t out_t; static constexpr int bytes = 3;
    static __device__ out_t load(const uint8_t *p) { return (int32_t)(((uint32_t)p[0] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 24)); } };
template <> struct WavIn<WAV_S32> { typedef int32_t out_t; static constexpr int bytes = 4;
    static __device__ out_t load(const uint8_t *p) { return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); } };
template <> struct WavIn<WAV_F32> { typedef float out_t; static constexpr int bytes = 4;
    static __device__ out_t load(const uint8_t *p) {
        return __uint_as_float((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); } };

template <int KIND>
__global__ void __launch_bounds__(256)
rg_deinterleave_kernel(const uint8_t *__restrict__ src, void *__restrict__ dst, uint64_t first, uint64_t frames, uint32_t channels) {
    typedef WavIn<KIND> W;
    typedef typename W::out_t T;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t f = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; f < frames; f += stride) {
        const uint8_t *p = src + f * channels * W::bytes;
        for (uint32_t c = 0; c < channels; ++c) reinterpret_cast<T *>(dst)[(uint64_t)c * frames + f] = W::load(p + c * W::bytes);
    }
}

// stereo, 4-byte samples (float and s32 share the bit copy): `quads` groups of four frames
__global__ void __launch_bounds__(256)
rg_deinterleave_stereo32_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ left, uint4 *__restrict__ right, uint64_t quads) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += stride) {
        const uint4 a = src[2 * q], b = src[2 * q + 1];  // L0 R0 L1 R1 | L2 R2 L3 R3
        left[q] = make_uint4(a.x, a.z, b.x, b.z);
        right[q] = make_uint4(a.y, a.w, b.y, b.w);
    }
}

__global__ void __launch_bounds__(256)
rg_deinterleave_stereo16_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ left, uint4 *__restrict__ right, uint64_t octs) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < octs; q += stride) {
        const uint4 a = src[2 * q], b = src[2 * q + 1];  // eight frames: each word = L (low half) | R (high half)
        auto lo = [](uint32_t x, uint32_t y) { return (x & 0xFFFFu) | (y << 16); };
        auto hi = [](uint32_t x, uint32_t y) { return (x >> 16) | (y & 0xFFFF0000u); };
        left[q] = make_uint4(lo(a.x, a.y), lo(a.z, a.w), lo(b.x, b.y), lo(b.z, b.w));
        right[q] = make_uint4(hi(a.x, a.y), hi(a.z, a.w), hi(b.x, b.y), hi(b.z, b.w));
    }
}

uint32_t grid_for(uint64_t items) {
    const uint64_t blocks = (items + 255) / 256;
    return (uint32_t)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));  // 8192 blocks = 32 per CU; grid-stride beyond
}

template <int KIND>
void launch_general(const uint8_t *src, void *dst, uint64_t first, uint64_t frames, uint32_t channels, hipStream_t s) {
    if (first >= frames) return;
    hipLaunchKernelGGL((rg_deinterleave_kernel<KIND>), dim3(grid_for(frames - first)), dim3(256), 0, s, src, dst, first, frames, channels);
}

hipError_t launch_deinterleave(int kind, const uint8_t *src, void *dst, uint64_t frames, uint32_t channels, hipStream_t s) {
    if (frames == 0) return hipSuccess;
    uint64_t done = 0;
    const bool aligned = (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
    if (channels == 2 && aligned && (kind == WAV_F32 || kind == WAV_S32) && (frames * 4) % 16 == 0) {
        const uint64_t quads = frames / 4;
        hipLaunchKernelGGL(rg_deinterleave_stereo32_kernel, dim3(grid_for(quads)), dim3(256), 0, s, (const uint4 *)src,
                           (uint4 *)dst, (uint4 *)((uint8_t *)dst + frames * 4), quads);
        done = quads * 4;
    } else if (channels == 2 && aligned && kind == WAV_S16 && (frames * 2) % 16 == 0) {
        const uint64_t octs = frames / 8;
        hipLaunchKernelGGL(rg_deinterleave_stereo16_kernel, dim3(grid_for(octs)), dim3(256), 0, s, (const uint4 *)src, (uint4 *)dst,
                           (uint4 *)((uint8_t *)dst + frames * 2), octs);
        done = octs * 8;
    }
    switch (kind) {
        case WAV_U8: launch_general<WAV_U8>(src, dst, done, frames, channels, s); break;
        case WAV_S16: launch_general<WAV_S16>(src, dst, done, frames, channels, s); break;
        case WAV_S24: launch_general<WAV_S24>(src, dst, done, frames, channels, s); break;
        case WAV_S32: launch_general<WAV_S32>(src, dst, done, frames, channels, s); break;
        default: launch_general<WAV_F32>(src, dst, done, frames, channels, s); break;
    }
    return hipGetLastError();
}

// =================================================================================================
// host plumbing
struct WavItem {
    const uint8_t *bytes;
    rg_wav_info info;
    int kind;
    uint64_t src_off;  // in the interleaved staging buffer
    uint64_t src_len;
};

// One input of the file layer after loading: either the bytes of a WAV stream, or planar f32 PCM from the MP3 decoder
struct LoadedAudio {
    std::vector<uint8_t> wav;
    std::vector<float> planar;  // [channels][frames]
    uint32_t sample_rate = 0, channels = 0;
    uint64_t frames = 0;
    bool decoded = false;       // planar is valid
    // split decode (tuning key 6): stage A ran on the host, stages B-E will run on the device into the arena
    std::vector<int16_t> is;
    std::vector<rg_mp3_unit> units;
    uint64_t n_units = 0;
    uint32_t lsf = 0;
    bool split = false;
    // tuning key 6 = 2: the host only walks the frames; scalefactors and Huffman run on the device as well
    std::vector<uint8_t> main_stream;
    std::vector<RgMp3HuffRec> recs;
    std::vector<uint8_t> file_bytes;  // the file as read
    bool is_mp4 = false;
    uint32_t n_audio_tracks = 1;  // an MP4 file: what its sample tables say (include/mp3rgain_amd_demux.h); anything else has one
    // tuning key 6 = 3: the loader pipeline has decoded the stream into the arena already (planar f32 at arena_off);
    // `frames` is what the device found decodable
    bool staged = false;
    uint64_t arena_off = 0;
    uint64_t walked_frames = 0;  // PCM frames if every walked frame decodes: what the arena is laid out for
    uint32_t result_index = 0;
    // ready for the next file; the vectors keep their capacity
    void reset() {
        wav.clear(); planar.clear(); is.clear(); units.clear(); main_stream.clear(); recs.clear(); file_bytes.clear();
        sample_rate = channels = 0; frames = 0; n_units = 0; lsf = 0;
        decoded = split = is_mp4 = staged = false;
        arena_off = 0; walked_frames = 0; result_index = 0; n_audio_tracks = 1;
    }
};

// the context's pool of LoadedAudio (rg_ctx::file_pool): entry i serves the i-th file of a call
std::vector<LoadedAudio> &file_pool(rg_ctx *c, size_t n) {
    if (!c->file_pool) {
        c->file_pool = new std::vector<LoadedAudio>();
        c->file_pool_free = [](void *p) { delete static_cast<std::vector<LoadedAudio> *>(p); };
    }
    std::vector<LoadedAudio> &pool = *static_cast<std::vector<LoadedAudio> *>(c->file_pool);
    if (pool.size() < n) pool.resize(n);
    for (size_t i = 0; i < n; ++i) pool[i].reset();
    return pool;
}

size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
size_t align64(size_t x) { return (x + 63) & ~(size_t)63; }

// Grow the arena to `need` bytes without losing its first `keep` bytes (PCM that chunks decoded earlier in the call).
// The device is idle when this returns from a growth.
int arena_reserve_keep(rg_ctx *c, size_t need, size_t keep) {
    if (need <= c->d_arena.cap) return RG_OK;
    if (keep == 0 || !c->d_arena.p) {
        RG_HIP(c, c->d_arena.reserve(need));
        return RG_OK;
    }
    RG_HIP(c, hipDeviceSynchronize());
    unsigned char *fresh = nullptr;
    const size_t want = need + need / 2 + 16;
    RG_HIP(c, hipMalloc((void **)&fresh, want));
    hipError_t e = hipMemcpy(fresh, c->d_arena.p, keep, hipMemcpyDeviceToDevice);
    if (e != hipSuccess) { (void)hipFree(fresh); RG_HIP(c, e); }
    (void)hipFree(c->d_arena.p);
    c->d_arena.p = fresh;
    c->d_arena.cap = want;
    return RG_OK;
}

// parse, copy to HBM, de-interleave: on return `descs` describe the planar arena c->d_arena
int stage_wavs(rg_ctx *c, const void *const *wav, const size_t *wav_len, size_t n, std::vector<rg_track_desc> *descs, size_t *arena_bytes) {
    std::vector<WavItem> items(n);
    size_t src_total = 0, dst_total = 0;
    descs->assign(n ? n : 1, rg_track_desc{});
    for (size_t i = 0; i < n; ++i) {
        WavItem &it = items[i];
        it.bytes = (const uint8_t *)wav[i];
        if (!wav[i] || rg_wav_parse(wav[i], wav_len[i], &it.info) != RG_OK)
            return rg_set_err(c, RG_ERR_FORMAT, "input %zu is not a RIFF/WAVE stream", i);
        it.kind = wav_kind(it.info);
        if (it.kind < 0)
            return rg_set_err(c, RG_ERR_FORMAT, "input %zu: unsupported WAV sample format (tag %u, %u bits)", i,
                              it.info.sample_format, it.info.bits_per_sample);
        it.src_off = src_total;
        it.src_len = it.info.frames * it.info.block_align;
        src_total = align16(src_total + it.src_len);
        rg_track_desc &d = (*descs)[i];
        d.offset_bytes = dst_total;
        d.frames = it.info.frames;
        d.sample_rate = it.info.sample_rate;
        d.channels = it.info.channels;
        d.format = planar_format(it.kind);
        dst_total = align16(dst_total + (size_t)it.info.frames * it.info.channels * rg_bytes_per_sample(d.format));
    }
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    // the staging buffers may still be read by an earlier batch
    for (int s = 0; s < c->n_slots; ++s) RG_HIP(c, hipStreamSynchronize(c->slots[s].stream));
    RG_HIP(c, c->d_wav.reserve(src_total ? src_total : 16));
    RG_HIP(c, c->d_arena.reserve(dst_total ? dst_total : 16));
    hipStream_t fs = c->user_attached ? c->user_stream : c->slot().stream;
    for (size_t i = 0; i < n; ++i) {
        const WavItem &it = items[i];
        if (it.src_len == 0) continue;
        RG_HIP(c, hipMemcpyAsync(c->d_wav.p + it.src_off, it.bytes + it.info.data_offset, it.src_len, hipMemcpyHostToDevice, fs));
        RG_HIP(c, launch_deinterleave(it.kind, c->d_wav.p + it.src_off, c->d_arena.p + (*descs)[i].offset_bytes, it.info.frames,
                                      it.info.channels, fs));
    }
    // every pipeline stream must see the arena: the next enqueue waits for this point (as rg_synth_fill_device)
    if (!c->user_attached) RG_HIP(c, hipEventRecord(c->user_ev, fs));
    c->user_dirty = true;
    *arena_bytes = dst_total;
    return RG_OK;
}

// the same for loaded files: WAV items take the de-interleave route, decoded MP3 items are planar f32 already and go
// straight into the arena
int stage_loaded(rg_ctx *c, const std::vector<LoadedAudio> &in, size_t n, std::vector<rg_track_desc> *descs, size_t *arena_bytes) {
    std::vector<WavItem> items(n);
    // streams the loader pipeline has decoded already sit in [0, keep) of the arena; everything else goes behind them
    size_t keep = 0;
    for (size_t i = 0; i < n; ++i)
        if (in[i].staged) keep = std::max(keep, (size_t)align16(in[i].arena_off + (size_t)in[i].walked_frames * in[i].channels * sizeof(float)));
    size_t src_total = 0, dst_total = keep;
    descs->assign(n ? n : 1, rg_track_desc{});
    for (size_t i = 0; i < n; ++i) {
        rg_track_desc &d = (*descs)[i];
        if (in[i].staged) {
            d.offset_bytes = in[i].arena_off;
            d.frames = in[i].frames;
            d.sample_rate = in[i].sample_rate;
            d.channels = (uint16_t)in[i].channels;
            d.format = RG_FMT_F32_PLANAR;
            continue;
        }
        d.offset_bytes = dst_total;
        if (in[i].decoded || in[i].split) {
            d.frames = in[i].frames;
            d.sample_rate = in[i].sample_rate;
            d.channels = (uint16_t)in[i].channels;
            d.format = RG_FMT_F32_PLANAR;
            dst_total = align16(dst_total + (size_t)in[i].frames * in[i].channels * sizeof(float));
            continue;
        }
        WavItem &it = items[i];
        it.bytes = in[i].wav.data();
        if (rg_wav_parse(in[i].wav.data(), in[i].wav.size(), &it.info) != RG_OK)
            return rg_set_err(c, RG_ERR_FORMAT, "input %zu is not a RIFF/WAVE stream", i);
        it.kind = wav_kind(it.info);
        if (it.kind < 0)
            return rg_set_err(c, RG_ERR_FORMAT, "input %zu: unsupported WAV sample format (tag %u, %u bits)", i,
                              it.info.sample_format, it.info.bits_per_sample);
        it.src_off = src_total;
        it.src_len = it.info.frames * it.info.block_align;
        src_total = align16(src_total + it.src_len);
        d.frames = it.info.frames;
        d.sample_rate = it.info.sample_rate;
        d.channels = it.info.channels;
        d.format = planar_format(it.kind);
        dst_total = align16(dst_total + (size_t)it.info.frames * it.info.channels * rg_bytes_per_sample(d.format));
    }
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    for (int s = 0; s < c->n_slots; ++s) RG_HIP(c, hipStreamSynchronize(c->slots[s].stream));
    RG_HIP(c, c->d_wav.reserve(src_total ? src_total : 16));
    rc = arena_reserve_keep(c, dst_total ? dst_total : 16, keep);
    if (rc != RG_OK) return rc;
    hipStream_t fs = c->user_attached ? c->user_stream : c->slot().stream;
    std::vector<RgMp3SplitItem> split;
    for (size_t i = 0; i < n; ++i) {
        unsigned char *dst = c->d_arena.p + (*descs)[i].offset_bytes;
        if (in[i].staged) continue;
        if (in[i].split) {
            RgMp3SplitItem it{};
            it.is = in[i].is.data();
            it.units = in[i].units.data();
            if (!in[i].recs.empty()) {
                it.recs = in[i].recs.data();
                it.main = in[i].main_stream.data();
                it.main_len = in[i].main_stream.size();
            }
            it.n_units = in[i].n_units;
            it.channels = in[i].channels;
            it.rate_row = (uint32_t)rg_mp3_rate_row(in[i].sample_rate);
            it.lsf = in[i].lsf;
            it.d_ch0 = reinterpret_cast<float *>(dst);
            it.d_ch1 = in[i].channels == 2 ? it.d_ch0 + in[i].frames : nullptr;
            split.push_back(it);
            continue;
        }
        if (in[i].decoded) {
            const size_t bytes = (size_t)in[i].frames * in[i].channels * sizeof(float);
            if (bytes) RG_HIP(c, hipMemcpyAsync(dst, in[i].planar.data(), bytes, hipMemcpyHostToDevice, fs));
            continue;
        }
        const WavItem &it = items[i];
        if (it.src_len == 0) continue;
        RG_HIP(c, hipMemcpyAsync(c->d_wav.p + it.src_off, it.bytes + it.info.data_offset, it.src_len, hipMemcpyHostToDevice, fs));
        RG_HIP(c, launch_deinterleave(it.kind, c->d_wav.p + it.src_off, dst, it.info.frames, it.info.channels, fs));
    }
    if (!split.empty()) {  // the device half of the MP3 decoder writes PCM straight into the arena
        rc = rg_mp3dev_decode(c, split.data(), split.size(), fs);
        if (rc != RG_OK) return rc;
    }
    // the host buffers are the caller's locals: the copies must have left them before this returns
    RG_HIP(c, hipStreamSynchronize(fs));
    if (!c->user_attached) RG_HIP(c, hipEventRecord(c->user_ev, fs));
    c->user_dirty = true;
    *arena_bytes = dst_total;
    return RG_OK;
}

// Host threads this process may really run: the affinity mask, cut by the cgroup CPU quota if there is one (a container
// with 16 CPUs' worth of quota on a 256-core host sees all 256 in its mask; 256 loader threads then only fight).
unsigned usable_cores() {
    unsigned n = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = (unsigned)CPU_COUNT(&set);
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
        char q[64];
        double period = 0.0;
        if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0.0) {
            const double cpus = atof(q) / period;
            if (cpus >= 1.0 && cpus < (double)n) n = (unsigned)(cpus + 0.5);
        }
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
        double quota = -1.0, period = 0.0;
        if (fscanf(g, "%lf", &quota) != 1) quota = -1.0;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(h, "%lf", &period) != 1) period = 0.0;
            fclose(h);
        }
        if (quota > 0.0 && period > 0.0 && quota / period >= 1.0 && quota / period < (double)n) n = (unsigned)(quota / period + 0.5);
    }
    return n < 1 ? 1 : n;
}

bool read_all(FILE *f, std::vector<uint8_t> *out) {
    uint8_t chunk[1 << 16];
    size_t n;
    while ((n = fread(chunk, 1, sizeof chunk, f)) > 0) out->insert(out->end(), chunk, chunk + n);
    return !ferror(f);
}

std::string shell_quote(const char *s) {
    std::string q = "'";
    for (; *s; ++s) {
        if (*s == '\'') q += "'\\''";
        else q += *s;
    }
    return q + "'";
}

// Load one file (no device work; safe to call from several threads at once as long as `err` is per call).
// RIFF/WAVE: the bytes; MPEG Layer III: decoded planar f32; anything else: the decoder command's stdout.
int load_audio_for_impl(const std::string &decoder_cmd, int gpu_decode, const char *path, LoadedAudio *out, std::string *err, int32_t track_index);
// The loaders run on host threads of the library's own: an allocation failure there must come back as a status, not end
// the process in std::terminate.
int load_audio_for(const std::string &decoder_cmd, int gpu_decode, const char *path, LoadedAudio *out, std::string *err, int32_t track_index) {
    try {
        return load_audio_for_impl(decoder_cmd, gpu_decode, path, out, err, track_index);
    } catch (const std::bad_alloc &) {
        *err = std::string("Out of memory while loading: ") + (path ? path : "");
        return RG_ERR_NOMEM;
    } catch (const std::exception &ex) {
        *err = std::string("Failed to load: ") + (path ? path : "") + " (" + ex.what() + ")";
        return RG_ERR_FORMAT;
    }
}
int load_audio_for_impl(const std::string &decoder_cmd, int gpu_decode, const char *path, LoadedAudio *out, std::string *err, int32_t track_index) {
    char msg[1024];
    auto fail = [&](int code, const char *fmt, const char *a, int b = 0) {
        snprintf(msg, sizeof msg, fmt, a, b);
        *err = msg;
        return code;
    };
    if (!path) return fail(RG_ERR_INVALID_ARG, "null path%s", "");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(RG_ERR_IO, "Failed to open: %s", path);  // src/replaygain.rs:804-805
    std::vector<uint8_t> &bytes = out->file_bytes;
    bytes.clear();
    const bool ok = read_all(f, &bytes);
    fclose(f);
    if (!ok) return fail(RG_ERR_IO, "Failed to read: %s", path);
    if (bytes.size() >= 12 && memcmp(bytes.data(), "RIFF", 4) == 0 && memcmp(bytes.data() + 8, "WAVE", 4) == 0) {
        out->wav.swap(bytes);
        return RG_OK;
    }
    const bool mp4 = bytes.size() >= 8 && memcmp(bytes.data() + 4, "ftyp", 4) == 0;
    out->is_mp4 = rg_mp4_is_mp4_data(bytes.data(), bytes.size()) != 0;  // detect_file_type, src/replaygain.rs:777-783
    int mp4_track = 0;
    bool mp4_mpeg_audio = false;  // the selected track of an MP4 file is MPEG audio: `bytes` now holds its elementary stream
    if (mp4) {
        // ---- ISO base media: the audio tracks the reference would see (src/replaygain.rs:827-836), the one it would pick
        // (:838-851), its rate (:854-857).  MPEG Layer III in MP4 is decoded here, from the sample table; AAC needs the
        // decoder command.
        rg_mp4_audio_track tr[32];
        size_t n_audio = 0;
        bool walked = rg_mp4_audio_tracks(bytes.data(), bytes.size(), tr, 32, &n_audio) == RG_DEMUX_OK;
        // A file this walker cannot read (no moov box), or in which it finds no track of a codec the reference's build
        // decodes, is still the decoder command's to try when there is one: the command is the user's decoder (ALAC in
        // M4A, say), and the authority on what it can read.
        if (!walked && decoder_cmd.empty()) return fail(RG_ERR_FORMAT, "Failed to probe format: %s", path);
        if (walked && n_audio == 0) {
            if (decoder_cmd.empty()) return fail(RG_ERR_FORMAT, "No audio track found%s", "");
            walked = false;
        }
        if (walked) {
            out->n_audio_tracks = (uint32_t)n_audio;
            if (track_index >= 0 && (size_t)track_index >= n_audio) {
                snprintf(msg, sizeof msg, "Track index %d out of range (file has %zu audio track(s))", track_index, n_audio);
                *err = msg;
                return RG_ERR_INVALID_ARG;
            }
            mp4_track = track_index < 0 ? 0 : track_index;
            if (mp4_track >= 32) return fail(RG_ERR_INVALID_ARG, "Track index %d: more audio tracks than this library lists%s", "", mp4_track);
            const rg_mp4_audio_track &t = tr[mp4_track];
            if (t.sample_rate == 0) return fail(RG_ERR_FORMAT, "Unknown sample rate%s", "");
            if (t.codec == RG_CODEC_MP3) {
                // the track's samples are MPEG audio frames: laid end to end they are the stream the library's decoder takes
                size_t n_au = 0;
                if (rg_mp4_access_units(bytes.data(), bytes.size(), (size_t)mp4_track, nullptr, nullptr, 0, &n_au) != RG_DEMUX_OK)
                    return fail(RG_ERR_FORMAT, "Failed to probe format: %s", path);
                std::vector<uint64_t> off(n_au);
                std::vector<uint32_t> sz(n_au);
                size_t got = 0;
                (void)rg_mp4_access_units(bytes.data(), bytes.size(), (size_t)mp4_track, off.data(), sz.data(), n_au, &got);
                std::vector<uint8_t> es;
                for (size_t i = 0; i < got && i < n_au; ++i) es.insert(es.end(), bytes.begin() + (ptrdiff_t)off[i], bytes.begin() + (ptrdiff_t)(off[i] + sz[i]));
                bytes.swap(es);
                mp4_mpeg_audio = true;
            }
        }
    }
    if (!mp4 || mp4_mpeg_audio) {
        // the probe (src/replaygain.rs:815-822) and the packet loop (:881-904) for an MPEG audio stream
        rg_mp3_stream_info si;
        if (rg_mp3_scan(bytes.data(), bytes.size(), &si) == RG_MP3DEC_OK && si.audio_frames > 0) {
            if (gpu_decode == 2) {  // only the frame walk here: side information and where each granule's bits are
                rg_mp3_stream_info di;
                const int rc = rg_mp3_index_stream(bytes.data(), bytes.size(), &out->main_stream, &out->recs, &di);
                if (rc != RG_MP3DEC_OK) return fail(RG_ERR_FORMAT, "Failed to decode: %s", path);
                out->n_units = out->recs.size();
                out->sample_rate = di.sample_rate;
                out->channels = di.channels;
                out->frames = di.frames;
                out->lsf = di.mpeg_version == 1 ? 0u : 1u;
                out->split = true;
                return RG_OK;
            }
            if (gpu_decode) {  // stage A here (frame walk, side info, reservoir, scalefactors, Huffman), the rest on the device
                const uint64_t cap = (uint64_t)si.audio_frames * (si.mpeg_version == 1 ? 2u : 1u) * si.channels;
                out->is.assign((size_t)cap * 576, 0);
                out->units.assign((size_t)cap, rg_mp3_unit{});
                rg_mp3_stream_info di;
                const int rc = rg_mp3_parse_units(bytes.data(), bytes.size(), out->is.data(), out->units.data(), cap, &out->n_units, &di);
                if (rc != RG_MP3DEC_OK) return fail(RG_ERR_FORMAT, "Failed to decode: %s", path);
                out->sample_rate = di.sample_rate;
                out->channels = di.channels;
                out->frames = di.frames;
                out->lsf = di.mpeg_version == 1 ? 0u : 1u;
                out->split = true;
                return RG_OK;
            }
            out->planar.assign((size_t)si.frames * si.channels, 0.0f);
            rg_mp3_stream_info di;
            const int rc = rg_mp3_decode_f32(bytes.data(), bytes.size(), out->planar.data(),
                                             si.channels == 2 ? out->planar.data() + si.frames : nullptr, si.frames, &di);
            if (rc != RG_MP3DEC_OK) return fail(RG_ERR_FORMAT, "Failed to decode: %s", path);
            if (si.channels == 2 && di.frames != si.frames)  // dropped frames shortened the track: close the gap between the planes
                memmove(out->planar.data() + di.frames, out->planar.data() + si.frames, sizeof(float) * (size_t)di.frames);
            out->sample_rate = di.sample_rate;
            out->channels = di.channels;
            out->frames = di.frames;
            out->decoded = true;
            return RG_OK;
        }
    }
    if (mp4_mpeg_audio)  // an MPEG audio track whose samples are not Layer III frames this decoder takes (Layer I / II, say)
        return fail(RG_ERR_FORMAT, "Failed to create decoder: %s (the selected track's MPEG audio is not Layer III)", path);
    if (decoder_cmd.empty())  // src/replaygain.rs:861-863 (AAC: the probe succeeded, the codec is missing) / :815-822 (the probe knows no such format)
        return fail(RG_ERR_FORMAT, mp4 ? "Failed to create decoder: %s (an AAC track; no AAC decoder is built into this library: set a decoder command, rg_set_decoder_command)"
                                       : "Failed to probe format: %s (neither MPEG Layer III nor RIFF/WAVE, and no decoder command is set: rg_set_decoder_command)",
                    path);
    std::string cmd = decoder_cmd;
    {   // "{track}" = index of the selected audio track (ffmpeg: -map 0:a:{track})
        const std::string tn = std::to_string(mp4_track);
        for (size_t at = cmd.find("{track}"); at != std::string::npos; at = cmd.find("{track}", at + tn.size())) cmd.replace(at, 7, tn);
    }
    const std::string q = shell_quote(path);
    size_t at = cmd.find("{}");
    if (at == std::string::npos) cmd += " " + q;
    else
        for (; at != std::string::npos; at = cmd.find("{}", at + q.size())) cmd.replace(at, 2, q);
    FILE *p = popen(cmd.c_str(), "r");
    if (!p) return fail(RG_ERR_IO, "Failed to run decoder: %s", strerror(errno));
    out->wav.clear();
    const bool rd = read_all(p, &out->wav);
    const int status = pclose(p);
    if (!rd || status != 0 || out->wav.empty())
        return fail(RG_ERR_FORMAT, "Failed to probe format: %s (decoder command exited with status %d)", path, status);
    return RG_OK;
}

// =================================================================================================
// The loader pipeline of tuning key 6 = 3 (the default).
//
// Host threads do the least an MPEG stream allows: read the file, walk its frame headers, and strip headers and side
// information from the main data (rg_mp3_compact_stream).  Each stream's main data and slots go into a pinned staging
// block; a block that is full (or holds enough granules to fill the GPU) is a chunk, and the calling thread sends chunks
// to the device as they close: one H2D copy on the copy stream, then the frame parser, Huffman and back-half
// kernels on the file stream, writing PCM straight into the analysis arena.  Three staging blocks and two device copies
// rotate, so reading files, copying chunk k + 1 and decoding chunk k overlap.  How many frames of a stream decode is the
// device's finding (rg_mp3_frames_kernel); the arena is laid out for "all of them" and the counts come back at the end.
struct Mp3Stage {
    uint8_t *p = nullptr;
    size_t cap = 0;
    hipEvent_t staged = nullptr;  // H2D of the block's last chunk
};
struct Mp3Scratch {
    uint8_t *p = nullptr;
    size_t cap = 0;
    std::vector<uint8_t> slots;
    std::vector<uint64_t> tiles;
};
struct Mp3Pipe {
    static constexpr int NSTAGE = 3;
    Mp3Stage stage[NSTAGE];
    std::vector<Mp3Scratch> scratch;  // one per loader thread
    std::vector<hipEvent_t> part_ev;  // album parts: per chunk [2k] its decode is done, [2k + 1] its frame counts are on the host
    ~Mp3Pipe() {
        for (Mp3Stage &st : stage) {
            if (st.p) (void)hipHostFree(st.p);
            if (st.staged) (void)hipEventDestroy(st.staged);
        }
        for (Mp3Scratch &sc : scratch) free(sc.p);
        for (hipEvent_t e : part_ev) (void)hipEventDestroy(e);
    }
};
// Album parts.  The decode of an album's files is a pipeline of chunks (below); with `PartsRun` the tracks of chunk k are
// analysed -- one enqueue on a pipeline stream that waits for the chunk's decode, album mode -- while chunk k + 1 is copied and
// decoded, instead of all together at the end: where the H2D copy is the longest stage (anything from 128 kb/s up) the analysis
// disappears behind it, and elsewhere it fills the decode kernels' tails.  Each part leaves its [histogram | peak] pack in
// c->d_album_packs and its per-track results in c->h_part_results; u32 adds commute, so the album is the fold of the packs
// (the streamed host ingest and albums larger than the device do the same).  Anything out of the ordinary -- a file that is not
// an MPEG stream or failed, an unsupported rate, a track the fast kernels flag -- drops the parts and the album is analysed the
// plain way from the PCM, which is in the arena either way.
constexpr size_t kMaxParts = 64;
struct PartsRun {
    int album = 1;  // 0: track mode (rg_analyze_tracks) -- the same parts without the packs
    bool broken = false;
    size_t n_parts = 0;
    std::vector<size_t> file_of;  // position in c->h_part_results -> file of the call
    std::vector<size_t> pending;  // files of decoded chunks not yet in a part (chunks the device, not the copy, was the longer stage of)
};
// A chunk whose H2D copy takes longer than its decode leaves the device idle: its tracks (and what is pending) become a part
// right away.  Where the decode is the longer stage a part only splits the analysis into smaller, less efficient launches
// (measured: 256 VBR files 20.6 -> 21.6 ms, 256 files of 320 kb/s 50.7 -> 38.4 ms), so such chunks wait -- for a later chunk
// that is copy-bound, or for the end of the album, where an album without a single part goes the plain way.
// Copy: ~50 GB/s; decode: ~0.5 ms per 256 K units = 1.9 ns per unit = 95 bytes' worth of copy.  At 104 bytes per unit (128 kb/s
// stereo) the two routes measure the same within their noise (album 24.9 -> 23.4 ms, track mode 21.6 -> 22.5), so the line is
// drawn at 120: 160 kb/s and up.
// (rg_ctx::parts_min_bpu(): RG_PARTS_MIN_BYTES_PER_UNIT as read at rg_create, or tuning key 11)
Mp3Pipe &mp3_pipe(rg_ctx *c) {
    if (!c->mp3_pipe) {
        c->mp3_pipe = new Mp3Pipe();
        c->mp3_pipe_free = [](void *p) { delete static_cast<Mp3Pipe *>(p); };
    }
    return *static_cast<Mp3Pipe *>(c->mp3_pipe);
}

// Granule-channels per chunk.  The Huffman kernel deals a chunk's units to its lanes heaviest first (rg_mp3_sort_*): a chunk
// has to be several generations of its blocks (256 CUs x 2 blocks x 512 threads = 262144 resident) for the light tail to fill
// in behind the heavy head, and the six small launches in front of it (frame parser, sort) are paid per chunk: per 256 K units
// the chain takes 0.68 / 0.63 / 0.56 / 0.56 ms in chunks of 256 K / 384 K / 768 K / 1 M on the dense 320 kb/s stream.
constexpr uint64_t kPipeChunkUnits = 786432;
constexpr size_t kPipeStageBytes = (size_t)128 << 20;  // staging block (a 3-minute 320 kb/s file is 7.2 MB and 27 600 granule-channels)

struct PipeChunk {
    int stage = 0;
    size_t used = 0;
    uint64_t units = 0;
    std::vector<size_t> files;
    int pending = 0;  // files still being copied into the block
    bool closed = false, issued = false;
};
struct PipeFile {
    uint64_t main_off = 0, main_len = 0, slots_off = 0, tiles_off = 0;
    uint32_t n_frames = 0;
};
struct PipeRun {
    std::mutex m;
    std::condition_variable cv;
    std::deque<PipeChunk> chunks;
    int open = -1;
    bool preparing = false;  // a loader is getting the next chunk's staging block ready, outside the lock
    size_t files_done = 0;
    size_t stage_want = 0;
    int hip_error = RG_OK;
    std::string hip_msg;
    bool starved = false;       // the device was found idle when the latest chunk became ready: the host's loaders are the longer stage
    bool tapering = false;      // the call's last chunks are being made smaller
    size_t files_placed = 0;    // files that have their place in a chunk
    uint64_t units_placed = 0;
};

bool read_whole_file(const char *path, Mp3Scratch *sc, size_t *len) {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    size_t want = (fstat(fd, &st) == 0 && st.st_size > 0) ? (size_t)st.st_size : 0;
    size_t got = 0;
    for (;;) {
        if (sc->cap < got + 65536 + 64 || sc->cap < want + 64) {
            size_t cap = std::max(std::max(sc->cap * 2, want + 64 + 65536), (size_t)1 << 20);
            uint8_t *q = static_cast<uint8_t *>(realloc(sc->p, cap));
            if (!q) { close(fd); return false; }
            sc->p = q;
            sc->cap = cap;
        }
        const ssize_t k = read(fd, sc->p + got, sc->cap - 64 - got);
        if (k < 0) {
            if (errno == EINTR) continue;
            close(fd);
            return false;
        }
        if (k == 0) break;
        got += (size_t)k;
    }
    close(fd);
    memset(sc->p + got, 0, 64);
    *len = got;
    return true;
}

// Loads `paths` into `out` (entry i <- file i) with the pipeline: MPEG streams are on their way through the device when
// this returns and their PCM sits in c->d_arena (LoadedAudio::staged), other inputs are loaded as load_audio_for loads
// them.  rcs / errs: per-file outcome.
// What one file of a list comes to before any analysis, in the order the reference meets its errors
// (src/replaygain.rs:804-873): open / read, track selection, probe, sample rate.  RG_OK, or the code with `msg` set.
int file_outcome(const LoadedAudio &la, int load_rc, const std::string &load_err, const char *path, int32_t track_index,
                        std::string *msg) {
    if (load_rc != RG_OK) {
        *msg = load_err;
        return load_rc;
    }
    if (track_index >= 0 && (uint32_t)track_index >= la.n_audio_tracks) {
        char m[128];
        snprintf(m, sizeof m, "Track index %d out of range (file has %u audio track(s))", track_index, la.n_audio_tracks);
        *msg = m;
        return RG_ERR_INVALID_ARG;
    }
    uint32_t rate = la.sample_rate;
    if (!la.decoded && !la.split && !la.staged) {
        rg_wav_info wi;
        rate = rg_wav_parse(la.wav.data(), la.wav.size(), &wi) == RG_OK ? wi.sample_rate : 0;
        if (rate == 0) {
            *msg = std::string("Failed to probe format: ") + path;
            return RG_ERR_FORMAT;
        }
    }
    if (!rg_supported_rate(rate)) {
        char m[256];
        snprintf(m, sizeof m, "Unsupported sample rate: %u Hz. Supported rates: 96000, 88200, 64000, 48000, 44100, 32000, 24000, "
                              "22050, 16000, 12000, 11025, 8000", rate);
        *msg = m;
        return RG_ERR_UNSUPPORTED_RATE;
    }
    return RG_OK;
}

int load_many_pipelined(rg_ctx *c, const char *const *paths, size_t n, std::vector<LoadedAudio> *out, std::vector<int> *rcs,
                        std::vector<std::string> *errs, PartsRun *parts) {
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    Mp3Pipe &P = mp3_pipe(c);
    unsigned workers = c->loader_threads ? c->loader_threads : usable_cores();
    if (workers > n) workers = (unsigned)n;
    if (P.scratch.size() < workers) P.scratch.resize(workers);
    for (Mp3Stage &st : P.stage)
        if (!st.staged) RG_HIP(c, hipEventCreateWithFlags(&st.staged, hipEventDisableTiming));
    // earlier batches may still read the arena and the chunk buffers
    for (int s = 0; s < c->n_slots; ++s) RG_HIP(c, hipStreamSynchronize(c->slots[s].stream));
    // The decode runs on the FIRST pipeline stream, whichever slot the last batch used, and the copies on the second
    // (rg_mp3dev_enqueue_chunk): the runtime maps streams onto four hardware queues, and with a fifth stream for the copies and
    // the decode on "the current slot's stream" every fourth call landed on the queue the copy stream shared and took 25 instead
    // of 20 ms (tools/seq_album.py).
    hipStream_t fs = c->user_attached ? c->user_stream : c->slots[0].stream;
    if (parts) {  // decode and parts share that stream: see PartsRun
        RG_HIP(c, hipStreamSynchronize(fs));
        if (P.part_ev.size() < 2 * kMaxParts) {
            const size_t have = P.part_ev.size();
            P.part_ev.resize(2 * kMaxParts, nullptr);
            for (size_t k = have; k < P.part_ev.size(); ++k) RG_HIP(c, hipEventCreateWithFlags(&P.part_ev[k], hipEventDisableTiming));
        }
        RG_HIP(c, c->h_mp3_part_counts.reserve(n * kMaxParts));
        RG_HIP(c, c->h_part_results.reserve(n));
        if (parts->album) RG_HIP(c, c->d_album_packs.reserve(kMaxParts * (size_t)RG_ALBUM_PACK_WORDS));
    }
    rc = rg_mp3dev_reserve_results(c, n, fs);
    if (rc != RG_OK) return rc;

    const bool trace = c->trace_files;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
    PipeRun R;
    {   // staging blocks no larger than the call needs: a single album of a dozen files should not pin 3 x 96 MB
        size_t total = 0;
        for (size_t i = 0; i < n; ++i) {
            struct stat st;
            if (paths[i] && stat(paths[i], &st) == 0 && st.st_size > 0) total += (size_t)st.st_size;
        }
        size_t cap = kPipeStageBytes;
        if (c->stage_bytes()) cap = c->stage_bytes();  // tests: tiny blocks, so that a handful of small files exercises the whole rotation
        R.stage_want = std::min(cap, total + total / 8 + ((size_t)1 << 16));
    }
    std::vector<PipeFile> pf(n);
    std::atomic<uint64_t> t_read{0}, t_compact{0}, t_wait{0}, t_copy{0};  // trace: microseconds summed over the loader threads
    const std::string cmd = c->decoder_cmd;
    const int32_t track_index = c->file_track_index;
    const int device = c->device;
    std::atomic<size_t> next_file{0};

    auto grow_stage = [&](Mp3Stage &st, size_t need) -> bool {  // the block is idle
        if (st.cap >= need) return true;
        if (st.p) (void)hipHostFree(st.p);
        st.p = nullptr;
        st.cap = 0;
        const size_t want = need + need / 8;
        if (hipHostMalloc((void **)&st.p, want, hipHostMallocDefault) != hipSuccess) return false;
        st.cap = want;
        return true;
    };
    auto hip_fail = [&](const char *what) {  // R.m held
        if (R.hip_error == RG_OK) { R.hip_error = RG_ERR_DEVICE; R.hip_msg = what; }
    };

    auto load_file = [&](size_t i, Mp3Scratch &sc) {
        LoadedAudio &la = (*out)[i];
        std::string &err = (*errs)[i];
        const char *path = paths[i];
        char msg[1024];
        if (!path) { (*rcs)[i] = RG_ERR_INVALID_ARG; err = "null path"; return; }
        size_t len = 0;
        const double tl0 = trace ? now() : 0.0;
        if (!read_whole_file(path, &sc, &len)) {
            snprintf(msg, sizeof msg, "Failed to open: %s", path);  // src/replaygain.rs:804-805
            (*rcs)[i] = RG_ERR_IO;
            err = msg;
            return;
        }
        if (len >= 12 && memcmp(sc.p, "RIFF", 4) == 0 && memcmp(sc.p + 8, "WAVE", 4) == 0) {
            la.wav.assign(sc.p, sc.p + len);
            return;
        }
        const bool mp4 = len >= 8 && memcmp(sc.p + 4, "ftyp", 4) == 0;
        la.is_mp4 = rg_mp4_is_mp4_data(sc.p, len) != 0;
        rg_mp3_stream_info si;
        uint64_t main_len = 0;
        const double tl1 = trace ? now() : 0.0;
        if (mp4 || rg_mp3_compact_stream(sc.p, len, &sc.slots, &sc.tiles, &main_len, &si) != RG_MP3DEC_OK || si.audio_frames == 0) {
            (*rcs)[i] = load_audio_for(cmd, 2, path, &la, &err, track_index);  // the decoder command, or the reference's probe error
            return;
        }
        la.sample_rate = si.sample_rate;
        la.channels = si.channels;
        la.lsf = si.mpeg_version == 1 ? 0u : 1u;
        la.walked_frames = si.frames;
        la.frames = si.frames;
        la.result_index = (uint32_t)i;
        la.staged = true;
        const uint64_t units = (uint64_t)si.audio_frames * (la.lsf ? 1u : 2u) * si.channels;
        const size_t slot_bytes = sc.slots.size(), tile_bytes = sc.tiles.size() * sizeof(uint64_t);
        const size_t need = align64((size_t)main_len + 8) + align64(slot_bytes) + align64(tile_bytes);
        PipeFile &f = pf[i];
        f.main_len = main_len;
        f.n_frames = si.audio_frames;
        uint8_t *dst = nullptr;
        PipeChunk *chunk = nullptr;
        const double tl2 = trace ? now() : 0.0;
        {
            std::unique_lock<std::mutex> lk(R.m);
            for (;;) {
                if (R.open >= 0) {
                    PipeChunk &ch = R.chunks[(size_t)R.open];
                    Mp3Stage &st = P.stage[ch.stage];
                    const size_t with = ch.used + need + rg_mp3dev_track_bytes(ch.files.size() + 1) + 64;
                    // (the call's first chunks are smaller: the device has nothing to do until the first one is complete)
                    uint64_t unit_cap = R.open < 3 ? kPipeChunkUnits >> (3 - R.open) : kPipeChunkUnits;  // 1/8, 1/4, 1/2, then whole chunks
                    // ... and where the device waits for the loaders, the call's last chunks the other way round (each at most
                    // half of what is left, by the files so far): what follows the last file is one chunk's copy, decode and
                    // analysis, 3.5 ms of a 26 ms call for a whole chunk (two loader threads, 256 VBR files)
                    if (R.starved && R.files_placed) {
                        const uint64_t left = ch.units + (uint64_t)((double)(n - R.files_placed) * ((double)R.units_placed / (double)R.files_placed));
                        const uint64_t taper = std::max(left / 2, kPipeChunkUnits >> 3);
                        if (taper < unit_cap) {
                            unit_cap = taper;
                            R.tapering = true;  // (small chunks follow each other quickly: the device being busy then says nothing)
                        }
                    }
                    if (with <= st.cap && ch.units + units <= unit_cap) break;
                    if (ch.files.empty()) {  // a stream larger than a block: the block grows (nothing is in flight from it)
                        if (!grow_stage(st, with)) { hip_fail("hipHostMalloc of a staging block failed"); (*rcs)[i] = RG_ERR_DEVICE; err = "out of pinned memory"; la.staged = false; return; }
                        break;
                    }
                    ch.closed = true;
                    R.open = -1;
                    R.cv.notify_all();
                }
                const size_t id = R.chunks.size();
                // every block is filling or waiting to be sent, or another loader is already preparing the next one
                if (R.preparing || (id >= (size_t)Mp3Pipe::NSTAGE && !R.chunks[id - Mp3Pipe::NSTAGE].issued)) {
                    R.cv.wait(lk);
                    continue;
                }
                // Waiting for the block's last H2D copy and pinning memory (up to 128 MB) happen WITHOUT the lock: every
                // loader and the drive thread take it for each file and each chunk, and one loader sitting on it stalled
                // file reads, copy completion and chunk issue for all the others.
                Mp3Stage &st = P.stage[id % Mp3Pipe::NSTAGE];
                R.preparing = true;
                lk.unlock();
                bool ev_ok = true, mem_ok = true;
                if (id >= (size_t)Mp3Pipe::NSTAGE) {
                    (void)hipSetDevice(device);
                    ev_ok = hipEventSynchronize(st.staged) == hipSuccess;
                }
                mem_ok = grow_stage(st, std::max(R.stage_want, (size_t)4096));
                lk.lock();
                R.preparing = false;
                if (!ev_ok) hip_fail("waiting for a staging block failed");
                if (!mem_ok) {
                    hip_fail("hipHostMalloc of a staging block failed");
                    (*rcs)[i] = RG_ERR_DEVICE;
                    err = "out of pinned memory";
                    la.staged = false;
                    R.cv.notify_all();
                    return;
                }
                R.chunks.emplace_back();
                R.chunks.back().stage = (int)(id % Mp3Pipe::NSTAGE);
                R.open = (int)id;
                R.cv.notify_all();
            }
            chunk = &R.chunks[(size_t)R.open];
            f.main_off = chunk->used;
            f.slots_off = f.main_off + align64((size_t)main_len + 8);
            f.tiles_off = f.slots_off + align64(slot_bytes);
            chunk->used = f.tiles_off + align64(tile_bytes);
            chunk->units += units;
            R.files_placed++;
            R.units_placed += units;
            chunk->files.push_back(i);
            chunk->pending++;
            dst = P.stage[chunk->stage].p;
        }
        const double tl3 = trace ? now() : 0.0;
        // (one large copy per stream: it leaves the cache-resident scratch buffer with streaming stores.  Gathering the frames'
        // main data straight into the block instead -- a frame list first, no compaction in place -- writes the block in
        // pieces of a few hundred bytes, each a read-for-ownership of lines nobody will read here, and was slower from
        // 128 kb/s up: commit a661d6a, profiles/r06_host_loader.txt)
        memcpy(dst + f.main_off, sc.p, (size_t)main_len);
        memset(dst + f.main_off + main_len, 0, (size_t)(f.slots_off - f.main_off - main_len));  // the bit reader looks a few bytes ahead
        memcpy(dst + f.slots_off, sc.slots.data(), slot_bytes);
        memcpy(dst + f.tiles_off, sc.tiles.data(), tile_bytes);
        if (trace) {
            const double tl4 = now();
            t_read += (uint64_t)((tl1 - tl0) * 1e6);
            t_compact += (uint64_t)((tl2 - tl1) * 1e6);
            t_wait += (uint64_t)((tl3 - tl2) * 1e6);
            t_copy += (uint64_t)((tl4 - tl3) * 1e6);
        }
        {
            std::lock_guard<std::mutex> lk(R.m);
            chunk->pending--;
        }
        R.cv.notify_all();
    };
    auto work = [&](unsigned w) {
        (void)hipSetDevice(device);  // the staging blocks a loader allocates or waits for belong to this context's GPU
        for (size_t i = next_file.fetch_add(1); i < n; i = next_file.fetch_add(1)) {
            load_file(i, P.scratch[w]);
            {
                std::lock_guard<std::mutex> lk(R.m);
                R.files_done++;
            }
            R.cv.notify_all();
        }
    };

    // ---- the calling thread: send chunks as they close ---------------------------------------------------------------
    size_t arena_used = 0;
    auto issue = [&](PipeChunk &ch, size_t index) -> int {
        std::vector<RgMp3StreamItem> items(ch.files.size());
        size_t top = arena_used;
        for (size_t k = 0; k < ch.files.size(); ++k) {
            LoadedAudio &la = (*out)[ch.files[k]];
            la.arena_off = top;
            top = align16(top + (size_t)la.walked_frames * la.channels * sizeof(float));
        }
        int r = arena_reserve_keep(c, top ? top : 16, arena_used);
        if (r != RG_OK) return r;
        arena_used = top;
        for (size_t k = 0; k < ch.files.size(); ++k) {
            const size_t i = ch.files[k];
            const LoadedAudio &la = (*out)[i];
            RgMp3StreamItem &it = items[k];
            it.main_off = pf[i].main_off;
            it.slots_off = pf[i].slots_off;
            it.tiles_off = pf[i].tiles_off;
            it.n_frames = pf[i].n_frames;
            it.channels = la.channels;
            it.rate_row = (uint32_t)rg_mp3_rate_row(la.sample_rate);
            it.lsf = la.lsf;
            it.result_index = la.result_index;
            it.d_ch0 = reinterpret_cast<float *>(c->d_arena.p + la.arena_off);
        }
        Mp3Stage &st = P.stage[ch.stage];
        const size_t tracks_off = (ch.used + 7) & ~(size_t)7;
        if (parts && index >= kMaxParts) parts->broken = true;
        const bool part = parts && !parts->broken;
        r = rg_mp3dev_enqueue_chunk(c, (int)(index & 1), st.p, tracks_off + rg_mp3dev_track_bytes(items.size()), tracks_off, st.staged,
                                    items.data(), items.size(), fs, part ? c->h_mp3_part_counts.p + index * n : nullptr, n,
                                    part ? P.part_ev[2 * index + 1] : nullptr);
        if (r == RG_OK && part) RG_HIP(c, hipEventRecord(P.part_ev[2 * index], fs));
        return r;
    };
    // the tracks of chunk `index` (decode enqueued, the chunk after it too) as one part of the album
    const double copy_bound_at = parts ? c->parts_min_bpu() : 0.0;
    // `ch` (may be null: nothing new) joins what is pending; `index`: the newest chunk whose files are pending or were
    // `starved`: when the chunk after `ch` was ready to be sent, the device had already finished `ch`'s decode, i.e. it is the
    // host's loaders the call is waiting for (few of them: one loader thread reads and walks 6 GB/s of VBR files, the device
    // takes 17): the analysis of the files so far costs nothing while it waits.
    auto analyze_part = [&](const PipeChunk *ch, size_t index, bool last, bool starved) -> int {
        if (!parts || parts->broken) return RG_OK;
        bool copy_bound = false;
        if (ch) {
            parts->pending.insert(parts->pending.end(), ch->files.begin(), ch->files.end());
            copy_bound = (ch->units && (double)ch->used / (double)ch->units >= copy_bound_at) || starved;
        }
        if (!copy_bound && !(last && parts->n_parts)) {
            if (last) parts->broken = true;  // no chunk of the album was copy-bound: the plain route, one launch over all of it
            return RG_OK;
        }
        if (parts->pending.empty()) return RG_OK;
        std::vector<size_t> files;
        files.swap(parts->pending);
        RG_HIP(c, hipEventSynchronize(P.part_ev[2 * index + 1]));  // the frame parser ran at the head of the chunk's work: long done
        const uint32_t *counts = c->h_mp3_part_counts.p + index * n;  // (the counts of every earlier chunk are in this copy as well)
        std::vector<rg_track_desc> descs(files.size());
        for (size_t k = 0; k < files.size(); ++k) {
            const size_t i = files[k];
            LoadedAudio &la = (*out)[i];
            std::string msg;
            if (!la.staged || file_outcome(la, (*rcs)[i], (*errs)[i], paths[i], track_index, &msg) != RG_OK) {
                parts->broken = true;  // the plain route reports it, in input order
                return RG_OK;
            }
            rg_track_desc &d = descs[k];
            d = rg_track_desc{};
            d.offset_bytes = la.arena_off;
            d.frames = (uint64_t)counts[la.result_index] * 576;
            d.sample_rate = la.sample_rate;
            d.channels = (uint16_t)la.channels;
            d.format = RG_FMT_F32_PLANAR;
        }
        // on the decode's stream (behind the decode of the chunk after this part's), with the buffers of the next pipeline slot:
        // one batch in flight at a time (cost model: one_shot); with three or more slots (the callers' condition for parts) the
        // slot taken here had its last descriptor copy two parts ago, so its pinned descriptors are not waited for
        c->enqueue_wait_ev = P.part_ev[2 * index];
        c->enqueue_stream = fs;
        const bool one_shot_before = c->one_shot;
        c->one_shot = true;
        const int r = rg_enqueue_impl(c, descs.data(), descs.size(), c->d_arena.p, arena_used, parts->album);
        c->one_shot = one_shot_before;
        c->enqueue_stream = nullptr;
        c->enqueue_wait_ev = nullptr;
        if (r != RG_OK) {
            parts->broken = true;
            return RG_OK;
        }
        RgSlot &S = c->slot();
        if (parts->album)
            RG_HIP(c, hipMemcpyAsync(c->d_album_packs.p + parts->n_parts * (size_t)RG_ALBUM_PACK_WORDS, S.d_album_hist.p,
                                     (size_t)RG_ALBUM_PACK_WORDS * sizeof(uint32_t), hipMemcpyDeviceToDevice, fs));
        RG_HIP(c, hipMemcpyAsync(c->h_part_results.p + parts->file_of.size(), S.d_results.p, descs.size() * sizeof(rg_track_result),
                                 hipMemcpyDeviceToHost, fs));
        parts->n_parts++;
        parts->file_of.insert(parts->file_of.end(), files.begin(), files.end());
        return RG_OK;
    };
    auto drive = [&]() -> int {
        int result = RG_OK;
        size_t next = 0;
        const PipeChunk *prev = nullptr;  // issued, not yet analysed as a part
        size_t prev_index = 0;
        std::unique_lock<std::mutex> lk(R.m);
        for (;;) {
            R.cv.wait(lk, [&] { return (next < R.chunks.size() && R.chunks[next].closed && R.chunks[next].pending == 0) || R.files_done == n; });
            if (!(next < R.chunks.size() && R.chunks[next].closed && R.chunks[next].pending == 0)) {
                if (R.open >= 0) {  // every file is in: the last chunk closes as it is
                    R.chunks[(size_t)R.open].closed = true;
                    R.open = -1;
                    continue;
                }
                if (next >= R.chunks.size()) {
                    if (result == RG_OK && parts) {
                        lk.unlock();
                        const int r = analyze_part(prev, prev_index, true, false);
                        lk.lock();
                        if (r != RG_OK) result = r;
                        prev = nullptr;
                    }
                    break;
                }
                continue;
            }
            PipeChunk &ch = R.chunks[next];
            const size_t done_now = R.files_done;
            lk.unlock();
            const double t_i = now();
            const bool observed = parts && !parts->broken && prev && c->parts_when_starved();
            const bool starved = observed && hipEventQuery(P.part_ev[2 * prev_index]) == hipSuccess;
            int r = (result == RG_OK && !ch.files.empty()) ? issue(ch, next) : RG_OK;
            if (r == RG_OK && result == RG_OK && prev) {  // the device has this chunk's decode to go on with
                r = analyze_part(prev, prev_index, false, starved);
                prev = nullptr;
            }
            if (r == RG_OK && result == RG_OK && !ch.files.empty()) {
                prev = &ch;
                prev_index = next;
            }
            if (trace)
                fprintf(stderr, "[pipeline] chunk %zu: %zu files, %.1f MB, %llu units, ready at %.1f ms (files done %zu)%s, enqueue took %.2f ms\n", next,
                        ch.files.size(), ch.used / 1e6, (unsigned long long)ch.units, (t_i - t_start) * 1e3, done_now, starved ? ", the device was idle" : "",
                        (now() - t_i) * 1e3);
            lk.lock();
            if (observed && !R.tapering) R.starved = starved;  // the latest finding counts
            if (r != RG_OK && result == RG_OK) result = r;
            if (r != RG_OK || ch.files.empty()) (void)hipEventRecord(P.stage[ch.stage].staged, fs);  // loaders wait on it before refilling the block
            ch.issued = true;
            ++next;
            R.cv.notify_all();
        }
        return result;
    };

    if (n == 1) {  // one stream is one chunk: nothing to overlap
        work(0);
        rc = drive();
    } else {
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < workers; ++w) pool.emplace_back(work, w);
        rc = drive();
        for (auto &t : pool) t.join();
    }
    if (rc != RG_OK) return rc;
    if (R.hip_error != RG_OK) return rg_set_err(c, R.hip_error, "%s", R.hip_msg.c_str());
    // the device's findings: how much of each stream decoded
    const double t_issued = now();
    rc = rg_mp3dev_fetch_results(c, n, fs);
    if (rc != RG_OK) return rc;
    RG_HIP(c, hipStreamSynchronize(fs));
    if (trace)
        fprintf(stderr, "[pipeline] all chunks enqueued at %.1f ms, device done at %.1f ms; %u loader threads, summed: read %.1f ms, compact %.1f ms, "
                        "waiting for a block %.1f ms, copy into the block %.1f ms\n", (t_issued - t_start) * 1e3, (now() - t_start) * 1e3, workers,
                t_read.load() / 1e3, t_compact.load() / 1e3, t_wait.load() / 1e3, t_copy.load() / 1e3);
    const uint32_t *granules = rg_mp3dev_results(c);
    for (size_t i = 0; i < n; ++i) {
        LoadedAudio &la = (*out)[i];
        if (la.staged) la.frames = (uint64_t)granules[la.result_index] * 576;
    }
    if (!c->user_attached) RG_HIP(c, hipEventRecord(c->user_ev, fs));
    c->user_dirty = true;
    return RG_OK;
}

int load_many(rg_ctx *c, const char *const *paths, size_t n, std::vector<LoadedAudio> *out, std::vector<int> *rcs_out = nullptr,
              std::vector<std::string> *errs_out = nullptr, PartsRun *parts = nullptr);

// `out` is entry 0 of the context's pool
int load_one(rg_ctx *c, const char *path, std::vector<LoadedAudio> *pool) { return load_many(c, &path, 1, pool); }

// The files of an album, decoded on the host's cores (decode is by far the longest stage of a real run: one core turns
// about 200 s of stereo audio into PCM per second, the GPU analyses 8 million).  Errors keep the reference's order: the
// first failing file in input order is the one reported (src/replaygain.rs:1055).
int load_many(rg_ctx *c, const char *const *paths, size_t n, std::vector<LoadedAudio> *out, std::vector<int> *rcs_out,
              std::vector<std::string> *errs_out, PartsRun *parts) {
    std::vector<int> rcs(n, RG_OK);
    std::vector<std::string> errs(n);
    if (c->gpu_mp3_decode >= 3 && n) {
        const int prc = load_many_pipelined(c, paths, n, out, &rcs, &errs, parts);
        if (prc != RG_OK) return prc;
    } else {
        unsigned workers = c->loader_threads ? c->loader_threads : usable_cores();
        if (workers > n) workers = (unsigned)n;
        std::atomic<size_t> next{0};
        const std::string cmd = c->decoder_cmd;
        const int32_t track_index = c->file_track_index;
        const int gpu_decode = c->gpu_mp3_decode;
        auto work = [&]() {
            for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) rcs[i] = load_audio_for(cmd, gpu_decode, paths[i], &(*out)[i], &errs[i], track_index);  // (*out) holds >= n entries
        };
        if (workers <= 1) {
            work();
        } else {
            std::vector<std::thread> pool;
            for (unsigned w = 0; w < workers; ++w) pool.emplace_back(work);
            for (auto &t : pool) t.join();
        }
    }
    if (rcs_out) {  // per-file outcome wanted: nothing aborts
        rcs_out->swap(rcs);
        errs_out->swap(errs);
        return RG_OK;
    }
    for (size_t i = 0; i < n; ++i)
        if (rcs[i] != RG_OK) return rg_set_err(c, rcs[i], "%s", errs[i].c_str());
    return RG_OK;
}

// Some(idx) selects among the audio tracks of a container (src/replaygain.rs:838-851); a WAV stream has one
int check_track_index(rg_ctx *c, int32_t track_index, uint32_t n_audio_tracks) {
    if (track_index >= 0 && (uint32_t)track_index >= n_audio_tracks)
        return rg_set_err(c, RG_ERR_INVALID_ARG, "Track index %d out of range (file has %u audio track(s))", track_index, n_audio_tracks);
    return RG_OK;
}

uint32_t file_type_of(const char *path) {  // detect_file_type, src/replaygain.rs:777-783
    return rg_mp4_is_mp4_file(path) ? RG_FILE_AAC : RG_FILE_MP3;
}

}  // namespace

unsigned rg_usable_cores() { return usable_cores(); }

// =================================================================================================
extern "C" int rg_set_decoder_command(rg_ctx *c, const char *command_template) {
    if (!c) return RG_ERR_INVALID_ARG;
    c->decoder_cmd = command_template ? command_template : "";
    return RG_OK;
}

extern "C" int rg_analyze_wav_batch(rg_ctx *c, const void *const *wav, const size_t *wav_len, size_t n, int album,
                                    rg_track_result *out, rg_album_result *album_out) {
    if (!c) return RG_ERR_INVALID_ARG;
    if (n && (!wav || !wav_len)) return rg_set_err(c, RG_ERR_INVALID_ARG, "null input array");
    std::vector<rg_track_desc> descs;
    size_t arena_bytes = 0;
    int rc = stage_wavs(c, wav, wav_len, n, &descs, &arena_bytes);
    if (rc != RG_OK) return rc;
    // the planar arena is on the device: the rest is rg_analyze_pcm_batch / rg_analyze_album_pcm, exact pass included
    if (album) return rg_analyze_album_pcm(c, descs.data(), n, c->d_arena.p, arena_bytes, 1, out, album_out, nullptr);
    return rg_analyze_pcm_batch(c, descs.data(), n, c->d_arena.p, arena_bytes, 1, out, nullptr);
}

extern "C" int rg_analyze_track(rg_ctx *c, const char *path, int32_t track_index, rg_track_result *out) {
    if (!c || !out) return RG_ERR_INVALID_ARG;
    std::vector<LoadedAudio> &in = file_pool(c, 1);
    c->file_track_index = track_index;
    int rc = load_one(c, path, &in);
    c->file_track_index = -1;
    if (rc != RG_OK) return rc;
    rc = check_track_index(c, track_index, in[0].n_audio_tracks);
    if (rc != RG_OK) return rc;
    std::vector<rg_track_desc> descs;
    size_t arena_bytes = 0;
    rc = stage_loaded(c, in, 1, &descs, &arena_bytes);
    if (rc == RG_ERR_FORMAT) return rg_set_err(c, RG_ERR_FORMAT, "Failed to probe format: %s", path);  // src/replaygain.rs:815-822
    if (rc != RG_OK) return rc;
    rc = rg_analyze_pcm_batch(c, descs.data(), 1, c->d_arena.p, arena_bytes, 1, out, nullptr);
    if (rc != RG_OK) return rc;
    out->file_type = in[0].is_mp4 ? RG_FILE_AAC : RG_FILE_MP3;
    return RG_OK;
}

// Files of a long list in groups whose PCM is estimated (24 bytes of planar f32 per byte of file: a 128 kb/s stereo MP3;
// denser files decode to less) to stay within a third of the free device memory, at most 64 GB.
static void file_groups(rg_ctx *c, const char *const *paths, size_t n, std::vector<std::pair<size_t, size_t>> *groups) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)48 << 30;
    size_t budget = std::min((size_t)64 << 30, (free_b + c->d_arena.cap) / 3);
    if (c->group_bytes()) budget = c->group_bytes();  // tests: small groups
    groups->clear();
    for (size_t first = 0; first < n;) {
        size_t last = first, est = 0;
        while (last < n) {
            struct stat st;
            const size_t sz = (paths[last] && stat(paths[last], &st) == 0 && st.st_size > 0) ? (size_t)st.st_size : 0;
            if (last > first && est + sz * 24 > budget) break;
            est += sz * 24;
            ++last;
        }
        groups->push_back(std::make_pair(first, last - first));
        first = last;
    }
}

// analyze_album_with_index (src/replaygain.rs:1044-1074) up to, not including, the album percentile: per-file results in
// input order on the host, the album's [histogram | peak] pack ready on the device (rg_album_finish reads it out; on a
// node with several GPUs rg_album_exchange comes first, rg_node.cpp).  The first failing file IN INPUT ORDER aborts
// the album (:1055) -- the files of a group are loaded together, so every file's outcome is looked at before anything is
// reported -- and *failed_index (if given) says which one it was.  An album whose PCM does not fit the device at once is
// analysed in parts and the parts' histograms and peaks are folded (u32 adds commute: the result does not depend on the
// partition).
extern "C" int rg_analyze_album_begin(rg_ctx *c, const char *const *paths, size_t n, int32_t track_index, rg_track_result *tracks_out,
                                      size_t *failed_index) {
    if (failed_index) *failed_index = (size_t)-1;
    if (!c || (n && (!paths || !tracks_out))) return RG_ERR_INVALID_ARG;
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    std::vector<std::pair<size_t, size_t>> groups;
    file_groups(c, paths, n, &groups);
    const bool trace = c->trace_files;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto fail_at = [&](size_t i, int code) {
        if (failed_index) *failed_index = i;
        return code;
    };
    for (size_t g = 0; g < std::max<size_t>(groups.size(), 1); ++g) {
        const size_t first = groups.empty() ? 0 : groups[g].first, cnt = groups.empty() ? 0 : groups[g].second;
        std::vector<LoadedAudio> &in = file_pool(c, cnt);
        const double t0 = now();
        std::vector<int> rcs;
        std::vector<std::string> errs;
        // album parts (PartsRun): one album that fits the device, decoded by the loader pipeline, nobody else's stream involved
        const bool parts_on = c->parts_on();  // RG_ALBUM_PARTS=0 / tuning key 10 = 1: never (tests, measurements)
        PartsRun parts;
        // (with fewer than three pipeline slots -- tuning key 3 -- a part's enqueue would find its slot's pinned descriptors still
        // being copied behind the decode just issued and the drive thread would sit that decode out: no parts then)
        const bool use_parts = parts_on && c->n_slots >= 3 && groups.size() <= 1 && cnt > 0 && c->gpu_mp3_decode >= 3 && !c->user_attached;
        if (use_parts)  // the parts use every slot's buffers on the decode's stream: nothing of an earlier batch may be in flight
            for (int k = 0; k < RG_SLOT_STREAMS; ++k) RG_HIP(c, hipStreamSynchronize(c->slots[k].stream));
        c->file_track_index = track_index;
        rc = load_many(c, paths + first, cnt, &in, &rcs, &errs, use_parts ? &parts : nullptr);
        c->file_track_index = -1;
        if (rc != RG_OK) return rc;  // not a file's failure: *failed_index stays (size_t)-1, so that a node prefers real file errors of other shares
        for (size_t i = 0; i < cnt; ++i) {
            std::string msg;
            const int frc = file_outcome(in[i], rcs[i], errs[i], paths[first + i], track_index, &msg);
            if (frc != RG_OK) return fail_at(first + i, rg_set_err(c, frc, "%s", msg.c_str()));
        }
        const double t1 = now();
        if (use_parts && !parts.broken && parts.file_of.size() == cnt) {
            // every file was analysed as a part of its chunk: the results are on their way to the host, the packs are on the device
            for (int k = 0; k < RG_SLOT_STREAMS; ++k) RG_HIP(c, hipStreamSynchronize(c->slots[k].stream));
            bool flagged = false;
            for (size_t j = 0; j < cnt; ++j) {
                const rg_track_result &r = c->h_part_results.p[j];
                flagged = flagged || (c->kernel_variant == 0 && (r.flags & RG_TRACK_FLAG_IMPRECISE));
                tracks_out[first + parts.file_of[j]] = r;
            }
            if (!flagged) {  // (a flagged track: the plain route below repeats it on the order-faithful kernel)
                rc = rg_album_parts_fold(c, parts.n_parts);
                if (rc != RG_OK) return rc;
                for (size_t i = 0; i < cnt; ++i) tracks_out[first + i].file_type = in[i].is_mp4 ? RG_FILE_AAC : RG_FILE_MP3;
                if (trace) fprintf(stderr, "[rg_analyze_album] load + decode + analysis in %zu parts %.1f ms, results %.1f ms\n", parts.n_parts, (t1 - t0) * 1e3, (now() - t1) * 1e3);
                return RG_OK;
            }
        }
        std::vector<rg_track_desc> descs;
        size_t arena_bytes = 0;
        rc = stage_loaded(c, in, cnt, &descs, &arena_bytes);
        if (trace) fprintf(stderr, "[rg_analyze_album] load %.1f ms, stage + device decode %.1f ms\n", (t1 - t0) * 1e3, (now() - t1) * 1e3);
        if (rc == RG_ERR_FORMAT) {  // "input i ..." -> the reference's text with the file's name
            size_t i = 0;
            if (sscanf(c->err.c_str(), "input %zu", &i) == 1 && i < cnt)
                return fail_at(first + i, rg_set_err(c, RG_ERR_FORMAT, "Failed to probe format: %s", paths[first + i]));
        }
        if (rc != RG_OK) return rc;  // not a file's failure: *failed_index stays (size_t)-1, so that a node prefers real file errors of other shares
        const double t2 = now();
        if (groups.size() <= 1) rc = rg_album_local_pcm(c, descs.data(), cnt, c->d_arena.p, arena_bytes, 1, tracks_out);
        else rc = rg_album_part(c, descs.data(), cnt, c->d_arena.p, arena_bytes, g, groups.size(), tracks_out + first);
        if (rc != RG_OK) return rc;  // not a file's failure: *failed_index stays (size_t)-1, so that a node prefers real file errors of other shares
        for (size_t i = 0; i < cnt; ++i) tracks_out[first + i].file_type = in[i].is_mp4 ? RG_FILE_AAC : RG_FILE_MP3;
        if (trace) fprintf(stderr, "[rg_analyze_album] analysis %.1f ms\n", (now() - t2) * 1e3);
    }
    if (groups.size() > 1) return rg_album_parts_fold(c, groups.size());
    return RG_OK;
}

extern "C" int rg_analyze_album(rg_ctx *c, const char *const *paths, size_t n, int32_t track_index, rg_track_result *tracks_out,
                                rg_album_result *album_out) {
    if (!c || (n && (!paths || !tracks_out)) || !album_out) return RG_ERR_INVALID_ARG;
    const int rc = rg_analyze_album_begin(c, paths, n, track_index, tracks_out, nullptr);
    if (rc != RG_OK) return rc;
    return rg_album_finish(c, album_out, nullptr);
}

// one group of rg_analyze_tracks: files [first, first + n) of the call; file_errors is indexed by the call's numbering
static int analyze_tracks_group(rg_ctx *c, const char *const *paths, size_t first, size_t n, int32_t track_index, rg_track_result *out,
                                int32_t *status_out) {
    paths += first;
    out += first;
    status_out += first;
    std::vector<LoadedAudio> &in = file_pool(c, n);
    std::vector<int> rcs;
    std::vector<std::string> errs;
    // parts (PartsRun), track mode: every file of the group has to come through the loader pipeline for them to count
    PartsRun parts;
    parts.album = 0;
    const bool use_parts = c->parts_on() && c->n_slots >= 3 && n > 0 && c->gpu_mp3_decode >= 3 && !c->user_attached;
    if (use_parts)
        for (int k = 0; k < RG_SLOT_STREAMS; ++k) RG_HIP(c, hipStreamSynchronize(c->slots[k].stream));
    c->file_track_index = track_index;
    int rc = load_many(c, paths, n, &in, &rcs, &errs, use_parts ? &parts : nullptr);
    c->file_track_index = -1;
    if (rc != RG_OK) return rc;
    if (use_parts && !parts.broken && parts.file_of.size() == n) {
        for (int k = 0; k < RG_SLOT_STREAMS; ++k) RG_HIP(c, hipStreamSynchronize(c->slots[k].stream));
        bool flagged = false;
        for (size_t j = 0; j < n; ++j) flagged = flagged || (c->kernel_variant == 0 && (c->h_part_results.p[j].flags & RG_TRACK_FLAG_IMPRECISE));
        if (!flagged) {  // (else: the plain route below, which repeats flagged tracks on the order-faithful kernel)
            for (size_t j = 0; j < n; ++j) {
                const size_t i = parts.file_of[j];
                out[i] = c->h_part_results.p[j];
                out[i].file_type = in[i].is_mp4 ? RG_FILE_AAC : RG_FILE_MP3;
                status_out[i] = RG_OK;
                c->file_errors[first + i].clear();
            }
            return RG_OK;
        }
    }
    // the batch holds the files that loaded and whose rate the analysis knows; `slot` maps them back
    std::vector<size_t> slot;
    for (size_t i = 0; i < n; ++i) {
        memset(&out[i], 0, sizeof out[i]);
        status_out[i] = rcs[i];
        c->file_errors[first + i] = errs[i];
        if (rcs[i] != RG_OK) continue;
        std::string msg;
        const int frc = file_outcome(in[i], RG_OK, errs[i], paths[i], track_index, &msg);
        if (frc != RG_OK) {
            status_out[i] = frc;
            c->file_errors[first + i] = msg;
            continue;
        }
        slot.push_back(i);
    }
    if (slot.empty()) return RG_OK;
    // compact the good files to the front of the pool (swap keeps every buffer alive for the next call)
    for (size_t k = 0; k < slot.size(); ++k)
        if (slot[k] != k) std::swap(in[k], in[slot[k]]);
    std::vector<rg_track_desc> descs;
    size_t arena_bytes = 0;
    rc = stage_loaded(c, in, slot.size(), &descs, &arena_bytes);
    if (rc == RG_OK) {
        std::vector<rg_track_result> res(slot.size());
        rc = rg_analyze_pcm_batch(c, descs.data(), slot.size(), c->d_arena.p, arena_bytes, 1, res.data(), nullptr);
        if (rc == RG_OK)
            for (size_t k = 0; k < slot.size(); ++k) {
                out[slot[k]] = res[k];
                out[slot[k]].file_type = in[k].is_mp4 ? RG_FILE_AAC : RG_FILE_MP3;
            }
    }
    if (rc != RG_OK) {  // a failure of the batch itself (a WAV of a kind the library cannot stage, a device error): every file in it carries it
        for (size_t k = 0; k < slot.size(); ++k) {
            status_out[slot[k]] = rc;
            c->file_errors[first + slot[k]] = c->err;
        }
        return RG_OK;
    }
    return RG_OK;
}

// `-r` over a whole library must not need the whole library's PCM in HBM at once: the files are taken in groups
// (file_groups).  Tracks are independent, so the groups are too.
extern "C" int rg_analyze_tracks(rg_ctx *c, const char *const *paths, size_t n, int32_t track_index, rg_track_result *out,
                                 int32_t *status_out) {
    if (!c || (n && (!paths || !out || !status_out))) return RG_ERR_INVALID_ARG;
    c->file_errors.assign(n, std::string());
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    std::vector<std::pair<size_t, size_t>> groups;
    file_groups(c, paths, n, &groups);
    for (const auto &g : groups) {
        rc = analyze_tracks_group(c, paths, g.first, g.second, track_index, out, status_out);
        if (rc != RG_OK) return rc;
    }
    return RG_OK;
}

extern "C" const char *rg_tracks_error(const rg_ctx *c, size_t i) {
    if (!c || i >= c->file_errors.size()) return "";
    return c->file_errors[i].c_str();
}

// find_peak_amplitude (src/replaygain.rs:1140-1249): max |x| over ALL channels, no loudness analysis
extern "C" int rg_find_peak_amplitude(rg_ctx *c, const char *path, rg_peak_result *out) {
    if (!c || !out) return RG_ERR_INVALID_ARG;
    std::vector<LoadedAudio> &in = file_pool(c, 1);
    int rc = load_one(c, path, &in);
    if (rc != RG_OK) return rc;
    std::vector<rg_track_desc> descs;
    size_t arena_bytes = 0;
    rc = stage_loaded(c, in, 1, &descs, &arena_bytes);
    if (rc == RG_ERR_FORMAT) return rg_set_err(c, RG_ERR_FORMAT, "Failed to probe format: %s", path);
    if (rc != RG_OK) return rc;
    // the arena was produced on the stream rg_find_peak_pcm uses, so no further ordering is needed
    return rg_find_peak_pcm(c, &descs[0], c->d_arena.p, arena_bytes, 1, out);
}

// Measurement hook (bench.py, tools/): the device decode chain alone.  `copies` copies of one MPEG Layer III stream form ONE
// chunk of the default route (compacted by the host once, staged in pinned memory, copied H2D per repetition on the copy
// stream), and the chunk's three stages -- frame parser (three launches), Huffman, back half -- are bracketed with HIP events
// on the stream they run on.  ms_out[0..2] = average duration of each stage over `reps` repetitions, ms_out[3] = first event
// to last (the chain), ms_out[4] = per chunk in the file route's own arrangement (parser and sort beside the chunk before);
// the PCM lands in the analysis arena as in a real call and is not copied back.
extern "C" int rg_mp3_decode_bench(rg_ctx *c, const void *data, size_t len, uint32_t copies, uint32_t reps, double *ms_out /* 5 */,
                                   uint64_t *units_out, uint64_t *compressed_bytes_out, uint64_t *frames_out) {
    if (!c || !data || !ms_out || copies == 0 || reps == 0) return RG_ERR_INVALID_ARG;
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    Mp3Pipe &P = mp3_pipe(c);
    if (P.scratch.empty()) P.scratch.resize(1);
    Mp3Scratch &sc = P.scratch[0];
    if (sc.cap < len + 64) {
        uint8_t *q = static_cast<uint8_t *>(realloc(sc.p, len + 64));
        if (!q) return rg_set_err(c, RG_ERR_IO, "out of memory");
        sc.p = q;
        sc.cap = len + 64;
    }
    memcpy(sc.p, data, len);
    memset(sc.p + len, 0, 64);
    rg_mp3_stream_info si;
    uint64_t main_len = 0;
    if (rg_mp3_compact_stream(sc.p, len, &sc.slots, &sc.tiles, &main_len, &si) != RG_MP3DEC_OK)
        return rg_set_err(c, RG_ERR_FORMAT, "%s", rg_mp3dec_last_error());
    for (int s = 0; s < c->n_slots; ++s) RG_HIP(c, hipStreamSynchronize(c->slots[s].stream));
    const size_t per_stream = (size_t)si.frames * si.channels * sizeof(float);
    RG_HIP(c, c->d_arena.reserve(per_stream * copies + 64));
    const size_t one = align64((size_t)main_len + 8) + align64(sc.slots.size()) + align64(sc.tiles.size() * sizeof(uint64_t));
    const size_t tracks_off = one * copies;
    const size_t total = tracks_off + rg_mp3dev_track_bytes(copies);
    Mp3Stage &st = P.stage[0];
    if (!st.staged) RG_HIP(c, hipEventCreateWithFlags(&st.staged, hipEventDisableTiming));
    if (st.cap < total) {
        if (st.p) (void)hipHostFree(st.p);
        st.p = nullptr;
        st.cap = 0;
        RG_HIP(c, hipHostMalloc((void **)&st.p, total + total / 8, hipHostMallocDefault));
        st.cap = total + total / 8;
    }
    std::vector<RgMp3StreamItem> items(copies);
    for (uint32_t k = 0; k < copies; ++k) {
        RgMp3StreamItem &it = items[k];
        it.main_off = one * k;
        it.slots_off = it.main_off + align64((size_t)main_len + 8);
        it.tiles_off = it.slots_off + align64(sc.slots.size());
        memcpy(st.p + it.main_off, sc.p, (size_t)main_len);
        memset(st.p + it.main_off + main_len, 0, (size_t)(it.slots_off - it.main_off - main_len));
        memcpy(st.p + it.slots_off, sc.slots.data(), sc.slots.size());
        memcpy(st.p + it.tiles_off, sc.tiles.data(), sc.tiles.size() * sizeof(uint64_t));
        it.n_frames = si.audio_frames;
        it.channels = si.channels;
        it.rate_row = (uint32_t)rg_mp3_rate_row(si.sample_rate);
        it.lsf = si.mpeg_version == 1 ? 0u : 1u;
        it.result_index = k;
        it.d_ch0 = reinterpret_cast<float *>(c->d_arena.p + per_stream * k);
    }
    hipStream_t fs = c->slots[0].stream;  // as the file route: never the stream the copies run on (rg_mp3dev_enqueue_chunk)
    // one set of events per repetition: the repetitions are enqueued back to back (a synchronise after each would let the
    // clocks fall between them) and read out at the end
    if (reps > 256) reps = 256;
    std::vector<hipEvent_t> ev((size_t)4 * (reps + 1), nullptr);
    for (hipEvent_t &e : ev) RG_HIP(c, hipEventCreate(&e));
    double sum[4] = {0, 0, 0, 0};
    rc = rg_mp3dev_reserve_results(c, copies, fs);
    for (uint32_t r = 0; r < reps + 1 && rc == RG_OK; ++r) {  // the first repetition is not counted
        c->mp3_bench_ev = &ev[(size_t)4 * r];
        rc = rg_mp3dev_enqueue_chunk(c, (int)(r & 1), st.p, total, tracks_off, st.staged, items.data(), copies, fs);
        c->mp3_bench_ev = nullptr;
    }
    if (rc == RG_OK && hipStreamSynchronize(fs) != hipSuccess) rc = rg_set_err(c, RG_ERR_DEVICE, "decode bench: stream synchronise failed");
    for (uint32_t r = 1; r < reps + 1 && rc == RG_OK; ++r) {
        for (int k = 0; k < 3; ++k) {
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, ev[(size_t)4 * r + k], ev[(size_t)4 * r + k + 1]);
            sum[k] += ms;
        }
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, ev[(size_t)4 * r], ev[(size_t)4 * r + 3]);
        sum[3] += ms;
    }
    // The production arrangement: the same chunk `reps` times the way the file route enqueues chunks -- frame parser and lane
    // sort on the copy stream behind the chunk's H2D, i.e. beside the Huffman / back-half kernels of the chunk before -- first
    // event to last on the chain's stream, per chunk (the first chunk's parser has nothing to run beside: 1 / reps of the figure).
    double piped = 0.0;
    if (rc == RG_OK) {
        for (uint32_t r = 0; r < reps + 2 && rc == RG_OK; ++r) {
            if (r == 2) rc = hipEventRecord(ev[0], fs) == hipSuccess ? RG_OK : RG_ERR_DEVICE;  // two chunks ahead: the pipeline is full
            if (rc == RG_OK) rc = rg_mp3dev_enqueue_chunk(c, (int)(r & 1), st.p, total, tracks_off, st.staged, items.data(), copies, fs);
        }
        if (rc == RG_OK && (hipEventRecord(ev[1], fs) != hipSuccess || hipStreamSynchronize(fs) != hipSuccess))
            rc = rg_set_err(c, RG_ERR_DEVICE, "decode bench: stream synchronise failed");
        float ms = 0.0f;
        if (rc == RG_OK) (void)hipEventElapsedTime(&ms, ev[0], ev[1]);
        piped = (double)ms / reps;
    }
    for (hipEvent_t &e : ev) (void)hipEventDestroy(e);
    if (rc != RG_OK) return rc;
    for (int k = 0; k < 4; ++k) ms_out[k] = sum[k] / reps;
    ms_out[4] = piped;
    const uint64_t per_frame = si.mpeg_version == 1 ? 2u : 1u;
    if (units_out) *units_out = (uint64_t)si.audio_frames * per_frame * si.channels * copies;
    if (compressed_bytes_out) *compressed_bytes_out = (uint64_t)(main_len + sc.slots.size()) * copies;
    if (frames_out) *frames_out = (uint64_t)si.frames * copies;
    return RG_OK;
}

// Decode one MPEG Layer III stream through the split decoder (stage A on the host, B-E on the device) and bring the PCM
// back: the parity hook of tests/test_gpu_mp3.py.  Same outputs as rg_mp3_decode_f32.
extern "C" int rg_mp3_decode_device(rg_ctx *c, const void *data, size_t len, float *ch0, float *ch1, uint64_t capacity,
                                    void *info) {
    rg_mp3_stream_info *out = static_cast<rg_mp3_stream_info *>(info);
    if (!c || !data || !out || !ch0) return RG_ERR_INVALID_ARG;
    if (c->gpu_mp3_decode >= 3) {  // the default route: the host strips headers and side information, nothing else
        int rc = rg_bind_device(c);
        if (rc != RG_OK) return rc;
        Mp3Pipe &P = mp3_pipe(c);
        if (P.scratch.empty()) P.scratch.resize(1);
        Mp3Scratch &sc = P.scratch[0];
        if (sc.cap < len + 64) {
            uint8_t *q = static_cast<uint8_t *>(realloc(sc.p, len + 64));
            if (!q) return rg_set_err(c, RG_ERR_IO, "out of memory");
            sc.p = q;
            sc.cap = len + 64;
        }
        memcpy(sc.p, data, len);
        memset(sc.p + len, 0, 64);
        uint64_t main_len = 0;
        if (rg_mp3_compact_stream(sc.p, len, &sc.slots, &sc.tiles, &main_len, out) != RG_MP3DEC_OK)
            return rg_set_err(c, RG_ERR_FORMAT, "%s", rg_mp3dec_last_error());
        if (out->frames > capacity) return rg_set_err(c, RG_ERR_INVALID_ARG, "capacity %llu < %llu frames", (unsigned long long)capacity, (unsigned long long)out->frames);
        if (out->channels == 2 && !ch1) return rg_set_err(c, RG_ERR_INVALID_ARG, "stereo stream needs a second output channel");
        for (int s = 0; s < c->n_slots; ++s) RG_HIP(c, hipStreamSynchronize(c->slots[s].stream));
        const size_t bytes = (size_t)out->frames * out->channels * sizeof(float);
        RG_HIP(c, c->d_arena.reserve(bytes ? bytes : 16));
        Mp3Stage &st = P.stage[0];
        if (!st.staged) RG_HIP(c, hipEventCreateWithFlags(&st.staged, hipEventDisableTiming));
        RgMp3StreamItem it{};
        it.main_off = 0;
        it.slots_off = align64((size_t)main_len + 8);
        it.tiles_off = it.slots_off + align64(sc.slots.size());
        const size_t tracks_off = it.tiles_off + align64(sc.tiles.size() * sizeof(uint64_t));
        const size_t total = tracks_off + rg_mp3dev_track_bytes(1);
        if (st.cap < total) {
            if (st.p) (void)hipHostFree(st.p);
            st.p = nullptr;
            st.cap = 0;
            RG_HIP(c, hipHostMalloc((void **)&st.p, total + total / 8, hipHostMallocDefault));
            st.cap = total + total / 8;
        }
        memcpy(st.p, sc.p, (size_t)main_len);
        memset(st.p + main_len, 0, (size_t)(it.slots_off - main_len));
        memcpy(st.p + it.slots_off, sc.slots.data(), sc.slots.size());
        memcpy(st.p + it.tiles_off, sc.tiles.data(), sc.tiles.size() * sizeof(uint64_t));
        it.n_frames = out->audio_frames;
        it.channels = out->channels;
        it.rate_row = (uint32_t)rg_mp3_rate_row(out->sample_rate);
        it.lsf = out->mpeg_version == 1 ? 0u : 1u;
        it.result_index = 0;
        it.d_ch0 = reinterpret_cast<float *>(c->d_arena.p);
        hipStream_t fs = c->slots[0].stream;
        rc = rg_mp3dev_reserve_results(c, 1, fs);
        if (rc != RG_OK) return rc;
        rc = rg_mp3dev_enqueue_chunk(c, 0, st.p, total, tracks_off, st.staged, &it, 1, fs);
        if (rc != RG_OK) return rc;
        rc = rg_mp3dev_fetch_results(c, 1, fs);
        if (rc != RG_OK) return rc;
        RG_HIP(c, hipStreamSynchronize(fs));
        const uint32_t granules = rg_mp3dev_results(c)[0];
        const uint32_t per_frame = it.lsf ? 1u : 2u;
        const uint32_t walked = out->audio_frames;
        out->frames = (uint64_t)granules * 576;
        out->audio_frames = granules / per_frame;
        out->skipped_frames = walked - out->audio_frames;
        if (out->frames) {
            RG_HIP(c, hipMemcpy(ch0, it.d_ch0, (size_t)out->frames * sizeof(float), hipMemcpyDeviceToHost));
            if (out->channels == 2) RG_HIP(c, hipMemcpy(ch1, it.d_ch0 + out->frames, (size_t)out->frames * sizeof(float), hipMemcpyDeviceToHost));
        }
        return RG_OK;
    }
    rg_mp3_stream_info si;
    if (rg_mp3_scan(data, len, &si) != RG_MP3DEC_OK) return rg_set_err(c, RG_ERR_FORMAT, "%s", rg_mp3dec_last_error());
    const uint64_t cap = (uint64_t)si.audio_frames * (si.mpeg_version == 1 ? 2u : 1u) * si.channels;
    std::vector<int16_t> is;
    std::vector<rg_mp3_unit> units;
    std::vector<uint8_t> main_stream;
    std::vector<RgMp3HuffRec> recs;
    uint64_t n_units = 0;
    if (c->gpu_mp3_decode == 2) {
        if (rg_mp3_index_stream(data, len, &main_stream, &recs, out) != RG_MP3DEC_OK)
            return rg_set_err(c, RG_ERR_FORMAT, "%s", rg_mp3dec_last_error());
        n_units = recs.size();
    } else {
        is.resize((size_t)cap * 576 + 1);
        units.resize((size_t)cap + 1);
        if (rg_mp3_parse_units(data, len, is.data(), units.data(), cap, &n_units, out) != RG_MP3DEC_OK)
            return rg_set_err(c, RG_ERR_FORMAT, "%s", rg_mp3dec_last_error());
    }
    if (out->frames > capacity) return rg_set_err(c, RG_ERR_INVALID_ARG, "capacity %llu < %llu frames", (unsigned long long)capacity, (unsigned long long)out->frames);
    if (out->channels == 2 && !ch1) return rg_set_err(c, RG_ERR_INVALID_ARG, "stereo stream needs a second output channel");
    int rc = rg_bind_device(c);
    if (rc != RG_OK) return rc;
    for (int s = 0; s < c->n_slots; ++s) RG_HIP(c, hipStreamSynchronize(c->slots[s].stream));
    const size_t bytes = (size_t)out->frames * out->channels * sizeof(float);
    RG_HIP(c, c->d_arena.reserve(bytes ? bytes : 16));
    RgMp3SplitItem it{};
    it.is = is.data();
    it.units = units.data();
    if (c->gpu_mp3_decode == 2) {
        it.recs = recs.data();
        it.main = main_stream.data();
        it.main_len = main_stream.size();
    }
    it.n_units = n_units;
    it.channels = out->channels;
    it.rate_row = (uint32_t)rg_mp3_rate_row(out->sample_rate);
    it.lsf = out->mpeg_version == 1 ? 0u : 1u;
    it.d_ch0 = reinterpret_cast<float *>(c->d_arena.p);
    it.d_ch1 = out->channels == 2 ? it.d_ch0 + out->frames : nullptr;
    hipStream_t fs = c->slot().stream;
    rc = rg_mp3dev_decode(c, &it, 1, fs);
    if (rc != RG_OK) return rc;
    if (out->frames) {
        RG_HIP(c, hipMemcpy(ch0, it.d_ch0, (size_t)out->frames * sizeof(float), hipMemcpyDeviceToHost));
        if (out->channels == 2) RG_HIP(c, hipMemcpy(ch1, it.d_ch1, (size_t)out->frames * sizeof(float), hipMemcpyDeviceToHost));
    }
    return RG_OK;
}
