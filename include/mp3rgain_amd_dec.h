/* mp3rgain_amd_dec.h -- C ABI of the PCM decode row: MPEG-1 / MPEG-2 / MPEG-2.5
 * Layer III bitstream -> planar f32 PCM, the form `process_audio_buffer` consumes (src/replaygain.rs:953-1029).
 *
 * What it replaces in the reference (citations are file:line under /root/reference):
 *   symphonia::default::get_probe().format(...)      src/replaygain.rs:815-822   container probe: ID3v2 skip, frame sync,
 *                                                                                 Xing/Info/VBRI header frame not decoded
 *   track.codec_params.{sample_rate, channels}       src/replaygain.rs:855-858   rg_mp3_stream_info
 *   format.next_packet() / decoder.decode(&packet)   src/replaygain.rs:881-900   the frame loop: a frame that cannot be
 *                                                                                 decoded is skipped (DecodeError -> continue),
 *                                                                                 the end of the data ends the track
 * The decoder crate (symphonia 0.5.5, Cargo.lock:230-311) is not part of the reference tree; this is an independent
 * implementation of ISO/IEC 11172-3 / 13818-3 Layer III with the same packet semantics: every audio frame yields
 * 1152 (MPEG-1) or 576 (MPEG-2 / 2.5) frames of PCM per channel, FormatOptions::default() => no gapless trimming
 * (encoder delay and padding are decoded and analysed like everything else), f32 samples nominally in [-1, 1].
 *
 * Host code, no GPU needed.  Plain C types only.
 */
#ifndef MP3RGAIN_AMD_DEC_H
#define MP3RGAIN_AMD_DEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rg_mp3dec_status {
    RG_MP3DEC_OK = 0,
    RG_MP3DEC_ERR_ARG = -1,
    RG_MP3DEC_ERR_NO_AUDIO = -2, /* no Layer III frame found: the reference's "Failed to probe format" */
    RG_MP3DEC_ERR_CAPACITY = -3  /* output buffers too small (frames needed are reported in the info) */
} rg_mp3dec_status;

typedef struct rg_mp3_stream_info {
    uint32_t sample_rate;     /* of the first audio frame; frames at another rate are skipped            */
    uint32_t channels;        /* 1 or 2, of the first audio frame                                        */
    uint64_t frames;          /* PCM frames per channel: scan = upper bound, decode = what was produced  */
    uint32_t audio_frames;    /* Layer III frames found (scan) / decoded (decode)                        */
    uint32_t skipped_frames;  /* frames that could not be decoded and were dropped (bit reservoir underflow,
                                 invalid side info, a channel count other than the stream's): DecodeError ->
                                 continue, src/replaygain.rs:896-899                                       */
    uint32_t info_frame;      /* 1: a Xing / Info / VBRI header frame was found and not decoded (lib.rs:388-408 skips it too) */
    uint32_t id3v2_bytes;     /* size of the ID3v2 tag skipped at the start                              */
    uint32_t mpeg_version;    /* 1, 2, or 25 (MPEG-2.5)                                                  */
    uint32_t samples_per_frame; /* 1152 or 576                                                           */
    uint64_t first_frame_offset;
    uint32_t junk_bytes;      /* bytes skipped while resynchronising                                     */
    uint32_t reserved;
} rg_mp3_stream_info;

/* Header-level scan (no decoding): stream parameters and an upper bound of the PCM length. */
int rg_mp3_scan(const void *data, size_t len, rg_mp3_stream_info *out);

/* Decode a whole stream.  ch0 / ch1: planar outputs of `capacity` frames each (ch1 may be NULL for a mono stream).
 * On RG_MP3DEC_OK out->frames holds the number of frames written per channel. */
int rg_mp3_decode_f32(const void *data, size_t len, float *ch0, float *ch1, uint64_t capacity, rg_mp3_stream_info *out);

/* ---- split decode: stage A on the host, stages B-E on the GPU (include/mp3rgain_amd.h: tuning key 6) -------------------
 * The serial part of Layer III decoding -- frame walk, side information, bit reservoir, scalefactors, Huffman -- is a
 * few percent of a decoder's work and strictly sequential per frame; everything after it (requantisation, joint stereo,
 * reordering, alias reduction, IMDCT + overlap, polyphase synthesis) has no recursion across granules and runs on the
 * device (rg_mp3dev.hip).  rg_mp3_parse_units is that serial part: per decoded granule and channel one `rg_mp3_unit`
 * and 576 quantised spectral values (int16), in decode order: unit = granule * channels + channel, granule counting the
 * granules of decodable frames only (a dropped frame contributes nothing, exactly as in rg_mp3_decode_f32). */
typedef struct rg_mp3_unit {
    uint8_t sf[40];          /* scalefactors, flat: long bands, then short bands x 3 windows (see rg_mp3dec.cpp)        */
    uint64_t illegal;        /* bit i: sf[i] is an LSF intensity position meaning "not intensity coded"                */
    uint16_t nz;             /* number of spectral lines decoded (the rest are zero)                                   */
    uint8_t global_gain;
    uint8_t block_type;      /* 0 normal, 1 start, 2 short, 3 stop                                                      */
    uint8_t mixed;
    uint8_t subblock_gain[3];
    uint8_t scalefac_scale;
    uint8_t preflag;
    uint8_t long_end;        /* long scalefactor bands in this granule (22, 8 / 6 for mixed, 0 for short)               */
    uint8_t short_start;     /* first short band (13 = none)                                                            */
    uint8_t mode_ext;        /* joint stereo only: bit 0 intensity, bit 1 mid/side; 0 otherwise                         */
    uint8_t intensity_scale; /* LSF: low bit of the right channel's scalefac_compress                                   */
    uint8_t reserved[2];     /* [0]: units written by the DEVICE Huffman stage only (mp3rgain_amd/csrc/rg_mp3dev.h): 4-line words of
                              *      the spectrum row's second byte plane worth reading; 0 from rg_mp3_parse_units               */
} rg_mp3_unit;               /* 64 bytes */

/* is_out: int16 [capacity_units][576]; units_out: [capacity_units].  out->frames / audio_frames / skipped_frames as
 * rg_mp3_decode_f32 reports them; *n_units = units written. */
int rg_mp3_parse_units(const void *data, size_t len, int16_t *is_out, rg_mp3_unit *units_out, uint64_t capacity_units,
                       uint64_t *n_units, rg_mp3_stream_info *out);

/* The frame walk alone (what the default device route runs on the host): how many units and PCM frames the stream
 * decodes to, decided from headers and side information only.  Must agree with rg_mp3_parse_units / rg_mp3_decode_f32 on
 * every input, damaged ones included (tests/test_mp3dec.py fuzzes that). */
int rg_mp3_index_units(const void *data, size_t len, uint64_t *n_units, rg_mp3_stream_info *out);

/* Test hook for the default device route (tuning key 6 = 3), where the host only strips headers and side information
 * from the stream and the device decides which frames decode: runs that route's host half and the frame logic the device
 * shares with it (mp3rgain_amd/csrc/rg_mp3_frame.h) on the CPU and compares with rg_mp3_index_units' view of the stream.
 * 0 = identical, 1 = different, < 0 = no audio. */
int rg_mp3_index_selfcheck(const void *data, size_t len);

/* Text of the last error of the calling thread ("" if none). */
const char *rg_mp3dec_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MP3RGAIN_AMD_DEC_H */
