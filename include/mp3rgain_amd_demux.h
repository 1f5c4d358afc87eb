/* mp3rgain_amd_demux.h -- container walkers in front of the analysis path: ISO base media (MP4 / M4A) and ADTS.
 *
 * The reference hands every file to symphonia (src/replaygain.rs:807-863): probe -> tracks -> the tracks whose codec it
 * can decode (`codec != CODEC_TYPE_NULL`; its build enables "mp3", "aac", "isomp4", Cargo.toml:24) -> the first of them, or
 * the one `-i` names (:838-851) -> its sample rate and channel count (:854-858) -> packets (:881-904).  symphonia's source is
 * not in the reference's tree, so what follows restates ISO/IEC 14496-12 (boxes, sample tables) and 14496-3 (AudioSpecificConfig,
 * ADTS) and keeps the reference's own error texts; its demuxer's corner-case behaviour is [unverified].
 *
 *   rg_mp4_audio_tracks   the audio tracks of a file in file order: `soun` handler, sample entry `mp4a` with an AAC
 *                         (0x40, 0x66-0x68) or MPEG audio (0x69, 0x6B) object type, or `.mp3`
 *   rg_mp4_access_units   file offset and size of every sample (AAC access unit / MP3 frame) of one of them, from
 *                         stsz|stz2 + stsc + stco|co64
 *   rg_adts_scan / rg_adts_access_units   the same for a raw ADTS stream (.aac)
 *
 * What the library does with them (rg_files.hip): an MP4 file's audio tracks are counted ("No audio track found",
 * "Track index {} out of range (file has {} audio track(s))"), the selected track's rate is known before anything is decoded,
 * MPEG Layer III in MP4 is decoded by the library's own decoder from the sample table, and AAC goes to the decoder command
 * (rg_set_decoder_command; "{track}" in it is replaced by the audio track index) -- an AAC-LC decoder is not built: nothing in
 * the build image can check one.
 * Host code; plain C types. */
#ifndef MP3RGAIN_AMD_DEMUX_H
#define MP3RGAIN_AMD_DEMUX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { RG_DEMUX_OK = 0, RG_DEMUX_ERR_ARG = -1, RG_DEMUX_ERR_FORMAT = -2 /* not the container / damaged */, RG_DEMUX_ERR_RANGE = -3 };
enum { RG_CODEC_AAC = 1, RG_CODEC_MP3 = 2 };

typedef struct rg_mp4_audio_track {
    uint32_t track_id;       /* tkhd */
    uint32_t codec;          /* RG_CODEC_* */
    uint32_t object_type;    /* esds objectTypeIndication (0 for a `.mp3` sample entry)                                  */
    uint32_t sample_rate;    /* AudioSpecificConfig's when there is one, else the sample entry's                         */
    uint32_t channels;       /* likewise                                                                                 */
    uint32_t timescale;      /* mdhd */
    uint64_t duration;       /* mdhd, in timescale units                                                                 */
    uint64_t n_samples;      /* access units in the sample table                                                         */
    uint32_t audio_object_type; /* AudioSpecificConfig (2 = AAC-LC), 0 when absent                                       */
    uint32_t asc_len;
    uint8_t asc[32];         /* AudioSpecificConfig bytes (what a decoder is configured with)                            */
} rg_mp4_audio_track;

/* out[0 .. min(cap, *n_audio)) filled; *n_audio = audio tracks in the file.  RG_DEMUX_ERR_FORMAT: no moov box. */
int rg_mp4_audio_tracks(const void *data, size_t len, rg_mp4_audio_track *out, size_t cap, size_t *n_audio);
/* offsets / sizes may be NULL to count; *n = samples of the track.  A sample that reaches past `len` ends the list (the
 * reference's loop ends at the reader's UnexpectedEof, src/replaygain.rs:884-888). */
int rg_mp4_access_units(const void *data, size_t len, size_t audio_index, uint64_t *offsets, uint32_t *sizes, size_t cap, size_t *n);

typedef struct rg_adts_info {
    uint32_t sample_rate, channels;
    uint32_t profile;            /* 2-bit field + 1 = audio object type (2 = AAC-LC)                                     */
    uint32_t mpeg_version;       /* ID bit: 0 = MPEG-4, 1 = MPEG-2                                                        */
    uint64_t frames;             /* ADTS frames (each carries number_of_raw_data_blocks + 1 blocks of 1024 samples)      */
    uint64_t raw_blocks;
    uint64_t first_frame_offset; /* ID3v2 tag / junk before the first frame                                              */
    uint64_t junk_bytes;         /* bytes skipped while resynchronising                                                   */
} rg_adts_info;
int rg_adts_scan(const void *data, size_t len, rg_adts_info *out);
/* payload (after the 7- or 9-byte header) offset and size of every frame */
int rg_adts_access_units(const void *data, size_t len, uint64_t *offsets, uint32_t *sizes, size_t cap, size_t *n);
const char *rg_demux_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MP3RGAIN_AMD_DEMUX_H */
