/* mp3rgain_amd_mp3.h -- C ABI of the lossless MP3 gain path:
 * the global_gain frame scanner / patcher and the APEv2 undo tags of mp3rgain's library crate.
 *
 * Host-only byte work (about one read-modify-write per 100 bytes of file, I/O bound): it is here so
 * that the step count computed by the GPU analysis path (rg_track_result.gain_steps) can actually be
 * applied, behind the reference's own function names.  Citations are file:line under /root/reference.
 *
 *   rg_mp3_analyze                 analyze                      src/lib.rs:470-514
 *   rg_mp3_apply_gain              apply_gain                   src/lib.rs:602-616
 *   rg_mp3_apply_gain_db           apply_gain_db                src/lib.rs:626-629
 *   rg_mp3_apply_gain_wrap         apply_gain_wrap              src/lib.rs:1232-1246
 *   rg_mp3_apply_gain_channel      apply_gain_channel           src/lib.rs:748-768
 *   rg_mp3_apply_gain_with_undo    apply_gain_with_undo         src/lib.rs:1280-1308
 *   rg_mp3_apply_gain_with_undo_wrap  apply_gain_with_undo_wrap src/lib.rs:1249-1277
 *   rg_mp3_apply_gain_channel_with_undo  apply_gain_channel_with_undo  src/lib.rs:771-812
 *   rg_mp3_undo_gain               undo_gain                    src/lib.rs:1311-1338
 *   rg_mp3_is_mono                 is_mono                      src/lib.rs:670-673
 *   rg_ape_get / rg_ape_set / rg_ape_remove / rg_ape_delete     ApeTag + read/write/delete_ape_tag  src/lib.rs:866-1163
 *   *_data variants                the in-memory cores: apply_gain_to_data (:544-592), apply_gain_to_channel_data
 *                                  (:677-737), iterate_frames (:412-461), read_ape_tag (:974-1027)
 *
 * Return value: >= 0 on success (frames modified where that is what the reference returns), negative
 * rg_mp3_status otherwise; rg_mp3_last_error() holds the anyhow-style message (thread-local).
 */
#ifndef MP3RGAIN_AMD_MP3_H
#define MP3RGAIN_AMD_MP3_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rg_mp3_status {
    RG_MP3_ERR_IO = -101,        /* "Failed to read: ..." / "Failed to write: ..." */
    RG_MP3_ERR_NO_FRAMES = -102, /* "No valid MP3 frames found"                       src/lib.rs:497-499 */
    RG_MP3_ERR_MONO = -103,      /* "Cannot apply channel-specific gain to mono file. Use -g for mono files." :757-759 */
    RG_MP3_ERR_NO_APE = -104,    /* "No APE tag found - cannot undo"                  :1312-1313 */
    RG_MP3_ERR_NO_UNDO = -105,   /* "No MP3GAIN_UNDO tag found - cannot undo"         :1315-1317 */
    RG_MP3_ERR_ARG = -106
} rg_mp3_status;

/* Mp3Analysis, src/lib.rs:58-75 (the two strings as enums + text) */
typedef struct rg_mp3_analysis {
    uint64_t frame_count;
    uint32_t mpeg_version;   /* 1 = "MPEG1", 2 = "MPEG2", 25 = "MPEG2.5" */
    uint32_t channel_mode;   /* 0 Stereo, 1 Joint Stereo, 2 Dual Channel, 3 Mono */
    uint8_t min_gain;
    uint8_t max_gain;
    uint8_t pad_[6];
    double avg_gain;
    int32_t headroom_steps;
    int32_t pad2_;
    double headroom_db;
    char mpeg_version_str[8];   /* "MPEG1" / "MPEG2" / "MPEG2.5" */
    char channel_mode_str[16];  /* "Stereo" / "Joint Stereo" / "Dual Channel" / "Mono" */
} rg_mp3_analysis;

/* parse_header's result, src/lib.rs:132-142, for tests and tools */
typedef struct rg_mp3_header {
    uint32_t mpeg_version; /* 1 / 2 / 25 */
    uint32_t has_crc;
    uint32_t bitrate_kbps;
    uint32_t sample_rate;
    uint32_t padding;
    uint32_t channel_mode;
    uint32_t frame_size;
} rg_mp3_header;

const char *rg_mp3_last_error(void);

/* ---- pure byte-level pieces (unit-tested by the reference, src/lib.rs:1340-1444) ---------------- */
int rg_mp3_parse_header(const uint8_t *hdr, size_t len, rg_mp3_header *out);               /* 1 = valid */
uint8_t rg_mp3_read_gain_at(const uint8_t *data, size_t len, size_t byte_offset, unsigned bit_offset);
void rg_mp3_write_gain_at(uint8_t *data, size_t len, size_t byte_offset, unsigned bit_offset, uint8_t value);
size_t rg_mp3_skip_id3v2(const uint8_t *data, size_t len);
size_t rg_mp3_find_audio_end(const uint8_t *data, size_t len);
int rg_mp3_is_xing_frame(const uint8_t *data, size_t len, size_t frame_offset);            /* header parsed at frame_offset */
/* byte/bit positions of the global_gain fields of the frame at frame_offset: up to 4 (gr x ch); returns the count */
int rg_mp3_gain_locations(const uint8_t *data, size_t len, size_t frame_offset, size_t *byte_offsets, unsigned *bit_offsets);

/* ---- in-memory cores ---------------------------------------------------------------------------- */
int64_t rg_mp3_analyze_data(const uint8_t *data, size_t len, rg_mp3_analysis *out);
int64_t rg_mp3_apply_gain_data(uint8_t *data, size_t len, int32_t gain_steps, int wrap);
int64_t rg_mp3_apply_gain_channel_data(uint8_t *data, size_t len, int channel /* 0 left, 1 right */, int32_t gain_steps);

/* ---- file level, the reference's public functions -------------------------------------------------- */
int64_t rg_mp3_analyze(const char *path, rg_mp3_analysis *out);
int64_t rg_mp3_apply_gain(const char *path, int32_t gain_steps);
int64_t rg_mp3_apply_gain_db(const char *path, double gain_db);
int64_t rg_mp3_apply_gain_wrap(const char *path, int32_t gain_steps);
int64_t rg_mp3_apply_gain_channel(const char *path, int channel, int32_t gain_steps);
int64_t rg_mp3_apply_gain_with_undo(const char *path, int32_t gain_steps);
int64_t rg_mp3_apply_gain_with_undo_wrap(const char *path, int32_t gain_steps);
int64_t rg_mp3_apply_gain_channel_with_undo(const char *path, int channel, int32_t gain_steps);
int64_t rg_mp3_undo_gain(const char *path);
int rg_mp3_is_mono(const char *path); /* 1 / 0, negative status on error */

/* ---- APEv2 tags ------------------------------------------------------------------------------------- */
/* value of `key` (case-insensitive) into buf (NUL-terminated, truncated to buflen); returns the value's
 * length, -1 when the file has no APE tag or no such item */
int64_t rg_ape_get(const char *path, const char *key, char *buf, size_t buflen);
int64_t rg_ape_get_data(const uint8_t *data, size_t len, const char *key, char *buf, size_t buflen);
int64_t rg_ape_item_count_data(const uint8_t *data, size_t len);                            /* -1: no tag */
int rg_ape_set(const char *path, const char *key, const char *value);                       /* ApeTag::set + write_ape_tag */
int rg_ape_remove(const char *path, const char *key);                                       /* ApeTag::remove + write/delete */
int rg_ape_delete(const char *path);                                                        /* delete_ape_tag */

#ifdef __cplusplus
}
#endif
#endif /* MP3RGAIN_AMD_MP3_H */
