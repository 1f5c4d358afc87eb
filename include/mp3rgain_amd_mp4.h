/* mp3rgain_amd_mp4.h -- C ABI of the MP4/M4A ReplayGain tag path (second half):
 * iTunes freeform ("----", mean = com.apple.iTunes) metadata atoms carrying replaygain_track_gain / _peak and
 * replaygain_album_gain / _peak, the on-disk form of a ReplayGain result for AAC files (the reference cannot
 * change AAC samples losslessly, so it writes tags: src/main.rs:2172-2215).
 *
 * Host-only byte work behind the reference's function names.  Citations are file:line under /root/reference.
 *
 *   rg_mp4_tags_set_track / _set_album   ReplayGainTags::set_track / set_album     src/mp4meta.rs:126-134
 *                                        ("{:+.2} dB" and "{:.6}" formats)
 *   rg_mp4_read_replaygain_tags          read_replaygain_tags                      src/mp4meta.rs:333-417
 *   rg_mp4_write_replaygain_tags         write_replaygain_tags                     src/mp4meta.rs:420-430
 *   rg_mp4_delete_replaygain_tags        delete_replaygain_tags                    src/mp4meta.rs:866-869
 *   rg_mp4_is_mp4_file                   is_mp4_file                               src/mp4meta.rs:872-889
 *   *_data variants                      the in-memory cores: update_mp4_metadata (:433-531), create_ilst_box
 *                                        (:621-675), update_chunk_offsets (:750-863), parse/serialize_freeform_tag
 *                                        (:236-330)
 *
 * Return value: 0 (or a byte count where stated) on success, negative rg_mp4_status otherwise;
 * rg_mp4_last_error() holds the anyhow-style message (thread-local).
 */
#ifndef MP3RGAIN_AMD_MP4_H
#define MP3RGAIN_AMD_MP4_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rg_mp4_status {
    RG_MP4_ERR_IO = -201,      /* "Failed to read: ..." / "Failed to write: ..."        src/mp4meta.rs:335,427 */
    RG_MP4_ERR_NO_MOOV = -202, /* "No moov box found in MP4 file"                        src/mp4meta.rs:436 */
    RG_MP4_ERR_ARG = -203,     /* null pointer, or the output buffer is too small (the needed size is returned by the sizing call) */
} rg_mp4_status;

#define RG_MP4_TAG_VALUE_MAX 64

/* ReplayGainTags, src/mp4meta.rs:113-119: four Option<String> */
typedef struct rg_mp4_rg_tags {
    uint8_t has_track_gain, has_track_peak, has_album_gain, has_album_peak;
    uint8_t pad_[4];
    char track_gain[RG_MP4_TAG_VALUE_MAX]; /* e.g. "+3.50 dB" */
    char track_peak[RG_MP4_TAG_VALUE_MAX]; /* e.g. "0.987650" */
    char album_gain[RG_MP4_TAG_VALUE_MAX];
    char album_peak[RG_MP4_TAG_VALUE_MAX];
} rg_mp4_rg_tags;

const char *rg_mp4_last_error(void);

/* ---- ReplayGainTags ------------------------------------------------------------------------------------ */
void rg_mp4_tags_clear(rg_mp4_rg_tags *t);                                  /* ReplayGainTags::new        :122 */
void rg_mp4_tags_set_track(rg_mp4_rg_tags *t, double gain_db, double peak); /* :126-129 */
void rg_mp4_tags_set_album(rg_mp4_rg_tags *t, double gain_db, double peak); /* :131-134 */
int rg_mp4_tags_is_empty(const rg_mp4_rg_tags *t);                          /* :136-141 */

/* ---- freeform atoms (unit-tested by the reference, src/mp4meta.rs:895-913) -------------------------------- */
/* whole "----" box for (namespace, name, value); returns its size; writes it when out != NULL and cap suffices */
size_t rg_mp4_serialize_freeform(const char *ns, const char *name, const char *value, uint8_t *out, size_t cap);
/* `data` = the CONTENT of a "----" box (after its 8-byte header); 1 when mean, name and data were all found */
int rg_mp4_parse_freeform(const uint8_t *data, size_t len, char *ns, size_t ns_cap, char *name, size_t name_cap,
                          char *value, size_t value_cap);

/* ---- in-memory cores ------------------------------------------------------------------------------------- */
int rg_mp4_read_replaygain_tags_data(const uint8_t *data, size_t len, rg_mp4_rg_tags *out);
/* the rewritten file: returns its length (call with out == NULL to size it), or a negative status */
int64_t rg_mp4_update_metadata_data(const uint8_t *data, size_t len, const rg_mp4_rg_tags *tags, uint8_t *out,
                                    size_t out_cap);
int rg_mp4_is_mp4_data(const uint8_t *data, size_t len);

/* ---- file level, the reference's public functions ----------------------------------------------------------- */
int rg_mp4_read_replaygain_tags(const char *path, rg_mp4_rg_tags *out);
int rg_mp4_write_replaygain_tags(const char *path, const rg_mp4_rg_tags *tags);
int rg_mp4_delete_replaygain_tags(const char *path);
int rg_mp4_is_mp4_file(const char *path); /* 1 / 0; unreadable files are 0 like the reference */

#ifdef __cplusplus
}
#endif
#endif /* MP3RGAIN_AMD_MP4_H */
