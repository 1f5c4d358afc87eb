/* mp3rgain_amd_node.h -- the file-level entry points over ALL GPUs of one node, in ONE process, behind the
 * reference's own signatures.
 *
 * The reference's album analysis is one blocking call in one process:
 *     analyze_album(&[&Path]) / analyze_album_with_index(&[&Path], Option<usize>)   src/replaygain.rs:1033-1074
 *     caller: src/main.rs:1317 (`-a`)
 * and its `-r` mode is a loop of analyze_track over the files (src/main.rs:1937-2001).  mp3rgain_amd.h's rg_analyze_album
 * / rg_analyze_tracks do that on ONE GPU.  A node (rg_node) owns one context per GPU of the machine and host threads to
 * drive them; the calls below have the same shape -- paths in, results in input order out -- and
 *   * deal the files out by size (longest first, each to the device with the least work so far: rg_node_partition),
 *   * run every device's share through its own context concurrently (file read, device MP3 decode, analysis),
 *   * album mode: agree that nobody failed (the first failing file IN INPUT ORDER ends the album, src/replaygain.rs:1055,
 *     whichever device it was on), then merge the devices' [histogram | peak] packs -- LoudnessHistogram::accumulate
 *     (:658-662) and album_peak.max (:1056) across GPUs -- and read the 95th percentile of the merged histogram,
 *   * put the per-file results back in input order (track_results.push order, :1061).
 * With one device a node call returns the bits of the corresponding single-context call.
 *
 * The merge is 48 KB per device and latency-bound either way; two forms:
 *   RG_NODE_EXCHANGE_HOST (default)  every device's pack comes to the host (one 48 KB D2H each), the host adds the bins;
 *   RG_NODE_EXCHANGE_RCCL            one communicator per device from ncclCommInitAll, ncclAllGather of the packs over
 *                                    xGMI on each device's batch stream + the device fold (rg_album_exchange), every
 *                                    device runs the percentile, device 0's answer is returned.
 *
 * Plain C types only.  No CPU fallback: rg_node_create fails when no gfx950 device is usable.
 */
#ifndef MP3RGAIN_AMD_NODE_H
#define MP3RGAIN_AMD_NODE_H

#include "mp3rgain_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rg_node rg_node;

enum { RG_NODE_EXCHANGE_HOST = 0, RG_NODE_EXCHANGE_RCCL = 1 };

/* devices: HIP ordinals; NULL (n ignored) = every visible device.  NULL on failure: rg_node_last_error(NULL). */
rg_node *rg_node_create(const int *devices, size_t n);
void rg_node_destroy(rg_node *node);
const char *rg_node_last_error(const rg_node *node);
size_t rg_node_devices(const rg_node *node);
/* the context of the node's i-th device, for rg_set_tuning / rg_set_kernel / rg_set_decoder_command (NULL when the node
 * runs on a caller-supplied backend) */
rg_ctx *rg_node_ctx(rg_node *node, size_t i);
/* RG_NODE_EXCHANGE_*; RCCL builds the communicators now (RG_ERR_COLLECTIVE when librccl.so cannot be resolved) */
int rg_node_set_exchange(rg_node *node, int mode);

/* Pure host helper, the rule every call here uses: item i of weight sizes[i] goes to device owner_out[i]; items are
 * taken heaviest first (ties: lower index first), each to the device with the least weight so far (ties: lower device).
 * (mp3rgain_amd/album.py shard_indices(frames=...) is the same rule for torchrun-launched ranks.) */
void rg_node_partition(const uint64_t *sizes, size_t n, size_t world, uint32_t *owner_out);

/* analyze_album_with_index (src/replaygain.rs:1044-1074) over all devices.  tracks_out[n] in input order. */
int rg_analyze_album_node(rg_node *node, const char *const *paths, size_t n, int32_t track_index,
                          rg_track_result *tracks_out, rg_album_result *album_out);
/* `-r` over all devices (rg_analyze_tracks per device, replicas only: no exchange); status_out[i] / rg_node_tracks_error
 * per file, input order */
int rg_analyze_tracks_node(rg_node *node, const char *const *paths, size_t n, int32_t track_index,
                           rg_track_result *out, int32_t *status_out);
const char *rg_node_tracks_error(const rg_node *node, size_t i);
/* how the last call dealt the files out: owner_out[i] = index of the device that had file i (n = that call's n) */
int rg_node_last_partition(const rg_node *node, uint32_t *owner_out, size_t n);

/* ---- engines ------------------------------------------------------------------------------------------------------
 * What a node needs from one device, as a table of functions.  The built-in table drives an rg_ctx
 * (rg_analyze_album_begin / rg_album_finish / rg_analyze_tracks); a host that brings its own per-device engine -- or
 * a test that wants to see the dealing, the ordering and the abort rule without a GPU -- passes another one. */
typedef struct rg_node_backend {
    void *(*open)(int device, void *user);                  /* NULL = failure */
    void (*close)(void *engine, void *user);
    /* album of these files up to the percentile: per-file results, *failed_index on error (index into `paths`) */
    int (*album_begin)(void *engine, const char *const *paths, size_t n, int32_t track_index, rg_track_result *out,
                       size_t *failed_index, void *user);
    /* this engine's [histogram u32 x 12000 | peak f64] of the album_begin before: RG_ALBUM_PACK_WORDS words */
    int (*album_pack)(void *engine, uint32_t *pack_out, void *user);
    int (*tracks)(void *engine, const char *const *paths, size_t n, int32_t track_index, rg_track_result *out,
                  int32_t *status_out, void *user);
    const char *(*tracks_error)(void *engine, size_t i, void *user);
    const char *(*last_error)(void *engine, void *user);
    void *user;
} rg_node_backend;
/* TEST SEAM, unsupported in production: refused (NULL, see rg_node_create_error) unless MP3RGAIN_AMD_TEST_SEAMS=1. */
rg_node *rg_node_create_backend(const rg_node_backend *backend, const int *devices, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* MP3RGAIN_AMD_NODE_H */
