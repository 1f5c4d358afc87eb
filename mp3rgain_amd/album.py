"""Album mode across GPUs: analyze_album_with_index (src/replaygain.rs:1044-1074) as a sharded job.

The reference walks the album's files one after the other, adds every track's histogram into the
album histogram (LoudnessHistogram::accumulate, :658-662) and keeps the maximum peak (:1056).  Tracks
are independent, so here rank r of `world` owns tracks r, r + world, r + 2*world, ... ; each rank's
context merges its own tracks on the GPU (rg_enqueue_pcm_batch(album=1)) and the only exchange is one
all-reduce(sum) of the 12 000-bin histogram and one all-reduce(max) of the peak -- 48 KB + 8 B per
rank, latency-bound on xGMI.  Bins are u32 in the reference; they travel as int32 (two's-complement
addition is the same bit pattern).  This module is plumbing over torch.distributed (backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests); the percentile itself is rg_hist_loudness / the GPU kernel.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from . import _capi


def shard_indices(n_tracks: int, world: int, rank: int, frames: Optional[Sequence[int]] = None) -> List[int]:
    """Tracks of the album that `rank` analyses, in ascending order.

    Without `frames`: round robin (rank r owns r, r + world, ...), which is balanced for like-sized tracks and is
    the split BASELINE configs[3] names.  With `frames` (one length per track): balanced by cumulative frames --
    longest track first, each to the rank with the least work so far (ties: the lower rank), so that a few long
    tracks in an album of short ones do not make one GPU the straggler.  Deterministic: every rank computes the
    same partition from the same list."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    if frames is None:
        return list(range(rank, n_tracks, world))
    if len(frames) != n_tracks:
        raise ValueError("frames must hold one length per track")
    load = [0] * world
    owner = [0] * n_tracks
    for t in sorted(range(n_tracks), key=lambda i: (-int(frames[i]), i)):
        r = min(range(world), key=lambda q: (load[q], q))
        owner[t] = r
        load[r] += int(frames[t])
    return [t for t in range(n_tracks) if owner[t] == rank]


def allreduce_album(hist_i32, peak_f64, group=None, even_if_alone: bool = False) -> None:
    """In-place: histogram bins summed, peak maximised over the ranks of `group`.

    hist_i32: int32 tensor [12000] (a view of the context's d_album_hist, or a CPU tensor in tests);
    peak_f64: float64 tensor [1]."""
    import torch.distributed as dist

    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not even_if_alone):
        return
    dist.all_reduce(hist_i32, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(peak_f64, op=dist.ReduceOp.MAX, group=group)


ALBUM_PACK_WORDS = _capi.HISTOGRAM_SIZE + 2  # 12000 bins + the f64 peak, contiguous in the context (RG_ALBUM_PACK_WORDS)


def allgather_album(pack_i32, gathered_i32, group=None) -> None:
    """One collective instead of two: every rank's [histogram | peak] pack (int32[12002]) into
    gathered_i32 (int32[world * 12002]); fold with Analyzer.album_reduce_gathered / fold_gathered."""
    import torch.distributed as dist

    dist.all_gather_into_tensor(gathered_i32, pack_i32, group=group)


def fold_gathered(gathered_u32: np.ndarray, world: int):
    """Host restatement of rg_album_reduce_gathered_kernel (tests): -> (hist uint32[12000], peak)."""
    g = np.ascontiguousarray(gathered_u32, dtype=np.uint32).reshape(world, ALBUM_PACK_WORDS)
    hist = g[:, :_capi.HISTOGRAM_SIZE].sum(axis=0, dtype=np.uint64).astype(np.uint32)
    peaks = g[:, _capi.HISTOGRAM_SIZE:].copy().view(np.float64).reshape(world)
    return hist, float(peaks.max()) if world else 0.0


def album_result_from_hist(hist_u32: np.ndarray, peak: float) -> dict:
    """The tail of analyze_album (src/replaygain.rs:1064-1073) on a merged histogram, host side."""
    lib = _capi.load()
    h = np.ascontiguousarray(hist_u32, dtype=np.uint32)
    if h.shape != (_capi.HISTOGRAM_SIZE,):
        raise ValueError("histogram must have 12000 bins")
    loud = lib.rg_hist_loudness(h.ctypes.data)
    gain = lib.rg_gain_from_loudness(loud)
    return {"album_loudness_db": loud, "album_gain_db": gain, "album_peak": float(peak),
            "album_gain_steps": lib.rg_gain_steps(gain), "windows": int(h.sum(dtype=np.uint64))}


def gather_track_results(local: Sequence, n_tracks: int, group=None, frames: Optional[Sequence[int]] = None) -> Optional[list]:
    """Per-track results back in input order (track_results.push order, src/replaygain.rs:1061):
    every rank contributes the results of shard_indices(n_tracks, world, rank); all ranks get the list."""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(local)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    buckets = [None] * world
    dist.all_gather_object(buckets, list(local), group=group)
    out = [None] * n_tracks
    for r in range(world):
        for k, t in enumerate(shard_indices(n_tracks, world, r, frames)):
            out[t] = buckets[r][k]
    return out


class AlbumAborted(RuntimeError):
    """An album analysis stopped because some rank could not analyse one of its tracks."""


def abort_if_any_failed(local_error: Optional[BaseException], group=None) -> None:
    """The reference's album loop ends at the first track that fails (`?` at src/replaygain.rs:1055): no album result,
    no per-track results.  Sharded over ranks that has to be a joint decision -- a rank that raised on its own would
    leave the others waiting in the exchange.  Call this between the per-rank analysis and the collective: every rank
    passes its own exception (or None); if any rank failed, every rank raises AlbumAborted carrying the message of the
    failing rank with the lowest number (its tracks come first in input order among equals), and nobody enters the
    collective."""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        if local_error is not None:
            raise AlbumAborted(str(local_error)) from local_error
        return
    msgs = [None] * dist.get_world_size(group)
    dist.all_gather_object(msgs, None if local_error is None else str(local_error), group=group)
    for r, m in enumerate(msgs):
        if m is not None:
            raise AlbumAborted(f"rank {r}: {m}")


def analyze_album_files_sharded(analyzer, files: Sequence, group=None, exchange_even_if_alone: bool = False):
    """analyze_album_with_index (src/replaygain.rs:1044-1074) over files, sharded across the ranks of `group`.

    Every rank passes the same list.  Files are dealt out by size (a proxy for their length: longest first, each to
    the least loaded rank); each rank decodes and analyses its own on its GPU (MP3 on the device decoder), the ranks
    agree that nobody failed, exchange the album histogram / peak over the analyzer's RCCL communicator
    (Analyzer.comm_init_torch must have been called) and every rank returns the whole album: per-file results in input
    order, album loudness / gain / peak from the merged histogram."""
    import os

    import torch.distributed as dist

    from .replaygain import AlbumGainResult, ReplayGainError

    files = [os.fspath(f) for f in files]
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    # ONE view of the sizes decides the partition: rank 0's, broadcast.  (Every rank stat'ing for itself can disagree
    # -- a lagging network filesystem, a file rewritten meanwhile -- and then tracks are analysed twice or not at all
    # and the gathered results no longer line up.)
    box = [None]
    if rank == 0:
        sizes = []
        for f in files:
            try:
                sizes.append(os.path.getsize(f))
            except OSError:
                sizes.append(0)  # the rank that owns it reports "Failed to open"
        box[0] = sizes
    if world > 1:
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    sizes = box[0]
    mine = shard_indices(len(files), world, rank, frames=sizes)
    err, local = None, None
    try:
        local = analyzer.analyze_album_files([files[i] for i in mine])
    except Exception as ex:  # not only ReplayGainError: a rank that raised alone would leave the others in the collective
        err = ex
    abort_if_any_failed(err, group)
    if world > 1 or exchange_even_if_alone:
        analyzer.album_exchange()
        analyzer.album_result_enqueue()
        alb = analyzer.album_finish()
        loud, gain, peak = alb.album_loudness_db, alb.album_gain_db, alb.album_peak
    else:
        loud, gain, peak = local.album_loudness_db, local.album_gain_db, local.album_peak
    tracks = gather_track_results(local.tracks, len(files), group, frames=sizes)
    return AlbumGainResult(tracks, loud, gain, peak)
