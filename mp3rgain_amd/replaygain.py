"""Host-side mirror of mp3rgain's `replaygain` module on decoded PCM, over the C ABI.

Names, fields, argument meaning and error behaviour follow the reference
(src/replaygain.rs, v1.5.0):

  REPLAYGAIN_REFERENCE_DB                          :37
  AudioFileType {Mp3, Aac}                         :48-53
  ReplayGainResult {loudness_db, gain_db, peak, sample_rate, file_type} + gain_steps()   :57-75
  AlbumGainResult {tracks, album_loudness_db, album_gain_db, album_peak} + album_gain_steps()  :79-95
  analyze_track / analyze_album                    :929-941 / :1033-1074
  PeakAmplitudeResult / find_peak_amplitude        :1125-1132 / :1140-1249
  is_available                                     :1119-1121

The reference's functions take a file path and decode with symphonia; the decoder is outside
this path, so the functions here take a `PcmTrack` -- what the
reference's decode loop hands to process_audio_buffer, as whole planar channels.  All
arithmetic happens in libmp3rgain_amd.so on the GPU; this file is plumbing.
"""
from __future__ import annotations

import ctypes as C
import os
import time
import enum
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _capi

REPLAYGAIN_REFERENCE_DB = 89.0
GAIN_STEP_DB = 1.5
SUPPORTED_RATES = (96000, 88200, 64000, 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000)


class ReplayGainError(RuntimeError):
    """The anyhow::Error of the reference: carries the library's message."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


class AudioFileType(enum.IntEnum):
    Mp3 = 0
    Aac = 1


@dataclass
class ReplayGainResult:
    loudness_db: float
    gain_db: float
    peak: float
    sample_rate: int
    file_type: AudioFileType = AudioFileType.Mp3
    windows: int = 0
    flags: int = 0  # RG_TRACK_FLAG_*: 1 = non-finite samples in the track, 2 = a window variant 2 could not resolve

    def gain_steps(self) -> int:
        return _capi.load().rg_gain_steps(self.gain_db)


@dataclass
class AlbumGainResult:
    tracks: List[ReplayGainResult]
    album_loudness_db: float
    album_gain_db: float
    album_peak: float

    def album_gain_steps(self) -> int:
        return _capi.load().rg_gain_steps(self.album_gain_db)


@dataclass
class PeakAmplitudeResult:
    peak: float
    peak_pcm: float
    sample_rate: int


_NP_FMT = {
    np.dtype(np.float32): _capi.FMT_F32_PLANAR,
    np.dtype(np.int16): _capi.FMT_S16_PLANAR,
    np.dtype(np.int32): _capi.FMT_S32_PLANAR,
}


@dataclass
class PcmTrack:
    """Decoded audio of one file: planar channels (float32 in [-1,1], int16 or int32)."""

    channels: Sequence[np.ndarray]
    sample_rate: int
    file_type: AudioFileType = AudioFileType.Mp3
    _fmt: int = field(init=False, default=0)

    def __post_init__(self):
        if len(self.channels) == 0:
            raise ValueError("No audio track found")  # src/replaygain.rs:834-836
        chans = [np.ascontiguousarray(c) for c in self.channels]
        dt = chans[0].dtype
        if dt not in _NP_FMT:
            raise TypeError(f"unsupported sample dtype {dt}")
        for c in chans:
            if c.dtype != dt or c.shape != chans[0].shape or c.ndim != 1:
                raise ValueError("channels must be 1-D arrays of one dtype and length")
        self.channels = chans
        self._fmt = _NP_FMT[dt]

    @property
    def frames(self) -> int:
        return int(self.channels[0].shape[0])


def is_available() -> bool:
    """replaygain::is_available (src/replaygain.rs:1119-1121): the library is present."""
    try:
        return bool(_capi.load().rg_is_available())
    except (ImportError, OSError):
        return False


def db_to_steps(db: float) -> int:
    return _capi.load().rg_db_to_steps(db)


def steps_to_db(steps: int) -> float:
    return _capi.load().rg_steps_to_db(steps)


def clip_limit_steps(steps: int, gain_db: float, peak: float, prevent_clipping: bool, wrap_gain: bool = False) -> int:
    """The -k rule of the CLI (src/main.rs:2033-2058)."""
    return _capi.load().rg_clip_limit_steps(steps, gain_db, peak, int(prevent_clipping), int(wrap_gain))


def pack_tracks(tracks: Sequence[PcmTrack]):
    """Lay the tracks out in one planar arena -> (uint8 arena, TrackDesc array).

    Track t, channel c lives at offset_bytes + c * frames * itemsize; each track starts on a
    16-byte boundary.  All channels are packed so that find_peak_amplitude sees every channel;
    the analysis itself reads channels 0 and 1 only (src/replaygain.rs:971).
    """
    descs = (_capi.TrackDesc * max(1, len(tracks)))()
    sizes, off = [], 0
    for t in tracks:
        nbytes = sum(c.nbytes for c in t.channels)
        sizes.append((off, nbytes))
        off = (off + nbytes + 15) & ~15
    arena = np.zeros(max(off, 16), dtype=np.uint8)
    for i, t in enumerate(tracks):
        o, _ = sizes[i]
        p = o
        for c in t.channels:
            arena[p:p + c.nbytes] = c.view(np.uint8)
            p += c.nbytes
        descs[i].offset_bytes = o
        descs[i].frames = t.frames
        descs[i].sample_rate = t.sample_rate
        descs[i].channels = len(t.channels)
        descs[i].format = t._fmt
    return arena, descs


class Analyzer:
    """One rg_ctx (one GPU).  Not thread-safe, like the single-threaded reference."""

    def __init__(self, device: int = 0):
        self._lib = _capi.load()
        self._ctx = self._lib.rg_create(device)
        if not self._ctx:
            msg = self._lib.rg_last_error(None).decode()
            raise ReplayGainError(_capi.RG_ERR_NO_DEVICE, msg)
        self.device = device

    # -- lifecycle ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            if not getattr(self, "_borrowed", False):  # a Node's context (Node.analyzer) dies with the node
                self._lib.rg_destroy(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != _capi.RG_OK:
            raise ReplayGainError(rc, self._lib.rg_last_error(self._ctx).decode())

    @property
    def handle(self):
        return self._ctx

    def set_stream(self, hip_stream: Optional[int]):
        """Attach a HIP stream handle (0 = the default stream, e.g. torch.cuda.current_stream().cuda_stream);
        None detaches."""
        if hip_stream is None:
            self._check(self._lib.rg_set_stream(self._ctx, None, 0))
        else:
            self._check(self._lib.rg_set_stream(self._ctx, hip_stream or None, 1))

    def batch_stream(self) -> int:
        """HIP stream handle of the most recent enqueue (rg_batch_stream): issue the album collective on it."""
        return int(self._lib.rg_batch_stream(self._ctx) or 0)

    def wait_user_stream(self):
        """Order the next enqueue behind what the attached stream has been given so far."""
        self._check(self._lib.rg_wait_user_stream(self._ctx))

    def set_kernel(self, variant: int):
        self._check(self._lib.rg_set_kernel(self._ctx, variant))

    def set_tuning(self, key: int, value: int):
        """key 1: segment length of variant 2 (frames); key 2: lane target; 0 = default."""
        self._check(self._lib.rg_set_tuning(self._ctx, key, value))

    # -- synchronous, host PCM ----------------------------------------------------------------
    def analyze_tracks(self, tracks: Sequence[PcmTrack], return_histograms: bool = False):
        """`-r` mode: analyze_track for each track (src/replaygain.rs:929-941)."""
        n = len(tracks)
        arena, descs = pack_tracks(tracks)
        out = (_capi.TrackResult * max(1, n))()
        hist = np.zeros((max(1, n), _capi.HISTOGRAM_SIZE), dtype=np.uint32) if return_histograms else None
        self._check(self._lib.rg_analyze_pcm_batch(
            self._ctx, descs, n, arena.ctypes.data, arena.nbytes, 0, out,
            hist.ctypes.data if hist is not None else None))
        res = [_to_result(out[i], tracks[i].file_type) for i in range(n)]
        return (res, hist[:n]) if return_histograms else res

    def analyze_track(self, track: PcmTrack) -> ReplayGainResult:
        return self.analyze_tracks([track])[0]

    def analyze_album(self, tracks: Sequence[PcmTrack], return_histogram: bool = False):
        """analyze_album (src/replaygain.rs:1033-1074) on one GPU."""
        n = len(tracks)
        arena, descs = pack_tracks(tracks)
        out = (_capi.TrackResult * max(1, n))()
        alb = _capi.AlbumResult()
        hist = np.zeros(_capi.HISTOGRAM_SIZE, dtype=np.uint32) if return_histogram else None
        self._check(self._lib.rg_analyze_album_pcm(
            self._ctx, descs, n, arena.ctypes.data, arena.nbytes, 0, out, C.byref(alb),
            hist.ctypes.data if hist is not None else None))
        res = AlbumGainResult([_to_result(out[i], tracks[i].file_type) for i in range(n)],
                              alb.album_loudness_db, alb.album_gain_db, alb.album_peak)
        return (res, hist) if return_histogram else res

    def find_peak_amplitude(self, track: PcmTrack) -> PeakAmplitudeResult:
        """find_peak_amplitude's scan over ALL channels (src/replaygain.rs:1210-1249)."""
        arena, descs = pack_tracks([track])
        pk = _capi.PeakResult()
        self._check(self._lib.rg_find_peak_pcm(self._ctx, descs, arena.ctypes.data, arena.nbytes, 0, C.byref(pk)))
        return PeakAmplitudeResult(pk.peak, pk.peak_pcm, pk.sample_rate)

    # -- file level: the reference's public functions (src/replaygain.rs:929-941, 1033-1074, 1140-1249) ----
    def set_decoder_command(self, command_template: Optional[str]):
        """Command (run by /bin/sh, `{}` = quoted path) that writes a WAV stream to stdout for files that are
        not RIFF/WAVE, e.g. "ffmpeg -v error -i {} -f wav -c:a pcm_f32le -".  None = WAV files only."""
        self._check(self._lib.rg_set_decoder_command(self._ctx, command_template.encode() if command_template else None))

    def analyze_track_file(self, file_path, track_index: Optional[int] = None) -> ReplayGainResult:
        """analyze_track_with_index (src/replaygain.rs:935-941)."""
        out = _capi.TrackResult()
        self._check(self._lib.rg_analyze_track(self._ctx, os.fsencode(os.fspath(file_path)),
                                               -1 if track_index is None else int(track_index), C.byref(out)))
        return _to_result(out, out.file_type)

    def analyze_album_files(self, files, track_index: Optional[int] = None, timing: Optional[dict] = None) -> AlbumGainResult:
        """analyze_album_with_index (src/replaygain.rs:1044-1074).  `timing` (measurement tools): receives `c_call_seconds`, the
        duration of the C call rg_analyze_album alone -- what a caller over the C ABI waits for; building the path array and the
        result objects of this wrapper costs a few microseconds per file on top."""
        n = len(files)
        paths = (C.c_char_p * max(1, n))(*[os.fsencode(os.fspath(f)) for f in files])
        out = (_capi.TrackResult * max(1, n))()
        alb = _capi.AlbumResult()
        t0 = time.perf_counter()
        rc = self._lib.rg_analyze_album(self._ctx, paths, n, -1 if track_index is None else int(track_index), out, C.byref(alb))
        if timing is not None:
            timing["c_call_seconds"] = time.perf_counter() - t0
        self._check(rc)
        return AlbumGainResult([_to_result(out[i], out[i].file_type) for i in range(n)], alb.album_loudness_db,
                               alb.album_gain_db, alb.album_peak)

    def analyze_track_files(self, files, track_index: Optional[int] = None, timing: Optional[dict] = None) -> list:
        """analyze_track for every file, as ONE GPU batch (files loaded on all host cores).  -> per file a
        ReplayGainResult, or the ReplayGainError analyze_track_file would have raised for it.  `timing`: as analyze_album_files."""
        n = len(files)
        paths = (C.c_char_p * max(1, n))(*[os.fsencode(os.fspath(f)) for f in files])
        out = (_capi.TrackResult * max(1, n))()
        status = (C.c_int32 * max(1, n))()
        t0 = time.perf_counter()
        rc = self._lib.rg_analyze_tracks(self._ctx, paths, n, -1 if track_index is None else int(track_index), out, status)
        if timing is not None:
            timing["c_call_seconds"] = time.perf_counter() - t0
        self._check(rc)
        res = []
        for i in range(n):
            if status[i] == 0:
                res.append(_to_result(out[i], out[i].file_type))
            else:
                res.append(ReplayGainError(int(status[i]), self._lib.rg_tracks_error(self._ctx, i).decode("utf-8", "replace")))
        return res

    def find_peak_amplitude_file(self, file_path) -> PeakAmplitudeResult:
        pk = _capi.PeakResult()
        self._check(self._lib.rg_find_peak_amplitude(self._ctx, os.fsencode(os.fspath(file_path)), C.byref(pk)))
        return PeakAmplitudeResult(pk.peak, pk.peak_pcm, pk.sample_rate)

    def decode_mp3_device(self, data: bytes):
        """The split MP3 decoder (stage A on the host, stages B-E on this GPU) -> (float32 [channels][frames], StreamInfo);
        bit for bit what mp3dec.decode returns."""
        from . import mp3dec

        info = mp3dec.scan(data)
        cap = int(info.frames)
        out = np.zeros((int(info.channels), max(1, cap)), dtype=np.float32)
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        di = mp3dec.StreamInfo()
        self._check(self._lib.rg_mp3_decode_device(self._ctx, C.cast(buf, C.c_void_p), len(data), out[0].ctypes.data,
                                                   out[1].ctypes.data if info.channels == 2 else None, cap, C.byref(di)))
        return out[:, :int(di.frames)], di

    def decode_mp3_bench(self, data: bytes, copies: int, reps: int = 5) -> dict:
        """rg_mp3_decode_bench: per-kernel HIP-event times of the device decode chain on `copies` copies of one stream."""
        ms = (C.c_double * 5)()
        units, cbytes, frames = C.c_uint64(), C.c_uint64(), C.c_uint64()
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        self._check(self._lib.rg_mp3_decode_bench(self._ctx, buf, len(data), copies, reps, ms, C.byref(units), C.byref(cbytes), C.byref(frames)))
        # chain_pipelined: per chunk in the file route's arrangement (frame parser and lane sort beside the chunk before)
        return {"ms": {"frames": ms[0], "huffman": ms[1], "backhalf": ms[2], "chain": ms[3], "chain_pipelined": ms[4]},
                "units": units.value, "compressed_bytes": cbytes.value, "frames": frames.value}

    def analyze_wav_bytes(self, wavs: Sequence[bytes], album: bool = False):
        """WAV streams already in memory -> [ReplayGainResult] (+ AlbumGainResult fields when album)."""
        n = len(wavs)
        keep = [C.c_char_p(bytes(w) if not isinstance(w, bytes) else w) for w in wavs]  # no copy: points into the bytes objects
        ptrs = (C.c_void_p * max(1, n))(*[C.cast(k, C.c_void_p).value for k in keep])
        lens = (C.c_size_t * max(1, n))(*[len(w) for w in wavs])
        out = (_capi.TrackResult * max(1, n))()
        alb = _capi.AlbumResult()
        self._check(self._lib.rg_analyze_wav_batch(self._ctx, ptrs, lens, n, int(album), out, C.byref(alb)))
        res = [_to_result(out[i], AudioFileType.Mp3) for i in range(n)]
        return AlbumGainResult(res, alb.album_loudness_db, alb.album_gain_db, alb.album_peak) if album else res

    # -- synchronous, PCM already resident on the device (one blocking call = the reference's API shape) ----
    def analyze_device(self, descs, n: int, d_pcm_base: int, pcm_bytes: int, want_hist: bool = False, out=None):
        """rg_analyze_pcm_batch with pcm_on_device = 1: ONE batch in flight, results (exact repeat of flagged tracks
        included) when the call returns.  With `out` (a ctypes TrackResult array) the raw records are left there and
        nothing is converted (timing loops)."""
        if out is not None:
            self._check(self._lib.rg_analyze_pcm_batch(self._ctx, descs, n, d_pcm_base, pcm_bytes, 1, out, None))
            return out
        out = (_capi.TrackResult * max(1, n))()
        hist = np.zeros((max(1, n), _capi.HISTOGRAM_SIZE), dtype=np.uint32) if want_hist else None
        self._check(self._lib.rg_analyze_pcm_batch(self._ctx, descs, n, d_pcm_base, pcm_bytes, 1, out,
                                                   hist.ctypes.data if hist is not None else None))
        res = [_to_result(out[i], AudioFileType.Mp3) for i in range(n)]
        return (res, hist[:n]) if want_hist else res

    # -- device-resident pipeline --------------------------------------------------------------
    def enqueue_device(self, descs, n: int, d_pcm_base: int, pcm_bytes: int, album: bool = False):
        self._check(self._lib.rg_enqueue_pcm_batch(self._ctx, descs, n, d_pcm_base, pcm_bytes, int(album)))

    def collect(self, n: int, want_hist: bool = False):
        out = (_capi.TrackResult * max(1, n))()
        hist = np.zeros((max(1, n), _capi.HISTOGRAM_SIZE), dtype=np.uint32) if want_hist else None
        self._check(self._lib.rg_collect(self._ctx, out, hist.ctypes.data if hist is not None else None))
        res = [_to_result(out[i], AudioFileType.Mp3) for i in range(n)]
        return (res, hist[:n]) if want_hist else res

    def collect_exact(self, descs, n: int, d_pcm_base: int, pcm_bytes: int, want_hist: bool = False):
        """collect() with the synchronous calls' guarantee: a batch that has a track flagged imprecise is run once more with
        that track on the order-faithful kernel (the PCM must still be in place)."""
        out = (_capi.TrackResult * max(1, n))()
        hist = np.zeros((max(1, n), _capi.HISTOGRAM_SIZE), dtype=np.uint32) if want_hist else None
        self._check(self._lib.rg_collect_exact(self._ctx, descs, n, d_pcm_base, pcm_bytes, out,
                                               hist.ctypes.data if hist is not None else None))
        res = [_to_result(out[i], AudioFileType.Mp3) for i in range(n)]
        return (res, hist[:n]) if want_hist else res

    def device_view(self) -> _capi.DeviceView:
        v = _capi.DeviceView()
        self._check(self._lib.rg_device_view_get(self._ctx, C.byref(v)))
        return v

    def album_allreduce(self, nccl_comm: Optional[int] = None):
        self._check(self._lib.rg_album_allreduce(self._ctx, nccl_comm))

    def album_reduce_gathered(self, d_gathered: int, world: int):
        self._check(self._lib.rg_album_reduce_gathered(self._ctx, d_gathered, world))

    # -- the library's own RCCL communicator (rg_comm_*): the exchange runs on the batch's stream -----------
    def comm_init(self, unique_id: bytes, world: int, rank: int):
        self._check(self._lib.rg_comm_init(self._ctx, unique_id, world, rank))

    def comm_init_torch(self, group=None, library: Optional[str] = None):
        """Bootstrap through torch.distributed: rank 0's ncclUniqueId is broadcast over the (already
        initialised) process group, then every rank joins.  torch's own librccl.so is the one used unless `library`
        (or the environment's MP3RGAIN_AMD_RCCL_LIBRARY) names another -- the tests' stand-in transport, which lets
        several ranks share one GPU (tests/standin_rccl).
        Every rank raises, or none does: failures are agreed on over the process group."""
        import torch
        import torch.distributed as dist

        lib = library or os.environ.get("MP3RGAIN_AMD_RCCL_LIBRARY") or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        explicit = bool(library or os.environ.get("MP3RGAIN_AMD_RCCL_LIBRARY"))
        # a library named explicitly must exist and load: silently going on with whatever librccl.so resolves instead (real
        # RCCL where the stand-in was meant refuses two ranks on one device) would fail on some ranks only
        if os.path.exists(lib):
            lib_rc = self._lib.rg_comm_library(os.fsencode(lib))
        else:
            lib_rc = -1 if explicit else 0
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        uid = b""
        if rank == 0:
            buf = C.create_string_buffer(_capi.COMM_ID_BYTES)
            if self._lib.rg_comm_unique_id(buf) == 0:
                uid = buf.raw
        box = [uid]
        dist.broadcast_object_list(box, src=src, group=group)
        if not box[0]:
            raise ReplayGainError(-7, "ncclGetUniqueId failed on rank 0 (librccl.so not found?)")
        err = None
        if explicit and lib_rc != 0:  # a library that was asked for by name and cannot be loaded: agreed on below, like any other failure
            err = ReplayGainError(int(lib_rc), f"refused {lib}: not a librccl.so[.N] (a stand-in transport needs MP3RGAIN_AMD_TEST_SEAMS=1)"
                                  if int(lib_rc) == -10 else f"cannot load {lib}")
        else:
            try:
                self.comm_init(box[0], world, rank)
            except ReplayGainError as ex:
                err = ex
        flags = [None] * world
        dist.all_gather_object(flags, err is None, group=group)
        if not all(flags):
            self.comm_destroy()
            raise err or ReplayGainError(-7, "ncclCommInitRank failed on another rank")

    def comm_info(self) -> dict:
        """{"ranks": ranks of the context's communicator (0 = none), "nccl_version": ncclGetVersion of the library behind it}"""
        w, v = C.c_int(0), C.c_int(0)
        self._check(self._lib.rg_comm_info(self._ctx, C.byref(w), C.byref(v)))
        return {"ranks": int(w.value), "nccl_version": int(v.value)}

    def comm_destroy(self):
        self._check(self._lib.rg_comm_destroy(self._ctx))

    def album_exchange(self):
        """Sum of histograms / max of peaks across the communicator's ranks, on the batch's stream."""
        self._check(self._lib.rg_album_exchange(self._ctx))

    def album_result_enqueue(self):
        self._check(self._lib.rg_album_result_enqueue(self._ctx))

    def album_finish(self, want_hist: bool = False):
        alb = _capi.AlbumResult()
        hist = np.zeros(_capi.HISTOGRAM_SIZE, dtype=np.uint32) if want_hist else None
        self._check(self._lib.rg_album_finish(self._ctx, C.byref(alb), hist.ctypes.data if hist is not None else None))
        return (alb, hist) if want_hist else alb

    def synth_fill_device(self, d_dst: int, seed: int, channel: int, sample_rate: int, first_frame: int, frames: int):
        self._check(self._lib.rg_synth_fill_device(self._ctx, d_dst, seed, channel, sample_rate, first_frame, frames))

    def timing_enable(self, on: bool = True):
        self._check(self._lib.rg_timing_enable(self._ctx, int(on)))

    def timing_read(self, reset: bool = True):
        """-> (sum of kernel durations [ms], launches, first-start-to-last-end span [ms])"""
        s, k, sp = C.c_double(), C.c_uint64(), C.c_double()
        self._check(self._lib.rg_timing_read(self._ctx, C.byref(s), C.byref(k), C.byref(sp), int(reset)))
        return s.value, k.value, sp.value


class Node:
    """All GPUs of this machine behind the reference's file-level signatures, in one process
    (include/mp3rgain_amd_node.h): one context and one host thread per device, files dealt out by size, the album's
    histogram merged across devices.  `devices` = HIP ordinals, None = every visible device."""

    EXCHANGE_HOST, EXCHANGE_RCCL = 0, 1

    def __init__(self, devices: Optional[Sequence[int]] = None, _backend=None):
        self._lib = _capi.load()
        self._node = None
        if _backend is not None:  # tests: a table of per-device functions instead of rg_ctx
            devs = (C.c_int * len(devices))(*devices)
            self._backend = _backend
            self._node = self._lib.rg_node_create_backend(C.addressof(_backend), devs, len(devices))
        elif devices is None:
            self._node = self._lib.rg_node_create(None, 0)
        else:
            devs = (C.c_int * max(1, len(devices)))(*devices)
            self._node = self._lib.rg_node_create(devs, len(devices))
        if not self._node:
            raise ReplayGainError(-4, self._lib.rg_node_last_error(None).decode("utf-8", "replace"))

    def close(self):
        if self._node:
            self._lib.rg_node_destroy(self._node)
            self._node = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise ReplayGainError(rc, self._lib.rg_node_last_error(self._node).decode("utf-8", "replace"))

    @property
    def devices(self) -> int:
        return int(self._lib.rg_node_devices(self._node))

    def set_exchange(self, mode: int):
        self._check(self._lib.rg_node_set_exchange(self._node, int(mode)))

    def analyzer(self, i: int) -> "Analyzer":
        """The i-th device's context as an Analyzer, for the PCM-level and asynchronous entry points (bench.py's one-process
        mode drives one from a host thread per device).  Owned by the node: closing it releases nothing."""
        ctx = self._lib.rg_node_ctx(self._node, int(i))
        if not ctx:
            raise ReplayGainError(-1, f"node has no context {i}")
        a = Analyzer.__new__(Analyzer)
        a._lib = self._lib
        a._ctx = ctx
        a._borrowed = True
        a.device = int(i)
        return a

    def set_tuning(self, key: int, value: int):
        """rg_set_tuning on every device's context."""
        for i in range(self.devices):
            ctx = self._lib.rg_node_ctx(self._node, i)
            if ctx and self._lib.rg_set_tuning(ctx, int(key), int(value)) != 0:
                raise ReplayGainError(-1, self._lib.rg_last_error(ctx).decode("utf-8", "replace"))

    def set_decoder_command(self, command_template: Optional[str]):
        for i in range(self.devices):
            ctx = self._lib.rg_node_ctx(self._node, i)
            if ctx:
                self._lib.rg_set_decoder_command(ctx, command_template.encode() if command_template else None)

    def analyze_album_files(self, files, track_index: Optional[int] = None) -> AlbumGainResult:
        """analyze_album_with_index (src/replaygain.rs:1044-1074) over all devices."""
        n = len(files)
        paths = (C.c_char_p * max(1, n))(*[os.fsencode(os.fspath(f)) for f in files])
        out = (_capi.TrackResult * max(1, n))()
        alb = _capi.AlbumResult()
        self._check(self._lib.rg_analyze_album_node(self._node, paths, n, -1 if track_index is None else int(track_index), out, C.byref(alb)))
        return AlbumGainResult([_to_result(out[i], out[i].file_type) for i in range(n)], alb.album_loudness_db,
                               alb.album_gain_db, alb.album_peak)

    def analyze_track_files(self, files, track_index: Optional[int] = None) -> list:
        """analyze_track for every file (`-r`), the files dealt out over all devices; per file a result or its error."""
        n = len(files)
        paths = (C.c_char_p * max(1, n))(*[os.fsencode(os.fspath(f)) for f in files])
        out = (_capi.TrackResult * max(1, n))()
        status = (C.c_int32 * max(1, n))()
        self._check(self._lib.rg_analyze_tracks_node(self._node, paths, n, -1 if track_index is None else int(track_index), out, status))
        res = []
        for i in range(n):
            if status[i] == 0:
                res.append(_to_result(out[i], out[i].file_type))
            else:
                res.append(ReplayGainError(int(status[i]), self._lib.rg_node_tracks_error(self._node, i).decode("utf-8", "replace")))
        return res

    def analyze_track_file(self, file_path, track_index: Optional[int] = None) -> ReplayGainResult:
        """analyze_track_with_index (src/replaygain.rs:935-941) on the node's first device."""
        ctx = self._lib.rg_node_ctx(self._node, 0)
        if not ctx:
            r = self.analyze_track_files([file_path], track_index)[0]
            if isinstance(r, ReplayGainError):
                raise r
            return r
        out = _capi.TrackResult()
        rc = self._lib.rg_analyze_track(ctx, os.fsencode(os.fspath(file_path)), -1 if track_index is None else int(track_index), C.byref(out))
        if rc != 0:
            raise ReplayGainError(rc, self._lib.rg_last_error(ctx).decode("utf-8", "replace"))
        return _to_result(out, out.file_type)

    def find_peak_amplitude_file(self, file_path) -> PeakAmplitudeResult:
        """find_peak_amplitude (src/replaygain.rs:1140-1249) on the node's first device."""
        ctx = self._lib.rg_node_ctx(self._node, 0)
        pk = _capi.PeakResult()
        rc = self._lib.rg_find_peak_amplitude(ctx, os.fsencode(os.fspath(file_path)), C.byref(pk))
        if rc != 0:
            raise ReplayGainError(rc, self._lib.rg_last_error(ctx).decode("utf-8", "replace"))
        return PeakAmplitudeResult(pk.peak, pk.peak_pcm, pk.sample_rate)

    def last_partition(self, n: int) -> List[int]:
        own = (C.c_uint32 * max(1, n))()
        self._check(self._lib.rg_node_last_partition(self._node, own, n))
        return [int(own[i]) for i in range(n)]


def node_partition(sizes: Sequence[int], world: int) -> List[int]:
    """rg_node_partition: device of every item (heaviest first, each to the least loaded device)."""
    lib = _capi.load()
    n = len(sizes)
    a = (C.c_uint64 * max(1, n))(*[int(x) for x in sizes])
    own = (C.c_uint32 * max(1, n))()
    lib.rg_node_partition(a, n, world, own)
    return [int(own[i]) for i in range(n)]


def _to_result(r: _capi.TrackResult, file_type: AudioFileType) -> ReplayGainResult:
    return ReplayGainResult(r.loudness_db, r.gain_db, r.peak, r.sample_rate, AudioFileType(int(file_type)), r.windows, r.flags)


_default: Optional[Analyzer] = None


def _default_analyzer() -> Analyzer:
    global _default
    if _default is None:
        _default = Analyzer(0)
    return _default


def _is_path(x) -> bool:
    return isinstance(x, (str, bytes, os.PathLike))


def analyze_track(track) -> ReplayGainResult:
    """analyze_track (src/replaygain.rs:929-932) for a file path, or for decoded PCM (PcmTrack)."""
    a = _default_analyzer()
    return a.analyze_track_file(track) if _is_path(track) else a.analyze_track(track)


def analyze_track_with_index(file_path, track_index: Optional[int]) -> ReplayGainResult:
    return _default_analyzer().analyze_track_file(file_path, track_index)


def analyze_album(tracks) -> AlbumGainResult:
    """analyze_album (src/replaygain.rs:1033-1036) for file paths, or for decoded PCM (PcmTracks)."""
    a = _default_analyzer()
    tracks = list(tracks)
    return a.analyze_album_files(tracks) if tracks and _is_path(tracks[0]) else a.analyze_album(tracks)


def analyze_album_with_index(files, track_index: Optional[int]) -> AlbumGainResult:
    return _default_analyzer().analyze_album_files(list(files), track_index)


def find_peak_amplitude(track) -> PeakAmplitudeResult:
    a = _default_analyzer()
    return a.find_peak_amplitude_file(track) if _is_path(track) else a.find_peak_amplitude(track)


def set_decoder_command(command_template: Optional[str]) -> None:
    _default_analyzer().set_decoder_command(command_template)
