"""ctypes mirror of include/mp3rgain_amd_demux.h: ISO base media (MP4 / M4A) sample tables and ADTS walkers.

What the reference gets from symphonia's probe (src/replaygain.rs:815-858): which audio tracks a file has, the selected
one's sample rate and channel count, and its packets.  Host code; no GPU involved."""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

from . import _capi

CODEC_AAC, CODEC_MP3 = 1, 2


class Mp4AudioTrack(C.Structure):
    _fields_ = [("track_id", C.c_uint32), ("codec", C.c_uint32), ("object_type", C.c_uint32), ("sample_rate", C.c_uint32),
                ("channels", C.c_uint32), ("timescale", C.c_uint32), ("duration", C.c_uint64), ("n_samples", C.c_uint64),
                ("audio_object_type", C.c_uint32), ("asc_len", C.c_uint32), ("asc", C.c_uint8 * 32)]


class AdtsInfo(C.Structure):
    _fields_ = [("sample_rate", C.c_uint32), ("channels", C.c_uint32), ("profile", C.c_uint32), ("mpeg_version", C.c_uint32),
                ("frames", C.c_uint64), ("raw_blocks", C.c_uint64), ("first_frame_offset", C.c_uint64), ("junk_bytes", C.c_uint64)]


class DemuxError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


_bound = False


def lib():
    global _bound
    L = _capi.load()
    if not _bound:
        L.rg_mp4_audio_tracks.restype = C.c_int
        L.rg_mp4_audio_tracks.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Mp4AudioTrack), C.c_size_t, C.POINTER(C.c_size_t)]
        L.rg_mp4_access_units.restype = C.c_int
        L.rg_mp4_access_units.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_size_t)]
        L.rg_adts_scan.restype = C.c_int
        L.rg_adts_scan.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(AdtsInfo)]
        L.rg_adts_access_units.restype = C.c_int
        L.rg_adts_access_units.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_size_t)]
        L.rg_demux_last_error.restype = C.c_char_p
        L.rg_demux_last_error.argtypes = []
        _bound = True
    return L


def _check(rc: int):
    if rc != 0:
        raise DemuxError(rc, lib().rg_demux_last_error().decode("utf-8", "replace"))


def _buf(data: bytes):
    return (C.c_char * max(1, len(data))).from_buffer_copy(data or b"\0")


def mp4_audio_tracks(data: bytes) -> List[Mp4AudioTrack]:
    L = lib()
    n = C.c_size_t()
    out = (Mp4AudioTrack * 32)()
    _check(L.rg_mp4_audio_tracks(_buf(data), len(data), out, 32, C.byref(n)))
    return [out[i] for i in range(min(32, n.value))]


def mp4_access_units(data: bytes, audio_index: int) -> List[Tuple[int, int]]:
    L = lib()
    n = C.c_size_t()
    b = _buf(data)
    _check(L.rg_mp4_access_units(b, len(data), audio_index, None, None, 0, C.byref(n)))
    k = n.value
    offs, sizes = (C.c_uint64 * max(1, k))(), (C.c_uint32 * max(1, k))()
    _check(L.rg_mp4_access_units(b, len(data), audio_index, offs, sizes, k, C.byref(n)))
    return [(int(offs[i]), int(sizes[i])) for i in range(min(k, n.value))]


def adts_scan(data: bytes) -> AdtsInfo:
    info = AdtsInfo()
    _check(lib().rg_adts_scan(_buf(data), len(data), C.byref(info)))
    return info


def adts_access_units(data: bytes) -> List[Tuple[int, int]]:
    L = lib()
    n = C.c_size_t()
    b = _buf(data)
    _check(L.rg_adts_access_units(b, len(data), None, None, 0, C.byref(n)))
    k = n.value
    offs, sizes = (C.c_uint64 * max(1, k))(), (C.c_uint32 * max(1, k))()
    _check(L.rg_adts_access_units(b, len(data), offs, sizes, k, C.byref(n)))
    return [(int(offs[i]), int(sizes[i])) for i in range(min(k, n.value))]
