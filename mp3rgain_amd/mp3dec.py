"""ctypes binding of include/mp3rgain_amd_dec.h: MPEG-1/2/2.5 Layer III bitstream -> planar f32 PCM.

The decode row of the reference (symphonia behind src/replaygain.rs:807-904) as a library call; host code,
no GPU needed.  Only loads the in-tree shared library -- there is no Python decoder behind it."""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

from . import _capi


class StreamInfo(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_uint32),
        ("channels", C.c_uint32),
        ("frames", C.c_uint64),
        ("audio_frames", C.c_uint32),
        ("skipped_frames", C.c_uint32),
        ("info_frame", C.c_uint32),
        ("id3v2_bytes", C.c_uint32),
        ("mpeg_version", C.c_uint32),
        ("samples_per_frame", C.c_uint32),
        ("first_frame_offset", C.c_uint64),
        ("junk_bytes", C.c_uint32),
        ("reserved", C.c_uint32),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "reserved"}


class Unit(C.Structure):
    _fields_ = [
        ("sf", C.c_uint8 * 40),
        ("illegal", C.c_uint64),
        ("nz", C.c_uint16),
        ("global_gain", C.c_uint8),
        ("block_type", C.c_uint8),
        ("mixed", C.c_uint8),
        ("subblock_gain", C.c_uint8 * 3),
        ("scalefac_scale", C.c_uint8),
        ("preflag", C.c_uint8),
        ("long_end", C.c_uint8),
        ("short_start", C.c_uint8),
        ("mode_ext", C.c_uint8),
        ("intensity_scale", C.c_uint8),
        ("reserved", C.c_uint8 * 2),
    ]


SYMBOLS = [
    ("rg_mp3_scan", C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(StreamInfo)]),
    ("rg_mp3_decode_f32", C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(StreamInfo)]),
    ("rg_mp3_parse_units", C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(StreamInfo)]),
    ("rg_mp3_index_units", C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(StreamInfo)]),
    ("rg_mp3_index_selfcheck", C.c_int, [C.c_void_p, C.c_size_t]),
    ("rg_mp3dec_last_error", C.c_char_p, []),
]

_lib = None


class Mp3DecodeError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


def lib():
    global _lib
    if _lib is None:
        raw = C.CDLL(str(_capi.LIB_PATH))
        for name, res, args in SYMBOLS:
            fn = getattr(raw, name)
            fn.restype = res
            fn.argtypes = args
        _lib = raw
    return _lib


def _check(rc: int):
    if rc != 0:
        raise Mp3DecodeError(rc, lib().rg_mp3dec_last_error().decode("utf-8", "replace"))


def scan(data: bytes) -> StreamInfo:
    info = StreamInfo()
    buf = (C.c_char * max(1, len(data))).from_buffer_copy(data or b"\0")
    _check(lib().rg_mp3_scan(C.cast(buf, C.c_void_p), len(data), C.byref(info)))
    return info


def decode(data: bytes) -> Tuple[np.ndarray, StreamInfo]:
    """-> (float32 array [channels][frames], StreamInfo)"""
    info = scan(data)
    cap = int(info.frames)
    out = np.zeros((int(info.channels), max(1, cap)), dtype=np.float32)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    di = StreamInfo()
    _check(lib().rg_mp3_decode_f32(C.cast(buf, C.c_void_p), len(data), out[0].ctypes.data,
                                   out[1].ctypes.data if info.channels == 2 else None, cap, C.byref(di)))
    return out[:, :int(di.frames)], di


def parse_units(data: bytes):
    """Stage A only -> (int16 [units][576], Unit array, StreamInfo): what the device back half consumes."""
    info = scan(data)
    cap = int(info.audio_frames) * (2 if info.mpeg_version == 1 else 1) * int(info.channels)
    is_ = np.zeros((max(1, cap), 576), dtype=np.int16)
    units = (Unit * max(1, cap))()
    n = C.c_uint64()
    di = StreamInfo()
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    _check(lib().rg_mp3_parse_units(C.cast(buf, C.c_void_p), len(data), is_.ctypes.data, C.cast(units, C.c_void_p), cap, C.byref(n), C.byref(di)))
    return is_[:n.value], units, di


def index_units(data: bytes):
    """The frame walk of the device route -> (units, StreamInfo)."""
    n = C.c_uint64()
    di = StreamInfo()
    buf = (C.c_char * max(1, len(data))).from_buffer_copy(data or b"\0")
    _check(lib().rg_mp3_index_units(C.cast(buf, C.c_void_p), len(data), C.byref(n), C.byref(di)))
    return int(n.value), di


def index_selfcheck(data: bytes) -> int:
    """0 when the device route's host half + shared frame logic agree with `index_units` (rg_mp3_index_selfcheck)."""
    buf = (C.c_char * max(1, len(data))).from_buffer_copy(data or b"\0")
    return int(lib().rg_mp3_index_selfcheck(C.cast(buf, C.c_void_p), len(data)))
