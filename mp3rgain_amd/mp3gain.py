"""Host-side mirror of mp3rgain's lossless-gain API (src/lib.rs) over include/mp3rgain_amd_mp3.h.

Same names and error behaviour as the reference crate root: analyze, apply_gain, apply_gain_db,
apply_gain_wrap, apply_gain_channel, apply_gain_with_undo, apply_gain_with_undo_wrap,
apply_gain_channel_with_undo, undo_gain, is_mono, db_to_steps, steps_to_db, Channel, Mp3Analysis,
read_ape_tag_value / write / delete.  All byte work happens in libmp3rgain_amd.so (host code)."""
from __future__ import annotations

import ctypes as C
import enum
import os
from dataclasses import dataclass
from typing import Optional

from . import _capi

GAIN_STEP_DB = 1.5   # src/lib.rs:48
MAX_GAIN = 255       # src/lib.rs:51
MIN_GAIN = 0         # src/lib.rs:54
TAG_MP3GAIN_UNDO = "MP3GAIN_UNDO"
TAG_MP3GAIN_MINMAX = "MP3GAIN_MINMAX"


class Mp3GainError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


class Channel(enum.IntEnum):  # src/lib.rs:643-667
    Left = 0
    Right = 1


class _Analysis(C.Structure):
    _fields_ = [("frame_count", C.c_uint64), ("mpeg_version", C.c_uint32), ("channel_mode", C.c_uint32),
                ("min_gain", C.c_uint8), ("max_gain", C.c_uint8), ("pad_", C.c_uint8 * 6), ("avg_gain", C.c_double),
                ("headroom_steps", C.c_int32), ("pad2_", C.c_int32), ("headroom_db", C.c_double),
                ("mpeg_version_str", C.c_char * 8), ("channel_mode_str", C.c_char * 16)]


class Header(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("mpeg_version", "has_crc", "bitrate_kbps", "sample_rate", "padding",
                                          "channel_mode", "frame_size")]


@dataclass
class Mp3Analysis:  # src/lib.rs:58-75
    frame_count: int
    mpeg_version: str
    channel_mode: str
    min_gain: int
    max_gain: int
    avg_gain: float
    headroom_steps: int
    headroom_db: float


_vp, _sz, _i32, _i64, _int, _dbl, _cp = C.c_void_p, C.c_size_t, C.c_int32, C.c_int64, C.c_int, C.c_double, C.c_char_p
SYMBOLS = [
    ("rg_mp3_last_error", _cp, []),
    ("rg_mp3_parse_header", _int, [_vp, _sz, C.POINTER(Header)]),
    ("rg_mp3_read_gain_at", C.c_uint8, [_vp, _sz, _sz, C.c_uint]),
    ("rg_mp3_write_gain_at", None, [_vp, _sz, _sz, C.c_uint, C.c_uint8]),
    ("rg_mp3_skip_id3v2", _sz, [_vp, _sz]),
    ("rg_mp3_find_audio_end", _sz, [_vp, _sz]),
    ("rg_mp3_is_xing_frame", _int, [_vp, _sz, _sz]),
    ("rg_mp3_gain_locations", _int, [_vp, _sz, _sz, C.POINTER(_sz), C.POINTER(C.c_uint)]),
    ("rg_mp3_analyze_data", _i64, [_vp, _sz, C.POINTER(_Analysis)]),
    ("rg_mp3_apply_gain_data", _i64, [_vp, _sz, _i32, _int]),
    ("rg_mp3_apply_gain_channel_data", _i64, [_vp, _sz, _int, _i32]),
    ("rg_mp3_analyze", _i64, [_cp, C.POINTER(_Analysis)]),
    ("rg_mp3_apply_gain", _i64, [_cp, _i32]),
    ("rg_mp3_apply_gain_db", _i64, [_cp, _dbl]),
    ("rg_mp3_apply_gain_wrap", _i64, [_cp, _i32]),
    ("rg_mp3_apply_gain_channel", _i64, [_cp, _int, _i32]),
    ("rg_mp3_apply_gain_with_undo", _i64, [_cp, _i32]),
    ("rg_mp3_apply_gain_with_undo_wrap", _i64, [_cp, _i32]),
    ("rg_mp3_apply_gain_channel_with_undo", _i64, [_cp, _int, _i32]),
    ("rg_mp3_undo_gain", _i64, [_cp]),
    ("rg_mp3_is_mono", _int, [_cp]),
    ("rg_ape_get", _i64, [_cp, _cp, _vp, _sz]),
    ("rg_ape_get_data", _i64, [_vp, _sz, _cp, _vp, _sz]),
    ("rg_ape_item_count_data", _i64, [_vp, _sz]),
    ("rg_ape_set", _int, [_cp, _cp, _cp]),
    ("rg_ape_remove", _int, [_cp, _cp]),
    ("rg_ape_delete", _int, [_cp]),
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        L = _capi.load()
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _p(path) -> bytes:
    return os.fsencode(os.fspath(path))


def _ok(rc: int) -> int:
    if rc < 0:
        raise Mp3GainError(rc, lib().rg_mp3_last_error().decode())
    return rc


def _analysis(a: _Analysis) -> Mp3Analysis:
    return Mp3Analysis(a.frame_count, a.mpeg_version_str.decode(), a.channel_mode_str.decode(), a.min_gain, a.max_gain,
                       a.avg_gain, a.headroom_steps, a.headroom_db)


def analyze(file_path) -> Mp3Analysis:
    a = _Analysis()
    _ok(lib().rg_mp3_analyze(_p(file_path), C.byref(a)))
    return _analysis(a)


def analyze_data(data: bytes) -> Mp3Analysis:
    a = _Analysis()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    _ok(lib().rg_mp3_analyze_data(buf, len(data), C.byref(a)))
    return _analysis(a)


def apply_gain(file_path, gain_steps: int) -> int:
    return _ok(lib().rg_mp3_apply_gain(_p(file_path), gain_steps))


def apply_gain_db(file_path, gain_db: float) -> int:
    return _ok(lib().rg_mp3_apply_gain_db(_p(file_path), gain_db))


def apply_gain_wrap(file_path, gain_steps: int) -> int:
    return _ok(lib().rg_mp3_apply_gain_wrap(_p(file_path), gain_steps))


def apply_gain_channel(file_path, channel: Channel, gain_steps: int) -> int:
    return _ok(lib().rg_mp3_apply_gain_channel(_p(file_path), int(channel), gain_steps))


def apply_gain_with_undo(file_path, gain_steps: int) -> int:
    return _ok(lib().rg_mp3_apply_gain_with_undo(_p(file_path), gain_steps))


def apply_gain_with_undo_wrap(file_path, gain_steps: int) -> int:
    return _ok(lib().rg_mp3_apply_gain_with_undo_wrap(_p(file_path), gain_steps))


def apply_gain_channel_with_undo(file_path, channel: Channel, gain_steps: int) -> int:
    return _ok(lib().rg_mp3_apply_gain_channel_with_undo(_p(file_path), int(channel), gain_steps))


def undo_gain(file_path) -> int:
    return _ok(lib().rg_mp3_undo_gain(_p(file_path)))


def is_mono(file_path) -> bool:
    return bool(_ok(lib().rg_mp3_is_mono(_p(file_path))))


def db_to_steps(db: float) -> int:
    return _capi.load().rg_db_to_steps(db)


def steps_to_db(steps: int) -> float:
    return _capi.load().rg_steps_to_db(steps)


def apply_gain_data(data: bytes, gain_steps: int, wrap: bool = False, channel: Optional[Channel] = None):
    """-> (patched bytes, frames modified); the in-memory cores (src/lib.rs:544-592, :677-737)"""
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    if channel is None:
        n = _ok(lib().rg_mp3_apply_gain_data(buf, len(data), gain_steps, int(wrap)))
    else:
        n = _ok(lib().rg_mp3_apply_gain_channel_data(buf, len(data), int(channel), gain_steps))
    return bytes(buf), n


def read_ape_tag_value(file_path, key: str) -> Optional[str]:
    buf = C.create_string_buffer(4096)
    n = lib().rg_ape_get(_p(file_path), key.encode(), buf, len(buf))
    if n < -1:
        _ok(n)
    return None if n < 0 else buf.value.decode("utf-8", "replace")


def write_ape_tag_value(file_path, key: str, value: str) -> None:
    _ok(lib().rg_ape_set(_p(file_path), key.encode(), value.encode()))


def remove_ape_tag_value(file_path, key: str) -> None:
    _ok(lib().rg_ape_remove(_p(file_path), key.encode()))


def delete_ape_tag(file_path) -> None:
    _ok(lib().rg_ape_delete(_p(file_path)))


def has_ape_tag(file_path) -> bool:
    """read_ape_tag_from_file(..)?.is_some() (src/lib.rs:1030-1038): raises like the reference when unreadable."""
    try:
        with open(file_path, "rb") as f:
            data = f.read()
    except OSError:
        raise Mp3GainError(-101, f"Failed to read: {os.fspath(file_path)}")
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data if data else b"\0")
    return lib().rg_ape_item_count_data(buf, len(data)) >= 0
