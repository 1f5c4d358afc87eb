"""Host-side mirror of mp3rgain's MP4/M4A ReplayGain tag module (src/mp4meta.rs) over
include/mp3rgain_amd_mp4.h: ReplayGainTags (set_track / set_album / is_empty), read_replaygain_tags,
write_replaygain_tags, delete_replaygain_tags, is_mp4_file, and the in-memory cores.  All byte work
happens in libmp3rgain_amd.so (host code)."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

from . import _capi

RG_TRACK_GAIN = "replaygain_track_gain"  # src/mp4meta.rs:26-29
RG_TRACK_PEAK = "replaygain_track_peak"
RG_ALBUM_GAIN = "replaygain_album_gain"
RG_ALBUM_PEAK = "replaygain_album_peak"
ITUNES_NAMESPACE = "com.apple.iTunes"   # src/mp4meta.rs:32
_VMAX = 64


class Mp4MetaError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


class _Tags(C.Structure):
    _fields_ = [("has_track_gain", C.c_uint8), ("has_track_peak", C.c_uint8), ("has_album_gain", C.c_uint8),
                ("has_album_peak", C.c_uint8), ("pad_", C.c_uint8 * 4), ("track_gain", C.c_char * _VMAX),
                ("track_peak", C.c_char * _VMAX), ("album_gain", C.c_char * _VMAX), ("album_peak", C.c_char * _VMAX)]


_vp, _sz, _int, _i64, _dbl, _cp = C.c_void_p, C.c_size_t, C.c_int, C.c_int64, C.c_double, C.c_char_p
_tp = C.POINTER(_Tags)
SYMBOLS = [
    ("rg_mp4_last_error", _cp, []),
    ("rg_mp4_tags_clear", None, [_tp]),
    ("rg_mp4_tags_set_track", None, [_tp, _dbl, _dbl]),
    ("rg_mp4_tags_set_album", None, [_tp, _dbl, _dbl]),
    ("rg_mp4_tags_is_empty", _int, [_tp]),
    ("rg_mp4_serialize_freeform", _sz, [_cp, _cp, _cp, _vp, _sz]),
    ("rg_mp4_parse_freeform", _int, [_vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz]),
    ("rg_mp4_read_replaygain_tags_data", _int, [_vp, _sz, _tp]),
    ("rg_mp4_update_metadata_data", _i64, [_vp, _sz, _tp, _vp, _sz]),
    ("rg_mp4_is_mp4_data", _int, [_vp, _sz]),
    ("rg_mp4_read_replaygain_tags", _int, [_cp, _tp]),
    ("rg_mp4_write_replaygain_tags", _int, [_cp, _tp]),
    ("rg_mp4_delete_replaygain_tags", _int, [_cp]),
    ("rg_mp4_is_mp4_file", _int, [_cp]),
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
        raise Mp4MetaError(rc, lib().rg_mp4_last_error().decode())
    return rc


def _buf(data: bytes):
    return (C.c_uint8 * max(1, len(data))).from_buffer_copy(data if data else b"\0")


@dataclass
class FreeformTag:  # src/mp4meta.rs:105-110
    namespace: str
    name: str
    value: str


@dataclass
class ReplayGainTags:  # src/mp4meta.rs:113-141
    track_gain: Optional[str] = None
    track_peak: Optional[str] = None
    album_gain: Optional[str] = None
    album_peak: Optional[str] = None

    def set_track(self, gain_db: float, peak: float) -> None:
        t = self._c()
        lib().rg_mp4_tags_set_track(C.byref(t), gain_db, peak)
        self.track_gain, self.track_peak = t.track_gain.decode(), t.track_peak.decode()

    def set_album(self, gain_db: float, peak: float) -> None:
        t = self._c()
        lib().rg_mp4_tags_set_album(C.byref(t), gain_db, peak)
        self.album_gain, self.album_peak = t.album_gain.decode(), t.album_peak.decode()

    def is_empty(self) -> bool:
        return bool(lib().rg_mp4_tags_is_empty(C.byref(self._c())))

    def _c(self) -> _Tags:
        t = _Tags()
        for f in ("track_gain", "track_peak", "album_gain", "album_peak"):
            v = getattr(self, f)
            if v is not None:
                setattr(t, f, v.encode()[:_VMAX - 1])
                setattr(t, "has_" + f, 1)
        return t

    @staticmethod
    def _from(t: _Tags) -> "ReplayGainTags":
        g = lambda f: getattr(t, f).decode(errors="replace") if getattr(t, "has_" + f) else None  # noqa: E731
        return ReplayGainTags(g("track_gain"), g("track_peak"), g("album_gain"), g("album_peak"))


def serialize_freeform_tag(tag: FreeformTag) -> bytes:
    a = (tag.namespace.encode(), tag.name.encode(), tag.value.encode())
    n = lib().rg_mp4_serialize_freeform(*a, None, 0)
    out = (C.c_uint8 * n)()
    lib().rg_mp4_serialize_freeform(*a, out, n)
    return bytes(out)


def parse_freeform_tag(content: bytes) -> Optional[FreeformTag]:
    ns, nm, va = (C.create_string_buffer(256) for _ in range(3))
    if not lib().rg_mp4_parse_freeform(_buf(content), len(content), ns, 256, nm, 256, va, 256):
        return None
    return FreeformTag(ns.value.decode(), nm.value.decode(), va.value.decode())


def read_replaygain_tags_data(data: bytes) -> ReplayGainTags:
    t = _Tags()
    _ok(lib().rg_mp4_read_replaygain_tags_data(_buf(data), len(data), C.byref(t)))
    return ReplayGainTags._from(t)


def update_mp4_metadata(data: bytes, tags: ReplayGainTags) -> bytes:
    t = tags._c()
    b = _buf(data)
    n = _ok(lib().rg_mp4_update_metadata_data(b, len(data), C.byref(t), None, 0))
    out = (C.c_uint8 * max(1, n))()
    _ok(lib().rg_mp4_update_metadata_data(b, len(data), C.byref(t), out, n))
    return bytes(out)[:n]


def is_mp4_data(data: bytes) -> bool:
    return bool(lib().rg_mp4_is_mp4_data(_buf(data), len(data)))


def read_replaygain_tags(file_path) -> ReplayGainTags:
    t = _Tags()
    _ok(lib().rg_mp4_read_replaygain_tags(_p(file_path), C.byref(t)))
    return ReplayGainTags._from(t)


def write_replaygain_tags(file_path, tags: ReplayGainTags) -> None:
    _ok(lib().rg_mp4_write_replaygain_tags(_p(file_path), C.byref(tags._c())))


def delete_replaygain_tags(file_path) -> None:
    _ok(lib().rg_mp4_delete_replaygain_tags(_p(file_path)))


def is_mp4_file(file_path) -> bool:
    return bool(lib().rg_mp4_is_mp4_file(_p(file_path)))
