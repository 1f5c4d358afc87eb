"""ctypes loader for the CPU oracle (oracle/librg_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under mp3rgain_amd/ imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "librg_oracle.so"
HIST = 12000
FMT_F32, FMT_S16, FMT_S32 = 0, 1, 2
_NP_FMT = {np.dtype(np.float32): FMT_F32, np.dtype(np.int16): FMT_S16, np.dtype(np.int32): FMT_S32}


class Result(C.Structure):
    _fields_ = [
        ("loudness_db", C.c_double),
        ("gain_db", C.c_double),
        ("peak", C.c_double),
        ("sample_rate", C.c_uint32),
        ("gain_steps", C.c_int32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(force: bool = False) -> Path:
    src_newer = (not LIB_PATH.exists()) or any(
        p.stat().st_mtime > LIB_PATH.stat().st_mtime
        for p in (HERE / "rg_oracle.c", HERE / "rg_oracle.h", HERE.parent / "include" / "rg_coeffs.h",
                  HERE.parent / "include" / "rg_synth.h")
    )
    if force or src_newer:
        subprocess.run(["make", "-C", str(HERE), "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(LIB_PATH))
        vp, sz, u32, i32, u64, dbl = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int32, C.c_uint64, C.c_double
        L.rgo_analyze_pcm.argtypes = [vp, vp, sz, C.c_uint, C.c_int, C.POINTER(Result), vp]
        L.rgo_analyze_pcm.restype = C.c_int
        L.rgo_hist_loudness.argtypes = [vp]
        L.rgo_hist_loudness.restype = dbl
        L.rgo_percentile_threshold.argtypes = [u64]
        L.rgo_percentile_threshold.restype = u64
        L.rgo_gain_from_loudness.argtypes = [dbl]
        L.rgo_gain_from_loudness.restype = dbl
        L.rgo_gain_steps.argtypes = [dbl]
        L.rgo_gain_steps.restype = i32
        L.rgo_clip_limit_steps.argtypes = [i32, dbl, dbl, C.c_int, C.c_int]
        L.rgo_clip_limit_steps.restype = i32
        L.rgo_unit_test_sine.argtypes = [C.c_uint, dbl, dbl, sz, vp]
        L.rgo_unit_test_sine.restype = dbl
        L.rgo_synth_fill_f32.argtypes = [vp, u64, C.c_uint, C.c_uint, u64, sz]
        L.rgo_synth_fill_f32.restype = None
        L.rgo_find_peak.argtypes = [C.POINTER(vp), C.c_uint, sz, C.c_int]
        L.rgo_find_peak.restype = dbl
        L.rgo_rate_coeffs.argtypes = [C.c_uint]
        L.rgo_rate_coeffs.restype = vp
        L.rgo_track_begin.argtypes = [vp, C.c_uint, C.c_uint]
        L.rgo_track_begin.restype = C.c_int
        L.rgo_process_buffer.argtypes = [vp, vp, vp, sz, C.c_int]
        L.rgo_process_buffer.restype = None
        L.rgo_track_finish.argtypes = [vp, C.POINTER(Result)]
        L.rgo_track_finish.restype = None
        L.rgo_filter_init.argtypes = [vp, C.c_uint]
        L.rgo_filter_init.restype = C.c_int
        L.rgo_filter_process.argtypes = [vp, dbl]
        L.rgo_filter_process.restype = dbl
        _lib = L
    return _lib


def supported_rate(rate: int) -> bool:
    return bool(lib().rgo_rate_coeffs(rate))


def analyze_pcm(ch0: np.ndarray, ch1: np.ndarray | None, rate: int):
    """-> (dict result, hist uint32[12000]); raises ValueError for an unsupported rate."""
    ch0 = np.ascontiguousarray(ch0)
    fmt = _NP_FMT[ch0.dtype]
    if ch1 is not None:
        ch1 = np.ascontiguousarray(ch1)
        assert ch1.dtype == ch0.dtype and ch1.shape == ch0.shape
    res = Result()
    hist = np.zeros(HIST, dtype=np.uint32)
    rc = lib().rgo_analyze_pcm(ch0.ctypes.data, ch1.ctypes.data if ch1 is not None else None,
                               ch0.shape[0], rate, fmt, C.byref(res), hist.ctypes.data)
    if rc != 0:
        raise ValueError(f"Unsupported sample rate: {rate} Hz")
    return res.as_dict(), hist


class TrackStream:
    """Packet-at-a-time driver mirroring the reference's decode loop (replaygain.rs:881-907)."""

    _STATE_BYTES = 2 * (8 + 28 * 8) + (4 * 8 + 12000 * 4) + 64  # generous upper bound

    def __init__(self, rate: int, channels: int):
        self._buf = C.create_string_buffer(self._STATE_BYTES + 4096)
        if lib().rgo_track_begin(self._buf, rate, channels) != 0:
            raise ValueError(f"Unsupported sample rate: {rate} Hz")

    def push(self, ch0: np.ndarray, ch1: np.ndarray | None):
        ch0 = np.ascontiguousarray(ch0)
        fmt = _NP_FMT[ch0.dtype]
        p1 = None
        if ch1 is not None:
            ch1 = np.ascontiguousarray(ch1)
            p1 = ch1.ctypes.data
        lib().rgo_process_buffer(self._buf, ch0.ctypes.data, p1, ch0.shape[0], fmt)

    def finish(self):
        res = Result()
        lib().rgo_track_finish(self._buf, C.byref(res))
        return res.as_dict()


def hist_loudness(hist: np.ndarray) -> float:
    hist = np.ascontiguousarray(hist, dtype=np.uint32)
    assert hist.shape == (HIST,)
    return lib().rgo_hist_loudness(hist.ctypes.data)


def album_from_hists(hists, peaks):
    """analyze_album's merge (replaygain.rs:1048-1066): u32 sum of histograms, max of peaks."""
    acc = np.zeros(HIST, dtype=np.uint32)
    for h in hists:
        acc += np.asarray(h, dtype=np.uint32)
    loud = hist_loudness(acc)
    return {"album_loudness_db": loud, "album_gain_db": lib().rgo_gain_from_loudness(loud),
            "album_peak": float(max(peaks)) if len(peaks) else 0.0}, acc


def unit_test_sine(rate: int, freq: float, amp: float, n: int):
    hist = np.zeros(HIST, dtype=np.uint32)
    loud = lib().rgo_unit_test_sine(rate, freq, amp, n, hist.ctypes.data)
    return loud, hist


def synth_f32(seed: int, channel: int, rate: int, frames: int, first_frame: int = 0) -> np.ndarray:
    out = np.empty(frames, dtype=np.float32)
    lib().rgo_synth_fill_f32(out.ctypes.data, seed, channel, rate, first_frame, frames)
    return out


def find_peak(chans, fmt_dtype=np.float32) -> float:
    arrs = [np.ascontiguousarray(c, dtype=fmt_dtype) for c in chans]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    return lib().rgo_find_peak(ptrs, len(arrs), arrs[0].shape[0], _NP_FMT[np.dtype(fmt_dtype)])


def n_threads_default() -> int:
    return os.cpu_count() or 1
