"""ctypes binding of include/mp3rgain_amd.h (the C-ABI drop-in boundary).

This module only loads the in-tree shared library and declares signatures; there is no
Python or CPU compute path behind it.  If the library is missing it raises, loudly.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
import os

# MP3RGAIN_AMD_LIB selects another build of the same library (kernel experiments); never a fallback
LIB_PATH = Path(os.environ.get("MP3RGAIN_AMD_LIB", PKG_DIR / "libmp3rgain_amd.so"))

HISTOGRAM_SIZE = 12000
COMM_ID_BYTES = 128
HISTOGRAM_OFFSET = 2000
FMT_F32_PLANAR, FMT_S16_PLANAR, FMT_S32_PLANAR = 0, 1, 2

RG_OK = 0
RG_ERR_INVALID_ARG = -1
RG_ERR_UNSUPPORTED_RATE = -2
RG_ERR_DEVICE = -3
RG_ERR_NO_DEVICE = -4
RG_ERR_NOMEM = -5
RG_ERR_STATE = -6
RG_ERR_COLLECTIVE = -7


class TrackDesc(C.Structure):
    _fields_ = [
        ("offset_bytes", C.c_uint64),
        ("frames", C.c_uint64),
        ("sample_rate", C.c_uint32),
        ("channels", C.c_uint16),
        ("format", C.c_uint16),
    ]


class TrackResult(C.Structure):
    _fields_ = [
        ("loudness_db", C.c_double),
        ("gain_db", C.c_double),
        ("peak", C.c_double),
        ("sample_rate", C.c_uint32),
        ("gain_steps", C.c_int32),
        ("windows", C.c_uint32),
        ("file_type", C.c_uint32),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class AlbumResult(C.Structure):
    _fields_ = [
        ("album_loudness_db", C.c_double),
        ("album_gain_db", C.c_double),
        ("album_peak", C.c_double),
        ("album_gain_steps", C.c_int32),
        ("windows", C.c_uint32),
    ]


class PeakResult(C.Structure):
    _fields_ = [
        ("peak", C.c_double),
        ("peak_pcm", C.c_double),
        ("sample_rate", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class DeviceView(C.Structure):
    _fields_ = [
        ("d_track_hist", C.c_void_p),
        ("d_track_result", C.c_void_p),
        ("d_album_hist", C.c_void_p),
        ("d_album_peak", C.c_void_p),
        ("n_tracks", C.c_uint64),
    ]


# every symbol include/mp3rgain_amd.h declares: (name, restype, argtypes)
_vp, _sz, _u32, _i32, _u64, _dbl, _int = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int32, C.c_uint64, C.c_double, C.c_int
_P = C.POINTER
class WavInfo(C.Structure):
    _fields_ = [("sample_rate", C.c_uint32), ("channels", C.c_uint16), ("bits_per_sample", C.c_uint16),
                ("sample_format", C.c_uint16), ("block_align", C.c_uint16), ("reserved", C.c_uint32),
                ("data_offset", C.c_uint64), ("frames", C.c_uint64)]


SYMBOLS = [
    ("rg_abi_version", _int, []),
    ("rg_is_available", _int, []),
    ("rg_supported_rate", _int, [_u32]),
    ("rg_window_samples", _u32, [_u32]),
    ("rg_hist_loudness", _dbl, [_vp]),
    ("rg_gain_from_loudness", _dbl, [_dbl]),
    ("rg_gain_steps", _i32, [_dbl]),
    ("rg_db_to_steps", _i32, [_dbl]),
    ("rg_steps_to_db", _dbl, [_i32]),
    ("rg_clip_limit_steps", _i32, [_i32, _dbl, _dbl, _int, _int]),
    ("rg_rate_design_info", _int, [_u32, _P(_int), _P(_u32), _P(_dbl)]),
    ("rg_create", _vp, [_int]),
    ("rg_destroy", None, [_vp]),
    ("rg_last_error", C.c_char_p, [_vp]),
    ("rg_set_stream", _int, [_vp, _vp, _int]),
    ("rg_wait_user_stream", _int, [_vp]),
    ("rg_batch_stream", _vp, [_vp]),
    ("rg_set_kernel", _int, [_vp, _int]),
    ("rg_set_tuning", _int, [_vp, _int, C.c_int64]),
    ("rg_tm_design_info", _int, [_u32, _u32, _P(_u32), _P(_u32), _P(_u32), _P(_dbl), _vp, _vp]),
    ("rg_tm_design_affine", _int, [_u32, _u32, _P(_int), _P(_dbl), _P(_dbl), _P(_dbl), _P(_dbl), _vp]),
    ("rg_analyze_pcm_batch", _int, [_vp, _P(TrackDesc), _sz, _vp, _sz, _int, _P(TrackResult), _vp]),
    ("rg_analyze_album_pcm", _int, [_vp, _P(TrackDesc), _sz, _vp, _sz, _int, _P(TrackResult), _P(AlbumResult), _vp]),
    ("rg_find_peak_pcm", _int, [_vp, _P(TrackDesc), _vp, _sz, _int, _P(PeakResult)]),
    ("rg_enqueue_pcm_batch", _int, [_vp, _P(TrackDesc), _sz, _vp, _sz, _int]),
    ("rg_device_view_get", _int, [_vp, _P(DeviceView)]),
    ("rg_collect", _int, [_vp, _P(TrackResult), _vp]),
    ("rg_collect_exact", _int, [_vp, _P(TrackDesc), _sz, _vp, _sz, _P(TrackResult), _vp]),
    ("rg_album_allreduce", _int, [_vp, _vp]),
    ("rg_album_finish", _int, [_vp, _P(AlbumResult), _vp]),
    ("rg_comm_library", _int, [C.c_char_p]),
    ("rg_comm_unique_id", _int, [_vp]),
    ("rg_comm_init", _int, [_vp, _vp, _int, _int]),
    ("rg_comm_destroy", _int, [_vp]),
    ("rg_comm_info", _int, [_vp, _P(_int), _P(_int)]),
    ("rg_album_exchange", _int, [_vp]),
    ("rg_album_reduce_gathered", _int, [_vp, _vp, _u32]),
    ("rg_album_result_enqueue", _int, [_vp]),
    ("rg_timing_enable", _int, [_vp, _int]),
    ("rg_timing_read", _int, [_vp, _P(_dbl), _P(_u64), _P(_dbl), _int]),
    ("rg_synth_fill_device", _int, [_vp, _vp, _u64, _u32, _u32, _u64, _u64]),
    ("rg_wav_parse", _int, [_vp, _sz, _P(WavInfo)]),
    ("rg_set_decoder_command", _int, [_vp, C.c_char_p]),
    ("rg_analyze_wav_batch", _int, [_vp, _P(_vp), _P(_sz), _sz, _int, _P(TrackResult), _P(AlbumResult)]),
    ("rg_analyze_track", _int, [_vp, C.c_char_p, _i32, _P(TrackResult)]),
    ("rg_analyze_album", _int, [_vp, _P(C.c_char_p), _sz, _i32, _P(TrackResult), _P(AlbumResult)]),
    ("rg_analyze_tracks", _int, [_vp, C.POINTER(C.c_char_p), C.c_size_t, C.c_int32, _P(TrackResult), C.POINTER(C.c_int32)]),
    ("rg_tracks_error", C.c_char_p, [_vp, C.c_size_t]),
    ("rg_find_peak_amplitude", _int, [_vp, C.c_char_p, _P(PeakResult)]),
    ("rg_mp3_decode_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    ("rg_mp3_decode_bench", _int, [_vp, _vp, _sz, _u32, _u32, _P(_dbl), _P(_u64), _P(_u64), _P(_u64)]),
    ("rg_analyze_album_begin", _int, [_vp, _P(C.c_char_p), _sz, _i32, _P(TrackResult), _P(_sz)]),
    # include/mp3rgain_amd_node.h
    ("rg_node_create", _vp, [_P(_int), _sz]),
    ("rg_node_create_backend", _vp, [_vp, _P(_int), _sz]),
    ("rg_node_destroy", None, [_vp]),
    ("rg_node_last_error", C.c_char_p, [_vp]),
    ("rg_node_devices", _sz, [_vp]),
    ("rg_node_ctx", _vp, [_vp, _sz]),
    ("rg_node_set_exchange", _int, [_vp, _int]),
    ("rg_node_partition", None, [_P(_u64), _sz, _sz, _P(_u32)]),
    ("rg_analyze_album_node", _int, [_vp, _P(C.c_char_p), _sz, _i32, _P(TrackResult), _P(AlbumResult)]),
    ("rg_analyze_tracks_node", _int, [_vp, _P(C.c_char_p), _sz, _i32, _P(TrackResult), _P(_i32)]),
    ("rg_node_tracks_error", C.c_char_p, [_vp, _sz]),
    ("rg_node_last_partition", _int, [_vp, _P(_u32), _sz]),
]

# rg_node_backend (include/mp3rgain_amd_node.h): a table of per-device functions
NODE_OPEN = C.CFUNCTYPE(_vp, _int, _vp)
NODE_CLOSE = C.CFUNCTYPE(None, _vp, _vp)
NODE_ALBUM_BEGIN = C.CFUNCTYPE(_int, _vp, _P(C.c_char_p), _sz, _i32, _P(TrackResult), _P(_sz), _vp)
NODE_ALBUM_PACK = C.CFUNCTYPE(_int, _vp, _P(_u32), _vp)
NODE_TRACKS = C.CFUNCTYPE(_int, _vp, _P(C.c_char_p), _sz, _i32, _P(TrackResult), _P(_i32), _vp)
NODE_TRACKS_ERROR = C.CFUNCTYPE(_vp, _vp, _sz, _vp)  # const char *: the callee keeps the bytes alive
NODE_LAST_ERROR = C.CFUNCTYPE(_vp, _vp, _vp)


class NodeBackend(C.Structure):
    _fields_ = [("open", NODE_OPEN), ("close", NODE_CLOSE), ("album_begin", NODE_ALBUM_BEGIN), ("album_pack", NODE_ALBUM_PACK),
                ("tracks", NODE_TRACKS), ("tracks_error", NODE_TRACKS_ERROR), ("last_error", NODE_LAST_ERROR), ("user", _vp)]


_lib = None


def load():
    """Load libmp3rgain_amd.so from the package directory (built by __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C mp3rgain_amd/csrc). There is no fallback path."
            )
        # When PyTorch shares the process (tests, bench.py), let it bring its bundled HIP runtime in
        # first: two HIP runtimes in one process cannot both own the device.
        import importlib.util
        import sys

        # A process that will never import torch (the command line, `python -m mp3rgain_amd`) says so with
        # MP3RGAIN_AMD_STANDALONE=1 and saves the 1.5 s that import costs: the library then runs on the system's HIP runtime.
        import os

        standalone = os.environ.get("MP3RGAIN_AMD_STANDALONE") == "1"
        if not standalone and "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
        L = C.CDLL(str(LIB_PATH))
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
