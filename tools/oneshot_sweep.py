#!/usr/bin/env python3
"""A synchronous album call (rg_analyze_album_pcm, PCM resident on the device) over N three-minute tracks: the library's own
choice of (segment, windows per lane) against forced ones.  One batch is in flight in such a call, so what matters is how
evenly its blocks fill the CUs' rounds (rg_enqueue.hip: cost model, one_shot).

    python tools/oneshot_sweep.py [tracks] [minutes]"""
import ctypes as C
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import _capi  # noqa: E402

ntr = int(sys.argv[1]) if len(sys.argv) > 1 else 256
minutes = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
RATE = 44100
frames = int(round(minutes * 60 * RATE))
an = rg.Analyzer(0)
lib = an._lib
pcm = torch.empty(2 * frames * ntr, dtype=torch.float32, device="cuda")
descs = (_capi.TrackDesc * ntr)()
for t in range(ntr):
    off = 2 * frames * t
    for c in range(2):
        an.synth_fill_device(pcm[off + c * frames:].data_ptr(), 0x5EED0000 + t, c, RATE, 0, frames)
    descs[t].offset_bytes = off * 4
    descs[t].frames = frames
    descs[t].sample_rate = RATE
    descs[t].channels = 2
    descs[t].format = _capi.FMT_F32_PLANAR
torch.cuda.synchronize()
out = (_capi.TrackResult * ntr)()
alb = _capi.AlbumResult()


def run():
    t0 = time.perf_counter()
    rc = lib.rg_analyze_album_pcm(an._ctx, descs, ntr, C.c_void_p(pcm.data_ptr()), C.c_size_t(pcm.numel() * 4), 1, out, C.byref(alb), None)
    assert rc == 0, lib.rg_last_error(an._ctx)
    return (time.perf_counter() - t0) * 1e3


for _ in range(12):  # every pipeline slot has been through its first use (allocations) before anything is timed
    run()
ref = None
for label, seg, m in ([("library's choice", 0, 0)] + [(f"L = {L}", L, 1) for L in (245, 441, 735)]
                     + [(f"W x {m}", 2205, m) for m in (1, 2, 4, 5, 8, 10, 12, 14, 16, 20, 24, 30, 36, 37, 38, 40, 45)]):
    an.set_tuning(1, seg)
    an.set_tuning(4, m)
    run()
    ms = min(run() for _ in range(5))
    key = (alb.album_loudness_db, alb.album_peak, out[0].loudness_db)
    ref = ref or key
    print(f"{label:18s} {ms:7.3f} ms  = {ntr * frames / ms / 1e6:7.1f} G stereo samples/s   {'same result' if key == ref else 'RESULT DIFFERS'}")
