#!/usr/bin/env python3
"""How much of a launch is waiting for HBM?  The same 1000 x 3 min synchronous call with every descriptor pointing at ONE track's
PCM (63.5 MB: it stays in the 256 MiB Infinity Cache) against the real batch (63.5 GB streamed).  Same instructions, same lanes.
    python tools/ubench/alias_tracks.py [tracks] [minutes]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import _capi  # noqa: E402

NT = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
minutes = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
rate = 44100
frames = int(round(minutes * 60 * rate))
an = rg.Analyzer(0)
pcm = torch.empty((NT, 2, frames), dtype=torch.float32, device="cuda")
for t in range(NT):
    for c in range(2):
        an.synth_fill_device(pcm[t, c].data_ptr(), 0x5EED0000 + t, c, rate, 0, frames)
torch.cuda.synchronize()
out = (_capi.TrackResult * NT)()
import os
_mode = os.environ.get("RG_ALIAS_MODE")  # "stream" / "alias": one mode only (counter passes)
_runs = {"stream": (("streamed from HBM", False),), "alias": (("one track's PCM for every descriptor", True),)}.get(
    _mode, (("streamed from HBM", False), ("one track's PCM for every descriptor", True)) * 2)
for label, alias in _runs:
    d = (_capi.TrackDesc * NT)()
    for t in range(NT):
        d[t].offset_bytes, d[t].frames, d[t].sample_rate, d[t].channels, d[t].format = (0 if alias else t * 2 * frames * 4), frames, rate, 2, 0
    an.timing_enable(True)
    for _ in range(4):
        an.analyze_device(d, NT, pcm.data_ptr(), pcm.numel() * 4, out=out)
    an.timing_read(reset=True)
    ms = []
    for _ in range(8):
        t0 = time.perf_counter()
        an.analyze_device(d, NT, pcm.data_ptr(), pcm.numel() * 4, out=out)
        ms.append((time.perf_counter() - t0) * 1e3)
    ks, kl, _ = an.timing_read(reset=True)
    print(f"{label:40s} call {min(ms):7.3f} ms (mean {sum(ms) / len(ms):7.3f}), main kernel {ks / max(1, kl):7.3f} ms")
