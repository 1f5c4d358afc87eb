#!/usr/bin/env python3
"""PCIe-inclusive rate of the synchronous host-buffer entry point (rg_analyze_pcm_batch with host PCM):
the number DESIGN.md quotes next to the HBM-resident `value` of bench.py.  Run on the GPU box."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import mp3rgain_amd as rg  # noqa: E402
from oracle import pyoracle as po  # noqa: E402  (only to synthesise the input on the host)

rate, frames = 44100, 44100 * 600
l, r = po.synth_f32(0x5EED0000, 0, rate, frames), po.synth_f32(0x5EED0000, 1, rate, frames)
an = rg.Analyzer(0)
tr = rg.PcmTrack([l, r], rate)
arena, descs = rg.replaygain.pack_tracks([tr])
from mp3rgain_amd import _capi  # noqa: E402

out = (_capi.TrackResult * 1)()
lib = _capi.load()
for _ in range(3):
    lib.rg_analyze_pcm_batch(an.handle, descs, 1, arena.ctypes.data, arena.nbytes, 0, out, None)
t0 = time.perf_counter()
K = 10
for _ in range(K):
    lib.rg_analyze_pcm_batch(an.handle, descs, 1, arena.ctypes.data, arena.nbytes, 0, out, None)
dt = (time.perf_counter() - t0) / K
print(f"host-buffer (pageable, H2D inclusive) 10-min stereo track: {dt*1e3:.2f} ms -> {frames/dt/1e9:.2f} G stereo samples/s, "
      f"{arena.nbytes/dt/1e9:.1f} GB/s over PCIe; loudness {out[0].loudness_db:.2f} dB")
