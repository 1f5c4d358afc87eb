#!/usr/bin/env python3
"""Per-block stage timeline of rg_tm_fix_kernel (diagnostic).  Usage on the GPU box:
python tools/ubench/fix_timeline.py [L] [tracks] [minutes]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import _capi  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 0
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 1
MIN = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
rate, frames = 44100, int(44100 * 60 * MIN)
an = rg.Analyzer(0)
if L:
    an.set_tuning(1, L)
an.set_tuning(3, 1)
pcm = torch.empty((NT, 2, frames), dtype=torch.float32, device="cuda")
d = (_capi.TrackDesc * NT)()
for t in range(NT):
    for c in range(2):
        an.synth_fill_device(pcm[t, c].data_ptr(), 0x5EED0000 + t, c, rate, 0, frames)
    d[t].offset_bytes, d[t].frames, d[t].sample_rate, d[t].channels, d[t].format = t * 2 * frames * 4, frames, rate, 2, 0
raw = C.CDLL(str(_capi.LIB_PATH))
raw.rg_tm_set_fix_debug_buffer.argtypes = [C.c_void_p]
nb = 1 << 17
dbg = torch.zeros(nb * 8, dtype=torch.int64, device="cuda")
for _ in range(3):
    an.enqueue_device(d, NT, pcm.data_ptr(), pcm.numel() * 4)
an.collect(NT)
raw.rg_tm_set_fix_debug_buffer(dbg.data_ptr())
an.enqueue_device(d, NT, pcm.data_ptr(), pcm.numel() * 4)
an.collect(NT)
raw.rg_tm_set_fix_debug_buffer(None)
a = dbg.cpu().numpy().reshape(-1, 8)
a = a[a[:, 0] != 0]
t0 = a[:, 0].min()
us = (a - t0) / 100.0
print(f"blocks {len(a)}; kernel span (first start .. last stamp) {us[a != 0].max():.1f} us")
names = ["start", "records loaded", "scan done", "segment sums", "histogram atomics", "arrival counter", "finisher: acquire", "finisher: result"]
for k, n in enumerate(names):
    col = us[:, k][a[:, k] != 0]
    if len(col):
        print(f"  {n:22s} n={len(col):6d}  min {col.min():7.1f}  med {np.median(col):7.1f}  max {col.max():7.1f} us")
