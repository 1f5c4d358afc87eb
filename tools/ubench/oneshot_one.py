#!/usr/bin/env python3
"""N synchronous rg_analyze_pcm_batch calls over resident equal tracks at a forced (L, m); for rocprofv3 passes.
    python tools/ubench/oneshot_one.py <tracks> <minutes> <m> [calls] [L]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import _capi  # noqa: E402

NT = int(sys.argv[1]); minutes = float(sys.argv[2]); M = int(sys.argv[3])
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 6
L = int(sys.argv[5]) if len(sys.argv) > 5 else 2205
rate = 44100
frames = int(round(minutes * 60 * rate))
an = rg.Analyzer(0)
if M:
    an.set_tuning(1, L)
    an.set_tuning(4, M)
pcm = torch.empty((NT, 2, frames), dtype=torch.float32, device="cuda")
d = (_capi.TrackDesc * NT)()
for t in range(NT):
    for c in range(2):
        an.synth_fill_device(pcm[t, c].data_ptr(), 0x5EED0000 + t, c, rate, 0, frames)
    d[t].offset_bytes, d[t].frames, d[t].sample_rate, d[t].channels, d[t].format = t * 2 * frames * 4, frames, rate, 2, 0
torch.cuda.synchronize()
ms = []
for _ in range(calls):
    t0 = time.perf_counter()
    an.analyze_device(d, NT, pcm.data_ptr(), pcm.numel() * 4)
    ms.append((time.perf_counter() - t0) * 1e3)
print(f"m = {M}: min {min(ms):.3f} ms, calls {['%.2f' % v for v in ms]}")
