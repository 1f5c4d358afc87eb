#!/usr/bin/env python3
"""Per-wave timeline of rg_tm_main_kernel for a batch of equal tracks in ONE synchronous call (diagnostic).
Usage on the GPU box: python tools/ubench/timeline_batch.py <tracks> <minutes> <m> [L]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import _capi  # noqa: E402

NT = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
minutes = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
M = int(sys.argv[3]) if len(sys.argv) > 3 else 0
L = int(sys.argv[4]) if len(sys.argv) > 4 else 2205
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
raw = C.CDLL(str(_capi.LIB_PATH))
raw.rg_tm_set_debug_buffer.argtypes = [C.c_void_p]
nw = 1 << 18
dbg = torch.zeros(nw * 6, dtype=torch.int64, device="cuda")
for _ in range(10):
    an.analyze_device(d, NT, pcm.data_ptr(), pcm.numel() * 4)
raw.rg_tm_set_debug_buffer(dbg.data_ptr())
an.analyze_device(d, NT, pcm.data_ptr(), pcm.numel() * 4)
raw.rg_tm_set_debug_buffer(None)
a = dbg.cpu().numpy().reshape(-1, 6)
a = a[a[:, 0] != 0]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0  # wall_clock64 ticks at 100 MHz -> us
hw = a[:, 2] & 0xFFFFFFFF
xcc = (a[:, 2] >> 32) & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
print(f"m = {M}  waves {len(a)}  kernel span {en.max() / 1000:.3f} ms")
print(f"start: min {st.min():.1f} med {np.median(st):.1f} p90 {np.quantile(st, 0.9):.1f} max {st.max():.1f} us; "
      f"duration ms: min {(en - st).min() / 1000:.3f} med {np.median(en - st) / 1000:.3f} p90 {np.quantile(en - st, 0.9) / 1000:.3f} max {(en - st).max() / 1000:.3f}")
key = xcc * 1000 + se * 100 + sh * 50 + cu
uk, cnt = np.unique(key, return_counts=True)
print(f"distinct (xcc,se,sh,cu) = {len(uk)}; waves per CU: min {cnt.min()} med {int(np.median(cnt))} max {cnt.max()}; CUs with > 12 waves: {int((cnt > 12).sum())}")
late = st > 1000.0
print(f"waves that started later than 1 ms: {int(late.sum())} (their start: {np.sort(np.unique((st[late] / 1000).round(1)))[:10]} ms)")
for q in (0.1, 0.5, 0.9, 0.99, 1.0):
    print(f"  {int(q * 100):3d}% of waves finished by {np.quantile(en, q) / 1000:8.3f} ms")
dur_us = (a[:, 1] - a[:, 0]) / 100.0
cyc = (a[:, 5] - a[:, 4]).astype(np.float64)
ok = dur_us > 5
print(f"shader clock: median {np.median(cyc[ok] / dur_us[ok]) / 1000:.3f} GHz (min {np.min(cyc[ok] / dur_us[ok]) / 1000:.3f}, max {np.max(cyc[ok] / dur_us[ok]) / 1000:.3f})")
print("per-XCD wave counts:", np.bincount(xcc.astype(int), minlength=8))
print("per-XCD median duration ms:", [round(float(np.median(dur_us[xcc == x])) / 1000, 3) for x in range(8) if (xcc == x).any()])
