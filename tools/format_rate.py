#!/usr/bin/env python3
"""Device-resident throughput by sample format (f32 / s16 / s32 planar), one 10-minute 44.1 kHz stereo track per batch."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import _capi  # noqa: E402

an = rg.Analyzer(0)
rate, frames = 44100, 44100 * 600
f32 = torch.empty((2, frames), dtype=torch.float32, device="cuda")
for c in range(2):
    an.synth_fill_device(f32[c].data_ptr(), 77, c, rate, 0, frames)
torch.cuda.synchronize()
bufs = {"f32": (f32, _capi.FMT_F32_PLANAR), "s16": ((f32 * 32767).round().to(torch.int16), _capi.FMT_S16_PLANAR),
        "s32": ((f32.double() * 2147483647).round().to(torch.int32), _capi.FMT_S32_PLANAR)}
for name, (buf, fmt) in bufs.items():
    d = (_capi.TrackDesc * 1)()
    d[0].offset_bytes, d[0].frames, d[0].sample_rate, d[0].channels, d[0].format = 0, frames, rate, 2, fmt
    nbytes = buf.numel() * buf.element_size()
    for _ in range(20):
        an.enqueue_device(d, 1, buf.data_ptr(), nbytes)
    an.collect(1)
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    for _ in range(K):
        an.enqueue_device(d, 1, buf.data_ptr(), nbytes)
    r = an.collect(1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{name}: {dt * 1e6:7.1f} us per 10-min track, {frames / dt / 1e9:7.1f} G stereo samples/s, loudness {r[0].loudness_db:.2f}")
