#!/usr/bin/env python3
"""Throughput sanity across sample rates / formats / channel counts (device-resident synthetic PCM)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import _capi  # noqa: E402

an = rg.Analyzer(0)
for rate in (96000, 64000, 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000):
    for ch in (2, 1):
        frames = rate * 600
        pcm = torch.empty((ch, frames), dtype=torch.float32, device="cuda")
        for c in range(ch):
            an.synth_fill_device(pcm[c].data_ptr(), 77, c, rate, 0, frames)
        d = (_capi.TrackDesc * 1)()
        d[0].offset_bytes, d[0].frames, d[0].sample_rate, d[0].channels, d[0].format = 0, frames, rate, ch, 0
        for _ in range(8):
            an.enqueue_device(d, 1, pcm.data_ptr(), pcm.numel() * 4)
        an.collect(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 40
        for _ in range(K):
            an.enqueue_device(d, 1, pcm.data_ptr(), pcm.numel() * 4)
        r = an.collect(1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        print(f"{rate:6d} Hz ch={ch}: {dt*1e6:8.1f} us per 10-min track, {frames/dt/1e9:7.1f} G frames/s, loudness {r[0].loudness_db:.2f}")
