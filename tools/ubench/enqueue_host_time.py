#!/usr/bin/env python3
"""Host time of rg_enqueue_pcm_batch against the device's (GPU box): 8 kHz three-minute tracks resident in HBM, nine untimed
enqueues (every pipeline slot sizes its buffers), then REPS times [six timed enqueues + rg_collect].

    NS=700,1100 REPS=4 [SLOTS=n] [DUMMY_GB=x] python tools/ubench/enqueue_host_time.py

Written to chase a 3x slow-down of tools/rate_sweep.py at 800+ tracks: the enqueues cost 0.03-0.09 ms of host time whatever the
track count; what varies is the device's first ~100 ms of work in a fresh process."""
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import _capi  # noqa: E402

rate, minutes = int(os.environ.get("RATE", "8000")), 3.0
an = rg.Analyzer(0)
if os.environ.get("SLOTS"):
    an.set_tuning(3, int(os.environ["SLOTS"]))
if os.environ.get("DUMMY_GB"):
    _dummy = torch.empty(int(float(os.environ["DUMMY_GB"]) * 2 ** 30), dtype=torch.uint8, device="cuda")
frames = int(rate * 60 * minutes)
for n in [int(x) for x in os.environ.get("NS", "700,770,790,1100").split(",")]:
    pcm = torch.empty((n, 2, frames), dtype=torch.float32, device="cuda")
    d = (_capi.TrackDesc * n)()
    for t in range(n):
        for c in range(2):
            an.synth_fill_device(pcm[t, c].data_ptr(), 77 + t, c, rate, 0, frames)
        d[t].offset_bytes, d[t].frames, d[t].sample_rate, d[t].channels, d[t].format = t * 2 * frames * 4, frames, rate, 2, 0
    torch.cuda.synchronize()
    t_w = time.perf_counter()
    for _ in range(9):
        an.enqueue_device(d, n, pcm.data_ptr(), pcm.numel() * 4)
    an.collect(n)
    torch.cuda.synchronize()
    print(f"{n} tracks: nine untimed enqueues + collect {1e3 * (time.perf_counter() - t_w):.1f} ms", flush=True)
    for rep in range(int(os.environ.get("REPS", "1"))):
        ts = []
        t0 = time.perf_counter()
        for _ in range(6):
            a = time.perf_counter()
            an.enqueue_device(d, n, pcm.data_ptr(), pcm.numel() * 4)
            ts.append((time.perf_counter() - a) * 1e3)
        a = time.perf_counter()
        an.collect(n)
        tc = (time.perf_counter() - a) * 1e3
        torch.cuda.synchronize()
        print(f"{n} tracks, repetition {rep}: enqueue host ms {['%.2f' % v for v in ts]} collect {tc:.2f} total {(time.perf_counter() - t0) * 1e3:.2f}"
              f" = {6 * n * frames / (time.perf_counter() - t0) / 1e9:.0f} G frames/s", flush=True)
    del pcm
