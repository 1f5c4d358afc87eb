#!/usr/bin/env python3
"""Throughput across sample rates and channel counts (device-resident synthetic PCM), auto mode (variant 2 at every stable
rate since round 3; until then the order-faithful kernel at 64 / 96 kHz).   python tools/rate_sweep.py [tracks] [minutes]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import _capi  # noqa: E402

ntr = int(sys.argv[1]) if len(sys.argv) > 1 else 1
minutes = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
an = rg.Analyzer(0)
import os  # noqa: E402
rates = [int(x) for x in os.environ["RATES"].split(",")] if os.environ.get("RATES") else (96000, 64000, 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000)
if os.environ.get("TM_SEGMENT"):  # tuning key 1: segment length; key 4: most windows per lane
    an.set_tuning(1, int(os.environ["TM_SEGMENT"]))
if os.environ.get("TM_WINDOWS"):
    an.set_tuning(4, int(os.environ["TM_WINDOWS"]))
for rate in rates:
    for ch in ((2,) if os.environ.get("STEREO_ONLY") else (2, 1)):
        frames = int(rate * 60 * minutes)
        pcm = torch.empty((ntr, ch, frames), dtype=torch.float32, device="cuda")
        d = (_capi.TrackDesc * ntr)()
        for t in range(ntr):
            for c in range(ch):
                an.synth_fill_device(pcm[t, c].data_ptr(), 77 + t, c, rate, 0, frames)
            d[t].offset_bytes, d[t].frames, d[t].sample_rate, d[t].channels, d[t].format = t * ch * frames * 4, frames, rate, ch, 0
        K = 40 if ntr == 1 else 6
        # Untimed: every one of the context's eight pipeline slots sizes its buffers for this batch, and the device gets a
        # quarter of a second of this work before anything is timed -- in a fresh process its first ~130 ms of heavy kernels run
        # three times slower than the steady state (tools/ubench/enqueue_host_time.py: 1100 tracks at 8 kHz, 68 ms for the first
        # six batches, 22 ms for every six after them), which with two warm-up enqueues read as a cliff at 800+ tracks.
        t_w = time.perf_counter()
        while True:
            for _ in range(8):
                an.enqueue_device(d, ntr, pcm.data_ptr(), pcm.numel() * 4)
            an.collect(ntr)
            torch.cuda.synchronize()
            if time.perf_counter() - t_w > 0.25:
                break
        dt = 1e9  # best of three timed repetitions (a tenant of the box, a clock step: one in ten repetitions is 2-3x off)
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(K):
                an.enqueue_device(d, ntr, pcm.data_ptr(), pcm.numel() * 4)
            r = an.collect(ntr)
            torch.cuda.synchronize()
            dt = min(dt, (time.perf_counter() - t0) / K)
        print(f"{rate:6d} Hz ch={ch}: {dt*1e6:10.1f} us per batch of {ntr} x {minutes:g} min, {ntr*frames/dt/1e9:7.1f} G frames/s, loudness {r[0].loudness_db:.2f}", flush=True)
        del pcm
