#!/usr/bin/env python3
"""Host-arena throughput of the synchronous API (PCIe inclusive; never bench.py's `value`).

    python tools/ingest_rate.py [--tracks 64] [--minutes 3] [--chunk-mib 0]

Reports, for a batch of synthetic 44.1 kHz stereo f32 tracks in PAGEABLE host memory (what a decoder hands over) and in
pinned memory: the raw hipMemcpy H2D rate of the same bytes, the one-shot path (chunk larger than the arena) and the
streamed path (tuning key 5).  Done-criterion of the round: streamed >= 0.9 x the raw copy rate."""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tracks", type=int, default=64)
ap.add_argument("--minutes", type=float, default=3.0)
ap.add_argument("--chunk-mib", type=int, default=512)
a = ap.parse_args()
frames = int(a.minutes * 60 * 44100)
n = a.tracks
an = rg.Analyzer(0)
lib = an._lib
bytes_per_track = 2 * frames * 4
arena_t = torch.empty(n * bytes_per_track // 4, dtype=torch.float32)
rng = np.random.default_rng(1)
base = (rng.standard_normal(2 * frames) * 0.1).astype(np.float32)
for t in range(n):
    arena_t[t * 2 * frames:(t + 1) * 2 * frames] = torch.from_numpy(base * (0.5 + 0.5 * t / n))
descs = (_capi.TrackDesc * n)()
for t in range(n):
    descs[t].offset_bytes = t * bytes_per_track
    descs[t].frames = frames
    descs[t].sample_rate = 44100
    descs[t].channels = 2
    descs[t].format = 0
out = (_capi.TrackResult * n)()
dev = torch.empty_like(arena_t, device="cuda")


def timed(fn, reps=3):
    fn()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


res = {}
for kind, host in (("pageable", arena_t), ("pinned", arena_t.pin_memory())):
    nbytes = host.numel() * 4
    res[kind, "raw_copy"] = nbytes / timed(lambda: dev.copy_(host, non_blocking=False)) / 1e9

    def run():
        rc = lib.rg_analyze_pcm_batch(an._ctx, descs, n, host.data_ptr(), nbytes, 0, out, None)
        assert rc == 0, rc

    an.set_tuning(5, 1 << 40 >> 10)  # one shot
    res[kind, "one_shot"] = nbytes / timed(run) / 1e9
    an.set_tuning(5, a.chunk_mib * 1024)
    res[kind, "streamed"] = nbytes / timed(run) / 1e9
    an.set_tuning(5, 0)
print(f"{n} x {a.minutes:g}-min tracks, {n * bytes_per_track / 1e9:.2f} GB, chunk {a.chunk_mib} MiB")
for (kind, what), v in res.items():
    print(f"  {kind:9s} {what:9s} {v:7.2f} GB/s  = {v / 8:6.3f} G stereo samples/s")
for kind in ("pageable", "pinned"):
    print(f"  {kind}: streamed / raw copy = {res[kind, 'streamed'] / res[kind, 'raw_copy']:.3f}, one-shot / raw copy = {res[kind, 'one_shot'] / res[kind, 'raw_copy']:.3f}")
