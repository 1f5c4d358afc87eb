#!/usr/bin/env python3
"""Determinism soak (GPU box): the same resident batch analysed again and again -- synchronous calls and the pipelined
enqueue / collect pair, every pipeline slot -- must give bit-identical results and histograms every time.
    python tools/soak.py [tracks] [minutes] [repeats]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import _capi  # noqa: E402

NT = int(sys.argv[1]) if len(sys.argv) > 1 else 200
minutes = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
rate = 44100
rng = np.random.default_rng(5)
frames = [int(round(minutes * 60 * rate * f)) for f in rng.uniform(0.4, 1.0, NT)]  # ragged lengths: partial windows, short lanes
an = rg.Analyzer(0)
total = sum(2 * f for f in frames)
pcm = torch.empty(total, dtype=torch.float32, device="cuda")
d = (_capi.TrackDesc * NT)()
off = 0
for t, f in enumerate(frames):
    for c in range(2):
        an.synth_fill_device(pcm[off + c * f:].data_ptr(), 0x5EED0000 + t, c, rate, 0, f)
    d[t].offset_bytes, d[t].frames, d[t].sample_rate, d[t].channels, d[t].format = off * 4, f, rate, 2, 0
    off += 2 * f
torch.cuda.synchronize()


def key(res, hist):
    return (tuple((r.loudness_db, r.peak, r.windows, r.flags) for r in res), hist.tobytes())


ref = key(*an.analyze_device(d, NT, pcm.data_ptr(), total * 4, want_hist=True))
bad = 0
for i in range(reps):
    if key(*an.analyze_device(d, NT, pcm.data_ptr(), total * 4, want_hist=True)) != ref:
        bad += 1
print(f"synchronous: {reps} repeats of {NT} ragged tracks, {bad} differ from the first")
ref2 = None
bad2 = 0
for i in range(reps):
    for _ in range(3):  # three batches in flight behind each other, the last one collected
        an.enqueue_device(d, NT, pcm.data_ptr(), total * 4)
    k = key(*an.collect(NT, want_hist=True))
    ref2 = ref2 or k
    bad2 += k != ref2
print(f"pipelined:   {reps} x 3 enqueues, {bad2} collected results differ from the first; same histograms as the synchronous call: {ref2[1] == ref[1]}")
sys.exit(1 if bad or bad2 or ref2[1] != ref[1] else 0)
