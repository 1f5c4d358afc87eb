#!/usr/bin/env python3
"""Fuzz the device MP3 routes against the host decoder (GPU box): mutated / truncated / spliced streams through
rg_mp3_decode_device on routes 3, 2 and 1; PCM, lengths and frame counts must be the host decoder's (PCM wherever the
host's is finite: damaged side information can ask for enormous gains).

    python tools/fuzz_mp3_routes.py [cases] [seed]
"""
import random
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401

import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import mp3dec  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
srcs = [p.read_bytes() for p in sorted((ROOT / "tests/golden/mp3").glob("*.mp3")) + sorted((ROOT / "tests/golden/fixtures").glob("*.mp3"))
        if p.stat().st_size < 70000]
an = rg.Analyzer(0)
same = skipped = no_audio = 0
for k in range(cases):
    d = bytearray(rng.choice(srcs))
    kind = rng.randrange(6)
    if kind == 0:
        for _ in range(rng.randint(1, 40)):
            d[rng.randrange(len(d))] = rng.randrange(256)
    elif kind == 1:
        d = d[:rng.randrange(8, len(d))]
    elif kind == 2:
        a = rng.randrange(len(d))
        del d[a:a + rng.randint(1, 1200)]
    elif kind == 3:
        a = rng.randrange(len(d))
        d[a:a] = bytes(rng.randrange(256) for _ in range(rng.randint(1, 300)))
    elif kind == 4:
        for _ in range(rng.randint(1, 8)):
            a = rng.randrange(len(d))
            d[a] ^= 1 << rng.randrange(8)
    else:  # splice two streams
        o = rng.choice(srcs)
        d = d[:rng.randrange(len(d))] + bytearray(o[rng.randrange(len(o)):])
    d = bytes(d)
    try:
        want, wi = mp3dec.decode(d)
    except mp3dec.Mp3DecodeError:
        no_audio += 1
        for route in (3, 2, 1):
            an.set_tuning(6, route)
            try:
                an.decode_mp3_device(d)
            except (rg.ReplayGainError, mp3dec.Mp3DecodeError):
                continue
            raise SystemExit(f"case {k}: host finds no audio, route {route} decodes")
        continue
    ok = np.isfinite(want)
    for route in (3, 2, 1):
        an.set_tuning(6, route)
        got, gi = an.decode_mp3_device(d)
        if (gi.frames, gi.audio_frames, gi.skipped_frames, gi.channels, gi.sample_rate) != (wi.frames, wi.audio_frames, wi.skipped_frames, wi.channels, wi.sample_rate):
            raise SystemExit(f"case {k} (kind {kind}): route {route} counts {gi.frames, gi.audio_frames, gi.skipped_frames} != host {wi.frames, wi.audio_frames, wi.skipped_frames}")
        if not np.array_equal(got[ok], want[ok]):
            raise SystemExit(f"case {k} (kind {kind}): route {route} PCM differs from the host decoder's")
    same += 1
    skipped += int(wi.skipped_frames > 0)
print(f"{cases} cases (seed {seed}): {same} identical to the host decoder on routes 3, 2 and 1 ({skipped} of them with dropped frames), "
      f"{no_audio} without audio (all routes refuse)")
