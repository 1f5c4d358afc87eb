#!/usr/bin/env python3
"""Fuzz the device MP3 routes against the host decoder (GPU box): mutated / truncated / spliced streams through
rg_mp3_decode_device on routes 3 and 2; PCM, lengths and frame counts must be the host decoder's wherever its output is
finite.  Frames whose channel count differs from the stream's are dropped by the device routes and spread / truncated by
the one-shot host decoder (documented): such cases are counted, not compared.

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
same = skipped = no_audio = chan = 0
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
        for route in (3, 2):
            an.set_tuning(6, route)
            try:
                an.decode_mp3_device(d)
            except (rg.ReplayGainError, mp3dec.Mp3DecodeError):
                continue
            raise SystemExit(f"case {k}: host finds no audio, route {route} decodes")
        continue
    outs = []
    for route in (3, 2):
        an.set_tuning(6, route)
        got, gi = an.decode_mp3_device(d)
        outs.append((got, gi))
    (g3, i3), (g2, i2) = outs
    if (i3.frames, i3.audio_frames, i3.skipped_frames) != (i2.frames, i2.audio_frames, i2.skipped_frames) or not np.array_equal(g3, g2, equal_nan=True):
        raise SystemExit(f"case {k} (kind {kind}): routes 3 and 2 differ")
    if i3.frames != wi.frames:
        chan += 1
        continue
    ok = np.isfinite(want)
    if not np.array_equal(g3[ok], want[ok]):
        raise SystemExit(f"case {k} (kind {kind}): device PCM differs from the host decoder's")
    same += 1
    skipped += int(wi.skipped_frames > 0)
print(f"{cases} cases (seed {seed}): {same} identical to the host decoder ({skipped} of them with dropped frames), {chan} with frames of another "
      f"channel count (device routes agree with each other), {no_audio} without audio (all routes refuse)")
