#!/usr/bin/env python3
"""Random albums through the file route with album parts forced, by rule, and off (GPU box): results and errors must be the
plain route's, field by field.  Files: the golden / fixture streams, whole, truncated or with a few bytes changed; staging
blocks of random size so that an album is anything from one chunk to dozens.
    python tools/fuzz_album_parts.py [albums] [seed]"""
import os, random, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401
import mp3rgain_amd as rg

n_albums = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260930
rnd = random.Random(seed)
srcs = sorted((ROOT / "tests/golden/mp3").glob("*.mp3")) + sorted((ROOT / "tests/golden/fixtures").glob("*.mp3"))
blobs = [p.read_bytes() for p in srcs]
tmp = Path(tempfile.mkdtemp())
an = rg.Analyzer(0)
an.set_kernel(0)


def run(f, files):
    try:
        r = f(files)
    except rg.ReplayGainError as e:
        return ("error", e.code, str(e))
    if isinstance(r, list):
        return [(x.code, str(x)) if isinstance(x, rg.ReplayGainError) else (x.loudness_db, x.gain_db, x.peak, x.sample_rate, x.windows, x.file_type) for x in r]
    return [(t.loudness_db, t.gain_db, t.peak, t.sample_rate, t.windows, t.file_type) for t in r.tracks] + [(r.album_loudness_db, r.album_gain_db, r.album_peak)]


bad = 0
with_parts = 0
for a in range(n_albums):
    k = rnd.choice([1, 2, 3, 5, 8, 13, 21, 34])
    files = []
    for i in range(k):
        b = blobs[rnd.randrange(len(blobs))]
        reps = rnd.choice([1, 1, 1, 2, 5])
        what = rnd.random()
        if what < 0.06:
            b = b[:rnd.randrange(0, len(b))]                       # truncated (maybe empty)
        elif what < 0.14:
            m = bytearray(b)
            for _ in range(rnd.choice([1, 3, 20])):
                m[rnd.randrange(len(m))] = rnd.randrange(256)       # damaged
            b = bytes(m)
        p = tmp / f"a{a:04d}_{i:02d}.mp3"
        p.write_bytes(b * reps)
        files.append(p)
    if rnd.random() < 0.05:
        files.insert(rnd.randrange(len(files) + 1), tmp / "missing.mp3")
    stage = rnd.choice(["16384", "65536", "262144", None])
    an.set_tuning(12, int(stage) if stage else 0)  # staging block bytes (tuning keys 10-13: the environment is read at rg_create only)
    an.set_tuning(7, rnd.choice([0, 1, 2]))       # loader threads: with one or two the device waits for the host, and the chunks it waited for are parts
    an.set_tuning(10, 1)
    want_album, want_tracks = run(an.analyze_album_files, files), run(an.analyze_track_files, files)
    an.set_tuning(10, 2)
    for rule in ("0", "120", "300", "1000000000"):
        an.set_tuning(11, int(rule) + 1)
        got_album, got_tracks = run(an.analyze_album_files, files), run(an.analyze_track_files, files)
        if got_album != want_album or got_tracks != want_tracks:
            bad += 1
            print(f"album {a} (rule {rule}, stage {stage}, {len(files)} files): DIFFERS\n  want {want_album}\n  got  {got_album}", flush=True)
            break
    for p in files:
        if p.exists():
            p.unlink()
print(f"{n_albums} random albums (seed {seed}; whole, truncated, damaged and missing files; staging blocks of 16 KB to 128 MB; loader threads all / 1 / 2; parts forced, by rule, at 300 bytes per unit, for starved chunks only): {bad} differ from the plain route (album and track mode, results and errors)")
sys.exit(1 if bad else 0)
