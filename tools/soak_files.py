#!/usr/bin/env python3
"""File-route soak (GPU box): albums of 1 / 12 / 40 / 256 files, album and track mode, parts forced / by rule / off, again and
again on one context, with all / one / two loader threads in turn: results identical every time, free device memory steady after the first rounds.
    python tools/soak_files.py [rounds]"""
import os, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import mp3rgain_amd as rg
from mp3rgain_amd import mp3dec
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
an = rg.Analyzer(0)
tmp = Path(tempfile.mkdtemp())
sets = {}
for label, src, n in (("1x320k", "tests/golden/mp3/v1_44k_stereo_long.mp3", 1), ("12x128k", "tests/golden/mp3/dense_44k_joint_128.mp3", 12),
                      ("40xvbr", "tests/golden/fixtures/test_vbr.mp3", 40), ("256x320k", "tests/golden/mp3/v1_44k_stereo_long.mp3", 256)):
    data = (ROOT / src).read_bytes()
    info = mp3dec.scan(data)
    body = data[int(info.first_frame_offset):]
    one = mp3dec.scan(body)
    files = []
    for k in range(n):
        reps = max(1, int((60 + 13 * (k % 11)) / (one.frames / one.sample_rate)))  # 1 to 3 minutes
        p = tmp / f"{label}_{k:03d}.mp3"
        p.write_bytes(body * reps)
        files.append(p)
    sets[label] = files
ref = {}
free0 = free_mid = None
for r in range(rounds):
    an.set_tuning(7, (0, 1, 2)[r % 3])  # loader threads: all / one / two (with few the device waits for the host: other chunks become parts)
    for label, files in sets.items():
        for mode in ("album", "tracks"):
            for env in ({10: 1, 11: 0}, {10: 2, 11: 1}, {10: 2, 11: 0}):  # tuning keys: parts never | every chunk a part | the default rule
                for k, v in env.items():
                    an.set_tuning(k, v)
                res = an.analyze_album_files(files) if mode == "album" else an.analyze_track_files(files)
                tr = res.tracks if mode == "album" else res
                key = [(t.loudness_db, t.peak, t.windows) for t in tr] + ([(res.album_loudness_db, res.album_peak)] if mode == "album" else [])
                if (label, mode) not in ref:
                    ref[(label, mode)] = key
                elif ref[(label, mode)] != key:
                    print(f"round {r}: {label} {mode} {env}: results changed")
                    sys.exit(1)
    free, total = torch.cuda.mem_get_info()
    if r == 2:
        free0 = free
    if r == rounds // 2:
        free_mid = free
    if r in (0, 2, rounds // 2, rounds - 1):
        print(f"round {r}: free device memory {free / 2**30:.3f} GiB", flush=True)
# Which chunks become parts depends on timing (a chunk the device had to wait for is one), and every pipeline slot's buffers grow to
# the largest part that slot has seen: a bounded warm-up (at most the plain route's buffers per slot), so "steady" is judged over
# the second half of the run.
print(f"{rounds} rounds x {len(sets)} albums x 2 modes x 3 settings: identical results; free memory after round 2 {free0 / 2**30:.3f} GiB, "
      f"after round {rounds // 2} {free_mid / 2**30:.3f} GiB, at the end {free / 2**30:.3f} GiB")
sys.exit(0 if free >= free_mid - (64 << 20) else 2)
