#!/usr/bin/env python3
"""The same album call over and over (GPU box): does its duration depend on which pipeline stream it lands on?
    python tools/seq_album.py [calls]   (RG_ALBUM_PARTS, MP3_RATE_STREAM as in tools/mp3_rate.py)"""
import os, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401
import mp3rgain_amd as rg
from mp3rgain_amd import mp3dec
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 16
an = rg.Analyzer(0)
for label, src in (("128k", "tests/golden/mp3/dense_44k_joint_128.mp3"), ("vbr", "tests/golden/fixtures/test_vbr.mp3"), ("320k", "tests/golden/mp3/v1_44k_stereo_long.mp3")):
    data = (ROOT / src).read_bytes()
    info = mp3dec.scan(data)
    body = data[int(info.first_frame_offset):]
    one = mp3dec.scan(body)
    stream = body * max(1, int(180 / (one.frames / one.sample_rate)))
    tmp = Path(tempfile.mkdtemp())
    files = []
    for k in range(int(os.environ.get("NFILES", "256"))):
        p = tmp / f"t{k:04d}.mp3"; p.write_bytes(stream); files.append(p)
    for mode in ("album", "tracks"):
        f = an.analyze_album_files if mode == "album" else an.analyze_track_files
        ms = []
        for rep in range(calls + 3):
            tm = {}
            f(files, timing=tm)
            if rep >= 3: ms.append(tm["c_call_seconds"] * 1e3)
        print(f"{label:5s} {mode:6s}: " + " ".join(f"{x:.1f}" for x in ms), flush=True)
    for p in files: p.unlink()
