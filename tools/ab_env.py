#!/usr/bin/env python3
"""Two settings of one environment variable, interleaved call by call in one process (GPU box): rg_analyze_album over 256
three-minute files of three streams.   python tools/ab_env.py VAR A B [calls]"""
import os, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401
import mp3rgain_amd as rg
from mp3rgain_amd import mp3dec
var, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 9
an = rg.Analyzer(0)
for label, src in (("vbr", "tests/golden/fixtures/test_vbr.mp3"), ("128k", "tests/golden/mp3/dense_44k_joint_128.mp3"), ("320k", "tests/golden/mp3/v1_44k_stereo_long.mp3")):
    data = (ROOT / src).read_bytes()
    info = mp3dec.scan(data)
    body = data[int(info.first_frame_offset):]
    one = mp3dec.scan(body)
    stream = body * max(1, int(180 / (one.frames / one.sample_rate)))
    tmp = Path(tempfile.mkdtemp())
    files = []
    for k in range(256):
        p = tmp / f"t{k:04d}.mp3"; p.write_bytes(stream); files.append(p)
    res = {va: [], vb: []}
    for rep in range(calls + (12 if label == "vbr" else 2)):
        for v in (va, vb):
            os.environ[var] = v
            tm = {}
            r = an.analyze_album_files(files, timing=tm)
            if rep >= (12 if label == "vbr" else 2): res[v].append(tm["c_call_seconds"] * 1e3)
    print(f"{label:5s}: " + " | ".join(f"{var}={n} median {sorted(v)[len(v) // 2]:.2f} (" + " ".join(f"{x:.1f}" for x in v) + ")" for n, v in res.items()) + f"  loudness {r.album_loudness_db:.2f}", flush=True)
    for p in files: p.unlink()
