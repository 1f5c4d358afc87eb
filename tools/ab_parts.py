#!/usr/bin/env python3
"""Album parts against the plain route, interleaved call by call in one process (GPU box): rg_analyze_album / rg_analyze_tracks
over 256 three-minute files of three streams; per configuration the C call's duration in ms (median, and the list).
    python tools/ab_parts.py [calls]"""
import os, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401
import mp3rgain_amd as rg
from mp3rgain_amd import mp3dec
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 9
# per configuration: tuning keys of the context (10: album parts 1 = never / 2 = on, 11: threshold + 1 of the copy-bound rule;
# the RG_ALBUM_PARTS / RG_PARTS_MIN_BYTES_PER_UNIT environment is only what a context reads when it is created)
CONFIGS = {"plain": {10: 1}, "parts": {10: 2}}
for extra in os.environ.get("AB_EXTRA", "").split(";"):  # e.g. AB_EXTRA="parts_all:11=1"
    if extra:
        name, kv = extra.split(":")
        CONFIGS[name] = dict(CONFIGS["parts"], **{int(x.split("=")[0]): int(x.split("=")[1]) for x in kv.split(",")})
keys = sorted({k for c in CONFIGS.values() for k in c})
an = rg.Analyzer(0)
only = os.environ.get("AB_STREAM", "")
for label, src in (("128k", "tests/golden/mp3/dense_44k_joint_128.mp3"), ("vbr", "tests/golden/fixtures/test_vbr.mp3"), ("320k", "tests/golden/mp3/v1_44k_stereo_long.mp3")):
    if only and only != label:
        continue
    data = (ROOT / src).read_bytes()
    info = mp3dec.scan(data)
    body = data[int(info.first_frame_offset):]
    one = mp3dec.scan(body)
    stream = body * max(1, int(180 / (one.frames / one.sample_rate)))
    tmp = Path(tempfile.mkdtemp())
    files = []
    for k in range(256):
        p = tmp / f"t{k:04d}.mp3"; p.write_bytes(stream); files.append(p)
    for mode in ("album", "tracks"):
        f = an.analyze_album_files if mode == "album" else an.analyze_track_files
        res = {n: [] for n in CONFIGS}
        for rep in range(calls + 2):
            for name, env in CONFIGS.items():
                for k in keys:
                    an.set_tuning(k, env.get(k, 0))
                tm = {}
                f(files, timing=tm)
                if rep >= 2: res[name].append(tm["c_call_seconds"] * 1e3)
        print(f"{label:5s} {mode:6s}: " + " | ".join(f"{n} median {sorted(v)[len(v) // 2]:.2f} (" + " ".join(f"{x:.1f}" for x in v) + ")" for n, v in res.items()), flush=True)
    for p in files: p.unlink()
