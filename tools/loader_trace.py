#!/usr/bin/env python3
"""RG_TRACE_FILES=1 THREADS=1,2,4 python tools/loader_trace.py [files]  (GPU box) -- the file route's own trace (chunks as they become
ready, the loaders' summed read / compact / wait / copy times) for rg_analyze_album over copies of the three bench streams, with
the loader threads limited (tuning key 7).  stderr carries the trace."""
import os
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401

import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import mp3dec  # noqa: E402

nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 256
an = rg.Analyzer(0)
if os.environ.get("PARTS"):
    an.set_tuning(10, int(os.environ["PARTS"]))
for label, src in (("vbr_fixture", "tests/golden/fixtures/test_vbr.mp3"), ("dense128_joint", "tests/golden/mp3/dense_44k_joint_128.mp3"),
                   ("dense320", "tests/golden/mp3/v1_44k_stereo_long.mp3")):
    data = (ROOT / src).read_bytes()
    body = data[int(mp3dec.scan(data).first_frame_offset):]
    one = mp3dec.scan(body)
    stream = body * max(1, int(180 / (one.frames / one.sample_rate)))
    tmp = Path(tempfile.mkdtemp(prefix="rg_lt_"))
    files = []
    for k in range(nfiles):
        p = tmp / f"t{k:04d}.mp3"
        p.write_bytes(stream)
        files.append(p)
    an.set_tuning(7, 0)
    for _ in range(3):
        an.analyze_album_files(files)
    for th in [int(x) for x in os.environ.get("THREADS", "1,2,4").split(",")]:
        an.set_tuning(7, th)
        for rep in range(3):
            print(f"== {label}, {th} loader threads, call {rep}", file=sys.stderr, flush=True)
            tm = {}
            an.analyze_album_files(files, timing=tm)
            print(f"   the call: {tm['c_call_seconds'] * 1e3:.2f} ms", file=sys.stderr, flush=True)
    for p in files:
        p.unlink()
    tmp.rmdir()
