#!/usr/bin/env python3
"""One file through rg_analyze_track (the shape `mp3rgain -r one.mp3` has), GPU box: the C call's duration, and where it goes
(RG_TRACE_FILES=1 prints the loader pipeline's own stamps).
    python tools/one_file.py [minutes]"""
import ctypes as C
import os, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401
import mp3rgain_amd as rg
from mp3rgain_amd import _capi, mp3dec
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
an = rg.Analyzer(0)
lib = an._lib
for label, src in (("320k", "tests/golden/mp3/v1_44k_stereo_long.mp3"), ("128k", "tests/golden/mp3/dense_44k_joint_128.mp3"), ("vbr", "tests/golden/fixtures/test_vbr.mp3")):
    data = (ROOT / src).read_bytes()
    info = mp3dec.scan(data)
    body = data[int(info.first_frame_offset):]
    one = mp3dec.scan(body)
    stream = body * max(1, int(minutes * 60 / (one.frames / one.sample_rate)))
    si = mp3dec.scan(stream)
    p = Path(tempfile.mkdtemp()) / "one.mp3"
    p.write_bytes(stream)
    out = _capi.TrackResult()
    path = os.fsencode(str(p))
    ms = []
    for rep in range(12):
        t0 = time.perf_counter()
        rc = lib.rg_analyze_track(an._ctx, path, -1, C.byref(out))
        ms.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0
    t0 = time.perf_counter(); raw = p.read_bytes(); t_read = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); mp3dec.scan(raw); t_scan = (time.perf_counter() - t0) * 1e3
    print(f"{label}: {len(stream) / 1e6:.2f} MB, {si.frames / si.sample_rate:.0f} s: rg_analyze_track " + " ".join(f"{x:.2f}" for x in ms[2:]) +
          f" ms (median {sorted(ms[2:])[5]:.2f}); python read {t_read:.2f} ms, header walk alone {t_scan:.2f} ms; loudness {out.loudness_db:.2f}", flush=True)
    p.unlink()
