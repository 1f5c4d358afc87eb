#!/usr/bin/env python3
"""What the host side of the file route costs (GPU box): rg_analyze_album over 256 three-minute files per stream, default route
(the host reads the files, walks the frame headers and strips them; everything else is kernels), with the loader threads
(tuning key 7) limited to 1, 2, 4, 8, 16 and all usable cores.  Prints the C call's duration (best of `calls`), stereo samples/s,
and samples/s per loader thread -- the figure that says what rg_analyze_album_node delivers with cores / 8 threads per GPU.

    [THREADS=1,2,4] [MODES=3,2] [MP3RGAIN_AMD_LIB=...] python tools/loader_threads.py [files] [calls]

THREADS: the thread counts to try (0 = all); MODES: values of tuning key 10 to compare call by call (3 = album parts for
copy-bound chunks only: no parts for the chunks a starved device waited for and no smaller chunks at the call's end; 2 = both)."""
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
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cores = len(os.sched_getaffinity(0))
print(f"usable cores: {cores} (of {os.cpu_count()})")
an = rg.Analyzer(0)
MODES = [int(x) for x in os.environ["MODES"].split(",")] if os.environ.get("MODES") else [None]
THREADS = [int(x) for x in os.environ.get("THREADS", "1,2,4,8,16,0").split(",")]
for label, src in (("vbr_fixture", "tests/golden/fixtures/test_vbr.mp3"), ("dense128_joint", "tests/golden/mp3/dense_44k_joint_128.mp3"),
                   ("dense320", "tests/golden/mp3/v1_44k_stereo_long.mp3")):
    data = (ROOT / src).read_bytes()
    body = data[int(mp3dec.scan(data).first_frame_offset):]
    one = mp3dec.scan(body)
    stream = body * max(1, int(180 / (one.frames / one.sample_rate)))
    si = mp3dec.scan(stream)
    tmp = Path(tempfile.mkdtemp(prefix="rg_lt_"))
    files = []
    for k in range(nfiles):
        p = tmp / f"t{k:04d}.mp3"
        p.write_bytes(stream)
        files.append(p)
    print(f"== {label}: {nfiles} files x {si.frames / si.sample_rate:.0f} s, {len(stream) / 1e6:.2f} MB each ({len(stream) * 8 / (si.frames / si.sample_rate) / 1e3:.0f} kb/s)")
    for _ in range(4):
        an.analyze_album_files(files)
    ref = None
    for threads in THREADS:
        if threads > cores:
            continue
        an.set_tuning(7, threads)
        times = {m: [] for m in MODES}
        for _ in range(calls):  # the modes call by call in turn: what disturbs one disturbs the other
            for m in MODES:
                if m is not None:
                    an.set_tuning(10, m)
                tm = {}
                res = an.analyze_album_files(files, timing=tm)
                times[m].append(tm["c_call_seconds"])
                ref = ref if ref is not None else res.album_loudness_db
                assert res.album_loudness_db == ref
        n = threads or cores
        for m in MODES:
            best, med = min(times[m]), sorted(times[m])[len(times[m]) // 2]
            rate = nfiles * si.frames / best
            print(f"   loader threads {('all = %d' % cores) if threads == 0 else threads:>9}{'' if m is None else ' (key 10 = %d)' % m}: {best * 1e3:8.2f} ms (median {med * 1e3:6.2f})  "
                  f"{rate / 1e9:7.2f} G stereo samples/s  {rate / n / 1e9:6.2f} G per thread  ({nfiles * len(stream) / best / 1e9:5.2f} GB/s of files)", flush=True)
    an.set_tuning(7, 0)
    for p in files:
        p.unlink()
    tmp.rmdir()
