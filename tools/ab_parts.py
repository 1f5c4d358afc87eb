import os, sys, time, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import mp3rgain_amd as rg
from mp3rgain_amd import mp3dec
an = rg.Analyzer(0)
for label, src in (("128k", "tests/golden/mp3/dense_44k_joint_128.mp3"), ("vbr", "tests/golden/fixtures/test_vbr.mp3"), ("320k", "tests/golden/mp3/v1_44k_stereo_long.mp3")):
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
        res = {"0": [], "1": []}
        for rep in range(7):
            for h in ("0", "1"):
                os.environ["RG_ALBUM_PARTS"] = h
                tm = {}
                f(files, timing=tm)
                if rep >= 2: res[h].append(tm["c_call_seconds"] * 1e3)
        print(f"{label:5s} {mode:6s}: plain " + " ".join(f"{x:.2f}" for x in res["0"]) + f" (median {sorted(res['0'])[2]:.2f}) | parts " + " ".join(f"{x:.2f}" for x in res["1"]) + f" (median {sorted(res['1'])[2]:.2f}) ms", flush=True)
    for p in files: p.unlink()
