#!/usr/bin/env python3
"""How often does decoded MUSIC pay the exact repeat?  (GPU box)

    python tools/flag_rate.py [copies]

The fast analysis kernel marks a track whose result it cannot vouch for (RG_TRACK_FLAG_IMPRECISE); the synchronous entry
points then run the batch once more with those tracks on the order-faithful kernel.  tools/fuzz_selfcheck.py says how often
that happens on random and pathological PCM; this says how often it happens on PCM that came out of an MP3 decoder: every
stream under tests/golden/mp3 and tests/golden/fixtures is decoded (host decoder), cut into 10-second tracks, analysed through
the asynchronous pair (which hands the flag over instead of repeating) and through the synchronous call, both timed."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import mp3dec  # noqa: E402
from mp3rgain_amd.replaygain import PcmTrack, pack_tracks  # noqa: E402

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1
an = rg.Analyzer(0)
tot_tracks = tot_flagged = 0
print("stream: tracks (10 s each), flagged imprecise, async ms, synchronous ms (with the exact repeat if any)")
for src in sorted(list((ROOT / "tests/golden/mp3").glob("*.mp3")) + list((ROOT / "tests/golden/fixtures").glob("*.mp3"))):
    try:
        pcm, info = mp3dec.decode(src.read_bytes())
    except mp3dec.Mp3DecodeError:
        continue
    if not rg._capi.load().rg_supported_rate(int(info.sample_rate)) or not np.isfinite(pcm).all() or np.abs(pcm).max() > 64:
        continue  # (the fixture committed with global_gain 255 decodes to 6e8)
    seg = 10 * int(info.sample_rate)
    pieces = [pcm[:, a:a + seg] for a in range(0, max(1, pcm.shape[1] - seg + 1), seg)] or [pcm]
    tracks = [PcmTrack([np.ascontiguousarray(p[c]) for c in range(p.shape[0])], int(info.sample_rate)) for p in pieces] * copies
    arena, descs = pack_tracks(tracks)
    d = torch.from_numpy(arena).cuda()
    n = len(tracks)
    for _ in range(2):
        an.enqueue_device(descs, n, d.data_ptr(), arena.nbytes)
        res = an.collect(n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    an.enqueue_device(descs, n, d.data_ptr(), arena.nbytes)
    res = an.collect(n)
    t_async = (time.perf_counter() - t0) * 1e3
    an.analyze_device(descs, n, d.data_ptr(), arena.nbytes)
    t0 = time.perf_counter()
    an.analyze_device(descs, n, d.data_ptr(), arena.nbytes)
    t_sync = (time.perf_counter() - t0) * 1e3
    flagged = sum(1 for r in res if r.flags & 2)
    tot_tracks += n
    tot_flagged += flagged
    print(f"  {src.name:40s} {n:5d} {flagged:5d}   {t_async:8.3f} {t_sync:8.3f}")
print(f"total: {tot_flagged} of {tot_tracks} tracks flagged")
