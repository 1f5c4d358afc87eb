#!/usr/bin/env python3
"""Large differential fuzz of variant 2 against the CPU oracle (GPU box): random and pathological tracks from
tests/test_gpu_parity.py with fresh seeds.  Reports, per set, how many tracks have a histogram bin that differs
from the oracle's, how many carry RG_TRACK_FLAG_IMPRECISE, and -- the property that must hold -- how many differ
WITHOUT carrying the flag.  Usage: python tools/fuzz_selfcheck.py [tracks_per_set] [first_seed] [lane_target] [lo|hi|all]
(lo = rates up to 48 kHz, the default; hi = 64 and 96 kHz only; all = every stable rate)"""
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402

import mp3rgain_amd as rg  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

per_set = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # tuning key 2: 0 = cost model, 1 = one segment per window, 1 << 40 = shortest segments
which = sys.argv[4] if len(sys.argv) > 4 else "lo"
keep = {"lo": lambda r: r <= 48000, "hi": lambda r: r >= 64000, "all": lambda r: True}[which]
an = rg.Analyzer(0)
an.set_kernel(2)
an.set_tuning(2, lanes)
pool = ThreadPoolExecutor(16)
bad_total = 0
for name, gen in (("random", lambda n, s: T._random_cases(n, s)), ("pathological", lambda n, s: [c[:2] for c in T._pathological_cases(n if which != "hi" else n // 5, s, rates=[96000, 64000] if which == "hi" else None)])):
    for seed in (seed0, seed0 + 1):
        cases = [c for c in gen(per_set, seed) if keep(c[0])]
        mism = flagged = unflagged = 0
        for lo in range(0, len(cases), 64):
            part = cases[lo:lo + 64]
            got, h = an.analyze_tracks([rg.PcmTrack(ch, r) for r, ch in part], return_histograms=True)
            wants = list(pool.map(lambda c: po.analyze_pcm(c[1][0], c[1][1] if len(c[1]) > 1 else None, c[0]), part))
            for k, (w, wh) in enumerate(wants):
                f = bool(got[k].flags & 2)
                flagged += f
                if not np.array_equal(h[k], wh) or got[k].peak != w["peak"]:
                    mism += 1
                    if not f:
                        unflagged += 1
                        print("UNFLAGGED", name, seed, lo + k, part[k][0], "Hz", len(part[k][1][0]), "frames, bins", np.nonzero(h[k] != wh)[0][:6], flush=True)
        bad_total += unflagged
        print(f"{name:12s} seed {seed}: {len(cases)} tracks, {mism} with a differing bin, {flagged} flagged, {unflagged} differing and not flagged", flush=True)
sys.exit(1 if bad_total else 0)
