#!/usr/bin/env python3
"""Writes tests/golden/oracle_vectors.json: inputs (seed/shape of the deterministic generator,
plus a hash of the generated samples) and the oracle's outputs (loudness, peak, non-zero bins).

The reference's tests hold no dB value or histogram for this path, and the Rust
reference cannot be run here, so these vectors freeze the behaviour of the restatement after it
passed tests/test_oracle.py; they guard the oracle and the generator against regressions."""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import pyoracle as po  # noqa: E402

CASES = [(44100, 44100 * 12, 0x5EED0000, 2), (48000, 48000 * 3 + 7, 0x5EED0001, 2), (44100, 30000, 0x5EED0002, 1),
         (22050, 22050 * 2, 0x5EED0003, 2), (96000, 96000, 0x5EED0004, 2), (8000, 8000 * 5, 0x5EED0005, 1),
         (44100, 44100 * 2, 0x5EED0006 | (1 << 40), 2)]
out = {"generator": "include/rg_synth.h", "cases": []}
for rate, n, seed, ch in CASES:
    l = po.synth_f32(seed, 0, rate, n)
    r = po.synth_f32(seed, 1, rate, n) if ch == 2 else None
    res, hist = po.analyze_pcm(l, r, rate)
    nz = np.nonzero(hist)[0]
    out["cases"].append({"rate": rate, "frames": n, "seed": seed, "channels": ch,
                         "sha256_ch0": hashlib.sha256(l.tobytes()).hexdigest(),
                         "loudness_db": res["loudness_db"], "gain_db": res["gain_db"], "peak": res["peak"],
                         "hist_nonzero": [[int(i), int(hist[i])] for i in nz]})
(ROOT / "tests" / "golden" / "oracle_vectors.json").write_text(json.dumps(out, indent=1) + "\n")
print("wrote", len(out["cases"]), "cases")
