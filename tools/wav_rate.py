#!/usr/bin/env python3
"""File-level path, measured (GPU box): one 10-minute 44.1 kHz stereo WAV stream in host memory through
rg_analyze_wav_batch -- H2D copy of the interleaved bytes, device de-interleave, analysis.  Prints the
end-to-end rate; run under `rocprofv3 --kernel-trace --stats` for the de-interleave kernel's own duration
(its traffic is 2 x the PCM bytes: read interleaved, write planar).  Usage: python tools/wav_rate.py [f32|s16] [reps]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
import torch  # noqa: F401,E402

import mp3rgain_amd as rg  # noqa: E402
from wavutil import wav_bytes  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "f32"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rate, frames = 44100, 44100 * 600
rng = np.random.default_rng(1)
x = (0.2 * rng.standard_normal((2, frames))).clip(-1, 1)
chans = [x[0].astype(np.float32), x[1].astype(np.float32)] if kind == "f32" else [np.round(x[0] * 32767), np.round(x[1] * 32767)]
wav = wav_bytes(chans, rate, kind, extra_chunks=False)
an = rg.Analyzer(0)
an.analyze_wav_bytes([wav])
t0 = time.perf_counter()
for _ in range(reps):
    r = an.analyze_wav_bytes([wav])[0]
dt = (time.perf_counter() - t0) / reps
print(f"{kind}: {len(wav) / 1e6:.1f} MB WAV, {dt * 1e3:.2f} ms per call = {frames / dt / 1e9:.2f} G stereo samples/s end to end "
      f"(PCIe-inclusive, pageable host memory); loudness {r.loudness_db:.2f} dB")
