#!/usr/bin/env python3
"""The device MP3 decode chain alone (GPU box): per-kernel HIP-event times through rg_mp3_decode_bench.

    python tools/mp3_chain.py [target_units]

For each stream (the dense 320 kb/s synthetic one bench.py uses, the dense music-like 128 kb/s joint-stereo encode, the
reference's VBR fixture) the frames are repeated to three minutes and as many copies as make about `target_units`
granule-channels form one chunk.  Prints ms per kernel, the same normalised to 256 K units, and the chain's rate against its
algorithmic bytes (compressed bytes in + 4 bytes per decoded sample out)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401

import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import mp3dec  # noqa: E402

import os  # noqa: E402

target = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
only = os.environ.get("MP3_CHAIN_ONLY", "")  # substring of a stream's label: that stream alone
an = rg.Analyzer(0)
out = {}
for label, src in (("dense320_synthetic", ROOT / "tests/golden/mp3/v1_44k_stereo_long.mp3"),
                   ("dense128_joint_music", ROOT / "tests/golden/mp3/dense_44k_joint_128.mp3"),
                   ("vbr_fixture_sine", ROOT / "tests/golden/fixtures/test_vbr.mp3")):
    if only and only not in label:
        continue
    data = src.read_bytes()
    info = mp3dec.scan(data)
    body = data[int(info.first_frame_offset):]
    one = mp3dec.scan(body)
    reps = max(1, int(180.0 / (one.frames / one.sample_rate)))
    stream = body * reps
    si = mp3dec.scan(stream)
    units_per = si.audio_frames * (2 if si.mpeg_version == 1 else 1) * si.channels
    copies = max(1, round(target / units_per))
    r = an.decode_mp3_bench(stream, copies, reps=40)
    k = (1 << 18) / r["units"]
    algo = r["compressed_bytes"] + 4 * r["frames"] * si.channels
    r["ms_per_256k_units"] = {n: v * k for n, v in r["ms"].items()}
    r["algorithmic_bytes"] = algo
    r["achieved_GBps"] = algo / (r["ms"]["chain"] * 1e-3) / 1e9
    r["hbm_frac"] = r["achieved_GBps"] / 8000.0
    r["stereo_samples_per_s"] = r["frames"] / (r["ms"]["chain"] * 1e-3)
    r["copies"] = copies
    out[label] = r
    print(f"{label}: {copies} x {si.frames / si.sample_rate:.0f} s = {r['units']} units, {r['compressed_bytes'] / 1e6:.1f} MB in")
    print("   ms per 256K units: " + "  ".join(f"{n} {v:.3f}" for n, v in r["ms_per_256k_units"].items()))
    print(f"   chain {r['ms']['chain']:.3f} ms -> {r['stereo_samples_per_s'] / 1e9:.1f} G samples/s, {r['achieved_GBps']:.0f} GB/s algorithmic = {r['hbm_frac']:.3f} of HBM")
print(json.dumps(out))
