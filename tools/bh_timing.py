#!/usr/bin/env python3
"""Where a back-half block's step goes (GPU box; needs a library built with -DRG_BH_TIMING=<block>):

    tools/build_variant.sh bht -DRG_BH_TIMING=2000
    MP3RGAIN_AMD_LIB=build_ab/libbht.so python tools/bh_timing.py

Per stream: each wave's busy time per pipeline step (stamp before its barrier minus stamp after the previous one) and the step
period, in shader-clock cycles, for block RG_BH_TIMING of the last launch."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401

import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import _capi, mp3dec  # noqa: E402

lib = _capi.load()
HUFFMAN = "--huffman" in sys.argv  # a library built with -DRG_HF_TIMING=<block>: the Huffman kernel's phases per wave
an = rg.Analyzer(0)
names = {0: "imdct", 1: "requant", 2: "dct32", 3: "window"}
for label, src in (("dense320_synthetic", ROOT / "tests/golden/mp3/v1_44k_stereo_long.mp3"),
                   ("dense128_joint_music", ROOT / "tests/golden/mp3/dense_44k_joint_128.mp3"),
                   ("vbr_fixture_sine", ROOT / "tests/golden/fixtures/test_vbr.mp3")):
    data = src.read_bytes()
    info = mp3dec.scan(data)
    body = data[int(info.first_frame_offset):]
    one = mp3dec.scan(body)
    reps = max(1, int(180.0 / (one.frames / one.sample_rate)))
    stream = body * reps
    si = mp3dec.scan(stream)
    units_per = si.audio_frames * (2 if si.mpeg_version == 1 else 1) * si.channels
    copies = max(1, round((1 << 18) / units_per))
    r = an.decode_mp3_bench(stream, copies, reps=3)
    if HUFFMAN:
        hb = np.zeros((8, 8), dtype=np.uint64)
        assert lib.rg_hf_dbg_read(C.c_void_p(hb.ctypes.data)) == 0
        hb = hb.astype(np.int64)
        t0 = hb[:, 0].min()
        print(f"== {label}: huffman {r['ms']['huffman'] * (1 << 18) / r['units']:.3f} ms per 256K units; cycles from the block's first stamp")
        print("   wave: tables in LDS | scalefactors | big_values | count1 | finish + unit  (phase lengths)   end")
        for w in range(8):
            d = np.diff(hb[w, :6])
            print(f"   {w}: " + " ".join(f"{int(x):7d}" for x in d) + f"   {int(hb[w, 5] - t0):8d}")
        continue
    buf = np.zeros((4, 40, 2), dtype=np.uint64)
    assert lib.rg_bh_dbg_read(C.c_void_p(buf.ctypes.data)) == 0
    t = buf.astype(np.int64)
    print(f"== {label}: backhalf {r['ms']['backhalf'] * (1 << 18) / r['units']:.3f} ms per 256K units")
    steps = slice(6, 30)  # the pipeline's steady part
    # Round 6: waves 1 and 2 requantise the even and the odd granules, two steps each (first half of granule k in step k,
    # second half in step k + 1: entry [k] of an even k is wave 1's first half, of an odd k its second half); wave 0 the
    # IMDCT, wave 3 matrixing + window.
    period = np.diff(t[0, :, 0])[steps]
    print(f"   step period: mean {period.mean():.0f} cycles (min {period.min()}, max {period.max()})")
    for w, name in ((1, "requant even"), (2, "requant odd"), (0, "imdct"), (3, "dct + window")):
        busy = (t[w, :, 1] - t[w, :, 0])[steps]
        print(f"   wave {w} {name:12s}: busy mean {busy.mean():7.0f}  min {busy.min():6d}  max {busy.max():6d}   = {busy.mean() / period.mean() * 100:4.0f} % of the step")
    ks = np.arange(40)
    own = np.where(ks % 2 == 0, 1, 2)  # the wave whose granule k is
    first = np.array([t[own[k], k, 1] - t[own[k], k, 0] for k in range(40)])[steps]
    second = np.array([t[own[k], k + 1, 1] - t[own[k], k + 1, 0] if k + 1 < 40 else 0 for k in range(40)])[steps]
    print(f"   a granule's requantisation: first half {first.mean():.0f}, second half {second.mean():.0f} cycles")
    d2 = np.zeros((40, 6), dtype=np.uint64)
    assert lib.rg_bh_dbg2_read(C.c_void_p(d2.ctypes.data)) == 0
    d2 = d2.astype(np.int64)
    t0 = np.array([t[own[k], k, 0] for k in range(40)])
    t1b = np.array([t[own[k], k + 1, 0] if k + 1 < 40 else 0 for k in range(40)])
    u0 = (d2[:, 4] - t0)[steps]
    u1 = (d2[:, 5] - d2[:, 4])[steps]
    u2 = (d2[:, 2] - d2[:, 5])[steps]
    print(f"   first half: the step's units in hand {u0.mean():.0f}, its spectra in hand and zeroed {u1.mean():.0f}, second-plane work {u2.mean():.0f},"
          f" big-value pre-pass {(d2[:, 3] - d2[:, 2])[steps].mean():.0f}, prefetch issue + gains {(d2[:, 0] - d2[:, 3])[steps].mean():.0f} cycles")
    print(f"   second half: three rounds {(d2[:, 1] - t1b)[steps].mean():.0f} cycles, then the special cases")
