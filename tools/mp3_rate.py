#!/usr/bin/env python3
"""End-to-end rate of the file-level path on MP3 input (GPU box): host decoder vs split decoder (tuning key 6).

    python tools/mp3_rate.py [files] [minutes]

Builds `files` MP3 files of `minutes` each by repeating the frames of a dense 320 kb/s golden stream (and of a real
encode, the VBR fixture), then times rg_analyze_album over them with both decoders, and the stages of the split decoder
on one file: stage A alone (rg_mp3_parse_units, one host thread), the host decoder (one thread), the device half."""
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401

import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import mp3dec  # noqa: E402

nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 64
minutes = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
an = rg.Analyzer(0)
for label, src in (("dense 320k synthetic", ROOT / "tests/golden/mp3/v1_44k_stereo_long.mp3"),
                   ("dense 128k joint stereo, music-like", ROOT / "tests/golden/mp3/dense_44k_joint_128.mp3"),
                   ("VBR fixture (440 Hz sine)", ROOT / "tests/golden/fixtures/test_vbr.mp3")):
    if os.environ.get("MP3_RATE_STREAM") and os.environ["MP3_RATE_STREAM"] not in label:
        continue
    data = src.read_bytes()
    info = mp3dec.scan(data)
    body = data[int(info.first_frame_offset):]
    if info.info_frame:  # drop the Info frame: its successor is the first audio frame
        first = mp3dec.scan(body)
        hdr = body[:4]
        # frame length of the Info frame from a scan of the remainder
        for cut in range(100, 1500):
            try:
                i2 = mp3dec.scan(body[cut:])
            except mp3dec.Mp3DecodeError:
                continue
            if i2.first_frame_offset == 0 and i2.info_frame == 0 and i2.audio_frames == info.audio_frames:
                body = body[cut:]
                break
    one = mp3dec.scan(body)
    secs = one.frames / one.sample_rate
    reps = max(1, int(minutes * 60 / secs))
    stream = body * reps
    si = mp3dec.scan(stream)
    audio_s = si.frames / si.sample_rate
    tmp = Path(tempfile.mkdtemp())
    files = []
    for k in range(nfiles):
        p = tmp / f"t{k:04d}.mp3"
        p.write_bytes(stream)
        files.append(p)
    print(f"== {label}: {nfiles} files x {audio_s:.1f} s ({len(stream) / 1e6:.2f} MB each, {si.channels} ch, {si.sample_rate} Hz)")
    t0 = time.perf_counter(); mp3dec.decode(stream); t_host = time.perf_counter() - t0
    t0 = time.perf_counter(); mp3dec.parse_units(stream); t_a = time.perf_counter() - t0
    an.decode_mp3_device(stream)
    t0 = time.perf_counter(); an.decode_mp3_device(stream); t_split = time.perf_counter() - t0
    print(f"   one file, one host thread: host decoder {t_host * 1e3:8.1f} ms ({audio_s / t_host:7.0f}x real time) | stage A {t_a * 1e3:7.1f} ms "
          f"({audio_s / t_a:7.0f}x) | split decoder incl. copies {t_split * 1e3:7.1f} ms ({audio_s / t_split:7.0f}x)")
    for key6 in ((3, 3, 3) if os.environ.get("MP3_RATE_ONLY") else (0, 1, 2, 3, 3)):  # MP3_RATE_ONLY: the default route alone
        an.set_tuning(6, key6)
        an.analyze_album_files(files[:2])
        tm = {}
        t0 = time.perf_counter()
        res = an.analyze_album_files(files, timing=tm)
        dt_py = time.perf_counter() - t0
        dt = tm["c_call_seconds"]  # the C call: what a caller over the C ABI waits for (the Python wrapper's own work on top is printed)
        name = ["host decoder                     ", "split: Huffman on host, B-E on GPU", "device: host parses side info     ", "device: host strips headers, piped"][key6]
        print(f"   rg_analyze_album, {name}: {dt:8.4f} s = {nfiles * audio_s / dt:9.0f}x real time, "
              f"{nfiles * si.frames / dt / 1e6:8.1f} M stereo samples/s, album loudness {res.album_loudness_db:.2f} dB"
              f"   (+ {1e3 * (dt_py - dt):.2f} ms in the Python wrapper)")
    an.set_tuning(6, 3)
    for _ in range(2):  # track mode (rg_analyze_tracks: `-r` over many files), default route
        tm = {}
        an.analyze_track_files(files, timing=tm)
    dt = tm["c_call_seconds"]
    print(f"   rg_analyze_tracks, device: host strips headers, piped: {dt:8.4f} s = {nfiles * audio_s / dt:9.0f}x real time, "
          f"{nfiles * si.frames / dt / 1e6:8.1f} M stereo samples/s")
