#!/usr/bin/env python3
"""Do the MP3 decode chain and the analysis kernel share the GPU better than they take turns?  (GPU box.)

Two contexts in one process: A repeats one 768 K-unit chunk of the VBR fixture through the decode chain (rg_mp3_decode_bench),
B analyses a resident batch of 32 x 3 min over and over.  Timed: A alone, B alone, A and B started together on two host
threads (ctypes releases the GIL).  If the joint wall time is near max(A, B) the file route gains by analysing the tracks of
chunk k while chunk k + 1 decodes; if it is near A + B it does not.

    python tools/ubench/overlap_decode_analysis.py [reps]"""
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import _capi, mp3dec  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
RATE, FRAMES, NTR = 44100, 44100 * 180, 32

a = rg.Analyzer(0)
b = rg.Analyzer(0)

data = (ROOT / "tests/golden/fixtures/test_vbr.mp3").read_bytes()
info = mp3dec.scan(data)
body = data[int(info.first_frame_offset):]
one = mp3dec.scan(body)
stream = body * max(1, int(180.0 / (one.frames / one.sample_rate)))
si = mp3dec.scan(stream)
units_per = si.audio_frames * (2 if si.mpeg_version == 1 else 1) * si.channels
copies = max(1, round(786432 / units_per))

pcm = torch.empty(NTR * 2 * FRAMES, dtype=torch.float32, device="cuda:0")
descs = (_capi.TrackDesc * NTR)()
for t in range(NTR):
    off = t * 2 * FRAMES
    for c in range(2):
        b.synth_fill_device(pcm[off + c * FRAMES:].data_ptr(), 0x5EED0000 + t, c, RATE, 0, FRAMES)
    descs[t].offset_bytes = off * 4
    descs[t].frames = FRAMES
    descs[t].sample_rate = RATE
    descs[t].channels = 2
    descs[t].format = _capi.FMT_F32_PLANAR
torch.cuda.synchronize()


def run_a(n):
    return a.decode_mp3_bench(stream, copies, reps=n)


def run_b(n):
    for _ in range(n):
        b.enqueue_device(descs, NTR, pcm.data_ptr(), pcm.numel() * 4, album=True)
        b.collect(NTR)


run_a(3)
run_b(3)
torch.cuda.synchronize()
t0 = time.perf_counter(); ra = run_a(reps); ta = time.perf_counter() - t0
# B's repetitions: as long as A's run
t0 = time.perf_counter(); run_b(10); tb10 = time.perf_counter() - t0
nb = max(1, int(round(10 * ta / tb10)))
t0 = time.perf_counter(); run_b(nb); tb = time.perf_counter() - t0
th = [threading.Thread(target=run_a, args=(reps,)), threading.Thread(target=run_b, args=(nb,))]
t0 = time.perf_counter()
for t in th:
    t.start()
for t in th:
    t.join()
tj = time.perf_counter() - t0
print(f"decode chain alone: {reps} chunks of {ra['units']} units in {ta * 1e3:.1f} ms (host wall; events say {ra['ms']['chain']:.3f} ms per chunk)")
print(f"analysis alone: {nb} batches of {NTR} x 3 min in {tb * 1e3:.1f} ms ({tb / nb * 1e3:.3f} ms per batch)")
print(f"together: {tj * 1e3:.1f} ms = {tj / (ta + tb):.3f} of the sum, {tj / max(ta, tb):.3f} of the longer one")
