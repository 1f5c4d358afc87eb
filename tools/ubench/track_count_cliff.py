import sys, time
sys.path.insert(0, '/root/repo')
import torch
import mp3rgain_amd as rg
from mp3rgain_amd import _capi
rate = int(sys.argv[1]); minutes = float(sys.argv[2])
an = rg.Analyzer(0)
frames = int(rate * 60 * minutes)
NMAX = max(int(x) for x in sys.argv[3:])
pcm = torch.empty((NMAX, 2, frames), dtype=torch.float32, device="cuda")
for t in range(NMAX):
    for c in range(2):
        an.synth_fill_device(pcm[t, c].data_ptr(), 77 + t, c, rate, 0, frames)
for n in [int(x) for x in sys.argv[3:]]:
    d = (_capi.TrackDesc * n)()
    for t in range(n):
        d[t].offset_bytes, d[t].frames, d[t].sample_rate, d[t].channels, d[t].format = t * 2 * frames * 4, frames, rate, 2, 0
    ms = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        an.analyze_device(d, n, pcm.data_ptr(), n * 2 * frames * 4)
        ms.append((time.perf_counter() - t0) * 1e3)
    print(f"{rate} Hz, {n} tracks x {minutes} min: sync call min {min(ms):.3f} ms = {n*frames/min(ms)/1e6:.1f} G frames/s   {['%.2f' % v for v in ms]}", flush=True)
