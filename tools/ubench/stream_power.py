#!/usr/bin/env python3
"""Socket power of a plain streaming read (torch.sum over a 32 GiB tensor, wide coalesced loads, every DRAM page read whole)
-- against which the analysis kernel's memory-side power can be set.  GPU box: python tools/ubench/stream_power.py"""
import re
import subprocess
import threading
import time

import torch

x = torch.ones(8 << 30, dtype=torch.float32, device="cuda")  # 32 GiB
torch.cuda.synchronize()
samples = []
stop = False


def poll():
    while not stop:
        out = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout
        row = out.strip().splitlines()[-1].split(",")
        nums = [float(m.group(1)) for m in (re.search(r"([0-9.]+)", f) for f in row[1:]) if m]
        samples.append((time.perf_counter(), nums))
        time.sleep(0.2)


th = threading.Thread(target=poll)
th.start()
time.sleep(1.0)
t_idle = time.perf_counter()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 6.0:
    for _ in range(10):
        x.sum()
    torch.cuda.synchronize()
    n += 10
t1 = time.perf_counter()
stop = True
th.join()
bw = n * x.numel() * 4 / (t1 - t0) / 1e12
busy = [s[1] for s in samples if t0 + 1.0 < s[0] < t1]
idle = [s[1] for s in samples if s[0] < t_idle]
med = lambda v: sorted(v)[len(v) // 2]
print(f"streaming read {bw:.2f} TB/s: socket power median {med([b[-1] for b in busy]):.0f} W, sclk median {med([b[4] for b in busy]):.0f} MHz; idle before: {med([b[-1] for b in idle]):.0f} W")
