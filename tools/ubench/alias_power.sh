#!/bin/bash
# usage (GPU box): tools/ubench/alias_power.sh <outfile>  -- rocm-smi (socket power, shader clock) sampled every 0.25 s while the
# 1000 x 3 min synchronous launch runs for ~6 s streamed from HBM, then ~6 s with every descriptor on one track's PCM.
OUTF=${1:-gpurun_out/alias_power.txt}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python - <<'PY' &
import sys, time
sys.path.insert(0, ".")
import torch
import mp3rgain_amd as rg
from mp3rgain_amd import _capi
NT, rate = 1000, 44100
frames = 180 * rate
an = rg.Analyzer(0)
pcm = torch.empty((NT, 2, frames), dtype=torch.float32, device="cuda")
for t in range(NT):
    for c in range(2):
        an.synth_fill_device(pcm[t, c].data_ptr(), 0x5EED0000 + t, c, rate, 0, frames)
torch.cuda.synchronize()
out = (_capi.TrackResult * NT)()
open("/tmp/alias_phase", "w").write("setup")
for label, alias in (("stream", False), ("alias", True)):
    d = (_capi.TrackDesc * NT)()
    for t in range(NT):
        d[t].offset_bytes, d[t].frames, d[t].sample_rate, d[t].channels, d[t].format = (0 if alias else t * 2 * frames * 4), frames, rate, 2, 0
    for _ in range(3):
        an.analyze_device(d, NT, pcm.data_ptr(), pcm.numel() * 4, out=out)
    open("/tmp/alias_phase", "w").write(label)
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 6.0:
        an.analyze_device(d, NT, pcm.data_ptr(), pcm.numel() * 4, out=out); k += 1
    print(f"{label}: {k} calls, {(time.perf_counter() - t0) / k * 1e3:.3f} ms per call", flush=True)
    open("/tmp/alias_phase", "w").write("idle")
    time.sleep(1.0)
open("/tmp/alias_phase", "w").write("done")
PY
PYPID=$!
echo "# phase, then rocm-smi --showpower --showclocks --csv (device 0)" > $OUTF
rocm-smi -d 0 --showpower --showclocks --showmaxpower --csv 2>/dev/null | head -3 >> $OUTF
while [ "$(cat /tmp/alias_phase 2>/dev/null)" != "done" ] && kill -0 $PYPID 2>/dev/null; do
  echo "$(cat /tmp/alias_phase 2>/dev/null) $(rocm-smi -d 0 --showpower --showclocks --csv 2>/dev/null | tail -n +2 | tr '\n' ' ')" >> $OUTF
  sleep 0.25
done
wait $PYPID
python - <<PY
import re, collections
rows = collections.defaultdict(list)
hdr = None
for line in open("$OUTF"):
    m = re.match(r"^(stream|alias) (.*)", line)
    if not m: continue
    f = m.group(2).split(",")
    rows[m.group(1)].append(f)
for k, v in rows.items():
    # columns: device, fclk, level, mclk, level, sclk, level, socclk, level, power
    def num(s):
        mm = re.search(r"([0-9.]+)", s); return float(mm.group(1)) if mm else float("nan")
    sclk = [num(r[5]) for r in v if len(r) > 9]; pw = [num(r[-1].split()[0]) for r in v if len(r) > 9]
    if sclk: print(f"{k}: {len(sclk)} samples, sclk median {sorted(sclk)[len(sclk)//2]:.0f} MHz, socket power median {sorted(pw)[len(pw)//2]:.0f} W")
PY
