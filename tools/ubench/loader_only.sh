#!/bin/bash
# usage (GPU box): tools/ubench/loader_only.sh   (needs build_ab/libloader.so: tools/build_variant.sh loader -DRG_TM_LOADER_ONLY)
# The headline kernel's memory-side floor: the 1000 x 3 min synchronous call (m = 37, one round of blocks) with the shipped
# library and with the loader-only build -- the same global loads, LDS staging and LDS reads, ONE operation per sample instead of
# the cascade.  Per library: call time and kernel time, FETCH_SIZE / WRITE_SIZE / TCP_TCC_READ_REQ / GRBM_GUI_ACTIVE of
# rg_tm_main_kernel (one --pmc pass per group), socket power and shader clock while the call repeats for 6 s.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/loader_only; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
for lib in default loader; do
  if [ $lib = default ]; then unset MP3RGAIN_AMD_LIB; else export MP3RGAIN_AMD_LIB=build_ab/lib$lib.so; fi
  python tools/ubench/oneshot_one.py 1000 3 37 10 > $OUT/time_$lib.txt 2>&1
  for g in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
    n=$(echo $g | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $g -d $OUT/pmc_${lib}_$n --output-format csv -- python tools/ubench/oneshot_one.py 1000 3 37 4 > $OUT/pmc_${lib}_$n.log 2>&1 || echo "$lib $n failed"
  done
  # power: repeat the call for 6 s, sample rocm-smi every 0.25 s
  python - <<'PY' &
import sys, time
sys.path.insert(0, ".")
import torch
import mp3rgain_amd as rg
from mp3rgain_amd import _capi
NT, rate = 1000, 44100
frames = 180 * rate
an = rg.Analyzer(0)
an.set_tuning(1, 2205); an.set_tuning(4, 37)
pcm = torch.empty((NT, 2, frames), dtype=torch.float32, device="cuda")
d = (_capi.TrackDesc * NT)()
for t in range(NT):
    for c in range(2):
        an.synth_fill_device(pcm[t, c].data_ptr(), 0x5EED0000 + t, c, rate, 0, frames)
    d[t].offset_bytes, d[t].frames, d[t].sample_rate, d[t].channels, d[t].format = t * 2 * frames * 4, frames, rate, 2, 0
torch.cuda.synchronize()
out = (_capi.TrackResult * NT)()
for _ in range(3):
    an.analyze_device(d, NT, pcm.data_ptr(), pcm.numel() * 4, out=out)
open("/tmp/lo_phase", "w").write("run")
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < 6.0:
    an.analyze_device(d, NT, pcm.data_ptr(), pcm.numel() * 4, out=out); k += 1
print(f"{k} calls, {(time.perf_counter() - t0) / k * 1e3:.3f} ms per call", flush=True)
open("/tmp/lo_phase", "w").write("done")
PY
  PYPID=$!
  echo idle > /tmp/lo_phase
  : > $OUT/power_$lib.txt
  while [ "$(cat /tmp/lo_phase 2>/dev/null)" != "done" ] && kill -0 $PYPID 2>/dev/null; do
    echo "$(cat /tmp/lo_phase 2>/dev/null) $(rocm-smi -d 0 --showpower --showclocks --csv 2>/dev/null | tail -n +2 | tr '\n' ' ')" >> $OUT/power_$lib.txt
    sleep 0.25
  done
  wait $PYPID
done
unset MP3RGAIN_AMD_LIB
python - <<'PY'
import csv, glob, re, json, collections
out = "gpurun_out/loader_only"
res = {}
for lib in ("default", "loader"):
    r = {}
    t = open(f"{out}/time_{lib}.txt").read()
    m = re.search(r"min ([0-9.]+) ms", t)
    r["call_ms_min"] = float(m.group(1)) if m else None
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/pmc_{lib}_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "rg_tm_main_kernel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        r[k] = sum(v) / len(v)
    sclk, pw = [], []
    for line in open(f"{out}/power_{lib}.txt"):
        if not line.startswith("run "):
            continue
        f = line[4:].split(",")
        def num(s):
            mm = re.search(r"([0-9.]+)", s); return float(mm.group(1)) if mm else float("nan")
        if len(f) > 9:
            sclk.append(num(f[5])); pw.append(num(f[-1].split()[0]))
    if sclk:
        r["sclk_mhz_median"] = sorted(sclk)[len(sclk) // 2]; r["socket_w_median"] = sorted(pw)[len(pw) // 2]; r["power_samples"] = len(sclk)
    res[lib] = r
algo = 8 * 1000 * 7938000
for lib, r in res.items():
    if r.get("FETCH_SIZE"):
        r["fetch_bytes_x1"] = r["FETCH_SIZE"] * 1024
        r["fetch_x1_over_algorithmic"] = r["FETCH_SIZE"] * 1024 / algo
        r["fetch_x2_over_algorithmic"] = 2 * r["FETCH_SIZE"] * 1024 / algo
    if r.get("TCP_TCC_READ_REQ_sum"):
        r["l1_to_l2_read_bytes_64B"] = r["TCP_TCC_READ_REQ_sum"] * 64
    if r.get("GRBM_GUI_ACTIVE"):
        r["kernel_cycles_per_xcd"] = r["GRBM_GUI_ACTIVE"] / 8
res["algorithmic_bytes_per_launch"] = algo
json.dump(res, open("gpurun_out/r05_loader_only.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
