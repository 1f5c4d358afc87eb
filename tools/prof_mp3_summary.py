#!/usr/bin/env python3
"""Summarise tools/prof_mp3.sh: per decode kernel the rocprofv3 average duration, HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE,
MI355X_MICROARCH.md: FETCH_SIZE tallies 128-byte requests at 64 B on gfx950) and instruction counts per launch, against the
algorithmic bytes of the chain (compressed bytes in + 4 bytes per decoded sample out)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

d, prof = sys.argv[1], sys.argv[2]
RND = os.environ.get("PROF_ROUND", "r03")


def newest(pattern):
    by = {}
    for f in glob.glob(pattern):
        key = f.split(os.sep)[-3]
        if key not in by or os.path.getmtime(f) > os.path.getmtime(by[key]):
            by[key] = f
    return sorted(by.values())


chain = json.load(open(os.path.join(prof, f"{RND}_mp3dev_chain.json")))
stats = {}
for f in newest(f"{d}/kt/*/*kernel_stats.csv"):
    shutil.copy(f, os.path.join(prof, f"{RND}_mp3dev_kernel_stats.csv"))
    print("== kernel stats (rocprofv3 --kernel-trace --stats -- python tools/mp3_chain.py)")
    for r in csv.DictReader(open(f)):
        name = r["Name"].split("(")[0].replace("void ", "")
        if "rg_mp3" in name:
            stats[name] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6}
            print(f"  {name:44s} calls {r['Calls']:>5s} avg_ms {float(r['AverageNs']) / 1e6:9.4f} min_ms {float(r['MinNs']) / 1e6:9.4f}")
acc = defaultdict(lambda: defaultdict(list))
for f in newest(f"{d}/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== PMC (mean per dispatch; three workloads of different size are in the mix: see per-unit figures below)")
out = {"round": RND, "command": "rocprofv3 --pmc <one group per pass> -- python tools/mp3_chain.py", "kernels": {},
       "correction": "FETCH_SIZE (KiB) x2: 128-byte requests tallied at 64 B on gfx950; WRITE_SIZE (KiB) as reported"}
units_mean = sum(v["units"] for v in chain.values()) / len(chain)
algo_mean = sum(v["algorithmic_bytes"] for v in chain.values()) / len(chain)
for k, cs in sorted(acc.items()):
    if "rg_mp3" not in k:
        continue
    print(" ", k)
    for c, v in sorted(cs.items()):
        print(f"      {c:28s} {sum(v) / len(v):16.1f}  (n={len(v)})")
    mean = lambda n: (sum(cs[n]) / len(cs[n])) if cs.get(n) else None  # noqa: E731
    fk, wk = mean("FETCH_SIZE"), mean("WRITE_SIZE")
    out["kernels"][k] = {"hbm_bytes_per_launch": (2 * fk + wk) * 1024 if fk is not None and wk is not None else None,
                         "valu_insts_per_launch": mean("SQ_INSTS_VALU"), "lds_insts_per_launch": mean("SQ_INSTS_LDS"),
                         "waves_per_launch": mean("SQ_WAVES"), "rocprof_avg_ms": stats.get(k, {}).get("avg_ms")}
tot = sum(v["hbm_bytes_per_launch"] or 0 for v in out["kernels"].values())
out["units_per_launch_mean"] = units_mean
out["algorithmic_bytes_per_launch_mean"] = algo_mean
out["hbm_bytes_per_launch_chain"] = tot
out["traffic_over_algorithmic"] = tot / algo_mean if algo_mean else None
out["hbm_bytes_per_unit"] = tot / units_mean if units_mean else None
json.dump(out, open(os.path.join(prof, f"{RND}_pmc_mp3.json"), "w"), indent=1)
print(f"== chain: {tot / 1e9:.3f} GB of HBM traffic per launch for {algo_mean / 1e9:.3f} GB algorithmic = x{tot / algo_mean:.2f}; "
      f"{tot / units_mean:.0f} bytes per unit")
print("== HIP-event numbers of the tool itself:", json.dumps({k: v["ms_per_256k_units"] for k, v in chain.items()}))
