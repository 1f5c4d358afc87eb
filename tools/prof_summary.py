#!/usr/bin/env python3
"""Summarise a tools/prof_pmc.sh output directory: per-kernel average duration and PMC counters."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
for f in glob.glob(f"{d}/kt/*/*kernel_stats.csv"):
    print("== kernel stats (rocprofv3 --kernel-trace --stats)")
    for r in csv.DictReader(open(f)):
        print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg_ns {float(r['AverageNs']):12.1f} pct {r['Percentage']}")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"{d}/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== PMC (mean per dispatch)")
for k, cs in acc.items():
    if "rg_" not in k:
        continue
    print(" ", k)
    for c, v in sorted(cs.items()):
        print(f"      {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
