#!/usr/bin/env python3
"""Summarise a tools/prof_pmc.sh output directory and write the files that are committed under profiles/.

    prof_summary.py <rawdir> <tag> <frames_per_launch> "<bench args>"

  profiles/<round>_<tag>_kernel_stats.csv  copy of rocprofv3's kernel_stats.csv
  profiles/<round>_<tag>_span.json         per kernel: launches, sum of durations, first-start-to-last-end span, concurrency
                                       (from the raw kernel trace: the stats file's average alone says nothing about
                                       throughput when launches of consecutive batches overlap)
  profiles/<round>_pmc_<tag>.json          per-launch PMC means of the dominant kernel; bench.py reads it
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

d, tag, frames, bench_args = sys.argv[1], sys.argv[2], int(sys.argv[3]), (sys.argv[4] if len(sys.argv) > 4 else "")
RND = os.environ.get("PROF_ROUND", "r03")
prof = os.environ.get("PROFILES_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
os.makedirs(prof, exist_ok=True)
MAIN = "rg_tm_main_kernel"


def newest(pattern):
    """gpurun merges the output of successive calls into one tree: of several runs of a pass keep the latest"""
    by_pass = {}
    for f in glob.glob(pattern):
        key = f.split(os.sep)[-3]  # the pass directory (kt, pmc_fetch, ...)
        if key not in by_pass or os.path.getmtime(f) > os.path.getmtime(by_pass[key]):
            by_pass[key] = f
    return sorted(by_pass.values())


for f in newest(f"{d}/kt/*/*kernel_stats.csv"):
    shutil.copy(f, os.path.join(prof, f"{RND}_{tag}_kernel_stats.csv"))
    print("== kernel stats (rocprofv3 --kernel-trace --stats)")
    for r in csv.DictReader(open(f)):
        print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg_ns {float(r['AverageNs']):12.1f} pct {r['Percentage']}")

# ---- span / concurrency per kernel from the raw trace ------------------------------------------------------
# The bench line printed under the profiler (kt.log) says how many timed steps there were and how many launch groups a
# step has; the timed region of a kernel is then its LAST steps x groups launches (pre-roll and warm-up come before, and
# a barrier + device synchronise separates them from the timed region).
bench_line = None
try:
    for ln in open(f"{d}/kt.log"):
        if ln.startswith('{"metric"'):
            bench_line = json.loads(ln)
except OSError:
    pass
if bench_line:
    json.dump(bench_line, open(os.path.join(prof, f"{RND}_{tag}_bench_under_rocprofv3.json"), "w"))
steps = bench_line["steps"] if bench_line else None
groups = bench_line["roofline"].get("launch_groups_per_step", 1) if bench_line else 1
spans = {}
for f in newest(f"{d}/kt/*/*kernel_trace.csv"):
    per = defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]  # all instantiations of a kernel together
        per[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    for k, iv in per.items():
        if "rg_" not in k:
            continue
        iv.sort()
        ntimed = steps * groups if steps and len(iv) >= steps * groups and ("tm_main" in k or "tm_fix" in k or "k1_halo" in k) else None
        keep = iv[-ntimed:] if ntimed else iv
        tot = sum(e - s for s, e in keep)
        span = max(e for _, e in keep) - min(s for s, _ in keep)
        spans[k] = {"launches_all": len(iv), "launches_counted": len(keep), "sum_duration_ms": tot / 1e6,
                    "avg_duration_ms": tot / 1e6 / len(keep), "span_ms": span / 1e6,
                    "concurrency": tot / span if span else 1.0, "span_per_launch_ms": span / 1e6 / len(keep)}
if spans:
    m = next((v for k, v in spans.items() if MAIN in k), None)
    doc = {"command": f"rocprofv3 --kernel-trace --stats -- python bench.py --cpu-seconds 0 --no-configs1 --no-mp3 {bench_args}".strip(),
           "frames_per_step": frames, "timed_steps": steps, "launch_groups_per_step": groups, "kernels": spans,
           "note": "launches_counted = the timed region's launches (the last steps x groups of the kernel); "
                   "span = first start to last end of those; concurrency = sum of durations / span"}
    if m and bench_line:
        algo = bench_line["roofline"]["algorithmic_bytes_per_launch"] * groups  # per step
        step_ms = m["span_ms"] / steps
        doc["dominant_kernel"] = {"name": MAIN, "algorithmic_bytes_per_step": algo,
                                  "span_per_step_ms": step_ms,
                                  "achieved_GBps_span": algo / (step_ms * 1e-3) / 1e9,
                                  "hbm_frac_span": algo / (step_ms * 1e-3) / 8e12,
                                  "bench_line_frac_same_run": bench_line["roofline"]["frac"],
                                  "bench_line_kernel_ms_same_run": bench_line["roofline"]["kernel_ms"]}
    # one launch alone (pass kt1): the synchronous entry point over the same resident batch (tools/ubench/oneshot_one.py,
    # rg_analyze_pcm_batch: one batch in flight, the library's own choice of windows per lane), or the bench command with one
    # pipeline slot; either way launches cannot overlap and the stats file's average IS a launch alone
    kt1_cmd = os.environ.get("KT1_CMD_TEXT") or (doc["command"] + " --slots 1")
    for f in newest(f"{d}/kt1/*/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if MAIN in r["Name"]:
                doc["one_launch_alone"] = {"command": kt1_cmd, "calls": int(r["Calls"]),
                                           "avg_duration_ms": float(r["AverageNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6,
                                           "achieved_GBps": frames * 8 / (float(r["AverageNs"]) * 1e-9) / 1e9,
                                           "hbm_frac": frames * 8 / (float(r["AverageNs"]) * 1e-9) / 8e12}
    json.dump(doc, open(os.path.join(prof, f"{RND}_{tag}_span.json"), "w"), indent=1)
    print("== span", json.dumps(doc.get("dominant_kernel")))

acc = defaultdict(lambda: defaultdict(list))
for f in newest(f"{d}/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== PMC (mean per dispatch)")
for k, cs in acc.items():
    if "rg_" not in k:
        continue
    print(" ", k)
    for c, v in sorted(cs.items()):
        print(f"      {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")

# ---- profiles/r02_pmc_<tag>.json ---------------------------------------------------------------------------
# MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies the 128-byte
# requests of a wide coalesced streaming read at 64 B, hence the x2; each counter comes from its own --pmc pass.
main = [k for k in acc if MAIN in k]
if main:
    cs = acc[main[0]]
    mean = lambda name: (sum(cs[name]) / len(cs[name])) if cs.get(name) else None  # noqa: E731
    fk, wk = mean("FETCH_SIZE"), mean("WRITE_SIZE")
    out = {"kernel": MAIN, "workload_tag": tag, "frames_per_launch": frames,
           "algorithmic_bytes_per_launch": frames * 8,
           "fetch_size_kib_reported": fk, "write_size_kib_reported": wk,
           "correction": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B), WRITE_SIZE as reported",
           "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0 if fk is not None and wk is not None else None,
           "valu_insts_per_launch": mean("SQ_INSTS_VALU"),
           "fma_f64_insts_per_launch": mean("SQ_INSTS_VALU_FMA_F64"),
           "add_f64_insts_per_launch": mean("SQ_INSTS_VALU_ADD_F64"),
           "mul_f64_insts_per_launch": mean("SQ_INSTS_VALU_MUL_F64"),
           "cvt_insts_per_launch": mean("SQ_INSTS_VALU_CVT"),
           "waves_per_launch": mean("SQ_WAVES"),
           "dispatches_averaged": {k: len(v) for k, v in cs.items() if k in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64")},
           "command": f"rocprofv3 --pmc <one group per pass> -- python bench.py --cpu-seconds 0 --no-configs1 --no-mp3 {bench_args} --pre-roll 0.01 --steps 6 --warmup 1".replace("  ", " "),
           "note": "PMC passes serialise the dispatches: these describe one launch alone"}
    json.dump(out, open(os.path.join(prof, f"{RND}_pmc_{tag}.json"), "w"), indent=1)
    print("wrote", f"profiles/{RND}_pmc_{tag}.json")
