#!/usr/bin/env python3
"""Device timeline of the LAST rg_analyze_album call in a rocprofv3 kernel trace (GPU box):

    cd /tmp && export TMPDIR=/tmp
    MP3_RATE_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/e2e -- python tools/mp3_rate.py 256 3
    python tools/e2e_timeline.py /tmp/e2e

Prints, for the window from the call's first frame-parser launch to the end of its last kernel: time per kernel name (sum of
durations and the union of their intervals), the time no kernel was running, and the gaps longer than 50 us."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
# the last call: from the first MP3 kernel after the analysis kernel of the call before to the end
mains = [i for i, r in enumerate(rows) if "rg_tm_main" in r[2]]
prev = mains[-2] if len(mains) > 1 else -1
i = next(j for j in range(prev + 1, len(rows)) if "rg_mp3_" in rows[j][2])
call = rows[i:]
t0, t1 = call[0][0], max(r[1] for r in call)
print(f"window {1e-6 * (t1 - t0):.3f} ms, {len(call)} launches")
by = {}
for s, e, n in call:
    by.setdefault(n, []).append((s, e))
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
for n, iv in sorted(by.items(), key=lambda kv: -union(kv[1])):
    print(f"  {n[:60]:60s} launches {len(iv):4d}  sum {1e-6 * sum(e - s for s, e in iv):8.3f} ms  union {1e-6 * union(iv):8.3f} ms")
allu = union([(s, e) for s, e, _ in call])
print(f"  some kernel running: {1e-6 * allu:.3f} ms; none: {1e-6 * (t1 - t0 - allu):.3f} ms")
iv = sorted((s, e) for s, e, _ in call)
ce = iv[0][1]
for s, e in iv[1:]:
    if s - ce > 50000:
        print(f"  gap of {1e-3 * (s - ce):7.1f} us at +{1e-6 * (ce - t0):.3f} ms")
    ce = max(ce, e)
