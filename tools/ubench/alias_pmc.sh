#!/bin/bash
# usage (GPU box): tools/ubench/alias_pmc.sh  -- counters of the 1000 x 3 min synchronous launch, streamed from HBM vs every
# descriptor on one track's PCM (Infinity Cache resident): which wait grows when the data comes from HBM?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/alias_pmc; rm -rf $OUT; mkdir -p $OUT
declare -A G
G[a]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
G[b]="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INST_LEVEL_LDS"
G[c]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
for mode in stream alias; do for g in a b c; do
  (cd $R && RG_ALIAS_MODE=$mode timeout 200 rocprofv3 --pmc ${G[$g]} -d $OUT/${mode}_$g --output-format csv -- python tools/ubench/alias_tracks.py 1000 3 > $OUT/${mode}_$g.log 2>&1) || echo "$mode $g failed"
done; done
cd $R && python - <<'PY'
import csv, glob, collections
out = "gpurun_out/alias_pmc"
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/*_*/*/*counter_collection.csv"):
    m = f.split("/")[2].split("_")[0]
    for r in csv.DictReader(open(f)):
        if "rg_tm_main_kernel" in r["Kernel_Name"]:
            rows[m][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({n for m in rows for n in rows[m]})
with open(f"{out}/summary.txt", "w") as fo:
    fo.write(f"{'counter (mean per launch)':40s}{'stream':>16s}{'alias':>16s}{'ratio':>9s}\n")
    for n in names:
        a = sum(rows['stream'][n]) / max(1, len(rows['stream'][n])); b = sum(rows['alias'][n]) / max(1, len(rows['alias'][n]))
        fo.write(f"{n:40s}{a:16.5g}{b:16.5g}{(a / b if b else float('nan')):9.3f}\n")
print(open(f"{out}/summary.txt").read())
PY
rm -rf $OUT/*_?/
