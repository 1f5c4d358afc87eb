#!/bin/bash
# usage (GPU box): tools/pmc_oneshot.sh "<m values>" [tracks] [minutes]
# Counter passes of ONE synchronous call shape (tools/ubench/oneshot_one.py) per forced m: where does a single launch lose time?
MS=${1:-"37 40 10"}; NT=${2:-1000}; MIN=${3:-3}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_oneshot; rm -rf $OUT; mkdir -p $OUT
declare -A G
G[sq]="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
G[tlb]="TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_THRASHING_STALL_sum"
G[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"
G[fetch]="FETCH_SIZE"
G[lat]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
G[vmem]="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VALU_FMA_F64"
for m in $MS; do
  for g in sq tlb tcc fetch lat vmem; do
    (cd $R && timeout 200 rocprofv3 --pmc ${G[$g]} -d $OUT/m${m}_$g --output-format csv -- python tools/ubench/oneshot_one.py $NT $MIN $m 4 > $OUT/m${m}_$g.log 2>&1) || echo "m=$m $g failed"
  done
done
cd $R && python - <<'PY'
import csv, glob, collections, os
out = "gpurun_out/pmc_oneshot"
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/m*_*/*/*counter_collection.csv"):
    m = f.split("/")[2].split("_")[0]
    for r in csv.DictReader(open(f)):
        if "rg_tm_main_kernel" in r["Kernel_Name"]:
            rows[m][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({n for m in rows for n in rows[m]})
ms = sorted(rows, key=lambda s: int(s[1:]))
with open(f"{out}/summary.txt", "w") as fo:
    fo.write(f"{'counter':44s}" + "".join(f"{m:>18s}" for m in ms) + "\n")
    for n in names:
        fo.write(f"{n:44s}" + "".join(f"{(sum(rows[m][n]) / len(rows[m][n]) if rows[m][n] else float('nan')):18.4g}" for m in ms) + "\n")
    for m in ms:
        for lg in sorted(glob.glob(f"{out}/{m}_sq.log")):
            fo.write(m + " " + [l for l in open(lg) if l.startswith("m =")][-1])
print(open(f"{out}/summary.txt").read())
PY
rm -rf $OUT/m*_*/
