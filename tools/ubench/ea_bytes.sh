#!/bin/bash
# usage (GPU box): tools/ubench/ea_bytes.sh <outfile> -- <command ...>
# Exact fabric-side byte counts per kernel: the L2's read requests by size (TCC_EA0_RDREQ_32B / _64B / _128B), its DRAM reads
# in 32-byte units (TCC_EA0_RDREQ_DRAM_32B), writes (TCC_EA0_WRREQ, _64B, WRREQ_WRITE_DRAM_32B) -- what FETCH_SIZE / WRITE_SIZE
# are derived from without saying which sizes they saw.  Three --pmc passes (four TCC counters each), mean per dispatch.
OUTF=$1; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ea_bytes_tmp; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
i=0
for g in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_IO_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $g -d $O/g$i --output-format csv -- "$@" > $O/g$i.log 2>&1 || echo "group $i failed"
done
python - $O > $OUTF <<'PY'
import sys, glob, csv, collections, json
o = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(o + "/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in acc.items():
    m = {n: sum(v) / len(v) for n, v in cs.items()}
    m["dispatches"] = max(len(v) for v in cs.values())
    rd = m.get("TCC_EA0_RDREQ_sum", 0); r32 = m.get("TCC_EA0_RDREQ_32B_sum", 0); r64 = m.get("TCC_EA0_RDREQ_64B_sum", 0); r128 = m.get("TCC_EA0_RDREQ_128B_sum", 0)
    m["read_bytes_by_size"] = 32 * r32 + 64 * r64 + 128 * r128
    m["read_requests_unsized"] = rd - r32 - r64 - r128
    m["dram_read_bytes"] = 32 * m.get("TCC_EA0_RDREQ_DRAM_32B_sum", 0)
    m["dram_write_bytes"] = 32 * m.get("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum", 0)
    m["fetch_size_bytes_x1"] = 1024 * m.get("FETCH_SIZE", 0)
    m["write_size_bytes"] = 1024 * m.get("WRITE_SIZE", 0)
    out[k] = m
print(json.dumps(out, indent=1))
PY
rm -rf $O
