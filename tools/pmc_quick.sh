#!/bin/bash
# tools/pmc_quick.sh KERNEL_SUBSTRING LIB [ENV=VAL ...] -- counters of one kernel of tools/mp3_chain.py (GPU box): one --pmc pass per
# group, mean per dispatch.  LIB = default | a name under build_ab/.
K=$1; LIB=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd $R
for e in "$@"; do export "$e"; done
[ "$LIB" = default ] || export MP3RGAIN_AMD_LIB=build_ab/lib$LIB.so
O=gpurun_out/pmcq_$LIB
rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $O/g$i --output-format csv -- python tools/mp3_chain.py ${UNITS:-262144} > $O/g$i.log 2>&1 || echo "group $i failed"
done
python - "$K" $O <<'PY'
import sys, glob, csv, collections
k, o = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(o + "/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if k in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for n in sorted(acc):
    v = acc[n]
    print(f"  {n:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
