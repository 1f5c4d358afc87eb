#!/bin/bash
# usage: tools/prof_mp3.sh [target_units]      (run on the GPU box through gpurun)
# The device MP3 decode chain alone (tools/mp3_chain.py) under rocprofv3: kernel trace + one --pmc pass per counter group
# (never combined with a trace domain).  Writes raw output under gpurun_out/prof_mp3/ and, under gpurun_out/profiles/ (copy
# into profiles/ to commit):  <round>_mp3dev_kernel_stats.csv, <round>_mp3dev_chain.json (the tool's own HIP-event numbers),
# <round>_pmc_mp3.json (per kernel: HBM bytes and instruction counts per launch; bench.py reads it), <round>_mp3dev_pmc_summary.txt
UNITS=${1:-393216}
export PROF_ROUND=${PROF_ROUND:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/prof_mp3
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/profiles
mkdir -p $R/$OUT $P
cd $R
run() { name=$1; shift; timeout ${PROF_TIMEOUT:-300} rocprofv3 "$@" -d $OUT/$name --output-format csv -- python tools/mp3_chain.py $UNITS > $OUT/$name.log 2>&1 || echo "$name failed/timeout"; }
python tools/mp3_chain.py $UNITS | tail -1 > $P/${PROF_ROUND}_mp3dev_chain.json
run kt --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run pmc_sq2 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU
run pmc_grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT
run pmc_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT
python tools/prof_mp3_summary.py $OUT $P > $P/${PROF_ROUND}_mp3dev_pmc_summary.txt 2>&1
tail -30 $P/${PROF_ROUND}_mp3dev_pmc_summary.txt
[ -n "$KEEP_RAW" ] || rm -rf $OUT
