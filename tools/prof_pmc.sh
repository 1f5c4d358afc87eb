#!/bin/bash
# usage: tools/prof_pmc.sh <tag> <frames_per_launch> <bench args...>     (run on the GPU box through gpurun)
#   e.g. tools/prof_pmc.sh cfg2 7938000000                      (bench.py's default workload, configs[2])
#        tools/prof_pmc.sh cfg1 26460000 --tracks-per-rank 1 --minutes 10
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>/ and the summaries that are committed under profiles/:
#   <round>_<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command
#   <round>_<tag>_span.json          per-kernel span / concurrency derived from the raw kernel trace
#   <round>_pmc_<tag>.json           per-launch PMC means of the dominant kernel (read by bench.py)
#   <round>_<tag>_pmc_summary.txt    every counter, every kernel
# Each counter group is its own rocprofv3 run (no trace domains combined with --pmc), each under a timeout.
TAG=$1; FRAMES=$2; shift 2
export PROF_ROUND=${PROF_ROUND:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
export PROFILES_DIR=$R/gpurun_out/profiles   # only gpurun_out/ travels back: copy from there into profiles/
mkdir -p $R/$OUT $PROFILES_DIR
cd $R
BENCH_ARGS="$*"
# (--no-one-shot: the bench's synchronous leg launches the same kernel with another choice of windows per lane; the passes
# below describe the pipelined workload's launches only, the pass kt1 the synchronous one)
run() { name=$1; shift; timeout ${PROF_TIMEOUT:-400} rocprofv3 "$@" -d $OUT/$name --output-format csv -- python bench.py --cpu-seconds 0 --no-configs1 --no-mp3 --no-one-shot $BENCH_ARGS $EXTRA > $OUT/$name.log 2>&1 || echo "$name failed/timeout"; }
EXTRA="${KT_EXTRA:-}"
run kt --kernel-trace --stats
# one launch alone.  ONESHOT="<tracks> <minutes>" (set for cfg2 / cfg1 below): the synchronous entry point on the same batch
case "$TAG" in cfg2) ONESHOT=${ONESHOT:-"1000 3"};; cfg1) ONESHOT=${ONESHOT:-"1 10"};; esac
if [ -n "$ONESHOT" ]; then
  export KT1_CMD_TEXT="rocprofv3 --kernel-trace --stats -- python tools/ubench/oneshot_one.py $ONESHOT 0 30"
  timeout ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace --stats -d $OUT/kt1 --output-format csv -- python tools/ubench/oneshot_one.py $ONESHOT 0 30 > $OUT/kt1.log 2>&1 || echo "kt1 failed/timeout"
else
  EXTRA="${KT_EXTRA:-} --slots 1"
  run kt1 --kernel-trace --stats   # one pipeline slot: every launch alone
fi
# the counter passes serialise the dispatches and write one row per dispatch and counter: a short pre-roll and few
# timed steps keep them small (the counters are per dispatch, they do not depend on how many there are)
EXTRA="${PMC_EXTRA:---pre-roll 0.01 --steps 6 --warmup 1}"
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run pmc_f64 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32
if [ -z "$PMC_SHORT" ]; then
run pmc_sq2 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU
run pmc_grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT
run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run pmc_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT
run pmc_lat --pmc SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES
fi
python tools/prof_summary.py $OUT $TAG $FRAMES "$BENCH_ARGS" > $PROFILES_DIR/${PROF_ROUND}_${TAG}_pmc_summary.txt 2>&1
tail -5 $PROFILES_DIR/${PROF_ROUND}_${TAG}_pmc_summary.txt
# only the summaries travel back (gpurun merges at most 64 MiB): the raw traces stay unless asked for
[ -n "$KEEP_RAW" ] || rm -rf $OUT
