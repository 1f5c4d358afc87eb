#!/bin/bash
# usage: tools/prof_pmc.sh <outdir> <bench args...>   (run on the GPU box through gpurun)
# Each counter group is its own rocprofv3 run (no trace domains combined with --pmc), each under a timeout.
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$OUT
cd $R
run() { name=$1; shift; timeout 150 rocprofv3 "$@" -d $OUT/$name --output-format csv -- python bench.py --cpu-reps 0 $BENCH_ARGS $EXTRA > $OUT/$name.log 2>&1 || echo "$name failed/timeout"; }
BENCH_ARGS="$*"
EXTRA=""
run kt --kernel-trace --stats
# the counter passes serialise the dispatches and write one row per dispatch and counter: a short pre-roll and 100
# timed steps keep them small (the counters are per dispatch, they do not depend on how many there are)
EXTRA="--pre-roll 0.01 --steps 100 --warmup 5"
run pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run pmc_sq2 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU
run pmc_grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run pmc_lat --pmc SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES
run pmc_sqc --pmc SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_REQ SQC_TC_STALL
run pmc_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT
find $OUT -name "*.csv" | head -40
