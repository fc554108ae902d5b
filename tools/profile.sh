#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + PMC passes (each in its own run, as the
# MI355X guide prescribes: FETCH_SIZE and WRITE_SIZE cannot share a pass).
# usage: tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof/<tag>/
set -u
TAG=${1:-c2}; shift || true
# the trace pass runs bench.py with its default --steps/--warmup (the command the bench line comes from, so the
# kernel's average duration is comparable); the counter passes need only a few launches
TRACE_ARGS="--no-cpu-baseline --no-live-traffic $*"
PMC_ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-live-traffic $*"
OUT=$PWD/gpurun_out/prof/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() { # name, rocprof options...
  local name=$1; shift
  local args=$PMC_ARGS
  [ "$name" = trace ] && args=$TRACE_ARGS
  ( cd /tmp && rocprofv3 "$@" -d $OUT/$name -o $name -f csv -- python $OLDPWD/bench.py $args ) > $OUT/$name.log 2>&1
  tail -1 $OUT/$name.log | cut -c1-300
}
run trace --kernel-trace --stats
run fetch --pmc FETCH_SIZE --kernel-include-regex "stage_kernel|limiter_kernel"
run write --pmc WRITE_SIZE --kernel-include-regex "stage_kernel|limiter_kernel"
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-include-regex "stage_kernel|limiter_kernel"
run sq2 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS --kernel-include-regex "stage_kernel|limiter_kernel"
run tcc --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-include-regex "stage_kernel|limiter_kernel"
run mf --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA --kernel-include-regex "stage_kernel|limiter_kernel"
python $PWD/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
