#!/bin/bash
# tools/rccl_probe.sh [outdir]: the RCCL latency probe once per channel setting (RCCL reads NCCL_*_NCHANNELS at communicator creation)
OUT=${1:-gpurun_out/rccl_probe}
mkdir -p $OUT scratch/probe
[ -x scratch/probe/rccl_probe ] || hipcc --offload-arch=gfx950 -O2 -o scratch/probe/rccl_probe tools/rccl_probe.hip -Ldflo_amd -ldflo_hip -Wl,-rpath,'$ORIGIN/../../dflo_amd' -ldl
export HSA_ENABLE_IPC_MODE_LEGACY=0
scratch/probe/rccl_probe > $OUT/default.txt 2>&1
for n in 1 2 4; do
  NCCL_MAX_NCHANNELS=$n NCCL_MIN_NCHANNELS=1 scratch/probe/rccl_probe > $OUT/nchannels_$n.txt 2>&1
done
grep -H "summary_json" $OUT/*.txt
