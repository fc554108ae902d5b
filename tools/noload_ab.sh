#!/bin/bash
# upper bound of the asynchronous LDS fill: the C2 stage kernel with u(s) "already in LDS" (no own-row loads, no halo gather)
# at 4 / 3 / 2 workgroups per CU, against the shipped kernel at the same occupancies
cp dflo_amd/libdflo_hip.so /tmp/base.so
for rep in 1 2; do
for v in pad noload; do
  cp scratch/variants/$v.so dflo_amd/libdflo_hip.so
  for pad in 0 8500 21500; do
    printf "%-7s DFLO_LDS_PAD=%-6s: " $v $pad; DFLO_VERBOSE=1 DFLO_LDS_PAD=$pad tools/bench_line.sh --no-secondary --steps 200 --warmup 50 2>/dev/null
  done
done
done
cp /tmp/base.so dflo_amd/libdflo_hip.so
