#!/bin/bash
# A/B/C... on one box: tools/ab_multi.sh "<bench args>" base variant1 variant2 ...   (names under scratch/variants/, "base" = the shipped library)
ARGS=$1; shift
cp dflo_amd/libdflo_hip.so /tmp/base.so
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then cp /tmp/base.so dflo_amd/libdflo_hip.so; else cp scratch/variants/$v.so dflo_amd/libdflo_hip.so; fi
    printf "%-12s: " $v; tools/bench_line.sh --no-secondary $ARGS
  done
done
cp /tmp/base.so dflo_amd/libdflo_hip.so
