#!/bin/bash
# A/B on one box: tools/ab.sh <variant.so> [bench args]   (base = dflo_amd/libdflo_hip.so)
V=$1; shift
cp dflo_amd/libdflo_hip.so /tmp/base.so
for rep in 1 2; do
  cp /tmp/base.so dflo_amd/libdflo_hip.so; echo -n "base   : "; tools/bench_line.sh "$@"
  cp $V dflo_amd/libdflo_hip.so;          echo -n "variant: "; tools/bench_line.sh "$@"
done
cp /tmp/base.so dflo_amd/libdflo_hip.so
