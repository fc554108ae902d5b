#!/bin/bash
# A/B of run-time switches on one box: tools/ab_env.sh "<bench args>" "VAR=1 OTHER=0" "VAR=0" ...   (two runs of each; "-" = no variables)
ARGS=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    printf "%-28s: " "$v"; env $e tools/bench_line.sh --no-secondary $ARGS
  done
done
