#!/bin/bash
# per-kernel times of several builds on one box: tools/ab_kernels.sh "<bench args>" base variant1 ...  (rocprofv3 --kernel-trace --stats)
ARGS=$1; shift
export TMPDIR=/tmp
cp dflo_amd/libdflo_hip.so /tmp/base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so dflo_amd/libdflo_hip.so; else cp scratch/variants/$v.so dflo_amd/libdflo_hip.so; fi
  OUT=/tmp/abk_$v; rm -rf $OUT
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o t -f csv -- python $OLDPWD/bench.py --no-cpu-baseline --no-secondary --no-live-traffic $ARGS ) > /tmp/abk_$v.log 2>&1
  echo "== $v: $(tail -1 /tmp/abk_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>/dev/null)"
  python - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("stage_kernel", "limiter", "finalize", "indicator", "dt_", "bc_eval")):
        print("   %-50s calls %5s avg %8.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
cp /tmp/base.so dflo_amd/libdflo_hip.so
