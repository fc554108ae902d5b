#!/bin/bash
# usage: tools/bench_line.sh <bench args...>  -> value ms_per_step kernel_ms
python bench.py --no-cpu-baseline --no-live-traffic "$@" 2>&1 | tail -1 | python -c "
import json,sys
l=sys.stdin.read()
try:
  d=json.loads(l); print('%.0f MDoF/s  %.4f ms/step  kernel %.4f ms  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))
except Exception: print(l[-400:])"
