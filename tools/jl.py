"""stdin: a bench.py JSON line -> value, ms/step, kernel ms, roofline fraction on one short line (developer tool)"""
import json, sys
l = sys.stdin.read()
try:
    d = json.loads(l)
    print('%.0f MDoF/s  %.4f ms/step  kernel %.4f ms  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))
except Exception:
    print(l[-600:])
