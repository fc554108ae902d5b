"""stdin: the tail of a bench.py run -> the host/device timing lines of the multi-device driver (DFLO_MULTI_VERBOSE) and the bench line, short"""
import json, sys
for l in sys.stdin.read().splitlines():
    if l.startswith("dflo_hip_multi_advance"):
        print("   ", l)
    elif l.startswith("{"):
        try:
            d = json.loads(l)
            print('%.0f MDoF/s  %.4f ms/step  kernel %.4f ms  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))
        except Exception:
            print(l[-300:])
    else:
        print("   ?", l[-200:])
