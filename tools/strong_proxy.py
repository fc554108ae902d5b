#!/usr/bin/env python
"""One-GPU proxy of STRONG scaling of the headline mesh (north_star: ">= 6x at 8 GPUs over 1 GPU on a 1024x1024 Q2 mesh"):
the share of one rank of the 1024^2 mesh cut N ways -- a (1024 / N) x 1024 periodic slab -- run through the whole rank schedule
with itself as both neighbours (bench.py --nx 1024/N --ny 1024 --self-halo T).  T(1024^2 plain) / T(slab, self-halo) bounds the
speed-up at N GPUs from above (no xGMI hop, no waiting for slower neighbours in it); the plain slab beside it shows how much of
the loss is the small launch itself.

  python tools/strong_proxy.py [--out profiles/r05/strong_proxy_c2.json]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary", "--no-live-traffic", "--steps", "200", "--warmup", "50"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True)
    for ln in reversed(out.stdout.splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    raise SystemExit("no bench line from %s\n%s" % (" ".join(cmd), out.stderr[-2000:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--transports", default="rccl,ipc,direct")
    args = ap.parse_args()
    full = min((line([]) for _ in range(2)), key=lambda d: d["ms_per_step"])
    t1 = full["ms_per_step"] * 1e3
    rec = {"mesh": "1024x1024 Q2 HLLC periodic (bench.py default)", "one_gpu_us_per_step": round(t1, 1), "one_gpu_mdof_s": round(full["value"], 1), "cuts": {}}
    for n in (2, 4, 8):
        nx = 1024 // n
        slab = ["--nx", str(nx), "--ny", "1024"]
        plain = min((line(slab) for _ in range(2)), key=lambda d: d["ms_per_step"])
        row = {"slab": "%dx1024" % nx, "plain_us_per_step": round(plain["ms_per_step"] * 1e3, 1), "ideal_us_per_step": round(t1 / n, 1), "self_halo": {}}
        for t in args.transports.split(","):
            d = min((line(slab + ["--self-halo", t]) for _ in range(2)), key=lambda d: d["ms_per_step"])
            us = d["ms_per_step"] * 1e3
            row["self_halo"][t] = {"us_per_step": round(us, 1), "speedup_bound": round(t1 / us, 2), "efficiency_bound": round(t1 / us / n, 3),
                                   "exchange_wait_us": d["config"]["exchange_wait_us"], "check": d["config"]["check"]}
        rec["cuts"][str(n)] = row
    txt = json.dumps(rec, indent=1)
    print(txt)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
