#!/usr/bin/env python
"""One-GPU proxy of the per-GPU efficiency of a multi-GPU run: bench.py's plain line against its --self-halo lines, run
alternately on THIS box (box-to-box spread is 3 %: only ratios taken on one box mean anything).

  python tools/selfhalo_ratio.py --config c2 [--reps 2] [--transports rccl,direct] [--out profiles/r05/selfhalo_c2.json]

The self-halo part is its own neighbour across a virtual cut and runs the complete schedule of a rank (bench.py --self-halo);
ratio = MDoF/s(self-halo) / MDoF/s(plain) bounds the weak-scaling efficiency of one GPU of an N-GPU run from above: what is
still missing in it is the latency of a real xGMI hop and the waiting for slower neighbours."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line(config, extra, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--no-cpu-baseline", "--no-secondary", "--no-live-traffic"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    for ln in reversed(out.stdout.splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    raise SystemExit("no bench line from %s\n%s\n%s" % (" ".join(cmd), out.stdout[-2000:], out.stderr[-2000:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--transports", default="rccl,direct")
    ap.add_argument("--steps", type=int, default=0)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--out", default="")
    args, rest = ap.parse_known_args()
    extra = list(rest)
    if args.steps:
        extra += ["--steps", str(args.steps)]
    if args.warmup:
        extra += ["--warmup", str(args.warmup)]
    transports = [t for t in args.transports.split(",") if t]
    runs = {"plain": []}
    for t in transports:
        runs[t] = []
    for _ in range(args.reps):
        runs["plain"].append(line(args.config, extra))
        for t in transports:
            runs[t].append(line(args.config, extra + ["--self-halo", t]))
    best = {k: max(v, key=lambda r: r["value"]) for k, v in runs.items()}
    rec = {
        "config": args.config, "workload": best["plain"]["config"]["workload"], "steps": best["plain"]["steps"], "warmup": best["plain"]["warmup"],
        "plain": {"mdof_s": [round(r["value"], 1) for r in runs["plain"]], "ms_per_step": best["plain"]["ms_per_step"],
                  "kernel_ms": best["plain"]["roofline"]["kernel_ms"]},
        "self_halo": {},
        "note": "ratio = best self-halo / best plain of the alternating runs on one box; an upper bound of the per-GPU weak-scaling "
                "efficiency (no xGMI hop, no waiting for slower neighbours in it)",
    }
    for t in transports:
        b = best[t]
        rec["self_halo"][t] = {
            "mdof_s": [round(r["value"], 1) for r in runs[t]], "ms_per_step": b["ms_per_step"], "kernel_ms": b["roofline"]["kernel_ms"],
            "ratio": round(b["value"] / best["plain"]["value"], 4), "transport": b["config"]["transport"],
            "exchange_wait_us": b["config"]["exchange_wait_us"], "check": b["config"]["check"],
            "extra_us_per_step": round((b["ms_per_step"] - best["plain"]["ms_per_step"]) * 1e3, 2),
        }
    txt = json.dumps(rec, indent=1)
    print(txt)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
