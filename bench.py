#!/usr/bin/env python
"""Headline benchmark: million DoF-updates/s of the explicit DG + SSP-RK path (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): BASELINE config C2 -- isentropic vortex on [-5,5]^2, 1024x1024 Cartesian
quads, Q2 (n_rk = 3), HLLC, periodic, cfl 0.9.  A "step" is one time step = all RK stages
(residual, dt*M^-1, SSP combine, cell averages, CFL reduction).  With N GPUs every rank owns a
1024x1024 slab of a (1024 N) x 1024 periodic mesh (weak scaling) and exchanges one layer of
face-neighbour cells per stage over RCCL.
value = n_dofs * n_rk * steps / wall_seconds / 1e6, inputs resident in HBM.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch


def cpu_baseline(threads, nx=160, steps=2):
    """Oracle (port of the reference loops; assembly threaded over cells like MeshWorker+TBB, the
    other passes serial as in the reference) on a bounded sample of the same workload."""
    import dflo_amd
    from dflo_amd import problems
    import oracle_lib
    mesh = dflo_amd.Mesh.cartesian(nx, nx, -5.0, -5.0, 10.0 / nx, [-1] * 4, 2)
    prm = dflo_amd.Parameters(flux="hllc", cfl=0.9)
    ora = oracle_lib.Oracle(mesh, prm, threads=threads)
    ora.set_solution(mesh.interpolate(problems.isentropic_vortex))
    t = 0.0
    dt = ora.compute_time_step(t)
    ora.step(dt)  # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        dt = ora.compute_time_step(t)
        ora.step(dt)
        t += dt
    sec = time.perf_counter() - t0
    n_dofs = mesh.n_cells * mesh.ndof
    return {
        "value": n_dofs * ora.n_rk * steps / sec / 1e6, "unit": "MDoF-updates/s", "cores": threads, "kind": "port",
        "sample": "%dx%d Q2 HLLC periodic vortex, %d RK3 steps (%.1f s), oracle/dflo_oracle.cc with %d OpenMP "
                  "threads over cells in assembly" % (nx, nx, steps, sec, threads),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nx", type=int, default=1024, help="cells per direction per GPU")
    ap.add_argument("--degree", type=int, default=2)
    ap.add_argument("--flux", default="hllc")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import dflo_amd
    from dflo_amd import problems

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the dflo HIP engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")

    nx, ny = args.nx * world, args.nx
    h = 10.0 / args.nx
    prm = dflo_amd.Parameters(flux=args.flux, cfl=0.9)
    mesh = dflo_amd.Mesh.cartesian(nx, ny, -5.0, -5.0, h, [-1] * 4, args.degree)
    n_rk = 2 if args.degree == 1 else 3
    n_dofs_total = mesh.n_cells * mesh.ndof

    if not distributed:
        claw = dflo_amd.ConservationLaw(mesh, prm, device=local_rank)
        claw.set_initial_condition(mesh.interpolate(problems.isentropic_vortex))
        claw.advance(args.warmup)
        claw.stage_timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        claw.advance(args.steps)  # dt and time stay on the device; returns after a stream sync
        torch.cuda.synchronize()
        sec = time.perf_counter() - t0
        kernel_ms, n_launch = claw.stage_timing(False)
        n_dofs_launch = n_dofs_total
    else:
        import torch.distributed as dist
        from dflo_amd.dist import DistributedConservationLaw
        dclaw = DistributedConservationLaw(mesh, prm, device_index=local_rank)
        # every rank evaluates the IC only on its own cells
        xy = dclaw.mesh.support_points()
        w = problems.isentropic_vortex(((xy[..., 0] + 5.0) % 10.0) - 5.0, xy[..., 1])
        u = np.ascontiguousarray(np.stack(w, axis=1)).reshape(-1)
        dclaw.claw.set_initial_condition(u)
        dclaw.exchange_solution()
        for _ in range(args.warmup):
            dclaw.iterate_explicit(dclaw.compute_time_step())
        dclaw.claw.stage_timing(True)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            dclaw.iterate_explicit(dclaw.compute_time_step())
        dist.barrier()
        torch.cuda.synchronize()
        sec = time.perf_counter() - t0
        kernel_ms, n_launch = dclaw.claw.stage_timing(False)
        tt = torch.tensor([sec], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        sec = float(tt.item())
        n_dofs_launch = dclaw.n_dofs_owned

    if rank == 0:
        value = n_dofs_total * n_rk * args.steps / sec / 1e6
        bytes_per_update = 24.0  # read u(s), read u(n), write u(s+1)   
        achieved = n_dofs_launch * bytes_per_update / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf))
                key = "q%d_%s_%d" % (args.degree, args.flux, args.nx)
                traffic = rec.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "million DoF-updates/s (explicit RK3, 2D Euler)", "value": value, "unit": "MDoF-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "isentropic_vortex, %dx%d quads per GPU (global %dx%d), Q%d, %s, periodic, SSP-RK %d stages"
                            % (args.nx, args.nx, nx, ny, args.degree, args.flux.upper(), n_rk),
                "n_dofs": n_dofs_total, "n_rk": n_rk, "parallelism": "x-slabs, %d rank(s)" % world,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "traffic": traffic, "kernel": "stage_kernel<%d,%s>" % (args.degree + 1, args.flux),
                "kernel_ms": kernel_ms, "launches": n_launch, "algorithmic_bytes_per_dof_update": bytes_per_update,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            # the better of a scalar run and an all-cores run is reported, with the threads it used
            runs = [cpu_baseline(threads=1, nx=128, steps=1)]
            if (os.cpu_count() or 1) > 1:
                runs.append(cpu_baseline(threads=os.cpu_count(), nx=192, steps=2))
            out["cpu_baseline"] = max(runs, key=lambda r: r["value"])
        print(json.dumps(out))
    if distributed:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
