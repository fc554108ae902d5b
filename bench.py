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


def cpu_twin(threads, nx=512, steps=10):
    """The optimised CPU twin (oracle/dflo_oracle.cc, last section): the same stage fused into one pass per stage and
    threaded over all cells -- the honest CPU figure next to the reference-style one."""
    import dflo_amd
    from dflo_amd import problems
    import oracle_lib
    mesh = dflo_amd.Mesh.cartesian(nx, nx, -5.0, -5.0, 10.0 / nx, [-1] * 4, 2)
    ora = oracle_lib.Oracle(mesh, dflo_amd.Parameters(flux="hllc", cfl=0.9), threads=threads)
    ora.set_solution(mesh.interpolate(problems.isentropic_vortex))
    ora.twin_advance(1)  # warm-up
    t0 = time.perf_counter()
    ora.twin_advance(steps)
    sec = time.perf_counter() - t0
    return {
        "value": mesh.n_cells * mesh.ndof * ora.n_rk * steps / sec / 1e6, "unit": "MDoF-updates/s", "cores": threads,
        "sample": "%dx%d Q2 HLLC periodic vortex, %d RK3 steps (%.1f s), fused collocation stage (scalar C++, no SIMD "
                  "intrinsics), %d OpenMP threads over cells" % (nx, nx, steps, sec, threads),
    }


def _cpu_quota():
    """CPUs this process may actually use: the cgroup CPU quota if there is one (the GPU box gives the container 16 of the
    host's 256 hardware threads), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def _settle_clocks(seconds=0.4):
    """The GPU sat idle while the host built the mesh and the initial data; it needs a few tenths of a second of
    fp64 work to come back to its sustained clock.  This is not a solver step: it touches none of the engine's data
    and leaves the workload of the W warm-up and K timed steps exactly as specified."""
    if os.environ.get("DFLO_BENCH_NO_PREHEAT") == "1":
        return
    x = torch.full((1 << 26,), 1.0000001, dtype=torch.float64, device="cuda")   # 512 MB: streams through HBM like the solver
    y = torch.zeros_like(x)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            y = torch.addcmul(y, x, x)
        torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nx", type=int, default=1024, help="cells per direction per GPU")
    ap.add_argument("--degree", type=int, default=2)
    ap.add_argument("--flux", default="hllc")
    ap.add_argument("--basis", default="Qk", choices=["Qk", "Pk"], help="c2 only; Pk: dflo's FE_DGP (modal) element")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tvb", action="store_true", help="c4 only: positivity limiter alone (BASELINE config 4 as written)")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="c2 (default, the headline): periodic vortex; c3: Sod tube 2048x256 Q1 Roe TVB+positivity; "
                         "c5: bilinear-cell mesh, Q3, KFVS, positivity (1-GPU stand-ins for BASELINE configs 3 and 5)")
    args = ap.parse_args()
    if args.config == "c3":
        args.degree, args.flux = 1, "roe"
    if args.config == "c5":
        args.degree, args.flux = 3, "kfvs"
    if args.config == "c4":
        args.degree, args.flux = 2, "hllc"

    import dflo_amd
    from dflo_amd import problems

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the dflo HIP engine has no CPU fallback")
    if os.environ.get("DFLO_BENCH_BACKEND") == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    distributed = world > 1 or os.environ.get("DFLO_BENCH_FORCE_DIST") == "1"   # developer switch: drive the multi-rank loop with one rank
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # DFLO_BENCH_BACKEND=gloo: developer switch to exercise the multi-rank path with several ranks on ONE gpu
        dist.init_process_group(os.environ.get("DFLO_BENCH_BACKEND", "nccl"))

    nx, ny = args.nx * world, args.nx
    h = 10.0 / args.nx
    prm = dflo_amd.Parameters(flux=args.flux, cfl=0.9)
    ic = problems.isentropic_vortex
    bc_fn = None
    if args.config == "c2":
        mesh = dflo_amd.Mesh.cartesian(nx, ny, -5.0, -5.0, h, [-1] * 4, args.degree)
        mesh.set_basis(args.basis)
    elif args.config == "c3":   # examples/sod_shock_tube: slip walls (0), outflow right (1), inflow left (2)
        nx, ny = 2048, 256
        mesh = dflo_amd.Mesh.cartesian(nx, ny, 0.0, 0.0, 1.0 / nx, [2, 1, 0, 0], 1)
        prm = dflo_amd.Parameters(flux="roe", limiter="TVB", char_lim=True, pos_lim=True, M=0.0, beta=2.0, cfl=0.9,
                                  final_time=1e9, boundary={0: "slip", 1: "outflow", 2: "inflow"})
        ic = bc_fn = problems.sod
    elif args.config == "c4":   # one GPU's share of C4: the left 501 x 1000 squares of the double Mach reflection, Q2, HLLC,
        # TVB(M=100, beta=1, char) + positivity, moving inflow state on the top wall evaluated by the device
        nyc = 1000
        dy = 1.0 / nyc
        n1 = int(np.ceil((1.0 / 6.0) / dy))
        nx, ny = 501, nyc
        mesh = dflo_amd.Mesh.cartesian(nx, ny, 1.0 / 6.0 - n1 * dy, 0.0, dy, [4, 2, 1, 3], 2)
        mesh.neighbors[:n1, 2] = -1 - 0
        prm = dflo_amd.Parameters(flux="hllc", limiter="none" if args.no_tvb else "TVB", char_lim=True, pos_lim=True, M=100.0, beta=1.0,
                                  cfl=0.9, final_time=1e9, boundary={0: "outflow", 1: "slip", 2: "outflow", 3: "inflow", 4: "inflow"})
        ic = bc_fn = lambda x, y: problems.double_mach(x, y)
    else:                       # c5 stand-in: fully unstructured quads (Delaunay triangles cut in three), q1 mapping, Q3 KFVS
        from dflo_amd import gmsh
        n = 295 if args.nx == 1024 else args.nx       # 6 n^2 cells: 522 150 by default (C5 has ~200 k cells per GPU)
        verts, quads, bed, side = gmsh.unstructured_quads(n, Lx=3.0, Ly=3.0, seed=1)
        bid = np.array([2, 3, 2, 1], dtype=np.int32)[side]   # bottom/top slip (2), right outflow (3), left inflow (1)
        mesh = dflo_amd.Mesh.from_quads(verts, quads, bed, bid, 3)
        nx = ny = n
        prm = dflo_amd.Parameters(flux="kfvs", pos_lim=True, cfl=0.5, final_time=1e9,
                                  boundary={1: "inflow", 2: "slip", 3: "outflow"})   # examples/forward_step/input.prm
        ic = bc_fn = problems.forward_step_inflow
    n_rk = 2 if args.degree == 1 else 3
    n_dofs_total = mesh.n_cells * mesh.ndof

    if not distributed:
        claw = dflo_amd.ConservationLaw(mesh, prm, device=local_rank)
        if bc_fn is not None:
            cell, face, bid, xy = claw.boundary_faces()
            bv = np.stack(bc_fn(xy[..., 0], xy[..., 1]), axis=-1)
            claw.set_boundary_values(0, bv)
            claw.set_boundary_values(1, bv)
        u0 = mesh.interpolate(ic)
        if args.config == "c5":   # a smooth bump on the free stream so that the fluxes see real jumps
            xy = mesh.support_points()
            bump = 1.0 + 0.1 * np.exp(-20.0 * ((xy[..., 0] - 1.5) ** 2 + (xy[..., 1] - 1.5) ** 2))
            u0 = (u0.reshape(mesh.n_cells, 4, -1) * bump[:, None, :]).reshape(-1)
        claw.set_initial_condition(u0)
        if args.config == "c4":
            sh = "(x<1.0/6.0+(1+20*t)/sqrt(3))"
            claw.set_boundary_function(3, ["57.1576766498*" + sh, "-33.0*" + sh, "8.0*%s + 1.4*(1-%s)" % (sh, sh),
                                           "563.5*%s + 2.5*(1-%s)" % (sh, sh)])
        if args.config in ("c3", "c4"):
            claw.apply_limiter()   # run() limits the initial condition, src/claw.cc:997-1001
        mass0 = claw.cell_average.sum(axis=0) if args.config == "c2" else None
        _settle_clocks()
        claw.advance(args.warmup)
        claw.stage_timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        claw.advance(args.steps)  # dt and time stay on the device; returns after a stream sync
        torch.cuda.synchronize()
        sec = time.perf_counter() - t0
        kernel_ms, n_launch = claw.stage_timing(False)
        n_dofs_launch = n_dofs_total
        drift = None
        check = None
        if args.config == "c2":
            m1 = claw.cell_average.sum(axis=0)
            drift = float(np.abs(m1 - mass0).max() / np.abs(mass0).max())
        elif args.config == "c3":   # the tube is one-dimensional: every row of cells has to carry the same averages
            a = claw.cell_average.reshape(ny, nx, 4)
            check = "rows of cells identical to %.1e, min density %.4f" % (float(np.abs(a - a[:1]).max()), float(a[..., 2].min()))
        elif args.config in ("c4", "c5"):
            a = claw.cell_average
            pr = 0.4 * (a[:, 3] - 0.5 * (a[:, 0] ** 2 + a[:, 1] ** 2) / a[:, 2])
            check = "state finite and admissible after the run: min density %.4f, min pressure %.4f" % (float(a[:, 2].min()), float(pr.min()))
    else:
        import torch.distributed as dist
        from dflo_amd.dist import DistributedConservationLaw
        dclaw = DistributedConservationLaw(mesh, prm, device_index=local_rank)
        # every rank evaluates the IC only on its own cells
        u = dclaw.mesh.interpolate(lambda x, y: problems.isentropic_vortex(((x + 5.0) % 10.0) - 5.0, y))
        dclaw.claw.set_initial_condition(u)
        dclaw.exchange_solution()
        n_own = dclaw.mesh.n_owned
        mass0 = dclaw.claw.cell_average[:n_own].sum(axis=0)          # outside the timed region: conservation check
        _settle_clocks()
        dclaw.advance(args.warmup)
        dclaw.claw.stage_timing(True)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dclaw.advance(args.steps)
        dist.barrier()
        torch.cuda.synchronize()
        sec = time.perf_counter() - t0
        kernel_ms, n_launch = dclaw.claw.stage_timing(False)
        tt = torch.tensor([sec], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        sec = float(tt.item())
        n_dofs_launch = dclaw.n_dofs_owned
        # periodic box: the sums of the conserved variables over all ranks must not move (checks the halo exchange
        # and the doubly evaluated partition faces of the run that was just timed)
        mm = torch.tensor(np.concatenate([mass0, dclaw.claw.cell_average[:n_own].sum(axis=0)]), dtype=torch.float64,
                          device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(mm, op=dist.ReduceOp.SUM)
        mm = mm.cpu().numpy()
        drift = float(np.abs(mm[4:] - mm[:4]).max() / np.abs(mm[:4]).max())
        check = None

    if rank == 0:
        value = n_dofs_total * n_rk * args.steps / sec / 1e6
        # read u(s), read u(n), write u(s+1); +16 with a limiter/positivity pass.
        # roofline.achieved prices the stage kernel alone, so it uses 24 B in every configuration.
        bytes_per_update = 24.0
        achieved = n_dofs_launch * bytes_per_update / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf))
                key = "%s_q%d_%s_%d" % (args.config, args.degree, args.flux, args.nx)
                traffic = rec.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "million DoF-updates/s (explicit RK3, 2D Euler)", "value": value, "unit": "MDoF-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": {"c2": "isentropic_vortex, %dx%d quads per GPU (global %dx%d), %s%d, %s, periodic, SSP-RK %d stages"
                                   % (args.nx, args.nx, nx, ny, args.basis[0], args.degree, args.flux.upper(), n_rk),
                             "c3": "sod_shock_tube, 2048x256 quads, Q1, ROE, TVB(M=0,beta=2,char)+positivity, SSP-RK 2 stages",
                             "c4": "double_mach_reflection, 501x1000 of the 4001x1000 squares (one of 8 slabs), Q2, HLLC, %spositivity, moving inflow on the device, SSP-RK 3 stages"
                                   % ("" if args.no_tvb else "TVB(M=100,beta=1,char)+"),
                             "c5": "free stream + bump on %d unstructured quads (q1 mapping), Q3, KFVS, positivity, SSP-RK 3 stages"
                                   % mesh.n_cells}[args.config],
                "n_dofs": n_dofs_total, "n_rk": n_rk, "parallelism": "x-slabs, %d rank(s)" % world,
                "check": check if drift is None else "periodic box: max relative drift of the conserved totals over the run = %.1e" % drift,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "traffic": traffic, "kernel": "stage_kernel<%d,%s,geo%d>" % (args.degree + 1, args.flux, int(args.config == "c5")),
                "kernel_ms": kernel_ms, "launches": n_launch, "algorithmic_bytes_per_dof_update": bytes_per_update,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            # Like dflo on deal.II's WorkStream, only the assembly sweep is threaded (the update, average and
            # limiter passes are serial in the reference), so more threads stop paying early; the best of a
            # few thread counts is reported together with the count it used (tools/cpu_scaling.py).
            ncpu = os.cpu_count() or 1
            quota = _cpu_quota()
            counts = sorted({1, quota, min(2 * quota, ncpu)})
            runs = [cpu_baseline(threads=t, nx=512, steps=3) for t in counts]
            out["cpu_baseline"] = max(runs, key=lambda r: r["value"])
            out["cpu_baseline"]["host_cpus"] = ncpu
            out["cpu_baseline"]["cpu_quota"] = quota   # what the container may use (cgroup cpu.max); threads beyond it only queue
            # the fused twin threads every pass, so it scales to the quota (GPU box, 16 CPUs: 47 / 349 / 661 MDoF/s with
            # 1 / 8 / 16 threads, less with more threads than CPUs)
            twins = [cpu_twin(threads=t) for t in sorted({quota, min(2 * quota, ncpu)})]
            out["cpu_baseline"]["optimised_twin"] = max(twins, key=lambda r: r["value"])
        result_line = json.dumps(out)
    else:
        result_line = None
    if distributed:
        import torch.distributed as dist
        dist.destroy_process_group()
    if result_line is not None:
        import ctypes
        sys.stderr.flush()
        try:
            ctypes.CDLL(None).fflush(None)   # RCCL prints its version banner through C stdio: flush it first
        except Exception:
            pass
        print(result_line, flush=True)


if __name__ == "__main__":
    main()
