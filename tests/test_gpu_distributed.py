"""GPU: two ranks (both on cuda:0, gloo transport with host staging -- RCCL refuses two ranks on one
device) drive two HIP engines through dflo_amd.dist and must reproduce the single-engine run and the
oracle.  The 8-GPU RCCL run uses the same code path with backend "nccl"."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _front(x, y):
    """oblique front in a flow with both velocity components away from zero (see test_gpu_parity._oblique_front)"""
    s = 0.5 * (1.0 + np.tanh((x + 0.5 * y - 0.8) / 0.004))
    rho, p = 1.0 + 0.6 * s, 1.0 + 0.9 * s
    u, v = 0.6, 0.35
    return [rho * u, rho * v, rho, p / 0.4 + 0.5 * rho * (u * u + v * v)]


def _mesh(dflo_amd, nx, ny, x0, h, side_bc, degree):
    if side_bc == "unstructured":   # C5's kind of mesh: irregular all-quad mesh, q1 mapping
        from dflo_amd import gmsh
        verts, quads, bed, bid = gmsh.unstructured_quads(nx, seed=2)
        return dflo_amd.Mesh.from_quads(verts, quads, bed, bid, degree)
    return dflo_amd.Mesh.cartesian(nx, ny, x0, x0, h, side_bc, degree)


def _smooth(x, y):
    from dflo_amd import problems
    return problems.smooth_perturbation(x, y, L=1.0)


def _worker(rank, world, port, case, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dflo_amd
    from dflo_amd import problems
    from dflo_amd.dist import DistributedConservationLaw

    nx, ny, degree, flux, limiter, pos, side_bc, bnd, ic_name = case[:9]
    ic = {"sod": problems.sod, "vortex": problems.isentropic_vortex, "front": _front, "smooth": _smooth}[ic_name]
    x0, h = (-5.0, 10.0 / nx) if ic_name == "vortex" else (0.0, 1.0 / nx)
    mesh = _mesh(dflo_amd, nx, ny, x0, h, side_bc, degree)
    prm = dflo_amd.Parameters(flux=flux, limiter=limiter, pos_lim=pos, boundary=bnd, beta=2.0, cfl=0.8 if side_bc != "unstructured" else 0.4,
                              shock_indicator=case[9] if len(case) > 9 else "limiter")
    u0 = mesh.interpolate(ic)
    d = DistributedConservationLaw(mesh, prm, device_index=0)
    cell, face, bid, xy = d.claw.boundary_faces()
    if len(cell):
        bv = np.stack(ic(xy[..., 0], xy[..., 1]), axis=-1)
        d.claw.set_boundary_values(0, bv)
        d.claw.set_boundary_values(1, bv)
    d.set_initial_condition(u0)
    d.exchange_solution()
    dts = []
    norms = None
    for it in range(4):
        dt = d.compute_time_step()
        d.iterate_explicit(dt)
        dts.append(dt)
        if it == 0:
            norms = d.residual_norms()      # summed over the ranks
    t_end = d.advance(2)        # device-resident dt, all-reduced across the ranks
    dts.append(t_end)
    u = d.gather_solution()
    if rank == 0:
        ret["u"] = u
        ret["dts"] = dts
        ret["norms"] = norms
    dist.destroy_process_group()


CASES = [
    (32, 16, 2, "hllc", "none", False, [-1, -1, -1, -1], None, "vortex"),
    (64, 8, 1, "roe", "TVB", True, [2, 1, 0, 0], {0: "slip", 1: "outflow", 2: "inflow"}, "sod"),
    (48, 40, 1, "hllc", "TVB", True, [0, 0, 0, 0], {0: "outflow"}, "front", "density"),   # KXRCF-gated limiter across the cut
    (64, 8, 2, "hllc", "TVB", True, [2, 1, 0, 0], {0: "slip", 1: "outflow", 2: "inflow"}, "sod"),   # C4 style: limiter marks from the stage kernel, rim / interior launches
    (10, 10, 3, "kfvs", "none", True, "unstructured", {0: "slip", 1: "outflow", 2: "slip", 3: "inflow"}, "smooth"),   # C5 style
]


@pytest.mark.parametrize("mode,index", [("0", 0), ("2", 0), ("0", 3), ("2", 3), ("2", 4)])
def test_other_exchange_orders(mode, index, monkeypatch):
    """The serial order (DFLO_OVERLAP=0) and the two-stream order (2) of the halo exchange give what the default does:
    plain run, TVB on Q2 (limiter marks through the split launches), positivity inside the stage kernel on bilinear cells."""
    monkeypatch.setenv("DFLO_OVERLAP", mode)
    test_two_engines_match_one(CASES[index])


@pytest.mark.parametrize("index", [0, 3])
def test_three_engines_match_one(index):
    """Three slabs: the middle rank has a neighbour on either side and, on the periodic box, the outer ranks are neighbours
    through the wrap -- every rank of the 8-GPU run is in that position."""
    test_two_engines_match_one(CASES[index], world=3)


@pytest.mark.parametrize("case", CASES)
def test_two_engines_match_one(case, world=2):
    import random
    import dflo_amd
    from dflo_amd import problems
    import oracle_lib as O
    port = 29500 + random.randint(0, 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, case, ret), nprocs=world, join=True)
    nx, ny, degree, flux, limiter, pos, side_bc, bnd, ic_name = case[:9]
    ic = {"sod": problems.sod, "vortex": problems.isentropic_vortex, "front": _front, "smooth": _smooth}[ic_name]
    x0, h = (-5.0, 10.0 / nx) if ic_name == "vortex" else (0.0, 1.0 / nx)
    mesh = _mesh(dflo_amd, nx, ny, x0, h, side_bc, degree)
    prm = dflo_amd.Parameters(flux=flux, limiter=limiter, pos_lim=pos, boundary=bnd, beta=2.0, cfl=0.8 if side_bc != "unstructured" else 0.4,
                              shock_indicator=case[9] if len(case) > 9 else "limiter")
    ora = O.Oracle(mesh, prm)
    cell, face, bid, xy = ora.boundary_faces()
    if len(cell):
        bv = np.stack(ic(xy[..., 0], xy[..., 1]), axis=-1)
        ora.set_boundary_values(0, bv)
        ora.set_boundary_values(1, bv)
    ora.set_solution(mesh.interpolate(ic))
    t = 0.0
    for it, dt2 in enumerate(ret["dts"][:-1]):
        dt = ora.compute_time_step(t)
        assert abs(dt - dt2) <= 1e-12 * dt
        r0, r1 = ora.step(dt)
        if it == 0:
            assert abs(ret["norms"][0] - r0) <= 1e-10 * r0 and abs(ret["norms"][1] - r1) <= 1e-10 * r1
        t += dt
    for it in range(2):
        dt = ora.compute_time_step(t)
        ora.step(dt)
        t += dt
    assert abs(t - ret["dts"][-1]) <= 1e-12 * t
    uo = ora.get_solution()
    tol = 1e-11 if limiter == "none" else 1e-8
    assert np.abs(ret["u"] - uo).max() / np.abs(uo).max() < tol
