"""Randomised differential test, device against oracle: random small meshes (squares / skewed bilinear cells / unstructured
quadrilaterals), degrees, fluxes, boundary kinds, limiter settings, time-step modes; a few steps each.
usage: python tools/fuzz_parity.py [n_cases] [seed] [max_degree = 3]     (GPU box; prints the failing configurations)"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import dflo_amd
from dflo_amd import problems, gmsh
import oracle_lib

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_degree = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rng = np.random.default_rng(seed)
KINDS = ["inflow", "outflow", "slip", "pressure", "farfield"]


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


last = {}


def one(i):
    degree = int(rng.integers(0, max_degree + 1))
    flux = str(rng.choice(["lxf", "sw", "kfvs", "roe", "hllc"]))
    geo = str(rng.choice(["cart", "cart", "skew", "unstr"]))
    basis = "Pk" if (geo == "cart" and rng.random() < 0.2) else "Qk"
    tvb = geo == "cart" and rng.random() < 0.5
    pos = rng.random() < 0.6
    local = rng.random() < 0.15
    gravity = float(rng.choice([0.0, 0.0, 0.4]))
    M = float(rng.choice([0.0, 1.0, 50.0]))
    char_lim = bool(rng.random() < 0.6)
    periodic = geo == "cart" and rng.random() < 0.4
    last.clear()
    desc = last
    desc.update(i=i, degree=degree, flux=flux, geo=geo, basis=basis, tvb=tvb, pos=pos, local=local, gravity=gravity, M=M,
                char_lim=char_lim, periodic=periodic)
    if geo == "cart":
        nx, ny = int(rng.integers(1, 23)), int(rng.integers(1, 19))
        h = 1.0 / max(nx, ny)
        side = [-1] * 4 if periodic else [int(b) for b in rng.integers(0, 4, 4)]
        if not periodic and rng.random() < 0.3:
            side[0] = side[1] = -1
        if not periodic and rng.random() < 0.3:
            side[2] = side[3] = -1
        mesh = dflo_amd.Mesh.cartesian(nx, ny, 0.0, 0.0, h, side, degree)
        desc.update(nx=nx, ny=ny, side=side)
    elif geo == "skew":
        n = int(rng.integers(3, 13))
        from test_gpu_parity import skewed_mesh
        mesh = skewed_mesh(n, degree)
        desc.update(n=n)
    else:
        n = int(rng.integers(3, 10))
        verts, quads, bed, bid = gmsh.unstructured_quads(n, seed=int(rng.integers(0, 100)))
        mesh = dflo_amd.Mesh.from_quads(verts, quads, bed, bid, degree)
        desc.update(n=n)
    mesh.set_basis(basis)
    bnd = {b: str(rng.choice(KINDS)) for b in range(4)}
    indicator = str(rng.choice(["limiter", "limiter", "density", "energy"])) if tvb else "limiter"
    desc.update(indicator=indicator)
    prm = dflo_amd.Parameters(flux=flux, limiter="TVB" if tvb else "none", char_lim=char_lim, pos_lim=pos, M=M, beta=float(rng.choice([1.0, 1.5, 2.0])),
                              boundary=bnd, cfl=0.5, gravity=gravity, time_step_type="local" if local else "global", shock_indicator=indicator)
    desc.update(bnd=bnd)
    ic = lambda x, y: problems.smooth_perturbation(x, y, L=1.0)
    claw, ora = dflo_amd.ConservationLaw(mesh, prm), oracle_lib.Oracle(mesh, prm)
    cell, face, bid, xy = claw.boundary_faces()
    if len(cell):
        bv = np.stack(ic(xy[..., 0], xy[..., 1]), axis=-1)
        bv[..., 3] = np.where(np.array([bnd[int(b)] == "pressure" for b in bid])[:, None], 1.0, bv[..., 3])
        for w in (0, 1):
            claw.set_boundary_values(w, bv)
            ora.set_boundary_values(w, bv)
    u0 = mesh.project(ic) if basis == "Pk" else mesh.interpolate(ic)
    if rng.random() < 0.5:   # kinks and rough cells, so that the limiters have work
        u = u0.reshape(mesh.n_cells, 4, -1).copy()
        k = rng.integers(0, mesh.n_cells, max(1, mesh.n_cells // 5))
        amp = float(rng.choice([0.5, 2.0]))
        u[k, 2] *= 1.0 + amp * rng.random((len(k), 1))
        u[k, 3] *= 1.0 + amp * rng.random((len(k), 1))
        if basis == "Qk" and rng.random() < 0.5:     # nodal roughness inside some cells (positivity limiter)
            k2 = rng.integers(0, mesh.n_cells, max(1, mesh.n_cells // 9))
            u[k2, 2] *= 1.0 + 0.6 * (rng.random((len(k2), u.shape[2])) - 0.5)
        u0 = u.reshape(-1)
        desc.update(kink=amp)
    claw.set_initial_condition(u0)
    ora.set_solution(u0)
    if os.environ.get("FUZZ_ONLY") == str(i):    # diagnostics for one case (the generator has to run through the others)
        print("case", desc)
        for fused in ("1", "0"):
            os.environ["DFLO_FUSE_POS"] = fused
            c2, o2 = dflo_amd.ConservationLaw(mesh, prm), oracle_lib.Oracle(mesh, prm)
            if len(cell):
                for w in (0, 1):
                    c2.set_boundary_values(w, bv)
                    o2.set_boundary_values(w, bv)
            c2.set_initial_condition(u0)
            o2.set_solution(u0)
            tt = 0.0
            for it in range(3):
                dt = o2.compute_time_step(tt)
                dtc = c2.compute_time_step()
                print("   dt", dt, dtc)
                o2.set_dt(dt) if not local else None
                for rk in range(o2.n_rk):
                    c2.stage(rk, dt)
                    o2.stage(rk)
                    a_, b_ = c2.current_solution.reshape(mesh.n_cells, -1), o2.get_solution().reshape(mesh.n_cells, -1)
                    nd, no = ~np.isfinite(a_).all(axis=1), ~np.isfinite(b_).all(axis=1)
                    ok = ~(nd | no)
                    d = np.abs(a_[ok] - b_[ok]).max(axis=1) if ok.any() else np.zeros(1)
                    print("fused", fused, "step", it, "stage", rk, "nan dev/ora", nd.sum(), no.sum(), np.where(nd)[0][:6], "max diff", d.max(), "at cell", np.where(ok)[0][np.argmax(d)])
                c2.end_step(); o2.end_step(); tt += dt
        os.environ.pop("DFLO_FUSE_POS")
    r1, r2 = claw.assemble_system(), ora.assemble()
    if not np.isfinite(r2).all():
        # rough data whose trace on a face has a negative pressure (the higher the degree, the wilder the extrapolation to the
        # faces): sw / kfvs / roe take the root of it.  The reference's dense lifting loops spread the NaN over every DoF of both
        # cells (0 * NaN), the collocated lifting only over the DoFs the face point feeds -- a subset; the finite rest must agree.
        c1, c2 = r1.reshape(mesh.n_cells, -1), r2.reshape(mesh.n_cells, -1)
        nd, no = ~np.isfinite(c1).all(axis=1), ~np.isfinite(c2).all(axis=1)
        assert (nd <= no).all(), ("residual: device NaN cells outside the oracle's", int(nd.sum()), int(no.sum()))
        ok = ~no
        if ok.any():
            assert np.abs(c1[ok] - c2[ok]).max() <= 1e-11 * np.abs(c2[ok]).max(), ("residual (finite cells)", np.abs(c1[ok] - c2[ok]).max())
        raise oracle_lib.OracleError(0, "NaN state")
    # (degrees 4 and 5: the entries of the derivative matrix grow with the degree -- max |D| = 11, 17, 23 for k = 3, 4, 5 -- and the
    #  round-off of the two orders of summation with them; seen: 1.0e-11 / 3e-10 at k = 5 on small distorted cells)
    loose = {4: 4.0, 5: 10.0}.get(degree, 1.0)   # (every bar below carries it, the "agreed after the first step" of the classifications too)
    # (a residual that vanishes -- one periodic cell that is its own neighbour -- is compared on the scale of the fluxes)
    rscale = max(np.abs(r2).max(), 1e-6 * np.abs(u0).max())
    assert np.abs(r1 - r2).max() < 1e-11 * loose * rscale, ("residual", np.abs(r1 - r2).max() / rscale)
    t = 0.0
    e1 = None   # agreement after the first step
    for it in range(3):
        if not np.isfinite(ora.get_solution()).all():   # the reference's own arithmetic has broken down: the NaN cells are compared below
            break
        dt = ora.compute_time_step(t)
        dtc = claw.compute_time_step()
        if it == 0:
            dt_first = dt
        if abs(dtc - dt) > (1e-9 if "kink" in desc else 1e-11) * loose * dt:
            if e1 is not None and e1 <= 1e-12 * loose and ("kink" in desc or dt < 1e-2 * dt_first):
                # (rough data, or a run whose time step has collapsed a hundredfold within two steps: it is blowing up)
                raise oracle_lib.OracleError(3, "round-off amplified by rough data")   # see below
            if np.abs(ora.get_solution()).max() > 1.0e3 * np.abs(u0).max():
                # the blow-up class of the end of this function, met at a time step already: the reference state has grown a
                # thousandfold (an unlimited run on rough data), and the time steps of two such states differ like the states
                raise oracle_lib.OracleError(2, "blow-up of the reference solution")
            assert False, ("dt", it, dtc, dt)
        claw.iterate_explicit(dt)
        ora.step(-1.0 if local else dt)   # local time stepping: keep the per-cell steps compute_time_step has left
        t += dt
        if it == 0 and np.isfinite(ora.get_solution()).all():
            e1 = rel(claw.current_solution, ora.get_solution())
    if not local and rng.random() < 0.5:   # two more steps with the time step resident on the device
        t2 = claw.advance(2)
        for it in range(2):
            dt = ora.compute_time_step(t)
            ora.step(dt)
            t += dt
        if np.isfinite(t) and np.isfinite(claw.current_solution).all():   # (a device NaN is classified below)
            if abs(t2 - t) > 1e-9 * t and e1 is not None and e1 <= 1e-12 * loose and "kink" in desc:
                raise oracle_lib.OracleError(3, "round-off amplified by rough data")
            assert abs(t2 - t) <= 1e-9 * loose * t, ("advance time", t2, t)
        desc.update(advance=True)
    tol = (1e-8 if (tvb or pos or "kink" in desc) else 1e-10) * loose     # (jumps amplify the round-off of the fluxes)
    ud, uo = claw.current_solution, ora.get_solution()
    if np.isfinite(uo).all() and np.abs(uo).max() > 1.0e3 * np.abs(u0).max():
        # an unlimited run on rough data that is blowing up (the state has grown a thousandfold in a few steps, the time step has
        # collapsed): round-off differences grow with it -- 1e-14 after the first step, 1e-9 after the second, O(1) after the third
        raise oracle_lib.OracleError(2, "blow-up of the reference solution")
    if not np.isfinite(uo).all():      # the reference's own arithmetic has produced NaNs: the device has to have them in the same cells
        nd = ~np.isfinite(ud.reshape(mesh.n_cells, -1)).all(axis=1)
        no = ~np.isfinite(uo.reshape(mesh.n_cells, -1)).all(axis=1)
        if not (nd <= no).all() and pos and flux in ("sw", "kfvs", "roe") and "kink" in desc:
            raise oracle_lib.OracleError(1, "device NaN earlier than the reference's (cold point)")   # see below
        assert (nd <= no).all(), ("device NaN cells outside the oracle's", int(nd.sum()), int(no.sum()))
        desc.update(nan_cells=(int(nd.sum()), int(no.sum())))
        raise oracle_lib.OracleError(0, "NaN state")
    if not np.isfinite(ud).all():
        # Seen with rough data and the positivity limiter on (sw / kfvs / roe): a point the limiter has left at p = 1e-13
        # gets a pressure of the other sign from the device's reciprocal-based arithmetic and the flux takes the root of it,
        # steps after both solutions agreed to 1e-15 (FUZZ_ONLY=<case> prints the stage-by-stage history).  Not a parity
        # statement either way; reported, not failed.
        raise oracle_lib.OracleError(1, "device NaN, reference finite (cold point)")
    e = rel(ud, uo)
    if e >= tol and e1 is not None and e1 <= 1e-12 * loose and "kink" in desc:
        # Rough data (cells scaled by up to 3, nodal noise) on a few cells is not a resolved flow: the solutions agree to round-off
        # after the first step (e1) and the difference then grows by one to two orders of magnitude per stage -- limiter switches,
        # points left at p = 1e-13 -- until it passes the bar in the second or third step.  Seen 3 times in 40 000 cases at degrees
        # 0-3, identically on the kernels of round 2 (same seeds, same cases), more often at degrees 4 and 5.  Reported, not failed.
        raise oracle_lib.OracleError(3, "round-off amplified by rough data")
    assert e < tol, ("solution", e)
    return desc


fails = 0
skipped = {}
t0 = time.time()
for i in range(n_cases):
    try:
        one(i)
    except oracle_lib.OracleError as e:
        skipped["oracle: " + str(e)[:40]] = skipped.get("oracle: " + str(e)[:40], 0) + 1   # inadmissible for the reference too
    except dflo_amd.DfloError as e:
        if e.code in (-3, -4, -7, -1):   # negative states / root failure / unsupported combination / refused parameters
            skipped["device %d" % e.code] = skipped.get("device %d" % e.code, 0) + 1
            continue
        fails += 1
        print("CASE %d device error: %s" % (i, e))
    except AssertionError as e:
        fails += 1
        print("CASE %d FAILED: %s  %s" % (i, e, {k: v for k, v in last.items() if k != "bnd"}))
print("%d cases, %d failures, %d not compared %s, %.1f s" % (n_cases, fails, sum(skipped.values()), skipped, time.time() - t0))
sys.exit(1 if fails else 0)
