"""GPU parity of the round-5 paths: two ways of doing the same thing, held to each other bit for bit (and, through the other test
files, to the oracle).

  * boundary programs taken along by the limiter pass behind stage 0 (stage 0 reads the table the previous step's later stages
    used) against bc_eval_kernel before every step -- src/claw.cc:736-745: the boundary values of stage 0 at t, of the later stages
    at t + dt;
  * a standalone positivity / TVB pass after a device-resident run whose last stage kept its cell averages to itself
    (src/positivity.cc:17-208 reads cell_average)."""
import numpy as np
import pytest

import dflo_amd
from dflo_amd import problems

pytestmark = pytest.mark.gpu


def _dmr(nx=257, ny=64, tvb=True):
    """the double Mach reflection in small: moving inflow state on the top wall (id 3) as device programs in t"""
    dy = 1.0 / ny
    n1 = int(np.ceil((1.0 / 6.0) / dy))
    mesh = dflo_amd.Mesh.cartesian(nx, ny, 1.0 / 6.0 - n1 * dy, 0.0, dy, [4, 2, 1, 3], 2)
    mesh.neighbors[:n1, 2] = -1 - 0
    prm = dflo_amd.Parameters(flux="hllc", limiter="TVB" if tvb else "none", char_lim=True, pos_lim=True, M=100.0, beta=1.0, cfl=0.9, final_time=1e9,
                              boundary={0: "outflow", 1: "slip", 2: "outflow", 3: "inflow", 4: "inflow"})
    sh = "(x<1.0/6.0+(1+20*t)/sqrt(3))"
    programs = {3: ["57.1576766498*" + sh, "-33.0*" + sh, "8.0*%s + 1.4*(1-%s)" % (sh, sh), "563.5*%s + 2.5*(1-%s)" % (sh, sh)]}
    return mesh, prm, programs


def _run_dmr(mesh, prm, programs, plan, unequal_tables=False):
    claw = dflo_amd.ConservationLaw(mesh, prm)
    cell, face, bid, xy = claw.boundary_faces()
    bv = np.stack(problems.double_mach(xy[..., 0], xy[..., 1]), axis=-1)
    claw.set_boundary_values(0, bv)
    claw.set_boundary_values(1, bv * (1.0 + 1e-9) if unequal_tables else bv)
    claw.set_initial_condition(mesh.interpolate(problems.double_mach))
    for b, exprs in programs.items():
        claw.set_boundary_function(b, exprs)
    claw.apply_limiter()
    ts = []
    for kind, n in plan:
        if kind == "advance":
            ts.append(claw.advance(n))
        elif kind == "step":
            for _ in range(n):
                dt = claw.compute_time_step()
                claw.iterate_explicit(dt)
                ts.append(dt)
        else:   # the host re-uploads one table in the middle of the run
            claw.set_boundary_values(1, bv)
    out = (ts, claw.current_solution.copy(), claw.get_boundary_values(0).copy(), claw.get_boundary_values(1).copy())
    claw.close()
    return out


@pytest.mark.parametrize("plan", [[("advance", 13)], [("advance", 4), ("step", 2), ("advance", 5)], [("advance", 3), ("upload", 0), ("advance", 4)]])
def test_boundary_programs_taken_along_by_the_limiter_pass(plan, monkeypatch):
    mesh, prm, programs = _dmr()
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("DFLO_BC_FUSE", flag)
        res.append(_run_dmr(mesh, prm, programs, plan))
    (t0, u0, a0, b0), (t1, u1, a1, b1) = res
    assert t0 == t1 and np.array_equal(u0, u1)
    # the tables as the next step would find them: "stage 0" holds the values at the clock's t, "later stages" those at t + dt of the
    # last step -- in both arrangements
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1)
    if plan == [("advance", 13)]:   # ... and the programs did run: the density on the top wall (id 3) at the clock's time, closed form
        probe = dflo_amd.ConservationLaw(mesh, prm)
        cell, face, bid, xy = probe.boundary_faces()
        probe.close()
        top = bid == 3
        foot = 1.0 / 6.0 + (1.0 + 20.0 * t1[-1]) / np.sqrt(3.0)
        want = np.where(xy[top][..., 0] < foot, 8.0, 1.4)
        assert (b1[top][..., 2] == want).mean() > 0.999 and set(np.unique(b1[top][..., 2])) == {1.4, 8.0}


def test_tables_that_differ_where_no_program_writes_are_not_swapped(monkeypatch):
    """the swap of the tables' roles is only right while the uploaded entries of the two are the same"""
    mesh, prm, programs = _dmr(nx=129, ny=32)
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("DFLO_BC_FUSE", flag)
        res.append(_run_dmr(mesh, prm, programs, [("advance", 7)], unequal_tables=True))
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])


@pytest.mark.parametrize("flux,degree", [("hllc", 1), ("hllc", 2), ("hllc", 3)])
def test_standalone_positivity_pass_after_a_resident_run_with_lazy_averages(flux, degree, monkeypatch):
    """the last stage of a device-resident run without limiter passes keeps its cell averages to itself: a standalone
    apply_positivity_limiter() afterwards has to form them before it scales towards them (round 4 read those of before the run)"""
    mesh = dflo_amd.Mesh.cartesian(48, 40, 0.0, 0.0, 1.0 / 48, [0, 0, 0, 0], degree)
    # (no limiter in the run itself: its last stage keeps the averages to itself whatever the flux, and the unlimited blast leaves
    #  point values below the bounds for the standalone pass to repair)
    prm = dflo_amd.Parameters(flux=flux, pos_lim=False, cfl=0.3, boundary={0: "outflow"})

    def blast(x, y):
        r2 = (x - 0.5) ** 2 + (y - 0.4) ** 2
        p = np.where(r2 < 0.01, 1.0, 0.1)
        rho = np.where(r2 < 0.01, 1.0, 0.5)
        return [0 * x, 0 * x, rho, p / 0.4]

    out = []
    for lazy in ("0", "1"):
        monkeypatch.setenv("DFLO_LAZY_AVG", lazy)
        claw = dflo_amd.ConservationLaw(mesh, prm)
        claw.set_initial_condition(mesh.interpolate(blast))
        claw.advance(2)
        u = claw.current_solution.copy()
        claw.apply_positivity_limiter()
        out.append((u, claw.current_solution.copy(), claw.cell_average.copy()))
        claw.close()
    assert np.array_equal(out[0][0], out[1][0])
    # The stored averages are those of BEFORE the in-kernel limiter (src/claw.cc:762-766: compute_cell_average, then the limiters), the
    # ones formed on demand come from the limited DoFs: the same numbers to rounding (the limiter keeps the average), not the same bits.
    # Stale averages -- those of before the resident run, what the pass read before round 5 -- would be off by O(1).
    scale = np.abs(out[0][1]).max()
    assert np.abs(out[0][1] - out[1][1]).max() <= 1e-12 * scale and np.abs(out[0][2] - out[1][2]).max() <= 1e-12 * np.abs(out[0][2]).max()
    assert np.abs(out[0][1] - out[0][0]).max() > 1e-3 * scale      # (the pass had work to do)
    # conservation of the pass: the averages are those of the state before it
    mesh_avg = out[1][2]
    claw = dflo_amd.ConservationLaw(mesh, prm)
    claw.set_initial_condition(out[1][0])
    assert np.abs(claw.cell_average - mesh_avg).max() <= 1e-13 * np.abs(mesh_avg).max()
    claw.close()


def test_delivery_by_the_stage_kernel_driven_by_hand_through_the_c_abi():
    """The seam a transport of the host program's own would be written against (include/dflo_hip.h: dflo_hip_set_deliver,
    dflo_hip_stage_deliver): two engines of this process on the two x-slabs of a periodic box, each told once where the traces of its cut
    faces go -- straight into the other engine's ghost-trace tables -- and then one launch per engine and stage; here the host waits
    between the stages where the native driver uses the sequence words.  Bit-identical to the single engine
    (update_ghost_values of src_mpi/claw.cc:793 is part of the kernel that produced the values)."""
    import ctypes as C
    from dflo_amd._lib import lib
    hip = C.CDLL("libamdhip64.so.7")          # the runtime the engine library is linked to (already loaded): the sequence words are plain device memory
    words_p = C.c_void_p()
    assert hip.hipMalloc(C.byref(words_p), 512) == 0 and hip.hipMemset(words_p, 0, 512) == 0
    mesh = dflo_amd.Mesh.cartesian(64, 48, -5.0, -5.0, 10.0 / 64, [-1, -1, -1, -1], 2)
    prm = dflo_amd.Parameters(flux="hllc", cfl=0.8)
    u0 = mesh.interpolate(problems.isentropic_vortex)
    one = dflo_amd.ConservationLaw(mesh, prm)
    one.set_initial_condition(u0)
    dts = []
    for _ in range(5):
        dts.append(one.compute_time_step())
        one.iterate_explicit(dts[-1])
    ref = one.current_solution.copy()
    one.close()

    ndof, N = mesh.ndof, 3
    parts, engs = [mesh.partition(2, r) for r in range(2)], []
    for sub in parts:
        e = dflo_amd.ConservationLaw(sub, prm)
        assert lib.dflo_hip_halo_traces(e._h) == 1
        e.set_initial_condition(u0.reshape(mesh.n_cells, ndof)[sub.global_ids].reshape(-1))
        engs.append(e)
    # which faces travel: (owned cell, face) whose neighbour is a ghost cell -- by cell, then face: the order in which the OTHER part's
    # plan numbers the traces of its ghost cells (ghost cells sorted by global id, owned cells keep the global order)
    for r, (sub, e) in enumerate(zip(parts, engs)):
        nb = np.asarray(sub.neighbors)
        cells, faces = [], []
        for c in range(sub.n_owned):
            for f in range(4):
                if nb[c, f] >= sub.n_owned:
                    cells.append(c)
                    faces.append(f)
        cells, faces = np.asarray(cells, dtype=np.int32), np.asarray(faces, dtype=np.int32)
        assert len(cells) == lib.dflo_hip_n_ghost_traces(engs[1 - r]._h) == 2 * 48
        assert lib.dflo_hip_set_send_faces(e._h, len(cells), cells.ctypes.data_as(C.POINTER(C.c_int32)), faces.ctypes.data_as(C.POINTER(C.c_int32))) == 0
        for area in range(2):
            tg = C.c_void_p()
            assert lib.dflo_hip_ghost_trace_buffer(engs[1 - r]._h, area, C.byref(tg)) == 0
            first = (C.c_int32 * 2)(0, len(cells))
            dst, fl = (C.c_void_p * 1)(tg.value), (C.c_void_p * 1)(words_p.value + 8 * r)
            assert lib.dflo_hip_set_deliver(e._h, area, 1, first, dst, fl) == 0, lib.dflo_hip_last_error(e._h)
    n = 0
    for dt in dts:
        for rk in range(3):
            area = (1 + n) & 1          # exchange n fills table (1 + n) & 1: never the one the engines read at that moment
            for e in engs:
                assert lib.dflo_hip_stage_open(e._h, rk, dt) == 0
                assert lib.dflo_hip_stage_deliver(e._h, area, n + 1) == 0
                assert lib.dflo_hip_stage_update_part(e._h, 0) == 0
                assert lib.dflo_hip_stage_finish(e._h) == 0
            for e in engs:
                assert lib.dflo_hip_synchronize(e._h) == 0
                assert lib.dflo_hip_use_ghost_traces(e._h, area) == 0
            n += 1
        for e in engs:
            assert lib.dflo_hip_end_step(e._h) == 0
    words = (C.c_uint64 * 2)()
    assert hip.hipMemcpy(words, words_p, 16, 2) == 0 and hip.hipFree(words_p) == 0
    assert words[0] == n and words[1] == n        # every exchange's number was published by the last delivering workgroup
    u = np.empty((mesh.n_cells, ndof))
    for sub, e in zip(parts, engs):
        u[sub.global_ids[: sub.n_owned]] = e.current_solution.reshape(sub.n_cells, ndof)[: sub.n_owned]
        e.close()
    assert np.array_equal(u.reshape(-1), ref)
