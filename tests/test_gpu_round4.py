"""GPU parity of the round-4 paths: two ways of doing the same thing, held to each other bit for bit (and, through the other
test files, to the oracle): the LxF flux's (u, v, c) of the cell averages from the DoFs against the stored arrays of averages."""
import numpy as np
import pytest

import dflo_amd
from dflo_amd import problems
import oracle_lib

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _run(claw, u0, bfun=None, steps=4, resident=9):
    if bfun is not None:
        cell, face, bid, xy = claw.boundary_faces()
        bv = np.stack(bfun(xy[..., 0], xy[..., 1]), axis=-1)
        claw.set_boundary_values(0, bv)
        claw.set_boundary_values(1, bv)
    claw.set_initial_condition(u0)
    hist = []
    for it in range(steps):
        dt = claw.compute_time_step()
        hist.append((dt,) + tuple(claw.iterate_explicit(dt)))
    hist.append(claw.advance(resident))
    hist.append(claw.compute_time_step())
    hist.append(claw.advance(2))
    return hist, claw.current_solution.copy(), claw.cell_average.copy()


def test_step_index_slots_under_graph_replay(monkeypatch):
    """The reductions of step s write the index of step s + 1 into the slot of the other parity (the slot the kernels of the step in
    flight read -- to name the step in a failure flag -- is never written while they run): a captured two-step graph names the
    slots in a fixed order and is replayed from the parity it was captured at only."""
    mesh = dflo_amd.Mesh.cartesian(40, 24, -5.0, -5.0, 0.25, [-1] * 4, 2)
    prm = dflo_amd.Parameters(flux="hllc", cfl=0.7)
    u0 = mesh.interpolate(problems.isentropic_vortex)
    out = []
    for flag in ("0", "1"):
        monkeypatch.setenv("DFLO_GRAPH", flag)
        claw = dflo_amd.ConservationLaw(mesh, prm)
        claw.set_initial_condition(u0)
        ts = [claw.advance(11)]
        dt = claw.compute_time_step()
        claw.iterate_explicit(dt)        # one plain step: the parity flips
        ts.append(claw.advance(8))
        out.append((ts, claw.current_solution.copy()))
        claw.close()
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("degree", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("walls", [False, True])
def test_lxf_wave_speeds_from_the_dofs_give_the_bits_of_the_stored_averages(degree, walls, monkeypatch):
    """lxf_flux takes its lambda from the two CELL AVERAGES (src/equation.h:357-359).  On squares without a limiter the stage
    kernel forms (u, v, c) of the averages of its own and of the halo cells from the DoFs it loads anyway, summing them exactly
    as the epilogue of the previous stage summed the stored average (cell_average_rows), and neither reads nor -- in intermediate
    stages -- writes the array of averages.  DFLO_LXF_FROM_DOFS=0 restores the arrays: residual, time steps, norms and states
    agree bit for bit, and both agree with the oracle."""
    nx, ny = 52, 37     # partial shards on two sides
    if walls:
        mesh = dflo_amd.Mesh.cartesian(nx, ny, 0.0, 0.0, 1.0 / nx, [2, 1, 0, 0], degree)
        prm = dflo_amd.Parameters(flux="lxf", cfl=0.6, boundary={0: "slip", 1: "outflow", 2: "inflow"})
        bfun = problems.sod
        ic = lambda x, y: [0.3 + 0.05 * np.sin(7 * x), 0.02 * np.cos(9 * y), 1.0 + 0.1 * np.sin(5 * x) * np.cos(3 * y), 2.5 + 0.1 * np.cos(4 * x + y)]
    else:
        mesh = dflo_amd.Mesh.cartesian(nx, ny, -5.0, -5.0, 10.0 / nx, [-1] * 4, degree)
        prm = dflo_amd.Parameters(flux="lxf", cfl=0.8)
        bfun = None
        ic = problems.isentropic_vortex
    u0 = mesh.interpolate(ic)
    out, res = [], []
    for flag in ("1", "0"):
        monkeypatch.setenv("DFLO_LXF_FROM_DOFS", flag)
        claw = dflo_amd.ConservationLaw(mesh, prm)
        if bfun is not None:
            cell, face, bid, xy = claw.boundary_faces()
            bv = np.stack(bfun(xy[..., 0], xy[..., 1]), axis=-1)
            claw.set_boundary_values(0, bv)
            claw.set_boundary_values(1, bv)
        claw.set_initial_condition(u0)
        res.append(claw.assemble_system().copy())
        out.append(_run(claw, u0, bfun, steps=3, resident=5))
        claw.close()
    assert np.array_equal(res[0], res[1])
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    ora = oracle_lib.Oracle(mesh, prm)
    if bfun is not None:
        cell, face, bid, xy = ora.boundary_faces()
        bv = np.stack(bfun(xy[..., 0], xy[..., 1]), axis=-1)
        ora.set_boundary_values(0, bv)
        ora.set_boundary_values(1, bv)
    ora.set_solution(u0)
    assert rel(res[0], ora.assemble()) < (1e-12 if degree < 4 else 4e-12)


@pytest.mark.parametrize("degree", [1, 2, 3])
def test_lxf_on_the_modal_basis_takes_the_average_from_mode_zero(degree):
    """Pk: the cell average is mode 0 (src/limiter.cc:412-420), so the LxF flux's wave speeds come from the modes the kernel loads
    anyway, for own and halo cells alike, with or without a limiter.  Against the oracle, 5 steps with the TVB limiter on."""
    mesh = dflo_amd.Mesh.cartesian(28, 20, -5.0, -5.0, 10.0 / 28, [-1] * 4, degree)
    mesh.set_basis("Pk")
    for lim in (False, True):
        prm = dflo_amd.Parameters(flux="lxf", cfl=0.5, **({"limiter": "TVB", "M": 10.0, "beta": 1.0, "pos_lim": True} if lim else {}))
        claw = dflo_amd.ConservationLaw(mesh, prm)
        ora = oracle_lib.Oracle(mesh, prm)
        u0 = mesh.project(problems.isentropic_vortex)
        claw.set_initial_condition(u0)
        ora.set_solution(u0)
        assert rel(claw.assemble_system(), ora.assemble()) < 1e-12
        t = 0.0
        for it in range(5):
            dt = claw.compute_time_step()
            dto = ora.compute_time_step(t)
            assert abs(dt - dto) <= 1e-12 * dto
            claw.iterate_explicit(dt)
            ora.step(dt)
            t += dt
        assert rel(claw.current_solution, ora.get_solution()) < 1e-10
        claw.close()
