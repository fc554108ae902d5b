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


# ---------------------------------------------------------------- the modal basis and the KXRCF indicator on bilinear cells
def _skewed(n, degree, basis):
    from test_gpu_parity import skewed_mesh
    mesh = skewed_mesh(n, degree)
    mesh.set_basis(basis)
    return mesh


def _walls(claw, ora, ic):
    cell, face, bid, xy = claw.boundary_faces()
    bv = np.stack(ic(xy[..., 0], xy[..., 1]), axis=-1)
    for w in (0, 1):
        claw.set_boundary_values(w, bv)
        ora.set_boundary_values(w, bv)


@pytest.mark.parametrize("degree", [0, 1, 2, 3])
@pytest.mark.parametrize("flux", ["lxf", "hllc", "kfvs"])
def test_modal_basis_on_bilinear_cells(degree, flux):
    """FE_DGP lives on the reference cell, so ConservationLaw's Pk constructor takes any mapping (src/claw.cc:91-119) -- the
    reference's parameter file refuses the combination (src/parameters.cc:546-547), the C ABI does not.  On bilinear cells the
    residual needs the metric terms, the mass matrix is the diagonal sum_q psi_m^2 JxW_q the reference keeps
    (src/claw.cc:228-258), the cell average is a quadrature of the expansion and no longer mode 0 (src/claw.cc:589-593) and
    the time step is compute_time_step_q (src/claw.cc:520-557).  Residual, averages, time steps and five RK steps against the
    oracle on a mesh of genuinely non-affine cells, with the positivity limiter on for the steps."""
    mesh = _skewed(9, degree, "Pk")
    ic = lambda x, y: problems.smooth_perturbation(x, y, L=1.0)
    bnd = {1: "inflow", 2: "slip", 3: "outflow"}
    u0 = mesh.project(ic)
    for pos in (False, True):
        prm = dflo_amd.Parameters(flux=flux, boundary=bnd, cfl=0.4, pos_lim=pos)
        claw, ora = dflo_amd.ConservationLaw(mesh, prm), oracle_lib.Oracle(mesh, prm)
        _walls(claw, ora, ic)
        claw.set_initial_condition(u0)
        ora.set_solution(u0)
        assert rel(claw.cell_average, ora.get_cell_average()) < 1e-13
        if not pos:
            assert rel(claw.assemble_system(), ora.assemble()) < 1e-12
        t = 0.0
        for it in range(5):
            dt, dto = claw.compute_time_step(), ora.compute_time_step(t)
            assert abs(dt - dto) <= 1e-12 * dto
            r0, r1 = claw.iterate_explicit(dt)
            q0, q1 = ora.step(dt)
            assert abs(r1 - q1) <= 1e-10 * q1
            t += dt
        assert rel(claw.current_solution, ora.get_solution()) < 1e-11
        assert rel(claw.cell_average, ora.get_cell_average()) < 1e-11
        t2 = claw.advance(3)        # the same with the time step resident on the device
        for it in range(3):
            dto = ora.compute_time_step(t)
            ora.step(dto)
            t += dto
        assert abs(t2 - t) <= 1e-11 * t
        assert rel(claw.current_solution, ora.get_solution()) < 1e-10
        claw.close()


@pytest.mark.parametrize("basis,degree", [("Qk", 1), ("Qk", 2), ("Qk", 3), ("Pk", 2)])
@pytest.mark.parametrize("kind", ["density", "energy"])
def test_kxrcf_indicator_on_bilinear_cells(basis, degree, kind):
    """compute_shock_indicator_kxrcf (src/indicator.cc:51-198) on non-affine cells: inflow faces by the cell-mean velocity
    against the edge normal, jumps weighted with the edge length, diameter^((k+1)/2) in the denominator.  (The limiters it
    gates run on Cartesian cells only, src/parameters.cc:543-544: here it is the diagnostic of src/claw.cc:763.)"""
    mesh = _skewed(12, degree, basis)
    prm = dflo_amd.Parameters(flux="roe", shock_indicator=kind, boundary={1: "outflow", 2: "outflow", 3: "outflow"})
    claw, ora = dflo_amd.ConservationLaw(mesh, prm), oracle_lib.Oracle(mesh, prm)
    def ic(x, y):
        s = 0.5 * (1.0 + np.tanh((x + 0.5 * y - 0.8) / 0.02))
        rho, p, u, v = 1.0 + 0.6 * s, 1.0 + 0.9 * s, 0.6, 0.35
        return [rho * u, rho * v, rho, p / 0.4 + 0.5 * rho * (u * u + v * v)]
    u0 = mesh.interpolate(ic)
    claw.set_initial_condition(u0)
    ora.set_solution(u0)
    s, so = claw.compute_shock_indicator(), ora.compute_shock_indicator()
    assert (np.isnan(s) == np.isnan(so)).all()
    ok = ~np.isnan(so)
    assert np.abs(s[ok] - so[ok]).max() <= 1e-11 * np.abs(so[ok]).max()
    assert (so[ok] > 0).sum() > mesh.n_cells // 2
