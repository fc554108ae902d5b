"""GPU parity of the matrix-pipe variants of the stage kernels (DFLO_MFMA=1; north_star: "MFMA used only for the dense per-element
basis contractions at higher order", BASELINE config 5: "per-element MFMA basis contraction path").

What runs on the matrix pipe (degree 3):
  * Qk, squares and bilinear cells: the eta-derivative of phase C -- the dense ndof x n_q loops of the reference's cell term
    (src/assemble_explicit.cc:85-115) after sum factorisation -- one v_mfma_f64_4x4x4_4b per cell;
  * Pk on squares: the two dense tables of FE_DGP (src/main.cc:44-48, src/claw.cc:91-119), modal -> nodal and nodal -> modal, as
    v_mfma_f64_16x16x4 with 16 cells along the columns.
Every case is held to the ORACLE (the bars of test_gpu_parity.py: residual 1e-12, RK solution 1e-11, limited runs 1e-9 / 1e-8)
and to the vector-unit path of the same engine (same sums in another order: rounding only)."""
import numpy as np
import pytest

import dflo_amd
from dflo_amd import problems
import oracle_lib

pytestmark = pytest.mark.gpu

FLUXES = ["lxf", "sw", "kfvs", "roe", "hllc"]


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _claw(mesh, prm, mfma, monkeypatch):
    monkeypatch.setenv("DFLO_MFMA", "1" if mfma else "0")
    c = dflo_amd.ConservationLaw(mesh, prm)
    assert c.uses_mfma == bool(mfma)
    return c


def _mapped(degree, flux, monkeypatch, mfma, **kw):
    """the skewed bilinear-cell mesh of test_gpu_parity.mapped_pair"""
    from test_gpu_parity import skewed_mesh
    mesh = skewed_mesh(10, degree)
    bnd = {1: "inflow", 2: "slip", 3: "outflow"}
    prm = dflo_amd.Parameters(flux=flux, boundary=bnd, cfl=0.5, **kw)
    claw = _claw(mesh, prm, mfma, monkeypatch)
    ora = oracle_lib.Oracle(mesh, prm)
    u0 = mesh.interpolate(lambda x, y: problems.smooth_perturbation(x, y, L=1.0))
    cell, face, bid, xy = claw.boundary_faces()
    bv = np.stack(problems.smooth_perturbation(xy[..., 0], xy[..., 1], L=1.0), axis=-1)
    for w in (0, 1):
        claw.set_boundary_values(w, bv)
        ora.set_boundary_values(w, bv)
    claw.set_initial_condition(u0)
    ora.set_solution(u0)
    return mesh, claw, ora


def test_the_switch_is_off_by_default_and_only_degree_3_has_the_variants(monkeypatch):
    monkeypatch.delenv("DFLO_MFMA", raising=False)
    mesh = dflo_amd.Mesh.cartesian(8, 8, 0.0, 0.0, 0.125, [-1] * 4, 3)
    c = dflo_amd.ConservationLaw(mesh, dflo_amd.Parameters(flux="hllc"))
    assert not c.uses_mfma
    c.close()
    monkeypatch.setenv("DFLO_MFMA", "1")
    for degree, want in [(1, False), (2, False), (3, True), (4, False)]:
        mesh = dflo_amd.Mesh.cartesian(8, 8, 0.0, 0.0, 0.125, [-1] * 4, degree)
        c = dflo_amd.ConservationLaw(mesh, dflo_amd.Parameters(flux="hllc"))
        assert c.uses_mfma == want
        c.close()


@pytest.mark.parametrize("flux", FLUXES)
def test_q3_residual_squares(flux, monkeypatch):
    mesh = dflo_amd.Mesh.cartesian(20, 12, -5.0, -5.0, 0.5, [-1] * 4, 3)
    prm = dflo_amd.Parameters(flux=flux)
    u0 = mesh.interpolate(problems.smooth_perturbation)
    ora = oracle_lib.Oracle(mesh, prm)
    ora.set_solution(u0)
    ro = ora.assemble()
    r = []
    for mf in (1, 0):
        claw = _claw(mesh, prm, mf, monkeypatch)
        claw.set_initial_condition(u0)
        r.append(claw.assemble_system())
        claw.close()
    assert rel(r[0], ro) < 1e-12 and rel(r[1], ro) < 1e-12
    assert rel(r[0], r[1]) < 1e-13


@pytest.mark.parametrize("flux", FLUXES)
def test_q3_residual_bilinear_cells(flux, monkeypatch):
    """the kernel BASELINE config 5 names: Q3 on unstructured-type (bilinear) cells"""
    mesh, claw, ora = _mapped(3, flux, monkeypatch, 1)
    assert rel(claw.assemble_system(), ora.assemble()) < 1e-12
    claw.close()


@pytest.mark.parametrize("flux,pos", [("kfvs", True), ("hllc", False)])
def test_q3_rk_solution_bilinear_cells(flux, pos, monkeypatch):
    """C5-style: q1 mapping, compute_time_step_q, positivity inside the stage kernel -- six steps against the oracle, then the
    device-resident loop; and the vector-unit path of the same run agrees to rounding"""
    mesh, claw, ora = _mapped(3, flux, monkeypatch, 1, pos_lim=pos)
    t = 0.0
    for it in range(6):
        dt = claw.compute_time_step()
        dto = ora.compute_time_step(t)
        assert abs(dt - dto) <= 1e-12 * dto
        claw.iterate_explicit(dt)
        ora.step(dt)
        t += dt
    assert rel(claw.current_solution, ora.get_solution()) < 1e-11
    t2 = claw.advance(2)
    for it in range(2):
        dt = ora.compute_time_step(t)
        ora.step(dt)
        t += dt
    assert abs(t2 - t) < 1e-12 * t
    assert rel(claw.current_solution, ora.get_solution()) < 1e-11
    u_mf = claw.current_solution.copy()
    claw.close()
    mesh, plain, _ = _mapped(3, flux, monkeypatch, 0, pos_lim=pos)
    for it in range(6):
        plain.iterate_explicit(plain.compute_time_step())
    plain.advance(2)
    assert rel(plain.current_solution, u_mf) < 1e-12
    plain.close()


@pytest.mark.parametrize("flux", ["kfvs", "hllc", "lxf"])
def test_q3_rk_solution_vortex_squares(flux, monkeypatch):
    mesh = dflo_amd.Mesh.cartesian(16, 16, -5.0, -5.0, 10.0 / 16, [-1] * 4, 3)
    prm = dflo_amd.Parameters(flux=flux, cfl=0.9)
    claw = _claw(mesh, prm, 1, monkeypatch)
    ora = oracle_lib.Oracle(mesh, prm)
    u0 = mesh.interpolate(problems.isentropic_vortex)
    claw.set_initial_condition(u0)
    ora.set_solution(u0)
    t = 0.0
    for it in range(10):
        dt = claw.compute_time_step()
        dto = ora.compute_time_step(t)
        assert abs(dt - dto) <= 1e-13 * dto
        r0, r1 = claw.iterate_explicit(dt)
        q0, q1 = ora.step(dt)
        assert abs(r0 - q0) <= 1e-10 * q0 and abs(r1 - q1) <= 1e-10 * q1
        t += dt
    assert rel(claw.current_solution, ora.get_solution()) < 1e-11
    assert rel(claw.cell_average, ora.get_cell_average()) < 1e-11
    claw.close()


def test_q3_sod_tvb_positivity_with_the_limiter_marks(monkeypatch):
    """TVB + positivity on Q3 squares: the stage kernel's marks (POS 2) ride in the matrix-pipe variant as well"""
    bnd = {0: "slip", 1: "outflow", 2: "inflow"}
    nx, ny = 64, 8
    mesh = dflo_amd.Mesh.cartesian(nx, ny, 0.0, 0.0, 1.0 / nx, [2, 1, 0, 0], 3)
    prm = dflo_amd.Parameters(flux="roe", limiter="TVB", char_lim=True, pos_lim=True, M=0.0, beta=2.0, boundary=bnd, final_time=0.2)
    claw = _claw(mesh, prm, 1, monkeypatch)
    ora = oracle_lib.Oracle(mesh, prm)
    u0 = mesh.interpolate(problems.sod)
    cell, face, bid, xy = claw.boundary_faces()
    bv = np.zeros(xy.shape[:2] + (4,))
    bv[..., 2] = 1.0
    bv[..., 3] = 2.5
    for w in (0, 1):
        claw.set_boundary_values(w, bv)
        ora.set_boundary_values(w, bv)
    claw.set_initial_condition(u0)
    ora.set_solution(u0)
    claw.apply_limiter()
    ora.apply_limiter()
    t = 0.0
    for it in range(20):
        dt = claw.compute_time_step()
        dto = ora.compute_time_step(t)
        assert abs(dt - dto) <= 1e-12 * dto
        claw.iterate_explicit(dt)
        ora.step(dt)
        t += dt
    assert rel(claw.cell_average, ora.get_cell_average()) < 1e-9
    assert rel(claw.current_solution, ora.get_solution()) < 1e-8
    claw.close()


# ---------------------------------------------------------------- P3: the dense tables of the modal element
def _pk(nx, ny, flux, side_bc=(-1, -1, -1, -1), boundary=None, h=None, x0=-5.0, y0=-5.0, **kw):
    h = 10.0 / nx if h is None else h
    mesh = dflo_amd.Mesh.cartesian(nx, ny, x0, y0, h, list(side_bc), 3)
    mesh.set_basis("Pk")
    return mesh, dflo_amd.Parameters(flux=flux, boundary=boundary, **kw)


@pytest.mark.parametrize("flux", FLUXES)
def test_p3_residual_periodic(flux, monkeypatch):
    mesh, prm = _pk(20, 12, flux, h=0.5)
    u0 = mesh.project(problems.smooth_perturbation)
    ora = oracle_lib.Oracle(mesh, prm)
    ora.set_solution(u0)
    ro = ora.assemble()
    r = []
    for mf in (1, 0):
        claw = _claw(mesh, prm, mf, monkeypatch)
        claw.set_initial_condition(u0)
        assert rel(claw.cell_average, ora.get_cell_average()) < 1e-14
        r.append(claw.assemble_system())
        claw.close()
    assert rel(r[0], ro) < 1e-12 and rel(r[1], ro) < 1e-12
    assert rel(r[0], r[1]) < 1e-13


@pytest.mark.parametrize("nx,ny", [(1, 1), (3, 5), (9, 8), (17, 33)])
def test_p3_ragged_meshes(nx, ny, monkeypatch):
    """partial shards: lanes of the matrix layout that hold no cell, tiles of 16 cells that are half empty"""
    mesh, prm = _pk(nx, ny, "hllc", h=10.0 / max(nx, ny))
    u0 = mesh.project(lambda x, y: problems.smooth_perturbation(x, y, L=10.0))
    claw = _claw(mesh, prm, 1, monkeypatch)
    ora = oracle_lib.Oracle(mesh, prm)
    claw.set_initial_condition(u0)
    ora.set_solution(u0)
    assert rel(claw.assemble_system(), ora.assemble()) < 1e-12
    dt = claw.compute_time_step()
    assert abs(dt - ora.compute_time_step(0.0)) <= 1e-13 * dt
    claw.iterate_explicit(dt)
    ora.step(dt)
    assert rel(claw.current_solution, ora.get_solution()) < 1e-12
    claw.close()


@pytest.mark.parametrize("flux", ["kfvs", "lxf", "roe"])
def test_p3_rk_solution_vortex(flux, monkeypatch):
    mesh, prm = _pk(16, 16, flux, cfl=0.9)
    claw = _claw(mesh, prm, 1, monkeypatch)
    ora = oracle_lib.Oracle(mesh, prm)
    u0 = mesh.project(problems.isentropic_vortex)
    claw.set_initial_condition(u0)
    ora.set_solution(u0)
    t = 0.0
    for it in range(10):
        dt = claw.compute_time_step()
        dto = ora.compute_time_step(t)
        assert abs(dt - dto) <= 1e-13 * dto
        r0, r1 = claw.iterate_explicit(dt)
        q0, q1 = ora.step(dt)
        assert abs(r0 - q0) <= 1e-10 * q0 and abs(r1 - q1) <= 1e-10 * q1
        t += dt
    assert rel(claw.current_solution, ora.get_solution()) < 1e-11
    assert rel(claw.cell_average, ora.get_cell_average()) < 1e-11
    # the device-resident loop (time step from the last stage's mode 0) against the vector-unit path of the same run
    t_mf = claw.advance(5)
    u_mf = claw.current_solution.copy()
    claw.close()
    plain = _claw(mesh, prm, 0, monkeypatch)
    plain.set_initial_condition(u0)
    for it in range(10):
        plain.iterate_explicit(plain.compute_time_step())
    t_pl = plain.advance(5)
    assert abs(t_mf - t_pl) <= 1e-13 * t_pl
    assert rel(plain.current_solution, u_mf) < 1e-12
    plain.close()


def test_p3_boundaries(monkeypatch):
    bnd = {0: "farfield", 1: "outflow", 2: "inflow", 3: "slip"}
    mesh, prm = _pk(12, 9, "roe", side_bc=(0, 1, 2, 3), boundary=bnd, h=1.0 / 12, x0=0.0, y0=0.0)
    claw = _claw(mesh, prm, 1, monkeypatch)
    ora = oracle_lib.Oracle(mesh, prm)
    u0 = mesh.project(lambda x, y: problems.smooth_perturbation(x, y, L=1.0))
    claw.set_initial_condition(u0)
    ora.set_solution(u0)
    cell, face, bid, xy = claw.boundary_faces()
    bv = np.stack(problems.smooth_perturbation(xy[..., 0] + 0.01, xy[..., 1] - 0.02, L=1.0), axis=-1)
    for which in (0, 1):
        claw.set_boundary_values(which, bv)
        ora.set_boundary_values(which, bv)
    assert rel(claw.assemble_system(0), ora.assemble(0)) < 1e-12
    claw.close()


def test_p3_sod_tvb_positivity(monkeypatch):
    """apply_limiter_TVB_Pk + the Pk branch of the positivity limiter behind the matrix-pipe stage kernel"""
    bnd = {0: "slip", 1: "outflow", 2: "inflow"}
    nx, ny = 64, 8
    mesh = dflo_amd.Mesh.cartesian(nx, ny, 0.0, 0.0, 1.0 / nx, [2, 1, 0, 0], 3)
    mesh.set_basis("Pk")
    prm = dflo_amd.Parameters(flux="roe", limiter="TVB", char_lim=True, pos_lim=True, M=0.0, beta=2.0, boundary=bnd, final_time=0.2)
    claw = _claw(mesh, prm, 1, monkeypatch)
    ora = oracle_lib.Oracle(mesh, prm)
    u0 = mesh.project(problems.sod)
    cell, face, bid, xy = claw.boundary_faces()
    bv = np.zeros(xy.shape[:2] + (4,))
    bv[..., 2] = 1.0
    bv[..., 3] = 2.5
    for w in (0, 1):
        claw.set_boundary_values(w, bv)
        ora.set_boundary_values(w, bv)
    claw.set_initial_condition(u0)
    ora.set_solution(u0)
    claw.apply_limiter()
    ora.apply_limiter()
    t = 0.0
    for it in range(20):
        dt = claw.compute_time_step()
        dto = ora.compute_time_step(t)
        assert abs(dt - dto) <= 1e-12 * dto
        claw.iterate_explicit(dt)
        ora.step(dt)
        t += dt
    assert rel(claw.cell_average, ora.get_cell_average()) < 1e-9
    assert rel(claw.current_solution, ora.get_solution()) < 1e-8
    claw.close()


def test_p3_local_time_stepping_and_gravity(monkeypatch):
    """the per-cell time step and the forcing reach the matrix layout's lanes (dt_cell of the cell a lane updates)"""
    mesh, prm = _pk(12, 10, "hllc", h=0.1, x0=0.0, y0=0.0, gravity=0.3, time_step_type="local", cfl=0.4)
    claw = _claw(mesh, prm, 1, monkeypatch)
    ora = oracle_lib.Oracle(mesh, prm)
    u0 = mesh.project(lambda x, y: problems.smooth_perturbation(x, y, L=1.2))
    claw.set_initial_condition(u0)
    ora.set_solution(u0)
    t = 0.0
    for it in range(3):
        dt = claw.compute_time_step()
        dto = ora.compute_time_step(t)
        assert abs(dt - dto) <= 1e-12 * dto          # (also fills the oracle's per-cell dt)
        claw.iterate_explicit(dt)
        ora.step(-1.0)                                # keep the per-cell time steps
        t += dt
    assert rel(claw.current_solution, ora.get_solution()) < 1e-11
    claw.close()


def test_p3_in_two_parts_agrees_with_the_single_engine(monkeypatch):
    """whole ghost cells (the modal basis travels as cells): the matrix-pipe kernel in the multi-device schedule"""
    mesh, prm = _pk(24, 16, "hllc", cfl=0.9)
    u0 = mesh.project(problems.isentropic_vortex)
    one = _claw(mesh, prm, 1, monkeypatch)
    one.set_initial_condition(u0)
    t1 = one.advance(4)
    two = dflo_amd.MultiConservationLaw(mesh, prm, devices=[0, 0])
    assert two.uses_mfma
    two.set_initial_condition(u0)
    t2 = two.advance(4)
    assert abs(t1 - t2) <= 1e-13 * t1
    assert rel(two.current_solution, one.current_solution) < 1e-13
    one.close()
    two.close()


def test_random_configurations_on_the_matrix_pipe_kernels():
    """a slice of the differential fuzzers with DFLO_MFMA=1: 200 random configurations against the oracle (every degree-3 case, Qk and Pk,
    squares / skewed / unstructured cells, every flux, limiter and boundary kind, runs the matrix-pipe variants), 100 in several parts
    against the single engine (profiles/r06/fuzz_mfma.txt: 4 500 + 600 + 80 cases on the round's build, no failure)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DFLO_MFMA="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "200", "77"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "200 cases, 0 failures" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_multi.py"), "100", "78"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "100 cases, 0 failures" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]
