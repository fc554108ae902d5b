"""GPU parity at degrees 4 and 5 (round 3: DFLO_MAX_DEGREE 3 -> 5; the reference takes any `degree`, src/main.cc:40,
src/claw.cc:141-159 -- three RK stages from degree 2 on): HIP engine against the CPU oracle, the same bars as
tests/test_gpu_parity.py -- residual <= 1e-12, RK solution <= 1e-11 on smooth data, limited runs <= 1e-8 -- over the
pieces that are templates of the degree: the stage kernels on squares and on bilinear cells (time step formed by the last
stage kernel, positivity inside the kernel), the modal (Pk) kernels, the limiter / indicator passes, the halo records of
a two-part run."""
import numpy as np
import pytest

import dflo_amd
from dflo_amd import problems
import oracle_lib
import test_gpu_parity as T

pytestmark = pytest.mark.gpu
rel = T.rel
DEGREES = [4, 5]


@pytest.mark.parametrize("degree", DEGREES)
@pytest.mark.parametrize("flux", T.FLUXES)
def test_residual_periodic(degree, flux):
    mesh, prm, claw, ora = T.make_pair(12, 9, degree, flux, h=0.5)
    u0 = mesh.interpolate(problems.smooth_perturbation)
    claw.set_initial_condition(u0)
    ora.set_solution(u0)
    assert rel(claw.cell_average, ora.get_cell_average()) < 1e-14
    assert rel(claw.assemble_system(), ora.assemble()) < 1e-12


@pytest.mark.parametrize("degree,flux", [(4, "hllc"), (5, "kfvs"), (4, "lxf"), (5, "roe")])
def test_rk_solution_vortex(degree, flux):
    mesh, prm, claw, ora = T.make_pair(10, 10, degree, flux)
    assert claw.n_rk == 3
    u0 = mesh.interpolate(problems.isentropic_vortex)
    claw.set_initial_condition(u0)
    ora.set_solution(u0)
    t = 0.0
    for it in range(6):
        dt = claw.compute_time_step()
        dto = ora.compute_time_step(t)
        assert abs(dt - dto) <= 1e-13 * dto
        r0, r1 = claw.iterate_explicit(dt)
        q0, q1 = ora.step(dt)
        assert abs(r0 - q0) <= 1e-10 * q0 and abs(r1 - q1) <= 1e-10 * q1
        t += dt
    assert rel(claw.current_solution, ora.get_solution()) < 1e-11
    t2 = claw.advance(3)           # device-resident time step
    for it in range(3):
        dt = ora.compute_time_step(t)
        ora.step(dt)
        t += dt
    assert abs(t2 - t) <= 1e-12 * t and rel(claw.current_solution, ora.get_solution()) < 1e-11


@pytest.mark.parametrize("degree,flux,pos", [(4, "kfvs", True), (5, "hllc", True), (4, "lxf", False)])
def test_mapped_cells(degree, flux, pos):
    """bilinear cells: residual, compute_time_step_q formed by the last stage kernel, positivity inside the kernel"""
    mesh, claw, ora = T.mapped_pair(degree, flux, pos_lim=pos)
    assert rel(claw.cell_average, ora.get_cell_average()) < 1e-13
    assert rel(claw.assemble_system(), ora.assemble()) < 1e-12
    t_end = claw.advance(4)
    t = 0.0
    for it in range(4):
        dt = ora.compute_time_step(t)
        ora.step(dt)
        t += dt
    assert abs(t_end - t) <= 1e-12 * t
    assert rel(claw.current_solution, ora.get_solution()) < 1e-10


@pytest.mark.parametrize("degree", DEGREES)
@pytest.mark.parametrize("flux", ["lxf", "hllc", "kfvs"])
def test_pk_residual_and_steps(degree, flux):
    mesh, prm, claw, ora = T.pk_pair(10, 8, degree, flux, h=0.5)
    u0 = mesh.project(problems.smooth_perturbation)
    assert u0.size == mesh.n_cells * 4 * (degree + 1) * (degree + 2) // 2
    claw.set_initial_condition(u0)
    ora.set_solution(u0)
    assert rel(claw.cell_average, ora.get_cell_average()) < 1e-14
    assert rel(claw.assemble_system(), ora.assemble()) < 1e-12
    t = 0.0
    for it in range(4):
        dt = claw.compute_time_step()
        assert abs(dt - ora.compute_time_step(t)) <= 1e-13 * dt
        claw.iterate_explicit(dt)
        ora.step(dt)
        t += dt
    assert rel(claw.current_solution, ora.get_solution()) < 1e-11


@pytest.mark.parametrize("basis,degree", [("Qk", 4), ("Qk", 5), ("Pk", 4), ("Pk", 5)])
def test_sod_tvb_positivity(basis, degree):
    """limiter passes at degrees 4 / 5 (all DoFs of a cell in one thread's registers): TVB (characteristic) + positivity"""
    bnd = {0: "slip", 1: "outflow", 2: "inflow"}
    nx, ny = 40, 8
    mesh = dflo_amd.Mesh.cartesian(nx, ny, 0.0, 0.0, 1.0 / nx, [2, 1, 0, 0], degree)
    mesh.set_basis(basis)
    prm = dflo_amd.Parameters(flux="hllc", limiter="TVB", char_lim=True, pos_lim=True, M=0.0, beta=2.0, boundary=bnd, final_time=0.2)
    claw, ora = dflo_amd.ConservationLaw(mesh, prm), oracle_lib.Oracle(mesh, prm)
    u0 = mesh.interpolate(problems.sod) if basis == "Qk" else mesh.project(problems.sod)
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
    for it in range(12):
        dt = claw.compute_time_step()
        dto = ora.compute_time_step(t)
        assert abs(dt - dto) <= 1e-12 * dto
        claw.iterate_explicit(dt)
        ora.step(dt)
        t += dt
    assert rel(claw.cell_average, ora.get_cell_average()) < 1e-9
    assert rel(claw.current_solution, ora.get_solution()) < 1e-8


@pytest.mark.parametrize("basis,degree", [("Qk", 4), ("Pk", 5)])
def test_kxrcf_indicator(basis, degree):
    nx, ny = 24, 16
    mesh = dflo_amd.Mesh.cartesian(nx, ny, 0.0, 0.0, 1.0 / nx, [0, 1, -1, -1], degree)
    mesh.set_basis(basis)
    prm = dflo_amd.Parameters(flux="roe", limiter="TVB", shock_indicator="density", boundary={0: "outflow", 1: "outflow"})
    claw, ora = dflo_amd.ConservationLaw(mesh, prm), oracle_lib.Oracle(mesh, prm)
    u0 = mesh.interpolate(T._rough_wave) if basis == "Qk" else mesh.project(T._rough_wave)
    claw.set_initial_condition(u0)
    ora.set_solution(u0)
    s, so = claw.compute_shock_indicator(), ora.compute_shock_indicator()
    ok = ~np.isnan(so)
    assert (np.isnan(s) == np.isnan(so)).all()
    assert np.abs(s[ok] - so[ok]).max() <= 1e-11 * np.abs(so[ok]).max()


@pytest.mark.parametrize("degree", DEGREES)
def test_two_parts_bit_identical(degree):
    """face-trace halo records of (k + 1) * 4 doubles per cut face at k = 4, 5"""
    mesh = dflo_amd.Mesh.cartesian(24, 16, -5.0, -5.0, 10.0 / 24, [-1, -1, -1, -1], degree)
    prm = dflo_amd.Parameters(flux="hllc", cfl=0.9)
    one = dflo_amd.ConservationLaw(mesh, prm)
    u0 = mesh.interpolate(problems.isentropic_vortex)
    one.set_initial_condition(u0)
    multi = dflo_amd.MultiConservationLaw(mesh, prm, devices=[0, 0], partitioner="slab")
    multi.set_initial_condition(u0)
    assert one.advance(5) == multi.advance(5)
    assert np.array_equal(one.current_solution, multi.current_solution)


def test_degree_six_is_refused():
    mesh = None
    with pytest.raises(Exception):
        mesh = dflo_amd.Mesh.cartesian(4, 4, 0.0, 0.0, 0.25, [-1, -1, -1, -1], 6)
        dflo_amd.ConservationLaw(mesh, dflo_amd.Parameters(flux="lxf"))
