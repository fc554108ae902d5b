"""CPU: anchors the oracle's assembled residual / RK path (parity at this level is UNPINNED by any
reference fixture -- the reference has none, and deal.II is absent -- so it rests on these analytic
properties of the scheme the reference implements)."""
import numpy as np
import pytest

import dflo_amd
from dflo_amd import problems
import oracle_lib as O

FLUXES = ["lxf", "sw", "kfvs", "roe", "hllc"]


def uniform_state(x, y):
    o = np.ones_like(x)
    rho, u, v, p = 1.3, 0.7, -0.4, 0.9
    return rho * u * o, rho * v * o, rho * o, (p / 0.4 + 0.5 * rho * (u * u + v * v)) * o


@pytest.mark.parametrize("degree", [1, 2, 3])
@pytest.mark.parametrize("flux", FLUXES)
def test_free_stream_periodic(degree, flux):
    mesh = dflo_amd.Mesh.cartesian(6, 5, 0.0, 0.0, 0.25, [-1] * 4, degree)
    ora = O.Oracle(mesh, dflo_amd.Parameters(flux=flux))
    ora.set_solution(mesh.interpolate(uniform_state))
    tol = 2e-7 if flux == "kfvs" else 1e-13   # KFVS is consistent only to the accuracy of its ERF polynomial
    assert np.abs(ora.assemble()).max() < tol


def test_free_stream_farfield_and_outflow():
    mesh = dflo_amd.Mesh.cartesian(5, 4, 0.0, 0.0, 0.2, [0, 1, 0, 1], 2)
    prm = dflo_amd.Parameters(flux="roe", boundary={0: "farfield", 1: "outflow"})
    ora = O.Oracle(mesh, prm)
    ora.set_solution(mesh.interpolate(uniform_state))
    cell, face, bid, xy = ora.boundary_faces()
    bv = np.stack(uniform_state(xy[..., 0], xy[..., 1]), axis=-1)
    ora.set_boundary_values(0, bv)
    assert np.abs(ora.assemble(0)).max() < 1e-13


@pytest.mark.parametrize("degree", [1, 2])
@pytest.mark.parametrize("flux", FLUXES)
def test_conservation_periodic(degree, flux):
    """sum_i rhs_i = 0 per component on a periodic mesh (sum of the basis functions is 1)."""
    mesh = dflo_amd.Mesh.cartesian(8, 6, -5.0, -5.0, 1.25, [-1] * 4, degree)
    ora = O.Oracle(mesh, dflo_amd.Parameters(flux=flux))
    ora.set_solution(mesh.interpolate(problems.smooth_perturbation))
    r = ora.assemble().reshape(mesh.n_cells, 4, -1)
    assert np.abs(r.sum(axis=(0, 2))).max() < 1e-12 * np.abs(r).sum()


def l2_error_after(nx, degree, t_end=0.25):
    mesh = dflo_amd.Mesh.cartesian(nx, nx, -5.0, -5.0, 10.0 / nx, [-1] * 4, degree)
    ora = O.Oracle(mesh, dflo_amd.Parameters(flux="lxf", cfl=0.5, final_time=t_end))
    u0 = mesh.interpolate(problems.isentropic_vortex_exact)
    ora.set_solution(u0)
    t = 0.0
    while t < t_end - 1e-14:
        dt = ora.compute_time_step(t)
        ora.step(dt)
        t += dt
    xy, jxw = ora.cell_quadrature()          # Qk: quadrature points = support points
    exact = np.stack(problems.isentropic_vortex_exact(xy[..., 0], xy[..., 1]), axis=1)   # steady solution
    u = ora.get_solution().reshape(mesh.n_cells, 4, -1)
    return np.sqrt((((u - exact) ** 2) * jxw[:, None, :]).sum())


@pytest.mark.parametrize("degree", [1, 2])
def test_vortex_convergence_order(degree):
    """The vortex of src_mpi/ic.cc:44-60 (p = rho^gamma/gamma) is an exact steady Euler solution: the
    DG error must drop at ~ order k+1.  (The serial tree's p = rho^gamma, src/ic.cc:56, is not steady.)"""
    e1, e2 = l2_error_after(16, degree), l2_error_after(32, degree)
    order = np.log2(e1 / e2)
    assert order > degree + 0.5, (e1, e2, order)


def test_inverse_mass_is_diagonal_collocation():
    mesh = dflo_amd.Mesh.cartesian(2, 2, 0.0, 0.0, 0.5, [-1] * 4, 2)
    ora = O.Oracle(mesh, dflo_amd.Parameters())
    x, w = O.gauss(3)
    expect = 1.0 / (np.outer(w, w).reshape(-1) * 0.25)     # 1/(w_a w_b h^2), j = a + 3 b
    assert np.allclose(ora.inv_mass().reshape(4, 4, 9), expect[None, None, :], rtol=1e-14)


def test_periodic_two_sided_equals_one_sided():
    """src_mpi integrates a periodic face from both sides (boundary callback); the engine treats it
    as an interior face.  The two differ by round-off only (conservation symmetry of the fluxes)."""
    mesh_p = dflo_amd.Mesh.cartesian(6, 6, -5.0, -5.0, 10.0 / 6, [-1] * 4, 2)
    u0 = mesh_p.interpolate(problems.smooth_perturbation)
    for flux in FLUXES:
        ora = O.Oracle(mesh_p, dflo_amd.Parameters(flux=flux))
        ora.set_solution(u0)
        r2 = ora.assemble().copy()
        # clear the periodic flag (+8) -> interior-face treatment
        nf = mesh_p.neighbor_faces
        saved = nf.copy()
        nf &= 7
        ora1 = O.Oracle(mesh_p, dflo_amd.Parameters(flux=flux))
        ora1.set_solution(u0)
        r1 = ora1.assemble()
        nf[:] = saved
        assert np.abs(r1 - r2).max() < 1e-12 * np.abs(r1).max(), flux


def test_cartesian_and_q1_mapping_agree_on_squares():
    mesh = dflo_amd.Mesh.cartesian(5, 4, 0.0, 0.0, 0.3, [0, 1, 2, 3], 2)
    prm = dflo_amd.Parameters(flux="hllc", boundary={0: "slip", 1: "outflow", 2: "inflow", 3: "farfield"})
    u0 = mesh.interpolate(lambda x, y: problems.smooth_perturbation(x, y, L=1.5))
    res = []
    for mapping in ("cartesian", "q1"):
        mesh.set_mapping(mapping)
        ora = O.Oracle(mesh, prm)
        ora.set_solution(u0)
        cell, face, bid, xy = ora.boundary_faces()
        bv = np.stack(problems.smooth_perturbation(xy[..., 0], xy[..., 1], L=1.5), axis=-1)
        ora.set_boundary_values(0, bv)
        res.append(ora.assemble(0))
    mesh.set_mapping("cartesian")
    assert np.abs(res[0] - res[1]).max() < 1e-12 * np.abs(res[0]).max()


def skewed_mesh(n=6, degree=2, amp=0.15):
    """n x n quads on [0,1]^2 with interior vertices displaced: genuinely bilinear (non-affine) cells."""
    xs = np.linspace(0, 1, n + 1)
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    h = 1.0 / n
    X = X + amp * h * np.sin(2 * np.pi * X) * np.sin(2 * np.pi * Y)
    Y = Y + amp * h * np.sin(3 * np.pi * X) * np.sin(1 * np.pi * Y)
    verts = np.stack([X.reshape(-1), Y.reshape(-1)], axis=1)
    vid = lambda i, j: i + (n + 1) * j
    quads = [[vid(i, j), vid(i + 1, j), vid(i + 1, j + 1), vid(i, j + 1)] for j in range(n) for i in range(n)]
    bed, bid = [], []
    for i in range(n):
        bed += [[vid(i, 0), vid(i + 1, 0)], [vid(i, n), vid(i + 1, n)], [vid(0, i), vid(0, i + 1)], [vid(n, i), vid(n, i + 1)]]
        bid += [0, 0, 0, 0]
    return dflo_amd.Mesh.from_quads(verts, quads, bed, bid, degree)


@pytest.mark.parametrize("flux", ["lxf", "roe", "hllc"])
def test_free_stream_on_mapped_quads(flux):
    mesh = skewed_mesh()
    prm = dflo_amd.Parameters(flux=flux, boundary={0: "farfield"})
    ora = O.Oracle(mesh, prm)
    ora.set_solution(mesh.interpolate(uniform_state))
    cell, face, bid, xy = ora.boundary_faces()
    ora.set_boundary_values(0, np.stack(uniform_state(xy[..., 0], xy[..., 1]), axis=-1))
    assert np.abs(ora.assemble(0)).max() < 1e-12


def test_tvb_keeps_linear_data_and_means():
    """TVB limiter (src/limiter.cc:225-370): a linear field passes unchanged, cell means are preserved."""
    mesh = dflo_amd.Mesh.cartesian(8, 8, 0.0, 0.0, 0.125, [0, 0, 0, 0], 2)
    prm = dflo_amd.Parameters(limiter="TVB", char_lim=True, M=0.0, beta=2.0, boundary={0: "outflow"})
    lin = lambda x, y: (0.3 + 0.1 * x, 0.2 - 0.05 * y, 1.0 + 0.2 * x + 0.1 * y, 2.5 + 0.1 * x)
    ora = O.Oracle(mesh, prm)
    u0 = mesh.interpolate(lin)
    ora.set_solution(u0)
    ora.apply_limiter()
    assert np.abs(ora.get_solution() - u0).max() < 1e-13
    # a field with a jump gets limited but keeps its means
    ora.set_solution(mesh.interpolate(problems.sod))
    a0 = ora.get_cell_average().copy()
    ora.apply_limiter()
    ora.compute_cell_average()
    assert np.abs(ora.get_cell_average() - a0).max() < 1e-13


def test_positivity_limiter_guards():
    """src/positivity.cc: negative mean state is fatal (:26-38); otherwise rho, p >= eps at the points."""
    mesh = dflo_amd.Mesh.cartesian(4, 4, 0.0, 0.0, 0.25, [-1] * 4, 2)
    prm = dflo_amd.Parameters(pos_lim=True)
    ora = O.Oracle(mesh, prm)
    u = mesh.interpolate(uniform_state).reshape(mesh.n_cells, 4, 9).copy()
    u[3, 2, 4] = -0.5        # a negative density node, mean stays positive
    ora.set_solution(u.reshape(-1))
    ora.apply_positivity_limiter()
    v = ora.get_solution().reshape(mesh.n_cells, 4, 9)
    assert v[3, 2].min() > 0 and np.abs(v[0] - u[0]).max() == 0
    u[5, 2, :] = -1.0        # negative mean density
    ora.set_solution(u.reshape(-1))
    with pytest.raises(O.OracleError) as e:
        ora.apply_positivity_limiter()
    assert e.value.code == -3


def test_nonsquare_cell_rejected():
    """'Cell is not square', src/claw.cc:219"""
    verts = np.array([[0, 0], [2, 0], [2, 1], [0, 1]], dtype=float)
    mesh = dflo_amd.Mesh.from_quads(verts, [[0, 1, 2, 3]], degree=1)
    mesh.set_mapping("cartesian")
    with pytest.raises(O.OracleError) as e:
        O.Oracle(mesh, dflo_amd.Parameters())
    assert e.value.code == -2


@pytest.mark.parametrize("degree", [1, 2, 3])
def test_pk_projection_matches_oracle_shape_tables(degree):
    """Mesh.project (host mirror of set_initial_condition_Pk, src/ic.cc:128-164) uses the same modal basis, mode
    order and mass matrix as the oracle's FE_DGP restatement; projecting a P_k polynomial reproduces it."""
    mesh = dflo_amd.Mesh.cartesian(5, 4, -1.0, 2.0, 0.25, [-1, -1, -1, -1], degree)
    mesh.set_basis("Pk")
    ora = O.Oracle(mesh, dflo_amd.Parameters(flux="lxf"))
    T, ww = mesh.modal_matrix()
    assert np.abs(T.T - ora.cell_shape()).max() < 1e-14
    xy, jxw = ora.cell_quadrature()
    assert np.abs(xy - mesh.support_points()).max() < 1e-14
    assert np.abs(jxw - ww * 0.25 ** 2).max() < 1e-16
    inv_mass = np.empty(mesh.n_cells * mesh.ndof)
    O._lib.dflo_oracle_get_inv_mass(ora._h, O._d(inv_mass))
    assert np.abs(inv_mass * 0.25 ** 2 - 1.0).max() < 1e-13   # orthonormal modes: M = |K| I
    k = degree

    def poly(x, y):
        f = 1.0 + 0.3 * x ** k - 0.2 * y ** k + (0.1 * x * y ** (k - 1) if k > 1 else 0.0)
        return [0.1 * f, -0.2 * f, 1.0 + 0.05 * f, 2.5 + 0.1 * f]

    u = mesh.project(poly).reshape(mesh.n_cells, 4, -1)
    back = np.einsum("ncm,qm->ncq", u, T)
    exact = np.stack(poly(xy[..., 0], xy[..., 1]), axis=1)
    assert np.abs(back - exact).max() < 1e-12
    # cell average = mode 0
    ora.set_solution(u.reshape(-1))
    assert np.abs(ora.get_cell_average() - u[:, :, 0]).max() < 1e-13


# ---------------------------------------------------------------- KXRCF indicator (src/indicator.cc)
def _moving_jump(x, y, x_jump=0.5, rho_l=1.0, rho_r=0.4, u=0.7):
    rho = np.where(x < x_jump, rho_l, rho_r)
    p = 1.0
    return [rho * u, 0.0 * x, rho, p / 0.4 + 0.5 * rho * u * u]


@pytest.mark.parametrize("degree", [1, 2, 3])
@pytest.mark.parametrize("basis", ["Qk", "Pk"])
def test_kxrcf_indicator_known_answer(degree, basis):
    """Piecewise-constant density jump moving right: only the cell right of the jump sees it through its inflow
    (left) face: ind = |rho_r - rho_l| h / (diam^((k+1)/2) h rho_r)  (src/indicator.cc:119-124, 177-182)."""
    nx, ny = 16, 4
    h = 1.0 / nx
    mesh = dflo_amd.Mesh.cartesian(nx, ny, 0.0, 0.0, h, [0, 0, 0, 0], degree)
    mesh.set_basis(basis)
    prm = dflo_amd.Parameters(flux="lxf", limiter="TVB", shock_indicator="density", boundary={0: "outflow"})
    ora = O.Oracle(mesh, prm)
    ora.set_solution(mesh.interpolate(_moving_jump))
    ind = ora.compute_shock_indicator().reshape(ny, nx)
    diam = h * np.sqrt(2.0)
    expect = abs(0.4 - 1.0) / (diam ** (0.5 * (degree + 1)) * 0.4)
    assert np.allclose(ind[:, 8], expect, rtol=1e-12)
    others = np.delete(ind, 8, axis=1)
    assert np.abs(others[:, 1:]).max() < 1e-12      # no jump on the inflow face
    assert np.isnan(others[:, 0]).all()             # first column: its inflow face is a boundary face -> 0/0
    # energy as the indicator variable
    prm_e = dflo_amd.Parameters(flux="lxf", limiter="TVB", shock_indicator="energy", boundary={0: "outflow"})
    ora_e = O.Oracle(mesh, prm_e)
    ora_e.set_solution(mesh.interpolate(_moving_jump))
    ind_e = ora_e.compute_shock_indicator().reshape(ny, nx)
    e_l, e_r = 2.5 + 0.5 * 1.0 * 0.49, 2.5 + 0.5 * 0.4 * 0.49
    assert np.allclose(ind_e[:, 8], abs(e_r - e_l) / (diam ** (0.5 * (degree + 1)) * e_r), rtol=1e-12)


def test_kxrcf_indicator_gates_the_tvb_limiter():
    """With "shock indicator = limiter" every cell is offered to the limiter; with "density" only the cells whose
    indicator exceeds 1 (src/limiter.cc:263): the cells changed are exactly those changed before AND flagged."""
    nx, ny = 32, 4
    h = 1.0 / nx
    mesh = dflo_amd.Mesh.cartesian(nx, ny, 0.0, 0.0, h, [0, 0, -1, -1], 1)

    def ic(x, y):   # a rough wave (clipped by beta = 1 nearly everywhere) with two jumps
        rho = 1.0 + 0.2 * np.sin(16 * np.pi * x) + np.where(x > 0.75, 0.8, 0.0) + np.where(x > 0.25, 0.6, 0.0)
        return [0.5 * rho, 0.0 * x, rho, 2.5 + 0.125 * rho]

    out = {}
    for kind in ("limiter", "density"):
        prm = dflo_amd.Parameters(flux="lxf", limiter="TVB", M=0.0, beta=1.0, shock_indicator=kind, boundary={0: "outflow"})
        ora = O.Oracle(mesh, prm)
        u0 = mesh.interpolate(ic)
        ora.set_solution(u0)
        ind = ora.compute_shock_indicator()
        ora.apply_limiter()
        out[kind] = (ind, np.abs(ora.get_solution() - u0).reshape(mesh.n_cells, -1).max(axis=1) > 1e-12)
    assert (out["limiter"][0] == 1e20).all()
    changed_all, (ind, changed_kx) = out["limiter"][1], out["density"]
    flagged = ind > 1.0
    assert changed_all.sum() > mesh.n_cells // 2
    assert 0 < flagged.sum() < mesh.n_cells // 4
    assert (changed_kx == (changed_all & flagged)).all()
    assert changed_kx.sum() > 0


@pytest.mark.parametrize("degree,flux,bnd", [(1, "lxf", False), (2, "hllc", False), (3, "kfvs", False), (2, "roe", True),
                                             (1, "sw", True), (2, "lxf", True)])
def test_optimised_cpu_twin_equals_the_restatement(degree, flux, bnd):
    """bench.py's second CPU figure (the fused, fully threaded twin at the end of oracle/dflo_oracle.cc) advances the
    same states as the reference-style restatement: 5 steps with the CFL step, periodic and with every boundary kind."""
    if bnd:
        mesh = dflo_amd.Mesh.cartesian(14, 10, -5.0, -5.0, 10.0 / 14, [0, 1, 2, 3], degree)
        prm = dflo_amd.Parameters(flux=flux, cfl=0.9, boundary={0: "farfield", 1: "outflow", 2: "slip", 3: "pressure"})
    else:
        mesh = dflo_amd.Mesh.cartesian(14, 10, -5.0, -5.0, 10.0 / 14, [-1] * 4, degree)
        prm = dflo_amd.Parameters(flux=flux, cfl=0.9)
    a, b = O.Oracle(mesh, prm, threads=2), O.Oracle(mesh, prm, threads=3)
    assert b.twin_supported
    u0 = mesh.interpolate(problems.isentropic_vortex)
    for o in (a, b):
        o.set_solution(u0)
        if bnd:
            cell, face, bid, xy = o.boundary_faces()
            bv = np.stack(problems.isentropic_vortex(xy[..., 0], xy[..., 1]), axis=-1)
            o.set_boundary_values(0, bv)
            o.set_boundary_values(1, bv)
    t = 0.0
    for _ in range(5):
        dt = a.compute_time_step(t)
        r = a.step(dt)
        t += dt
    t2, r2 = b.twin_advance(5)
    assert abs(t - t2) <= 1e-14 * t and abs(r[1] - r2) <= 1e-12 * r[1]
    assert np.abs(a.get_solution() - b.get_solution()).max() <= 1e-12
    assert np.abs(a.get_cell_average() - b.get_cell_average()).max() <= 1e-12
    # and it says no where it does not apply (limiters, bilinear cells)
    c = O.Oracle(mesh, dflo_amd.Parameters(flux=flux, pos_lim=True), threads=1)
    assert not c.twin_supported
    with pytest.raises(O.OracleError):
        c.twin_advance(1)


# ---------------------------------------------------------------- the assembled path against the 60-digit derivation
def _residual_fixture():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "residual_fixture.json")))["cases"]


def run_fixture_case(case, make_solver, basis="Qk"):
    """shared by the oracle (here) and the device (tests/test_gpu_golden.py): right-hand side, cell averages, CFL time step
    and the state after one SSP-RK step against tests/golden/residual_fixture.json (make_residual_fixture.py)"""
    f = lambda v: np.array([float(x) for x in v])
    mesh = dflo_amd.Mesh.cartesian(case["nx"], case["ny"], 0.0, 0.0, float(case["h"]), case["side"], case["degree"])
    if basis != "Qk":
        mesh.set_basis(basis)
    lim = case.get("limiter")
    extra = dict(limiter="TVB", M=float(lim["M"]), beta=float(lim["beta"]), char_lim=lim["char_lim"], pos_lim=True,
                 shock_indicator=lim.get("indicator", "limiter")) if lim else {}
    prm = dflo_amd.Parameters(flux=case["flux"], cfl=float(case["cfl"]), boundary={int(k): v for k, v in case["kinds"].items()},
                              gravity=float(case.get("gravity", 0.0)), time_step_type="local" if case.get("local") else "global", **extra)
    s = make_solver(mesh, prm)
    cell, face, bid, xy = s.boundary_faces()
    bf = case["boundary_faces"]
    assert [(b["cell"], b["face"], b["id"]) for b in bf] == list(zip(cell.tolist(), face.tolist(), bid.tolist()))
    if bf:
        bv = np.array([[[float(x) for x in pt] for pt in b["values"]] for b in bf])
        s.set_boundary_values(0, bv)
        s.set_boundary_values(1, bv)
    return mesh, s, f(case["U0"]), f(case["residual"]), f(case["cell_average"]).reshape(-1, 4), float(case["dt"]), f(case["U1"])


@pytest.mark.parametrize("case", _residual_fixture(), ids=[c["name"] for c in _residual_fixture()])
def test_oracle_assembly_matches_the_independent_derivation(case):
    mesh, ora, U0, R, A, dt, U1 = run_fixture_case(case, lambda m, p: O.Oracle(m, p))
    ora.set_solution(U0)
    assert np.abs(ora.get_cell_average() - A).max() <= 1e-14 * np.abs(A).max()
    assert np.abs(ora.assemble() - R).max() <= 1e-12 * np.abs(R).max()
    assert abs(ora.compute_time_step(0.0) - dt) <= 1e-13 * dt
    ora.step(dt)
    assert np.abs(ora.get_solution() - U1).max() <= 1e-12 * np.abs(U1).max()


def _limiter_fixture():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "residual_fixture.json")))["limiter_cases"]


def _pk_limiter_fixture():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "residual_fixture.json")))["pk_limiter_cases"]


def run_limiter_case(case, make_solver, basis="Qk"):
    """TVB (src/limiter.cc:225-370) and positivity (src/positivity.cc:17-208) limiters against the 60-digit derivation of
    tests/golden/make_residual_fixture.py; shared by the oracle and the device.  Returns (limited state, expected)."""
    f = lambda v: np.array([float(x) for x in v])
    mesh = dflo_amd.Mesh.cartesian(case["nx"], case["ny"], 0.0, 0.0, float(case["h"]), case["side"], case["degree"])
    if basis != "Qk":
        mesh.set_basis(basis)
    if case["kind"] == "tvb":
        prm = dflo_amd.Parameters(flux="hllc", limiter="TVB", char_lim=case["char_lim"], M=float(case["M"]), beta=float(case["beta"]),
                                  conserve_angular_momentum=bool(case.get("conserve_angular_momentum", False)), boundary={0: "outflow"})
    else:
        prm = dflo_amd.Parameters(flux="hllc", pos_lim=True, boundary={0: "outflow"})
    s = make_solver(mesh, prm)
    set_state = s.set_solution if hasattr(s, "set_solution") else s.set_initial_condition
    set_state(f(case["U0"]))
    if case["kind"] == "tvb":
        s.apply_limiter()
    else:
        s.apply_positivity_limiter()
    got = s.get_solution() if hasattr(s, "get_solution") else s.current_solution
    return got, f(case["U1"]), f(case["U0"])


@pytest.mark.parametrize("case", _limiter_fixture(), ids=[c["name"] for c in _limiter_fixture()])
def test_oracle_limiters_match_the_independent_derivation(case):
    got, want, before = run_limiter_case(case, lambda m, p: O.Oracle(m, p))
    assert np.abs(want - before).max() > 1e-3          # the case limits something
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()


def _bilinear_fixture():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "residual_fixture.json")))["bilinear_cases"]


def run_bilinear_case(case, make_solver):
    """general quadrilaterals (MappingQ1, the C5 path): metric terms, normals, lumped mass, compute_time_step_q against the
    60-digit derivation; shared by the oracle and the device"""
    f = lambda v: np.array([float(x) for x in v])
    nx, ny = case["nx"], case["ny"]
    verts = np.array([[float(a), float(b)] for a, b in case["vertices"]])
    vid = lambda i, j: i + (nx + 1) * j
    quads = np.array([[vid(i, j), vid(i + 1, j), vid(i + 1, j + 1), vid(i, j + 1)] for j in range(ny) for i in range(nx)], dtype=np.int32)
    bed, bid = [], []
    for i in range(nx):
        bed += [[vid(i, 0), vid(i + 1, 0)], [vid(i, ny), vid(i + 1, ny)]]
        bid += [case["side"][2], case["side"][3]]
    for j in range(ny):
        bed += [[vid(0, j), vid(0, j + 1)], [vid(nx, j), vid(nx, j + 1)]]
        bid += [case["side"][0], case["side"][1]]
    mesh = dflo_amd.Mesh.from_quads(verts, quads, np.array(bed, dtype=np.int32), np.array(bid, dtype=np.int32), case["degree"])
    nodes = np.array([[float(a), float(b)] for a, b in case["nodes"]]).reshape(mesh.n_cells, -1, 2)
    assert np.abs(mesh.support_points() - nodes).max() < 1e-14        # the same cells, the same node order
    prm = dflo_amd.Parameters(flux=case["flux"], cfl=float(case["cfl"]), boundary={int(k): v for k, v in case["kinds"].items()})
    s = make_solver(mesh, prm)
    cell, face, b, xy = s.boundary_faces()
    bf = case["boundary_faces"]
    assert [(x["cell"], x["face"], x["id"]) for x in bf] == list(zip(cell.tolist(), face.tolist(), b.tolist()))
    bv = np.array([[[float(x) for x in pt] for pt in x["values"]] for x in bf])
    s.set_boundary_values(0, bv)
    s.set_boundary_values(1, bv)
    return mesh, s, f(case["U0"]), f(case["residual"]), f(case["cell_average"]).reshape(-1, 4), float(case["dt"]), f(case["U1"])


@pytest.mark.parametrize("case", _bilinear_fixture(), ids=[c["name"] for c in _bilinear_fixture()])
def test_oracle_on_bilinear_cells_matches_the_independent_derivation(case):
    mesh, ora, U0, R, A, dt, U1 = run_bilinear_case(case, lambda m, p: O.Oracle(m, p))
    ora.set_solution(U0)
    assert np.abs(ora.get_cell_average() - A).max() <= 1e-14 * np.abs(A).max()
    assert np.abs(ora.assemble() - R).max() <= 1e-12 * np.abs(R).max()
    assert abs(ora.compute_time_step(0.0) - dt) <= 1e-13 * dt
    ora.step(dt)
    assert np.abs(ora.get_solution() - U1).max() <= 1e-12 * np.abs(U1).max()


def _pk_fixture():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "residual_fixture.json")))["pk_cases"]


@pytest.mark.parametrize("case", _pk_fixture(), ids=[c["name"] for c in _pk_fixture()])
def test_oracle_pk_assembly_matches_the_independent_derivation(case):
    """the modal (FE_DGP) path: L2-projected state, residual against every mode, M = |K| I, one SSP-RK step"""
    mesh, ora, U0, R, A, dt, U1 = run_fixture_case(case, lambda m, p: O.Oracle(m, p), basis="Pk")
    ora.set_solution(U0)
    assert np.abs(ora.get_cell_average() - A).max() <= 1e-14 * np.abs(A).max()
    assert np.abs(ora.assemble() - R).max() <= 1e-12 * np.abs(R).max()
    assert abs(ora.compute_time_step(0.0) - dt) <= 1e-13 * dt
    ora.step(dt)
    assert np.abs(ora.get_solution() - U1).max() <= 1e-12 * np.abs(U1).max()


def _kxrcf_fixture():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "residual_fixture.json")))["kxrcf_cases"]


def run_kxrcf_case(case, make_solver):
    """KXRCF troubled-cell indicator (src/indicator.cc:51-198) against the 60-digit derivation; shared with the device"""
    f = lambda v: np.array([float(x) for x in v])
    mesh = dflo_amd.Mesh.cartesian(case["nx"], case["ny"], 0.0, 0.0, float(case["h"]), [0, 0, 0, 0], case["degree"])
    prm = dflo_amd.Parameters(flux="hllc", limiter="TVB", shock_indicator=case["variable"], boundary={0: "outflow"})
    s = make_solver(mesh, prm)
    set_state = s.set_solution if hasattr(s, "set_solution") else s.set_initial_condition
    set_state(f(case["U0"]))
    return s.compute_shock_indicator(), f(case["indicator"])


@pytest.mark.parametrize("case", _kxrcf_fixture(), ids=[c["name"] for c in _kxrcf_fixture()])
def test_oracle_kxrcf_matches_the_independent_derivation(case):
    got, want = run_kxrcf_case(case, lambda m, p: O.Oracle(m, p))
    ok = np.isfinite(want)
    assert (np.isfinite(got) == ok).all() and ok.sum() >= 20 and (want[ok] > 1).sum() >= 4
    assert np.abs(got[ok] - want[ok]).max() <= 1e-12 * np.abs(want[ok]).max()


@pytest.mark.parametrize("case", _pk_limiter_fixture(), ids=[c["name"] for c in _pk_limiter_fixture()])
def test_oracle_pk_limiters_match_the_independent_derivation(case):
    """apply_limiter_TVB_Pk (src/limiter.cc:377-516, with and without the angular-momentum correction) and the Pk branch of the
    positivity limiter against the 60-digit derivation"""
    got, want, before = run_limiter_case(case, lambda m, p: O.Oracle(m, p), basis="Pk")
    assert np.abs(want - before).max() > 1e-3
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()


def _forcing_fixture():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "residual_fixture.json")))["forcing_cases"]


@pytest.mark.parametrize("case", _forcing_fixture(), ids=[c["name"] for c in _forcing_fixture()])
def test_oracle_forcing_and_local_time_steps_match_the_independent_derivation(case):
    """the gravity source (src/equation.h:831-850) in the right-hand side and "time step type = local" (every cell advances
    with its own dt(c), src/claw.cc:486-511) against the 60-digit derivation"""
    mesh, ora, U0, R, A, dt, U1 = run_fixture_case(case, lambda m, p: O.Oracle(m, p))
    ora.set_solution(U0)
    assert np.abs(ora.assemble() - R).max() <= 1e-12 * np.abs(R).max()
    assert abs(ora.compute_time_step(0.0) - dt) <= 1e-13 * dt
    ora.step(-1.0 if case["local"] else dt)
    assert np.abs(ora.get_solution() - U1).max() <= 1e-12 * np.abs(U1).max()


def _extra_fixture():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "residual_fixture.json")))["extra_cases"]


@pytest.mark.parametrize("case", _extra_fixture(), ids=[c["name"] for c in _extra_fixture()])
def test_oracle_degree_0_and_limited_steps_match_the_independent_derivation(case):
    """degree 0 (one stage, src/claw.cc:141-145) and whole SSP-RK steps with the TVB and positivity limiters applied after every
    stage (iterate_explicit, src/claw.cc:726-772) against the 60-digit derivation"""
    mesh, ora, U0, R, A, dt, U1 = run_fixture_case(case, lambda m, p: O.Oracle(m, p))
    ora.set_solution(U0)
    assert np.abs(ora.assemble() - R).max() <= 1e-12 * np.abs(R).max()
    assert abs(ora.compute_time_step(0.0) - dt) <= 1e-13 * dt
    ora.step(dt)
    assert np.abs(ora.get_solution() - U1).max() <= 1e-11 * np.abs(U1).max()


def _pk_step_fixture():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "residual_fixture.json")))["pk_step_cases"]


@pytest.mark.parametrize("case", _pk_step_fixture(), ids=[c["name"] for c in _pk_step_fixture()])
def test_oracle_pk_limited_steps_match_the_independent_derivation(case):
    """a whole SSP-RK step on the modal basis with TVB-Pk and the positivity limiter after every stage"""
    mesh, ora, U0, R, A, dt, U1 = run_fixture_case(case, lambda m, p: O.Oracle(m, p), basis="Pk")
    ora.set_solution(U0)
    assert np.abs(ora.assemble() - R).max() <= 1e-12 * np.abs(R).max()
    assert abs(ora.compute_time_step(0.0) - dt) <= 1e-13 * dt
    ora.step(dt)
    assert np.abs(ora.get_solution() - U1).max() <= 1e-11 * np.abs(U1).max()


def _moving_bc_fixture():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "residual_fixture.json")))["moving_bc_cases"]


def run_moving_bc_case(case, make_solver):
    """boundary states that move in time: table 0 holds them at t (first stage), table 1 at t + dt (later stages), src/claw.cc:733-745"""
    mesh, s, U0, R, A, dt, U1 = run_fixture_case(case, make_solver)
    later = np.array([[[float(x) for x in pt] for pt in b["values_later"]] for b in case["boundary_faces"]])
    s.set_boundary_values(1, later)
    return mesh, s, U0, R, np.array([float(x) for x in case["residual_later"]]), dt, U1


@pytest.mark.parametrize("case", _moving_bc_fixture(), ids=[c["name"] for c in _moving_bc_fixture()])
def test_oracle_moving_boundary_states_match_the_independent_derivation(case):
    mesh, ora, U0, R0, R1, dt, U1 = run_moving_bc_case(case, lambda m, p: O.Oracle(m, p))
    ora.set_solution(U0)
    assert np.abs(ora.assemble() - R0).max() <= 1e-12 * np.abs(R0).max()
    assert np.abs(R1 - R0).max() > 1e-3 * np.abs(R0).max()          # the later table matters
    ora.step(dt)
    assert np.abs(ora.get_solution() - U1).max() <= 1e-12 * np.abs(U1).max()


def run_rotated_bilinear_case(case, make_solver, seed=5):
    """The bilinear fixture once more with the LOCAL numbering of most cells rotated (vertex order shifted by 1, 2 or 3): the same
    cells and the same physics, but the cells now meet with every relative orientation -- faces that run opposite on their two
    sides (flips), different local faces and node orders -- the data of an unstructured mesh.  State, right-hand side and the
    stepped state of the 60-digit derivation are carried over node by node through the support points."""
    f = lambda v: np.array([float(x) for x in v])
    nx, ny = case["nx"], case["ny"]
    verts = np.array([[float(a), float(b)] for a, b in case["vertices"]])
    vid = lambda i, j: i + (nx + 1) * j
    quads = np.array([[vid(i, j), vid(i + 1, j), vid(i + 1, j + 1), vid(i, j + 1)] for j in range(ny) for i in range(nx)], dtype=np.int32)
    rot = np.random.default_rng(seed).integers(0, 4, len(quads))
    rot[0] = 0
    quads_r = np.array([np.roll(q, -r) for q, r in zip(quads, rot)], dtype=np.int32)
    bed, bid = [], []
    for i in range(nx):
        bed += [[vid(i, 0), vid(i + 1, 0)], [vid(i, ny), vid(i + 1, ny)]]
        bid += [case["side"][2], case["side"][3]]
    for j in range(ny):
        bed += [[vid(0, j), vid(0, j + 1)], [vid(nx, j), vid(nx, j + 1)]]
        bid += [case["side"][0], case["side"][1]]
    bed, bid = np.array(bed, dtype=np.int32), np.array(bid, dtype=np.int32)
    plain = dflo_amd.Mesh.from_quads(verts, quads, bed, bid, case["degree"])
    mesh = dflo_amd.Mesh.from_quads(verts, quads, bed, bid, case["degree"])
    # from_quads starts every cell's vertex loop at its lower-left vertex; the rotated numbering is written into the mesh tables
    # directly (what a deal.II mesh with arbitrarily oriented cells hands over): vertices in lexicographic order v0 v1 v2 v3 of
    # the rotated loop, face neighbours and the neighbour's face (+4: the face runs the other way there) by matching edges
    lex = quads_r[:, [0, 1, 3, 2]]
    face_verts = [(0, 2), (1, 3), (0, 1), (2, 3)]
    edges = {}
    for c in range(len(lex)):
        for fc, (a, b) in enumerate(face_verts):
            edges.setdefault(frozenset((int(lex[c, a]), int(lex[c, b]))), []).append((c, fc, int(lex[c, a])))
    old_nbr = plain.neighbors.copy()
    old_faces = [[frozenset((int(quads[c, [0, 1, 3, 2]][a]), int(quads[c, [0, 1, 3, 2]][b]))) for a, b in face_verts] for c in range(len(quads))]
    for c in range(len(lex)):
        mesh.vertices[c] = verts[lex[c]]
        for fc, (a, b) in enumerate(face_verts):
            e = frozenset((int(lex[c, a]), int(lex[c, b])))
            hs = edges[e]
            if len(hs) == 2:
                other = hs[1] if hs[0][0] == c and hs[0][1] == fc else hs[0]
                me = hs[0] if other is hs[1] else hs[1]
                mesh.neighbors[c, fc] = other[0]
                mesh.neighbor_faces[c, fc] = other[1] | (4 if me[2] != other[2] else 0)
            else:
                mesh.neighbors[c, fc] = old_nbr[c, old_faces[c].index(e)]     # the boundary code of that edge
                mesh.neighbor_faces[c, fc] = 0
    assert (mesh.neighbor_faces & 4).any()        # some faces run opposite on their two sides now
    # node permutation per cell: the rotated mesh's support points among the plain ones
    P0, P1 = plain.support_points(), mesh.support_points()
    ns = P0.shape[1]
    perm = np.empty((mesh.n_cells, ns), dtype=int)
    for c in range(mesh.n_cells):
        d = np.abs(P1[c][:, None, :] - P0[c][None, :, :]).max(axis=2)
        perm[c] = d.argmin(axis=1)
        assert d.min(axis=1).max() < 1e-13 and len(set(perm[c])) == ns
    carry = lambda v: np.take_along_axis(f(v).reshape(mesh.n_cells, 4, ns), perm[:, None, :].repeat(4, axis=1), axis=2).reshape(-1)
    prm = dflo_amd.Parameters(flux=case["flux"], cfl=float(case["cfl"]), boundary={int(k): v for k, v in case["kinds"].items()})
    s0, s = make_solver(plain, prm), make_solver(mesh, prm)
    # boundary values by the coordinates of their face points
    c0, f0, b0, xy0 = s0.boundary_faces()
    bv0 = np.array([[[float(x) for x in pt] for pt in x["values"]] for x in case["boundary_faces"]])
    key = lambda p: (round(p[0], 11), round(p[1], 11))
    table = {key(xy0[i, q]): bv0[i, q] for i in range(len(c0)) for q in range(xy0.shape[1])}
    c1, f1, b1, xy1 = s.boundary_faces()
    bv = np.array([[table[key(xy1[i, q])] for q in range(xy1.shape[1])] for i in range(len(c1))])
    s.set_boundary_values(0, bv)
    s.set_boundary_values(1, bv)
    return mesh, s, carry(case["U0"]), carry(case["residual"]), f(case["cell_average"]).reshape(-1, 4), float(case["dt"]), carry(case["U1"])


@pytest.mark.parametrize("case", _bilinear_fixture(), ids=[c["name"] for c in _bilinear_fixture()])
def test_oracle_on_rotated_cells_matches_the_independent_derivation(case):
    mesh, ora, U0, R, A, dt, U1 = run_rotated_bilinear_case(case, lambda m, p: O.Oracle(m, p))
    ora.set_solution(U0)
    assert np.abs(ora.get_cell_average() - A).max() <= 1e-14 * np.abs(A).max()
    assert np.abs(ora.assemble() - R).max() <= 1e-12 * np.abs(R).max()
    assert abs(ora.compute_time_step(0.0) - dt) <= 1e-13 * dt
    ora.step(dt)
    assert np.abs(ora.get_solution() - U1).max() <= 1e-12 * np.abs(U1).max()
