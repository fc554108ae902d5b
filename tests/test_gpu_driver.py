"""Mini-driver (python -m dflo_amd input.prm) end to end on the GPU: .prm + .msh in, solution-NNN.vtu out,
against the oracle driven with the same parsed input."""
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np
import pytest

import dflo_amd
from dflo_amd import gmsh, vtu
from dflo_amd.prm import InputDeck
from dflo_amd.run import Run, main
import oracle_lib
from test_frontend import SOD_PRM

pytestmark = pytest.mark.gpu


def _oracle_run(deck, mesh, n_steps):
    ora = oracle_lib.Oracle(mesh, deck.parameters)
    cell, face, bid, xy = ora.boundary_faces()
    bv = np.zeros(xy.shape[:2] + (4,))
    for b in np.unique(bid):
        bv[bid == b] = np.stack(deck.boundary_values[int(b)](xy[bid == b][..., 0], xy[bid == b][..., 1], 0.0), axis=-1)
    ora.set_boundary_values(0, bv)
    ora.set_boundary_values(1, bv)
    ora.set_solution(mesh.interpolate(deck.initial_conditions))
    ora.apply_limiter()
    t = 0.0
    for it in range(n_steps):
        dt = ora.compute_time_step(t)
        ora.step(dt)
        t += dt
    return ora, t


@pytest.mark.parametrize("basis", ["Qk", "Pk"])
def test_sod_from_prm_and_msh(tmp_path, basis):
    gmsh.sod_tube(str(tmp_path / "tube.msh"), nx=41, ny=5)
    prm = tmp_path / "input.prm"
    prm.write_text(SOD_PRM.replace("set basis = Qk", "set basis = " + basis))
    out = tmp_path / "out"
    rc = main([str(prm), "4", "--outdir", str(out), "--max-steps", "12", "--quiet"])
    assert rc == 0
    files = sorted(os.listdir(out))
    assert files == ["shock.vtu", "solution-000.vtu", "solution-001.vtu", "solution-002.vtu"]   # IC + every 5 iterations
    deck = InputDeck.read(str(prm))
    run = Run(deck, str(tmp_path / "out2"), quiet=True)
    run.run(max_steps=10)
    ora, t = _oracle_run(deck, run.mesh, 10)
    assert abs(run.claw.elapsed_time - t) < 1e-12 * t
    assert np.abs(run.claw.current_solution - ora.get_solution()).max() < 1e-8
    # solution-002.vtu is the state after 10 iterations
    root = ET.parse(str(out / "solution-002.vtu")).getroot()
    da = [d for d in root.iter("DataArray") if d.get("Name") == "Density"][0]
    rho = vtu.decode_data_array(da.text, np.float64)
    _, _, fields = vtu.patch_fields(run.mesh, ora.get_solution())
    assert np.abs(rho - dict(fields)["Density"]).max() < 1e-8


def test_fast_mode_and_final_time(tmp_path):
    """--fast advances in device-resident chunks; the last step is clipped to `final time` (src/claw.cc:473-474)."""
    gmsh.vortex_square(str(tmp_path / "grid.msh"), n=17, L=10.0)
    text = """
set mesh file = grid.msh
set degree = 2
set mapping = cartesian
subsection boundary_1
   set type = periodic
   set pair = 3
   set direction = y
end
subsection boundary_2
   set type = periodic
   set pair = 4
   set direction = x
end
subsection boundary_3
   set type = periodic
   set pair = 1
   set direction = y
end
subsection boundary_4
   set type = periodic
   set pair = 2
   set direction = x
end
subsection initial condition
   set function = isenvort
end
subsection time stepping
  set cfl = 0.9
  set final time = 0.5
end
subsection refinement
  set refinement = false
end
subsection flux
  set flux = hllc
end
subsection output
  set iter step = 1000
end
"""
    prm = tmp_path / "input.prm"
    prm.write_text(text)
    deck = InputDeck.read(str(prm))
    a = Run(deck, str(tmp_path / "a"), quiet=True)
    a.run()
    b = Run(deck, str(tmp_path / "b"), quiet=True)
    b.run(fast=True)
    assert abs(a.claw.elapsed_time - 0.5) < 1e-13 and abs(b.claw.elapsed_time - 0.5) < 1e-13
    assert np.abs(a.claw.current_solution - b.claw.current_solution).max() < 1e-12
    assert a.time_iter == b.time_iter       # no chunk runs past final_time: the iteration count is the reference loop's
    assert sorted(os.listdir(tmp_path / "a")) == ["shock.vtu", "solution-000.vtu", "solution-001.vtu"]   # IC and final time
    # periodic box: the mean of every conserved variable is kept to round-off
    u0 = a.mesh.interpolate(dflo_amd.problems.isentropic_vortex).reshape(a.mesh.n_cells, 4, -1)
    w = np.tile(np.outer(*(2 * [np.polynomial.legendre.leggauss(3)[1] / 2])).reshape(-1), (a.mesh.n_cells, 4, 1))
    m0 = (u0 * w).sum(axis=(0, 2))
    m1 = (a.claw.current_solution.reshape(a.mesh.n_cells, 4, -1) * w).sum(axis=(0, 2))
    assert np.abs(m1 - m0).max() < 1e-10 * np.abs(m0).max()


def test_time_dependent_boundary_values(tmp_path):
    """Boundary expressions in t are re-evaluated at t (stage 0) and t + dt (later stages), src/claw.cc:736-745."""
    gmsh.sod_tube(str(tmp_path / "tube.msh"), nx=21, ny=5)
    text = SOD_PRM.replace("set w_2 value = 1.0\n   set w_3 value = 2.5", "set w_2 value = 1.0 + 0.5*t\n   set w_3 value = 2.5*(1+t)")
    text = text.replace("set type = TVB", "set type = none").replace("set positivity limiter = true", "set positivity limiter = false")
    text = text.replace("set w_2 value = 1.0*(x<=0.5) + 0.125*(x>0.5)", "set w_2 value = 1.0 + 0.1*sin(2*pi*x)")
    text = text.replace("set w_3 value = 2.5*(x<=0.5) + 0.250*(x>0.5)", "set w_3 value = 2.5 + 0.2*cos(2*pi*x)")
    prm = tmp_path / "input.prm"
    prm.write_text(text)
    deck = InputDeck.read(str(prm))
    assert deck.boundary_values[2].time_dependent
    run = Run(deck, str(tmp_path / "o"), quiet=True)
    assert run.bc_time_dependent
    run.run(max_steps=6)
    ora = oracle_lib.Oracle(run.mesh, deck.parameters)
    cell, face, bid, xy = ora.boundary_faces()

    def bvals(t):
        bv = np.zeros(xy.shape[:2] + (4,))
        for b in np.unique(bid):
            bv[bid == b] = np.stack(deck.boundary_values[int(b)](xy[bid == b][..., 0], xy[bid == b][..., 1], t), axis=-1)
        return bv

    ora.set_solution(run.mesh.interpolate(deck.initial_conditions))
    t = 0.0
    for it in range(6):
        dt = ora.compute_time_step(t)
        ora.set_boundary_values(0, bvals(t))
        ora.set_boundary_values(1, bvals(t + dt))
        ora.step(dt)
        t += dt
    assert np.isfinite(ora.get_solution()).all()
    assert np.abs(run.claw.current_solution - ora.get_solution()).max() < 1e-10
    # and the time dependence was really in play
    ora.set_boundary_values(1, bvals(0.0))
    assert np.abs(bvals(t) - bvals(0.0)).max() > 1e-3


def test_driver_reports_errors_like_main(tmp_path, capsys):
    prm = tmp_path / "input.prm"
    prm.write_text("set mesh file = missing.msh\nsubsection time stepping\n set cfl = 0.5\nend\nsubsection refinement\n set refinement = false\nend\n")
    assert main([str(prm), "--quiet"]) == 1
    assert "Exception on processing" in capsys.readouterr().err


DMR_PRM = """
set mesh file = grid.msh
set degree = %(degree)d
set mapping = cartesian
set basis = %(basis)s
subsection boundary_0
   set type = outflow
end
subsection boundary_1
   set type = slip
end
subsection boundary_2
   set type = outflow
end
subsection boundary_3
   set type = inflow
   set w_0 value =  57.1576766498*(x<1.0/6.0+(1+20*t)/sqrt(3)) + 0.0
   set w_1 value =  -33.0*(x<1.0/6.0+(1+20*t)/sqrt(3)) + 0.0
   set w_2 value =  8.0*(x<1.0/6.0+(1+20*t)/sqrt(3)) + 1.4*(x>=1.0/6.0+(1+20*t)/sqrt(3))
   set w_3 value =  563.5*(x<1.0/6.0+(1+20*t)/sqrt(3)) + 2.5*(x>=1.0/6.0+(1+20*t)/sqrt(3))
end
subsection boundary_4
   set type = inflow
   set w_0 value =  57.1576766498
   set w_1 value =  -33.0
   set w_2 value =  8.0
   set w_3 value =  563.5
end
subsection initial condition
   set w_0 value =  57.1576766498*(x<1.0/6.0+y/sqrt(3)) + 0.0
   set w_1 value =  -33.0*(x<1.0/6.0+y/sqrt(3)) + 0.0
   set w_2 value =  8.0*(x<1.0/6.0+y/sqrt(3)) + 1.4*(x>=1.0/6.0+y/sqrt(3))
   set w_3 value =  563.5*(x<1.0/6.0+y/sqrt(3)) + 2.5*(x>=1.0/6.0+y/sqrt(3))
end
subsection time stepping
  set cfl = 0.9
  set final time = 0.2
end
subsection refinement
  set refinement = false
end
subsection flux
 set flux = hllc
end
subsection limiter
   set type = TVB
   set shock indicator = limiter
   set characteristic limiter = true
   set positivity limiter = %(pos)s
   set M = 100.0
   set beta = 1.0
end
"""


@pytest.mark.parametrize("degree,basis,pos", [(1, "Pk", "false"), (2, "Qk", "true")])
def test_double_mach_reflection_c4_style(tmp_path, degree, basis, pos):
    """C4 path: Mach-10 shock, slip / outflow walls, the inflow state on the top boundary moving with the shock
    (time-dependent boundary function), HLLC + TVB (+ positivity); state of the shipped input and of BASELINE's C4."""
    gmsh.double_mach(str(tmp_path / "grid.msh"), ny=13)
    prm = tmp_path / "input.prm"
    prm.write_text(DMR_PRM % {"degree": degree, "basis": basis, "pos": pos})
    deck = InputDeck.read(str(prm))
    run = Run(deck, str(tmp_path / "o"), quiet=True)
    assert run.bc_time_dependent and run.mesh.n_cells == 49 * 12
    n = 12
    run.run(max_steps=n)
    ora = oracle_lib.Oracle(run.mesh, deck.parameters)
    cell, face, bid, xy = ora.boundary_faces()

    def bvals(t):
        bv = np.zeros(xy.shape[:2] + (4,))
        for b in np.unique(bid):
            bv[bid == b] = np.stack(deck.boundary_values[int(b)](xy[bid == b][..., 0], xy[bid == b][..., 1], t), axis=-1)
        return bv

    ora.set_boundary_values(0, bvals(0.0))
    ora.set_boundary_values(1, bvals(0.0))
    ora.set_solution(run.mesh.interpolate(deck.initial_conditions))
    ora.apply_limiter()
    t = 0.0
    for it in range(n):
        dt = ora.compute_time_step(t)
        ora.set_boundary_values(0, bvals(t))
        ora.set_boundary_values(1, bvals(t + dt))
        ora.step(dt)
        t += dt
    assert abs(run.claw.elapsed_time - t) < 1e-10 * t
    u, uo = run.claw.current_solution, ora.get_solution()
    scale = np.abs(uo).max()
    assert np.abs(run.claw.cell_average - ora.get_cell_average()).max() < 1e-8 * scale
    assert np.abs(u - uo).max() < 1e-6 * scale


def test_device_boundary_programs_match_host_evaluation(tmp_path):
    """dflo_hip_set_boundary_program: the engine evaluates the boundary expressions itself at t and t + dt."""
    mesh = dflo_amd.Mesh.cartesian(12, 9, 0.0, 0.0, 1.0 / 12, [0, 1, 2, 3], 2)
    prm = dflo_amd.Parameters(flux="roe", cfl=0.5, boundary={0: "inflow", 1: "outflow", 2: "farfield", 3: "slip"})
    claw = dflo_amd.ConservationLaw(mesh, prm)
    from dflo_amd.expr import VectorFunction
    exprs = {0: ["0.3*(1+0.1*sin(3*t+y))", "0.05*cos(pi*y)*t", "1.0 + 0.2*(y<0.4+t)", "2.5 + 0.5*exp(-t)*y^2"],
             2: ["0.1", "if(x>0.5, 0.2, -0.2*t)", "1.0 + 0.1*max(x, 0.3)", "2.5 + sqrt(abs(x-0.5))"]}
    cell, face, bid, xy = claw.boundary_faces()
    claw.set_initial_condition(mesh.interpolate(lambda x, y: dflo_amd.problems.smooth_perturbation(x, y, L=1.0)))
    base = np.full(xy.shape[:2] + (4,), 7.0)
    claw.set_boundary_values(0, base)
    claw.set_boundary_values(1, base)
    for b, e in exprs.items():
        claw.set_boundary_function(b, e)
    t0 = 0.123
    claw.elapsed_time = t0
    dt = claw.compute_time_step()          # sets the device clock to t0
    claw.iterate_explicit(dt)
    for which, t in ((0, t0), (1, t0 + dt)):
        got = claw.get_boundary_values(which)
        want = base.copy()
        for b, e in exprs.items():
            sel = bid == b
            want[sel] = np.stack(VectorFunction(e)(xy[sel][..., 0], xy[sel][..., 1], t), axis=-1)
        assert np.abs(got - want).max() < 1e-13
    # same step with host-evaluated values
    ref = dflo_amd.ConservationLaw(mesh, prm)
    ref.set_initial_condition(mesh.interpolate(lambda x, y: dflo_amd.problems.smooth_perturbation(x, y, L=1.0)))
    ref.set_boundary_values(0, claw.get_boundary_values(0))
    ref.set_boundary_values(1, claw.get_boundary_values(1))
    ref.iterate_explicit(dt)
    assert np.abs(ref.current_solution - claw.current_solution).max() < 1e-13
    # removing a program freezes the last values; a malformed program is refused
    claw.set_boundary_function(0, [None] * 4)
    bad = np.array([[5, 0]], dtype=np.int32)   # ADD on an empty stack
    import ctypes as C
    from dflo_amd import _lib
    rc = _lib.lib.dflo_hip_set_boundary_program(claw._h, 1, 0, 1, _lib.iptr(bad), 0, None)
    assert rc != 0


def test_double_mach_fast_mode_runs_on_device_programs(tmp_path):
    gmsh.double_mach(str(tmp_path / "grid.msh"), ny=13)
    prm = tmp_path / "input.prm"
    prm.write_text(DMR_PRM % {"degree": 2, "basis": "Qk", "pos": "true"})
    deck = InputDeck.read(str(prm))
    a = Run(deck, str(tmp_path / "a"), quiet=True, host_bc=True)
    assert a.bc_time_dependent and not a.bc_on_device
    a.run(max_steps=16)
    b = Run(deck, str(tmp_path / "b"), quiet=True)
    assert b.bc_on_device
    b.run(max_steps=16, fast=True)          # one advance(16): no host round trip, programs evaluated every step
    assert abs(a.claw.elapsed_time - b.claw.elapsed_time) < 1e-13
    assert np.abs(a.claw.current_solution - b.claw.current_solution).max() < 1e-10 * np.abs(a.claw.current_solution).max()


def test_command_line_entry_point(tmp_path):
    """`python -m dflo_amd input.prm 4` -- dflo's own command line (src/main.cc:22-27: input file, thread count)."""
    import subprocess
    gmsh.sod_tube(str(tmp_path / "tube.msh"), nx=21, ny=5)
    (tmp_path / "input.prm").write_text(SOD_PRM)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "dflo_amd", "input.prm", "4", "--max-steps", "6"], cwd=str(tmp_path),
                       env=dict(os.environ, PYTHONPATH=root), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Number of degrees of freedom: 1280" in r.stdout and "It=6" in r.stdout and "Writing file solution-001.vtu" in r.stdout
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith(".vtu")) == ["shock.vtu", "solution-000.vtu", "solution-001.vtu"]


# ---------------------------------------------------------------- dflo_hip_run: the stand-alone C++ driver
RUN_BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dflo_amd", "dflo_hip_run")


def _vtu_field(path, name):
    root = ET.parse(str(path)).getroot()
    da = [d for d in root.iter("DataArray") if d.get("Name") == name][0]
    a = vtu.decode_data_array(da.text, np.float64)
    nc = int(da.get("NumberOfComponents", "1"))
    return a.reshape(-1, nc) if nc > 1 else a


@pytest.mark.parametrize("case", ["sod-Qk", "sod-Pk", "dmr", "vortex", "step"])
def test_cxx_driver_matches_python_driver(tmp_path, case):
    import subprocess
    if case == "step":   # unstructured quadrilaterals, mapping q1, KFVS, positivity alone (C5's path)
        gmsh.forward_step(str(tmp_path / "step.msh"), cl=0.1, seed=2)
        text, steps = STEP_PRM % {"degree": 2}, 8
    elif case.startswith("sod"):
        gmsh.sod_tube(str(tmp_path / "tube.msh"), nx=41, ny=5)
        text, steps = SOD_PRM.replace("set basis = Qk", "set basis = " + case[4:]), 12
    elif case == "dmr":
        gmsh.double_mach(str(tmp_path / "grid.msh"), ny=13)
        text, steps = DMR_PRM % {"degree": 2, "basis": "Qk", "pos": "true"}, 12
    else:
        gmsh.vortex_square(str(tmp_path / "grid.msh"), n=17, L=10.0)
        text = ("set mesh file = grid.msh\nset degree = 3\nset mapping = cartesian\n"
                + "".join("subsection boundary_%d\n set type = periodic\n set pair = %d\n set direction = %s\nend\n" % (b, p, d)
                          for b, p, d in [(1, 3, "y"), (2, 4, "x"), (3, 1, "y"), (4, 2, "x")])
                + "subsection initial condition\n set function = isenvort\nend\nsubsection time stepping\n set cfl = 0.5\n set final time = 0.3\nend\n"
                  "subsection refinement\n set refinement = false\nend\nsubsection flux\n set flux = kfvs\nend\nsubsection output\n set iter step = 7\nend\n")
        steps = 10
    (tmp_path / "input.prm").write_text(text)
    for fast in ([], ["--fast"]):
        out = tmp_path / ("cxx" + "".join(fast))
        r = subprocess.run([RUN_BIN, str(tmp_path / "input.prm"), "4", "--outdir", str(out), "--max-steps", str(steps)] + fast,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        ref = tmp_path / ("py" + "".join(fast))
        assert main([str(tmp_path / "input.prm"), "--outdir", str(ref), "--max-steps", str(steps), "--quiet"] + fast) == 0
        assert sorted(os.listdir(out)) == sorted(os.listdir(ref))
        last = sorted(f for f in os.listdir(out) if f.startswith("solution"))[-1]
        for name in ["Density", "Energy", "Pressure", "XMomentum__YMomentum", "XVelocity__YVelocity"] + (["schlieren_plot"] if "schlieren plot = true" in text else []):
            a, b = _vtu_field(out / last, name), _vtu_field(ref / last, name)
            assert a.shape == b.shape and np.abs(a - b).max() <= 1e-9 * max(np.abs(b).max(), 1.0), (case, name)
        assert np.array_equal(_vtu_field(out / "shock.vtu", "shock_indicator"), _vtu_field(ref / "shock.vtu", "shock_indicator"))
    assert "Number of degrees of freedom:" in r.stdout and "Writing file solution-000.vtu" in r.stdout


def test_cxx_driver_reports_errors_like_main(tmp_path):
    import subprocess
    (tmp_path / "input.prm").write_text("set mesh file = missing.msh\nsubsection time stepping\n set cfl = 0.5\nend\nsubsection refinement\n set refinement = false\nend\n")
    r = subprocess.run([RUN_BIN, str(tmp_path / "input.prm")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "Exception on processing" in r.stderr


STEP_PRM = """
set mesh type = gmsh
set mesh file = step.msh
set degree = %(degree)d
set mapping = q1
subsection boundary_1
   set type = inflow
   set w_0 value =        4.20000
   set w_1 value =        0.00000
   set w_2 value =        1.40000
   set w_3 value =        8.80000
end
subsection boundary_2
   set type = slip
end
subsection boundary_3
   set type = outflow
end
subsection initial condition
   set w_0 value =        4.20000
   set w_1 value =        0.00000
   set w_2 value =        1.40000
   set w_3 value =        8.80000
end
subsection time stepping
  set time step type = global
  set cfl = 0.5
  set final time = 4.0
end
subsection linear solver
  set method         = rk3
end
subsection output
  set iter step      = 100
  set schlieren plot = true
end
subsection refinement
  set refinement = false
end
subsection flux
 set flux = kfvs
end
subsection limiter
   set type = none
   set positivity limiter = true
end
"""


def test_forward_step_c5_style(tmp_path):
    """BASELINE config 5 on its own geometry: the Mach 3 wind tunnel with a step (examples/forward_step/input.prm with the
    changes C5 names: unstructured quadrilaterals, mapping q1, KFVS, positivity limiter alone -- the limiter runs inside
    the stage kernel here), through the .prm / .msh front end.  Q3 for the first steps against the oracle; then, with
    the degree of the shipped input (1), a longer run for the properties: the flow upstream of the step is still the free
    stream, everything stays admissible, and the mass in the tunnel grows by what the inflow brings and the (still
    undisturbed) outflow takes.  (For k >= 2 the impulsive start at the step face ends in NaNs after ~20 steps in the
    reference's algorithm itself, oracle and device alike: the limiter leaves p = 1e-13 at its worst point and the next
    flux evaluation there takes the root of a pressure that rounding has made negative -- the shipped input avoids it
    with k = 1 and TVB.)"""
    gmsh.forward_step(str(tmp_path / "step.msh"), cl=0.1, seed=2)
    prm = tmp_path / "input.prm"
    for degree in (3, 1):
        prm.write_text(STEP_PRM % {"degree": degree})
        deck = InputDeck.read(str(prm))
        run = Run(deck, str(tmp_path / ("o%d" % degree)), quiet=True)
        assert run.mesh.n_cells == 1512 and run.mesh.degree == degree
        n = 6
        run.run(max_steps=n)
        ora, t = _oracle_run(deck, run.mesh, n)
        assert abs(run.claw.elapsed_time - t) < 1e-10 * t
        scale = np.abs(ora.get_solution()).max()
        assert np.abs(run.claw.cell_average - ora.get_cell_average()).max() < 1e-9 * scale
        assert np.abs(run.claw.current_solution - ora.get_solution()).max() < 1e-7 * scale
    # longer, device-resident (k = 1)
    v = run.mesh.vertices
    x, y = v[:, :, 0], v[:, :, 1]
    area = 0.5 * np.abs((x[:, 0] * y[:, 1] - x[:, 1] * y[:, 0]) + (x[:, 1] * y[:, 3] - x[:, 3] * y[:, 1]) +
                        (x[:, 3] * y[:, 2] - x[:, 2] * y[:, 3]) + (x[:, 2] * y[:, 0] - x[:, 0] * y[:, 2]))
    assert abs(area.sum() - 2.52) < 1e-12
    m0, t0 = (run.claw.cell_average[:, 2] * area).sum(), run.claw.elapsed_time
    run.claw.advance(150)
    t1 = run.claw.elapsed_time
    assert 0.05 < t1 < 0.5                        # the corner's disturbance (speed u + c = 4) has not reached the outlet
    a = run.claw.cell_average
    assert np.isfinite(a).all() and a[:, 2].min() > 0.1
    p = 0.4 * (a[:, 3] - 0.5 * (a[:, 0] ** 2 + a[:, 1] ** 2) / a[:, 2])
    assert p.min() > 0.1 and p.max() > 5.0        # and there is a shock in front of the step (p2/p1 = 10.3 at Mach 3)
    up = x.max(axis=1) < 0.25                     # Mach 3: nothing travels upstream, the bow shock forms at the step
    assert np.abs(a[up] - np.array([4.2, 0.0, 1.4, 8.8])).max() < 1e-9
    dm = (a[:, 2] * area).sum() - m0
    assert abs(dm - 4.2 * (1.0 - 0.8) * (t1 - t0)) < 2e-3 * 4.2 * (t1 - t0)


@pytest.mark.parametrize("extra", [[], ["--config", "c3"], ["--config", "c5", "--nx", "8"], ["--scaling", "strong", "--nx", "128", "--parts-per-gpu", "2"]])
def test_bench_line_contract(extra):
    """bench.py prints ONE JSON line with the fields of the driver's contract (metric / value / unit / n_gpus / steps / warmup /
    ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus `roofline` and, at N = 1,
    `cpu_baseline` unless it is switched off; value = n_dofs * n_rk * steps / time."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-secondary"] + extra
    if not extra:
        cmd += ["--nx", "96"]       # (this one also takes roofline.traffic live: two rocprofv3 --pmc passes of the same command)
    else:
        cmd += ["--no-live-traffic"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["dtype"] == "f64"
    assert d["vs_baseline"] is None and d["scaling"] == ("strong" if "strong" in extra else "weak") and "workload" in d["config"] and "model" not in d["config"]
    if "strong" in extra:
        assert "strong scaling" in d["config"]["workload"] and d["config"]["n_dofs"] == 128 * 128 * 36
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    if not extra:
        import shutil
        if shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3"):
            live = str(r["traffic_source"]).startswith("live: rocprofv3 --pmc")
            if not live:   # a box whose counters cannot be read must not fail the suite: bench.py says why on stderr and names its fallback
                assert "live PMC traffic not collected" in out.stderr or r["traffic_source"] is None or "profiles/" in str(r["traffic_source"])
                import warnings
                warnings.warn("bench.py could not collect roofline.traffic live on this box: " + out.stderr[-300:])
            else:
                assert r["traffic"] > 0
                # the rate the counters saw, next to the algorithmic one
                assert abs(r["hbm_gbs"] - r["traffic"] / (r["kernel_ms"] * 1e-3) / 1e9) <= 1e-9 * r["hbm_gbs"] and abs(r["hbm_frac"] - r["hbm_gbs"] / 8000.0) < 1e-12
    assert "hbm_gbs" in r and "hbm_frac" in r
    assert d["value"] > 0 and abs(d["value"] - d["config"]["n_dofs"] * d["config"]["n_rk"] / (d["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * d["value"]


@pytest.mark.parametrize("transport,extra", [("rccl", []), ("ipc", ["--config", "c3"]), ("direct", ["--nx", "64", "--ny", "256"])])
def test_bench_self_halo_line(transport, extra):
    """bench.py --self-halo T: the one part is its own neighbour and runs the whole rank schedule (tests/test_gpu_selfhalo.py holds
    its results to the single engine bit for bit); the line says so, carries the exchange waits, and its own check holds"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--self-halo", transport] + (extra or ["--nx", "128"])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    c = d["config"]
    assert d["n_gpus"] == 1 and c["self_halo"] == transport and "self-halo" in c["transport"]
    assert ("ncclSend" in c["transport"]) == (transport == "rccl") and ("sequence words" in c["transport"]) == (transport == "ipc")
    assert c["exchange_samples"][0] > 0 and c["exchange_wait_us"][0] > 0.0
    assert "cpu_baseline" not in d and "secondary" not in d      # a measurement of the schedule, not the headline line
    if "--config" not in extra:
        assert float(c["check"].split("=")[-1]) < 1e-12
    if "--ny" in extra:
        assert "64x256 quads per GPU" in c["workload"]


def test_bench_line_survives_a_transport_that_hangs():
    """a transport that never returns (a collective stalled inside a library) must not cost the line of the transports that ran
    before it -- nor the attempts behind it: it runs in child processes of the ranks, which end it at its deadline"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29900 + os.getpid() % 90), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--nx", "128"]
    env = dict(os.environ, DFLO_BENCH_TRANSPORTS="ipc_gloo,gloo", DFLO_BENCH_ALL_TRANSPORTS="1", DFLO_BENCH_ONE_GPU="1", DFLO_BENCH_TEST_HANG="gloo",
               DFLO_BENCH_ATTEMPT_S="60", DFLO_BENCH_NO_STRONG="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["transport_used"] == "ipc_gloo" and d["value"] > 0
    tried = {t["transport"]: t for t in d["config"]["transports"]}
    assert not tried["gloo"]["ok"] and "did not return within" in tried["gloo"]["check"], tried


def test_bench_line_survives_a_hang_outside_the_child_processes():
    """with the isolation off (DFLO_BENCH_ISOLATE=0) the ranks run the attempts themselves: the run's own watchdog then ends a stalled one,
    rank 0 prints what has been measured and says which attempt hung"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29300 + os.getpid() % 90), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--nx", "128"]
    env = dict(os.environ, DFLO_BENCH_TRANSPORTS="ipc_gloo,gloo", DFLO_BENCH_ALL_TRANSPORTS="1", DFLO_BENCH_ONE_GPU="1", DFLO_BENCH_TEST_HANG="gloo",
               DFLO_BENCH_ATTEMPT_S="25", DFLO_BENCH_ISOLATE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["transport_used"] == "ipc_gloo" and d["value"] > 0
    assert "'transport gloo' hung" in d["config"]["watchdog"]
    assert "did not return within" in out.stderr


def test_bench_line_survives_an_ipc_attempt_that_kills_a_process():
    """the IPC transport's attempts run in child processes of the ranks: one that takes a process down (os.abort() here; a memory fault
    on a mapped window on a real node) is a failed attempt, and the line comes from the transport that ran before it"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + os.getpid() % 90), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--nx", "128"]
    env = dict(os.environ, DFLO_BENCH_TRANSPORTS="gloo,ipc_gloo", DFLO_BENCH_ALL_TRANSPORTS="1", DFLO_BENCH_ONE_GPU="1", DFLO_BENCH_TEST_CRASH="ipc_gloo",
               DFLO_BENCH_ATTEMPT_S="120", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["transport_used"] == "gloo" and d["value"] > 0
    tried = {t["transport"]: t for t in d["config"]["transports"]}
    assert not tried["ipc_gloo"]["ok"] and "isolated attempt" in tried["ipc_gloo"]["check"], tried


def test_bench_line_survives_an_ipc_attempt_that_stalls():
    """... and one that never returns is ended at its own deadline (not the run's), the IPC attempts behind it are not tried, and the
    line comes from the transport that ran before"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29400 + os.getpid() % 90), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--nx", "128"]
    env = dict(os.environ, DFLO_BENCH_TRANSPORTS="gloo,ipc_gloo,ipc_coarse", DFLO_BENCH_ALL_TRANSPORTS="1", DFLO_BENCH_ONE_GPU="1", DFLO_BENCH_TEST_HANG="ipc_gloo",
               DFLO_BENCH_ATTEMPT_S="55", DFLO_BENCH_NO_STRONG="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["transport_used"] == "gloo" and d["value"] > 0
    tried = {t["transport"]: t for t in d["config"]["transports"]}
    assert not tried["ipc_gloo"]["ok"] and "did not return within" in tried["ipc_gloo"]["check"], tried
    assert not tried["ipc_coarse"]["ok"] and "skipped" in tried["ipc_coarse"]["check"], tried


def test_bench_line_survives_a_transport_one_rank_cannot_set_up():
    """the first transport of the N > 1 order (IPC over the gloo rendezvous) cannot be set up on ONE rank -- a neighbour's window that
    does not map -- while its peers' set-up has succeeded: the ranks agree on that before anybody goes on, the peers close their drivers
    (which have sent nothing: no barrier to hang in), and the next transport gives the line"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29800 + os.getpid() % 90), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--nx", "128"]
    env = dict(os.environ, DFLO_BENCH_TRANSPORTS="ipc_gloo,gloo", DFLO_BENCH_ONE_GPU="1", DFLO_BENCH_TEST_FAIL_CREATE="1",
               DFLO_BENCH_ATTEMPT_S="60", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["transport_used"] == "gloo" and d["value"] > 0
    tried = {t["transport"]: t for t in d["config"]["transports"]}
    assert not tried["ipc_gloo"]["ok"] and "could not be made on some rank" in tried["ipc_gloo"]["check"], tried
    assert "hung" not in str(d["config"].get("watchdog"))


@pytest.mark.parametrize("extra", [[], ["--scaling", "strong"], ["--config", "c3"]])
def test_bench_line_of_two_ranks_launched_the_drivers_way(extra):
    """bench.py --gpus 2 as the round driver launches it (python -m torch.distributed.run, one rank per GPU, 127.0.0.1), on this
    one-GPU box: both ranks on GPU 0 with the host-staged gloo transport (developer switches; RCCL refuses two ranks on one
    device) -- everything else is the N > 1 path: rendezvous, partition, per-rank initial data, the rank schedule, barriers around the
    timed region, the maximum over ranks, ONE JSON line from rank 0 with the whole-job value and the per-rank transport record."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29600 + (os.getpid() + len(extra) * 7) % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--nx", "128"] + extra
    # two transports in one run, as on the 8-GPU node (there: rccl and ipc): here the host-staged gloo transport and the IPC transport
    # set up over gloo -- real hipIpc handles between the two processes
    env = dict(os.environ, DFLO_BENCH_TRANSPORTS="gloo,ipc_gloo", DFLO_BENCH_ALL_TRANSPORTS="1", DFLO_BENCH_ONE_GPU="1", DFLO_BENCH_WATCHDOG_S="300",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == ("strong" if "strong" in extra else "weak")
    c = d["config"]
    assert c["comm_ranks_seen"] == [2, 2] and sorted(c["comm_rank_seen"]) == [0, 1] and len(c["sec_per_rank"]) == 2
    assert "cpu_baseline" not in d and "secondary" not in d
    # every transport that ran is on the line with the verdict of its own check; `value` is the best one that passed
    tr = {t["transport"]: t for t in c["transports"]}
    assert sorted(tr) == ["gloo", "ipc_gloo"] and all(t["ok"] for t in tr.values()), c["transports"]
    # the IPC attempt counts because its totals are those of the host-staged run (the transports compute the same bits)
    assert "equal the gloo run's" in tr["ipc_gloo"]["validated"] and tr["gloo"]["validated"] is None
    assert c["transport_used"] in tr and abs(d["value"] - max(t["value"] for t in tr.values())) <= 0.06
    assert ("gloo" in c["parallelism"]) and (c["transport"].startswith("IPC: ") == (c["transport_used"] == "ipc_gloo"))
    assert c["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0" and "DFLO_BENCH_TRANSPORTS" in c["env"]     # the settings RCCL / the runtime were given
    if not extra:   # the weak-scaling line carries the strong-scaling reading of north_star as well: the one-GPU mesh cut N ways
        st = d["secondary_strong"]
        assert st["ok"] and st["scaling"] == "strong" and st["n_gpus"] == 2 and st["n_dofs"] == 128 * 128 * 36 and st["steps"] == 4
        assert st["transport"] == c["transport_used"] and len(st["exchange_wait_us"]) == 2 and st["ms_per_step"] > 0
        assert abs(st["value"] - st["n_dofs"] * c["n_rk"] / (st["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * st["value"]
        assert float(st["check"].split("=")[-1]) < 1e-12
    else:
        assert "secondary_strong" not in d
    if "--config" not in extra:
        n = 128 * 128 * 36 * (1 if "strong" in extra else 2)
        assert c["n_dofs"] == n
        drift = float(c["check"].split("=")[-1])
        assert drift < 1e-12, c["check"]      # the cut faces are evaluated by both ranks with the same bits
    assert abs(d["value"] - c["n_dofs"] * c["n_rk"] / (d["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * d["value"]


def test_an_ipc_attempt_whose_totals_do_not_hold_is_not_counted_and_is_run_again_with_the_strict_protocol():
    """bench.py --gpus 2 on one GPU (gloo, then the IPC transport over gloo).  The IPC attempt's totals are made to differ from the
    host-staged run's in the 7th digit (test hook) -- what a run that read stale halos would look like on a configuration whose own
    check is only "finite and admissible": it is not counted, the line is the library transport's, and one more attempt runs with
    DFLO_IPC_STRICT=1 (a system-scope release per delivering workgroup), which holds and may win."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29300 + os.getpid() % 200), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--config", "c3"]
    env = dict(os.environ, DFLO_BENCH_TRANSPORTS="gloo,ipc_gloo", DFLO_BENCH_ALL_TRANSPORTS="1", DFLO_BENCH_ONE_GPU="1", DFLO_BENCH_WATCHDOG_S="400",
               DFLO_BENCH_TEST_BAD_TOTALS="ipc_gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    tr = {t["transport"]: t for t in d["config"]["transports"]}
    assert sorted(tr) == ["gloo", "ipc_gloo", "ipc_strict"], sorted(tr)
    assert tr["gloo"]["ok"] and not tr["ipc_gloo"]["ok"] and "NOT COUNTED" in tr["ipc_gloo"]["check"] and "differ from the gloo run's" in tr["ipc_gloo"]["check"]
    assert tr["ipc_strict"]["ok"] and "equal the gloo run's" in tr["ipc_strict"]["validated"]
    assert d["config"]["transport_used"] in ("gloo", "ipc_strict")
    assert abs(d["value"] - max(tr["gloo"]["value"], tr["ipc_strict"]["value"])) <= 0.06
