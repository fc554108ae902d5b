"""GPU: the native multi-device driver (dflo_hip_multi_*, dflo_amd/csrc/multi.hip) with 2 and 3 engines on ONE
device -- the same stage schedule, streams, events, pack / peer copy / unpack and time-step reduction an 8-GPU node
runs, with hipMemcpyPeerAsync copying to the same device -- against a single engine and against the oracle.

What replaces what: update_ghost_values (src_mpi/claw.cc:793, src_mpi/limiter.cc:232), Utilities::MPI::min
(src_mpi/claw.cc:579), right_hand_side.l2_norm (src_mpi/claw.cc:777).  Bars: smooth runs bit-identical to the single
engine (every face flux is evaluated from the same two traces by whoever owns either side, and the additions of a cell
happen in a fixed order whatever shard the cell lives in); limited runs (minmod switches) <= 1e-8.
"""
import os
import sys

import numpy as np
import pytest

import dflo_amd
from dflo_amd import problems
import oracle_lib

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _front(x, y):
    s = 0.5 * (1.0 + np.tanh((x + 0.5 * y - 0.8) / 0.004))
    rho, p = 1.0 + 0.6 * s, 1.0 + 0.9 * s
    u, v = 0.6, 0.35
    return [rho * u, rho * v, rho, p / 0.4 + 0.5 * rho * (u * u + v * v)]


def _smooth(x, y):
    return problems.smooth_perturbation(x, y, L=1.0)


def _case(name):
    """(mesh, parameters, initial/boundary function) of the named configuration"""
    if name == "c2":      # BASELINE config 2 in small: periodic vortex, Q2, HLLC
        mesh = dflo_amd.Mesh.cartesian(32, 24, -5.0, -5.0, 10.0 / 32, [-1, -1, -1, -1], 2)
        return mesh, dflo_amd.Parameters(flux="hllc", cfl=0.9), problems.isentropic_vortex
    if name == "c1":      # config 1: Q1, LxF (cell averages of the ghosts feed the flux)
        mesh = dflo_amd.Mesh.cartesian(24, 24, -5.0, -5.0, 10.0 / 24, [-1, -1, -1, -1], 1)
        return mesh, dflo_amd.Parameters(flux="lxf", cfl=0.9), problems.isentropic_vortex
    if name == "c3":      # config 3: Sod, Q1, Roe, TVB + positivity
        mesh = dflo_amd.Mesh.cartesian(64, 8, 0.0, 0.0, 1.0 / 64, [2, 1, 0, 0], 1)
        prm = dflo_amd.Parameters(flux="roe", limiter="TVB", char_lim=True, pos_lim=True, M=0.0, beta=2.0, cfl=0.8,
                                  boundary={0: "slip", 1: "outflow", 2: "inflow"})
        return mesh, prm, problems.sod
    if name == "c4":      # config 4 style: Q2, HLLC, TVB + positivity (limiter marks from the stage kernel)
        mesh = dflo_amd.Mesh.cartesian(64, 8, 0.0, 0.0, 1.0 / 64, [2, 1, 0, 0], 2)
        prm = dflo_amd.Parameters(flux="hllc", limiter="TVB", char_lim=True, pos_lim=True, M=0.0, beta=2.0, cfl=0.8,
                                  boundary={0: "slip", 1: "outflow", 2: "inflow"})
        return mesh, prm, problems.sod
    if name == "c5":      # config 5 style: unstructured quads, q1 mapping, Q3, KFVS, positivity inside the stage kernel
        from dflo_amd import gmsh
        verts, quads, bed, bid = gmsh.unstructured_quads(10, seed=2)
        mesh = dflo_amd.Mesh.from_quads(verts, quads, bed, bid, 3)
        prm = dflo_amd.Parameters(flux="kfvs", pos_lim=True, cfl=0.4, boundary={0: "slip", 1: "outflow", 2: "slip", 3: "inflow"})
        return mesh, prm, _smooth
    if name == "kxrcf":   # KXRCF-gated TVB across the cuts
        mesh = dflo_amd.Mesh.cartesian(48, 40, 0.0, 0.0, 1.0 / 48, [0, 0, 0, 0], 1)
        prm = dflo_amd.Parameters(flux="hllc", limiter="TVB", pos_lim=True, beta=2.0, cfl=0.8, boundary={0: "outflow"},
                                  shock_indicator="density")
        return mesh, prm, _front
    if name == "pk":      # Pk basis, positivity as a separate pass (sep_limiter path)
        mesh = dflo_amd.Mesh.cartesian(24, 16, 0.0, 0.0, 1.0 / 24, [2, 1, 0, 0], 2)
        mesh.set_basis("Pk")
        prm = dflo_amd.Parameters(flux="hllc", pos_lim=True, cfl=0.5, boundary={0: "slip", 1: "outflow", 2: "inflow"})
        return mesh, prm, problems.sod
    if name == "pkq1":    # the modal basis on bilinear cells (round 4), LxF: the ghost cells' averages are no longer their mode 0
        from dflo_amd import gmsh
        verts, quads, bed, bid = gmsh.unstructured_quads(9, seed=4)
        mesh = dflo_amd.Mesh.from_quads(verts, quads, bed, bid, 2)
        mesh.set_basis("Pk")
        prm = dflo_amd.Parameters(flux="lxf", cfl=0.4, boundary={0: "slip", 1: "outflow", 2: "slip", 3: "inflow"})
        return mesh, prm, _smooth
    raise KeyError(name)


def _setup(claw, mesh, ic):
    cell, face, bid, xy = claw.boundary_faces()
    if len(cell):
        bv = np.stack(ic(xy[..., 0], xy[..., 1]), axis=-1)
        claw.set_boundary_values(0, bv)
        claw.set_boundary_values(1, bv)
    claw.set_initial_condition(mesh.interpolate(ic))


def _run(claw, limited, steps=4, resident=3):
    """IC limiting as run() does, `steps` host-driven steps, then `resident` device-resident ones"""
    if limited:
        claw.apply_limiter()
    out = {"dt": [], "norms": []}
    for _ in range(steps):
        dt = claw.compute_time_step()
        out["dt"].append(dt)
        out["norms"].append(claw.iterate_explicit(dt))
    out["t"] = claw.advance(resident)
    out["u"] = claw.current_solution
    out["avg"] = claw.cell_average
    return out


CASES = [("c2", 2, "slab"), ("c2", 3, "slab"), ("c2", 4, "rcb"), ("c1", 3, "slab"), ("c3", 2, "slab"), ("c4", 2, "slab"), ("c4", 3, "slab"),
         ("c5", 2, "rcb"), ("c5", 3, "rcb"), ("kxrcf", 2, "slab"), ("kxrcf", 3, "rcb"), ("pk", 2, "slab"),
         ("pkq1", 2, "rcb"), ("pkq1", 3, "rcb")]


@pytest.mark.parametrize("name,n_parts,method", CASES)
def test_engines_on_one_device_match_the_single_engine(name, n_parts, method):
    mesh, prm, ic = _case(name)
    limited = prm.limiter == "TVB"
    one = dflo_amd.ConservationLaw(mesh, prm)
    _setup(one, mesh, ic)
    ref = _run(one, limited)
    multi = dflo_amd.MultiConservationLaw(mesh, prm, devices=[0] * n_parts, partitioner=method)
    assert multi.n_parts == n_parts and multi.n_local == n_parts
    owned = np.sort(multi.owned_cells())
    assert (owned == np.arange(mesh.n_cells)).all()
    _setup(multi, mesh, ic)
    got = _run(multi, limited)
    if mesh.basis == "Pk" and not limited and not prm.pos_lim:
        # (the modal kernel forms the trace of an own cell from its nodal image and the trace of a halo cell from its modes: the
        #  same number in two roundings, so where the shard rims lie shows in the last digits -- the parts are not bit-identical
        #  to the single engine, as they are for Qk)
        assert np.allclose(got["dt"], ref["dt"], rtol=1e-12, atol=0) and abs(got["t"] - ref["t"]) <= 1e-12 * ref["t"]
        assert rel(got["u"], ref["u"]) < 1e-11 and rel(got["avg"], ref["avg"]) < 1e-12
        return
    assert got["dt"] == ref["dt"]                 # the minimum of the parts' minima is the global minimum, exactly
    assert got["t"] == ref["t"]                   # ... also when it never leaves the devices
    for (a0, a1), (b0, b1) in zip(got["norms"], ref["norms"]):     # sums over shards: another order, same value to rounding
        assert abs(a0 - b0) <= 1e-12 * b0 and abs(a1 - b1) <= 1e-12 * b1
    if not limited and not prm.pos_lim:
        assert np.array_equal(got["u"], ref["u"]) and np.array_equal(got["avg"], ref["avg"])
    else:
        assert rel(got["u"], ref["u"]) < 1e-8 and rel(got["avg"], ref["avg"]) < 1e-9


@pytest.mark.parametrize("switch", ["DFLO_MULTI_STRICT", "DFLO_MULTI_COPY", "DFLO_PEER_FINEGRAINED", "DFLO_MULTI_THREADS=0", "DFLO_MULTI_GROUP=part"])
def test_one_exchange_per_tvb_stage_under_the_drivers_switches(switch, monkeypatch):
    """the one-exchange form with the sender waiting for the receiver's "consumed" event, through the staging buffer, with the
    receive areas fine-grained, driven by one host thread, one stream pair per part: the bits of the default arrangement"""
    mesh, prm, ic = _case("c4")
    monkeypatch.setenv("DFLO_TVB_ONE_EXCHANGE", "1")
    out = []
    for on in (False, True):
        if on:
            k, _, v = switch.partition("=")
            monkeypatch.setenv(k, v or "1")
        multi = dflo_amd.MultiConservationLaw(mesh, prm, devices=[0] * 3, partitioner="slab")
        assert "one exchange per stage" in multi.comm_info()[2]
        _setup(multi, mesh, ic)
        out.append(_run(multi, True))
        multi.close()
    a, b = out
    assert a["dt"] == b["dt"] and a["t"] == b["t"]
    assert np.array_equal(a["u"], b["u"]) and np.array_equal(a["avg"], b["avg"])


@pytest.mark.parametrize("name,n_parts,method", [("c3", 2, "slab"), ("c4", 3, "slab"), ("c4", 4, "rcb")])
def test_one_exchange_per_tvb_stage_gives_the_bits_of_two(name, n_parts, method, monkeypatch):
    """several parts, TVB between update and update_ghost_values: the cut cells sent unlimited with their neighbours' averages and
    limited by the receiver (one exchange per stage) against the reference's two (src_mpi/limiter.cc:232, src_mpi/claw.cc:793) --
    every bit.  Where a cut cell borders on two other parts (the corners of an RCB partition) the driver keeps the two exchanges."""
    mesh, prm, ic = _case(name)
    if method == "rcb":     # (a lattice on which four RCB blocks meet in a corner)
        mesh = dflo_amd.Mesh.cartesian(32, 32, 0.0, 0.0, 1.0 / 32, [2, 1, 0, 0], 2)
    out = []
    for one in ("1", "0"):
        monkeypatch.setenv("DFLO_TVB_ONE_EXCHANGE", one)
        multi = dflo_amd.MultiConservationLaw(mesh, prm, devices=[0] * n_parts, partitioner=method)
        what = multi.comm_info()[2]
        assert ("one exchange per stage" in what) == (one == "1" and method == "slab"), what
        _setup(multi, mesh, ic)
        out.append(_run(multi, True))
        multi.close()
    a, b = out
    assert a["dt"] == b["dt"] and a["t"] == b["t"]
    assert np.array_equal(a["u"], b["u"]) and np.array_equal(a["avg"], b["avg"])


@pytest.mark.parametrize("name,n_parts", [("c2", 3), ("c4", 2), ("c5", 3)])
def test_engines_on_one_device_match_the_oracle(name, n_parts):
    mesh, prm, ic = _case(name)
    limited = prm.limiter == "TVB"
    multi = dflo_amd.MultiConservationLaw(mesh, prm, devices=[0] * n_parts, partitioner="rcb" if name == "c5" else "slab")
    _setup(multi, mesh, ic)
    ora = oracle_lib.Oracle(mesh, prm)
    cell, face, bid, xy = ora.boundary_faces()
    if len(cell):
        bv = np.stack(ic(xy[..., 0], xy[..., 1]), axis=-1)
        ora.set_boundary_values(0, bv)
        ora.set_boundary_values(1, bv)
    ora.set_solution(mesh.interpolate(ic))
    assert rel(multi.assemble_system(), ora.assemble()) < 1e-12
    if limited:
        multi.apply_limiter()
        ora.apply_limiter()
    t = 0.0
    for it in range(4):
        dt = multi.compute_time_step()
        dto = ora.compute_time_step(t)
        assert abs(dt - dto) <= 1e-12 * dto
        r0, r1 = multi.iterate_explicit(dt)
        q0, q1 = ora.step(dt)
        assert abs(r0 - q0) <= 1e-10 * q0 and abs(r1 - q1) <= 1e-10 * q1
        t += dt
    tol = 1e-8 if (limited or prm.pos_lim) else 1e-11
    assert rel(multi.current_solution, ora.get_solution()) < tol


def test_one_part_is_the_plain_loop():
    """n_devices = 1 (bench.py --gpus 1): no peers, the driver issues exactly the launches of dflo_hip_advance."""
    mesh, prm, ic = _case("c2")
    one = dflo_amd.ConservationLaw(mesh, prm)
    _setup(one, mesh, ic)
    multi = dflo_amd.MultiConservationLaw(mesh, prm, devices=[0])
    _setup(multi, mesh, ic)
    assert one.advance(7) == multi.advance(7)
    assert np.array_equal(one.current_solution, multi.current_solution)
    # the same through the one-process-per-GPU entry with a world of one (no communicator needed)
    rank = dflo_amd.MultiConservationLaw.for_rank(mesh, prm, 0, 0, 1, None)
    _setup(rank, mesh, ic)
    assert rank.advance(7) == one.elapsed_time
    assert np.array_equal(one.current_solution, rank.current_solution)


def test_moving_boundary_programs_on_every_part():
    """C4's top wall: the boundary state is a function of (x, t) evaluated by every engine from its own device clock"""
    nx, ny = 40, 16
    mesh = dflo_amd.Mesh.cartesian(nx, ny, 0.0, 0.0, 1.0 / ny, [4, 2, 1, 3], 2)
    prm = dflo_amd.Parameters(flux="hllc", limiter="TVB", char_lim=True, pos_lim=True, M=100.0, beta=1.0, cfl=0.5,
                              boundary={1: "slip", 2: "outflow", 3: "inflow", 4: "inflow"})
    sh = "(x<1.0/6.0+(1+20*t)/sqrt(3))"
    top = ["57.1576766498*" + sh, "-33.0*" + sh, "8.0*%s + 1.4*(1-%s)" % (sh, sh), "563.5*%s + 2.5*(1-%s)" % (sh, sh)]
    res = []
    for claw in (dflo_amd.ConservationLaw(mesh, prm), dflo_amd.MultiConservationLaw(mesh, prm, devices=[0, 0, 0])):
        _setup(claw, mesh, problems.double_mach)
        claw.set_boundary_function(3, top)
        claw.apply_limiter()
        t = claw.advance(12)
        res.append((t, claw.current_solution))
    assert res[0][0] == res[1][0]
    assert rel(res[1][1], res[0][1]) < 1e-8


@pytest.mark.parametrize("n_parts", [1, 2])
def test_fixed_time_step_advances_the_clock(n_parts):
    """"time step type = global" with cfl = 0 (the reference's default cfl) and `time step` given (src/claw.cc:455-460):
    advance(n) is n steps of that dt -- on the device-resident loop too."""
    mesh = dflo_amd.Mesh.cartesian(16, 16, -5.0, -5.0, 10.0 / 16, [-1, -1, -1, -1], 2)
    prm = dflo_amd.Parameters(flux="hllc", cfl=0.0, time_step=2.0e-3)
    u0 = mesh.interpolate(problems.isentropic_vortex)
    a = dflo_amd.ConservationLaw(mesh, prm)
    a.set_initial_condition(u0)
    for _ in range(6):
        assert a.compute_time_step() == 2.0e-3
        a.iterate_explicit(2.0e-3)
    b = dflo_amd.ConservationLaw(mesh, prm) if n_parts == 1 else dflo_amd.MultiConservationLaw(mesh, prm, devices=[0] * n_parts)
    b.set_initial_condition(u0)
    t = b.advance(6)
    assert abs(t - 6 * 2.0e-3) < 1e-15
    assert np.array_equal(a.current_solution, b.current_solution)


def test_failure_is_reported_with_its_step():
    """A run that goes negative: step-by-step and resident loops report the same error, the resident loop knows the step"""
    mesh = dflo_amd.Mesh.cartesian(32, 4, 0.0, 0.0, 1.0 / 32, [0, 0, 0, 0], 2)
    prm = dflo_amd.Parameters(flux="lxf", pos_lim=True, cfl=3.0, boundary={0: "outflow"})   # cfl far beyond stability

    def blast(x, y):
        z = np.zeros_like(x)
        return z, z, np.ones_like(x), np.where(np.abs(x - 0.5) < 0.1, 1.0e4, 1.0e-2)
    a = dflo_amd.ConservationLaw(mesh, prm)
    _setup(a, mesh, blast)
    step = None
    for it in range(400):
        try:
            a.iterate_explicit(a.compute_time_step())
        except dflo_amd.DfloError as e:
            step, code = it, e.code
            break
    assert step is not None
    for claw in (dflo_amd.ConservationLaw(mesh, prm), dflo_amd.MultiConservationLaw(mesh, prm, devices=[0, 0])):
        _setup(claw, mesh, blast)
        with pytest.raises(dflo_amd.DfloError) as ei:
            claw.advance(400)
        assert ei.value.code == code
    import ctypes as C
    from dflo_amd._lib import lib
    b = dflo_amd.ConservationLaw(mesh, prm)
    _setup(b, mesh, blast)
    with pytest.raises(dflo_amd.DfloError):
        b.advance(400)
    st = C.c_int64()
    lib.dflo_hip_failure_step(b._h, C.byref(st))
    assert st.value == step


def test_rccl_loopback_transport(monkeypatch):
    """The RCCL transport on one GPU: the halo copies go through grouped ncclSend / ncclRecv on a one-rank communicator
    (self send/recv), which exercises the library loading, the communicator, the group calls and their stream order --
    everything of the one-process-per-GPU path except a second rank."""
    monkeypatch.setenv("DFLO_MULTI_TRANSPORT", "rccl_loopback")
    mesh, prm, ic = _case("c2")
    one = dflo_amd.ConservationLaw(mesh, prm)
    _setup(one, mesh, ic)
    multi = dflo_amd.MultiConservationLaw(mesh, prm, devices=[0, 0, 0])
    _setup(multi, mesh, ic)
    assert one.advance(5) == multi.advance(5)
    assert np.array_equal(one.current_solution, multi.current_solution)


def test_face_trace_records_are_what_travels(monkeypatch):
    """Qk without the KXRCF indicator: the halo record of a cut face is its trace, (k+1)*4 doubles, received
    straight into the engine's trace table; whole cells (DFLO_HALO_CELLS=1, and always for Pk / KXRCF) give the same bits."""
    import ctypes as C
    from dflo_amd._lib import lib
    mesh, prm, ic = _case("c2")
    res = []
    for cells in ("0", "1"):
        monkeypatch.setenv("DFLO_HALO_CELLS", cells)
        multi = dflo_amd.MultiConservationLaw(mesh, prm, devices=[0, 0])
        eng = lib.dflo_hip_multi_engine(multi._h, 1)
        assert lib.dflo_hip_halo_traces(eng) == (1 if cells == "0" else 0)
        if cells == "0":   # a slab of 16 x 24 cells of the periodic box: two cuts of 24 faces, one trace each
            assert lib.dflo_hip_n_ghost_traces(eng) == 48 and lib.dflo_hip_n_ghost_cells(eng) == 48
        _setup(multi, mesh, ic)
        multi.advance(6)
        res.append(multi.current_solution)
    assert np.array_equal(res[0], res[1])
    monkeypatch.setenv("DFLO_HALO_CELLS", "0")
    pk = _case("pk")
    multi = dflo_amd.MultiConservationLaw(pk[0], pk[1], devices=[0, 0])
    assert lib.dflo_hip_halo_traces(lib.dflo_hip_multi_engine(multi._h, 0)) == 0


def test_random_configurations_in_parts_against_the_single_engine():
    """tools/fuzz_multi.py: 200 random configurations (mesh kind and size, degree, basis, flux, boundary kinds, limiter and
    indicator switches, rough or smooth data) cut into 2-4 parts as slabs or RCB blocks, host-driven and device-resident steps:
    on the nodal basis the parts carry the bits of the single engine on any lattice (the parts' plans take the whole mesh's cell
    size, plan.h: build_plan's h_hint), on the modal basis they agree to rounding (1e-13)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_multi.py"), "200", "21"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "200 cases, 0 failures" in r.stdout
    assert "nodal basis, limited, not bit-identical: 0\n" in r.stdout, r.stdout[-1500:]     # (limited runs included)


@pytest.mark.parametrize("mode", ["self_rccl", "self_ipc", "self_direct"])
def test_random_configurations_against_themselves(mode):
    """the same generator, ONE part that is its own neighbour across a virtual cut (dflo_hip_multi_create_self) through the rank
    schedule with RCCL / the IPC sequence words, or the one-process schedule: every kind of record across the cut of random
    meshes, the single engine's bits on the nodal basis"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_multi.py"), "80", "23"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, FM_MODE=mode))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "80 cases (%s), 0 failures" % mode in r.stdout
    assert "nodal basis, limited, not bit-identical: 0\n" in r.stdout, r.stdout[-1500:]


def test_random_switch_settings_give_the_same_bits():
    """tools/fuzz_switches.py: 150 random configurations, each with the defaults and with 1-4 random run-time switches of
    tunables.h thrown (one engine or 2-4 parts): bit-identical, but for the two switches that are not bit-neutral by construction
    (DFLO_FUSE_POS=0: 1e-13; the average handed to the caller on bilinear cells: 1e-15)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_switches.py"), "150", "41"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "150 cases, 0 failures" in r.stdout
