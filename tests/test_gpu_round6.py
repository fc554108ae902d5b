"""GPU tests of round 6: the soundness switches of the exchange that rides inside the kernels (one process per GPU over mapped
tables: update_ghost_values of src_mpi/claw.cc:793 as stores and polls of the stage kernel's own workgroups).

  * DFLO_IPC_TIMEOUT_S: a workgroup that polls a neighbour's sequence word gives up after that many seconds, raises the failure
    word -- and then neither computes from the stale records nor delivers or counts itself as having delivered: the exchange never
    completes, every rank ends in an error instead of going on with wrong numbers;
  * DFLO_IPC_STRICT=1: every delivering workgroup fences at system scope before it counts itself (the formally complete
    release / acquire protocol); the same bits as the default."""
import ctypes as C
import time

import numpy as np
import pytest

import dflo_amd
from dflo_amd import problems
from dflo_amd._lib import lib

pytestmark = pytest.mark.gpu


def _half_of_a_periodic_box(monkeypatch, timeout_s):
    """part 0 of two x-slabs of a periodic box, told to deliver the traces of its cut faces into a scratch table and to wait for a
    neighbour that is not there"""
    monkeypatch.setenv("DFLO_IPC_TIMEOUT_S", str(timeout_s))
    hip = C.CDLL("libamdhip64.so.7")
    mesh = dflo_amd.Mesh.cartesian(64, 48, -5.0, -5.0, 10.0 / 64, [-1, -1, -1, -1], 2)
    prm = dflo_amd.Parameters(flux="hllc", cfl=0.8)
    sub = mesh.partition(2, 0)
    e = dflo_amd.ConservationLaw(sub, prm)
    assert lib.dflo_hip_halo_traces(e._h) == 1
    u0 = mesh.interpolate(problems.isentropic_vortex)
    e.set_initial_condition(u0.reshape(mesh.n_cells, mesh.ndof)[sub.global_ids].reshape(-1))
    nb = np.asarray(sub.neighbors)
    cells, faces = [], []
    for c in range(sub.n_owned):
        for f in range(4):
            if nb[c, f] >= sub.n_owned:
                cells.append(c)
                faces.append(f)
    cells, faces = np.asarray(cells, dtype=np.int32), np.asarray(faces, dtype=np.int32)
    assert lib.dflo_hip_set_send_faces(e._h, len(cells), cells.ctypes.data_as(C.POINTER(C.c_int32)), faces.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    buf = C.c_void_p()
    nbytes = len(cells) * 4 * 3 * 8 + 4096
    assert hip.hipMalloc(C.byref(buf), nbytes) == 0 and hip.hipMemset(buf, 0, nbytes) == 0
    words = buf.value + len(cells) * 4 * 3 * 8      # [0] the "neighbour's" word this engine publishes into, [1] the word it waits on, [2] failure
    for area in range(2):
        first = (C.c_int32 * 2)(0, len(cells))
        dst, fl = (C.c_void_p * 1)(buf.value), (C.c_void_p * 1)(words)
        assert lib.dflo_hip_set_deliver(e._h, area, 1, first, dst, fl) == 0
    wt = (C.c_void_p * 1)(words + 8)
    assert lib.dflo_hip_set_arrival_words(e._h, 1, wt, C.c_void_p(words + 16)) == 0
    return hip, e, buf, words


def _words(hip, words):
    w = (C.c_uint64 * 3)()
    assert hip.hipMemcpy(w, C.c_void_p(words), 24, 2) == 0
    return [int(x) for x in w]


def _one_stage(monkeypatch, neighbour_answers):
    hip, e, buf, words = _half_of_a_periodic_box(monkeypatch, timeout_s=1)
    if neighbour_answers:
        one = (C.c_uint64 * 1)(1)
        assert hip.hipMemcpy(C.c_void_p(words + 8), one, 8, 1) == 0      # the neighbour "has delivered" exchange 1
    dt = e.compute_time_step()
    assert lib.dflo_hip_stage_open(e._h, 0, dt) == 0
    assert lib.dflo_hip_stage_deliver(e._h, 1, 1) == 0
    assert lib.dflo_hip_stage_await(e._h, 1) == 0
    t0 = time.perf_counter()
    assert lib.dflo_hip_stage_update_part(e._h, 0) == 0
    assert lib.dflo_hip_synchronize(e._h) == 0
    sec = time.perf_counter() - t0
    published, _, failed = _words(hip, words)
    assert lib.dflo_hip_stage_finish(e._h) == 0
    u = e.current_solution.copy()
    hip.hipFree(buf)
    e.close()
    return sec, published, failed, u


def test_a_neighbour_that_never_answers_ends_the_exchange_instead_of_the_run_going_on_with_stale_traces(monkeypatch):
    sec_ok, pub_ok, fail_ok, u_ok = _one_stage(monkeypatch, True)
    assert fail_ok == 0 and pub_ok == 1 and sec_ok < 0.9      # the neighbour's word in place: the launch delivers and publishes
    sec, published, failed, u = _one_stage(monkeypatch, False)   # the neighbour's exchange number 1 never comes
    assert 0.9 < sec < 20.0, sec                          # DFLO_IPC_TIMEOUT_S=1, not the default 120
    assert failed == 1          # the failure word is up ...
    assert published == 0       # ... and the exchange was never published: the workgroups on the cut did not count themselves
    # they did not compute from the stale traces either: the shards that read ghost traces were left alone, the others computed
    # what they compute in the run whose neighbour answers
    same = (u.reshape(-1, 36) == u_ok.reshape(-1, 36)).all(axis=1)
    assert same.any() and not same.all()


@pytest.mark.parametrize("config", ["c2", "c4"])
def test_strict_delivery_gives_the_bits_of_the_default(config, monkeypatch):
    """self-halo over the IPC transport (the stage kernel / the limiter pass deliver): DFLO_IPC_STRICT=1 against the default and
    against the single engine"""
    if config == "c2":
        mesh = dflo_amd.Mesh.cartesian(128, 64, -5.0, -5.0, 10.0 / 128, [-1] * 4, 2)
        prm = dflo_amd.Parameters(flux="hllc", cfl=0.9)
        ic, lim = problems.isentropic_vortex, False
    else:
        mesh = dflo_amd.Mesh.cartesian(128, 32, 0.0, 0.0, 1.0 / 128, [2, 1, 0, 0], 2)
        prm = dflo_amd.Parameters(flux="hllc", limiter="TVB", char_lim=True, pos_lim=True, M=10.0, beta=1.0, cfl=0.9, final_time=1e9,
                                  boundary={0: "slip", 1: "outflow", 2: "inflow"})
        ic, lim = problems.sod, True
    u0 = mesh.interpolate(ic)

    def run(make):
        c = make()
        if lim:
            cell, face, bid, xy = c.boundary_faces()
            bv = np.stack(problems.sod(xy[..., 0], xy[..., 1]), axis=-1)
            c.set_boundary_values(0, bv)
            c.set_boundary_values(1, bv)
        c.set_initial_condition(u0)
        if lim:
            c.apply_limiter()
        t = c.advance(6)
        u = c.current_solution.copy()
        c.close()
        return t, u

    ref = run(lambda: dflo_amd.ConservationLaw(mesh, prm))
    out = []
    for strict in ("0", "1"):
        monkeypatch.setenv("DFLO_IPC_STRICT", strict)
        out.append(run(lambda: dflo_amd.MultiConservationLaw.for_self(mesh, prm, 0, transport="ipc", n_virtual=1 if config == "c2" else 2)))
    for t, u in out:
        assert t == ref[0] and np.array_equal(u, ref[1])
