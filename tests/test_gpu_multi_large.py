"""GPU: the multi-device driver on meshes whose parts have REAL interior shards -- the two-stream rim || interior schedule,
the ring of shards beside the rim for TVB (rim2) and the alternating receive areas at the sizes an 8-GPU run uses, not
only on the small meshes of test_gpu_multi.py where every shard is rim.

  C2-style   512 x 512 squares, Q2, HLLC, periodic vortex: 2 and 3 x-slabs and 4 RCB blocks; a part of the 2-slab run has
             32 x 64 shards of which 2 x 64 touch a cut.  Bar: bit-identical to the single engine (np.array_equal) after
             host-driven steps and >= 10 device-resident ones.
  C4-style   1001 x 1000 squares of the double Mach reflection (two of the eight x-slabs of BASELINE config 4), Q2, HLLC,
             TVB + positivity, the moving inflow state evaluated by the device's boundary programs.  Bar: <= 1e-8.

Each in the one-process driver (one host thread per part, the single-threaded driver, and the strict mode in which a sender
waits for an explicit "consumed" event of the receive area) and as one process per part with the host program's transport
(dflo_hip_multi_create_rank_custom + gloo staging, as test_gpu_multi_ranks.py).
What replaces what: update_ghost_values (src_mpi/claw.cc:793, src_mpi/limiter.cc:232), Utilities::MPI::min
(src_mpi/claw.cc:579)."""
import os
import sys

import numpy as np
import pytest

import dflo_amd
from dflo_amd import problems

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

HOST_STEPS, RESIDENT = 2, 12


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def case(name):
    """(mesh, parameters, initial/boundary function, device boundary programs)"""
    if name == "c2":
        mesh = dflo_amd.Mesh.cartesian(512, 512, -5.0, -5.0, 10.0 / 512, [-1, -1, -1, -1], 2)
        return mesh, dflo_amd.Parameters(flux="hllc", cfl=0.9), problems.isentropic_vortex, {}
    if name == "c4":      # bench.py --config c4 with two slabs: examples/double_mach_reflection, h = 1/1000
        nyc = 1000
        dy = 1.0 / nyc
        n1 = int(np.ceil((1.0 / 6.0) / dy))
        mesh = dflo_amd.Mesh.cartesian(1001, nyc, 1.0 / 6.0 - n1 * dy, 0.0, dy, [4, 2, 1, 3], 2)
        mesh.neighbors[:n1, 2] = -1 - 0
        prm = dflo_amd.Parameters(flux="hllc", limiter="TVB", char_lim=True, pos_lim=True, M=100.0, beta=1.0, cfl=0.9, final_time=1e9,
                                  boundary={0: "outflow", 1: "slip", 2: "outflow", 3: "inflow", 4: "inflow"})
        sh = "(x<1.0/6.0+(1+20*t)/sqrt(3))"
        programs = {3: ["57.1576766498*" + sh, "-33.0*" + sh, "8.0*%s + 1.4*(1-%s)" % (sh, sh), "563.5*%s + 2.5*(1-%s)" % (sh, sh)]}
        return mesh, prm, lambda x, y: problems.double_mach(x, y), programs
    raise KeyError(name)


def setup(claw, mesh, ic, programs, part=None):
    cell, face, bid, xy = claw.boundary_faces()
    if len(cell):
        bv = np.stack(ic(xy[..., 0], xy[..., 1]), axis=-1)
        claw.set_boundary_values(0, bv)
        claw.set_boundary_values(1, bv)
    if part is None:
        claw.set_initial_condition(mesh.interpolate(ic))
    else:   # one process per part: every rank evaluates the data on its own cells (owned + ghost) only, as bench.py does
        claw.set_part_initial_condition(0, claw.part_mesh(0).interpolate(ic))
    for b, exprs in programs.items():
        claw.set_boundary_function(b, exprs)


def run(claw, limited):
    if limited:
        claw.apply_limiter()      # run() limits the initial condition, src/claw.cc:997-1001
    out = {"dt": []}
    for _ in range(HOST_STEPS):
        dt = claw.compute_time_step()
        out["dt"].append(dt)
        claw.iterate_explicit(dt)
    out["t"] = claw.advance(RESIDENT)
    out["u"] = claw.current_solution
    out["avg"] = claw.cell_average
    return out


_REF = {}


def reference(name):
    """the single engine's run, once per session"""
    if name not in _REF:
        mesh, prm, ic, programs = case(name)
        one = dflo_amd.ConservationLaw(mesh, prm)
        setup(one, mesh, ic, programs)
        _REF[name] = run(one, prm.limiter == "TVB")
        one.close()
    return _REF[name]


def _plan_has_interior(multi):
    """every part must have shards that touch no cut (what the small meshes of test_gpu_multi.py lack)"""
    from dflo_amd._lib import lib
    for i in range(multi.n_local):
        eng = lib.dflo_hip_multi_engine(multi._h, i)
        n_owned = len(multi.part_cells(i)[0])
        rim = lib.dflo_hip_n_rim_shards(eng)
        assert 0 < rim < 0.25 * (n_owned / 64), (i, rim, n_owned // 64)


# how the one-process driver is arranged: "shared" is what it does by itself when the parts share a device (one stream pair and
# one host thread per DEVICE); "threads" gives every part its own stream pair and host thread -- what every part gets when each
# has a device of its own -- so that the cross-thread sequencing runs here; "copy": staging buffer + hipMemcpyPeerAsync instead
# of pack kernels that write into the peers' receive areas; "strict": senders wait for an explicit "consumed" event
MODES = {"shared": {}, "threads": {"DFLO_MULTI_GROUP": "part"}, "one_thread": {"DFLO_MULTI_GROUP": "part", "DFLO_MULTI_THREADS": "0"},
         "strict": {"DFLO_MULTI_GROUP": "part", "DFLO_MULTI_STRICT": "1"}, "shared_strict": {"DFLO_MULTI_STRICT": "1"},
         "copy": {"DFLO_MULTI_GROUP": "part", "DFLO_MULTI_COPY": "1"},
         "one_thread_strict": {"DFLO_MULTI_GROUP": "part", "DFLO_MULTI_THREADS": "0", "DFLO_MULTI_STRICT": "1"},
         # "fine": every buffer a peer's kernel writes (ghost-trace tables, time-step tables, receive areas) in fine-grained device
         # memory, with the strict mode -- the belt-and-braces arrangement bench.py falls back to when the default misbehaves across xGMI
         "fine": {"DFLO_MULTI_GROUP": "part", "DFLO_PEER_FINEGRAINED": "1"},
         "fine_strict": {"DFLO_MULTI_GROUP": "part", "DFLO_PEER_FINEGRAINED": "1", "DFLO_MULTI_STRICT": "1"}}


@pytest.mark.parametrize("n_parts,method,mode", [(2, "slab", "shared"), (3, "slab", "shared"), (4, "rcb", "shared"),
                                                 (2, "slab", "threads"), (3, "slab", "threads"), (4, "rcb", "threads"),
                                                 (2, "slab", "one_thread"), (4, "rcb", "one_thread"), (3, "slab", "strict"),
                                                 (4, "rcb", "shared_strict"), (3, "slab", "copy"), (4, "rcb", "one_thread_strict"),
                                                 (3, "slab", "fine"), (4, "rcb", "fine_strict")])
def test_c2_512_parts_bit_identical_to_the_single_engine(n_parts, method, mode, monkeypatch):
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    ref = reference("c2")
    mesh, prm, ic, programs = case("c2")
    multi = dflo_amd.MultiConservationLaw(mesh, prm, devices=[0] * n_parts, partitioner=method)
    _plan_has_interior(multi)
    setup(multi, mesh, ic, programs)
    got = run(multi, False)
    multi.close()
    assert got["dt"] == ref["dt"] and got["t"] == ref["t"]
    assert np.array_equal(got["avg"], ref["avg"])
    assert np.array_equal(got["u"], ref["u"])


@pytest.mark.parametrize("mode", ["shared", "threads", "one_thread", "strict", "copy", "fine_strict"])
def test_c4_slab_pair_matches_the_single_engine(mode, monkeypatch):
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    ref = reference("c4")
    mesh, prm, ic, programs = case("c4")
    multi = dflo_amd.MultiConservationLaw(mesh, prm, devices=[0, 0], partitioner="slab")
    _plan_has_interior(multi)
    setup(multi, mesh, ic, programs)
    got = run(multi, True)
    multi.close()
    assert got["dt"] == ref["dt"] and got["t"] == ref["t"]
    assert rel(got["avg"], ref["avg"]) < 1e-9
    assert rel(got["u"], ref["u"]) < 1e-8


def test_a_second_run_after_set_solution_repeats_the_first():
    """set_solution in the middle of a run (counters of the exchange schedule start again, receive areas change roles)"""
    mesh, prm, ic, programs = case("c2")
    multi = dflo_amd.MultiConservationLaw(mesh, prm, devices=[0] * 3, partitioner="slab")
    setup(multi, mesh, ic, programs)
    multi.advance(3)              # an odd number of exchanges behind us
    multi.elapsed_time = 0.0
    setup(multi, mesh, ic, programs)
    got = run(multi, False)
    multi.close()
    ref = reference("c2")
    assert got["dt"] == ref["dt"]
    assert np.array_equal(got["u"], ref["u"])


# ------------------------------------------------------------------ one process per part
def _worker(rank, world, port, name, ret, transport="callbacks"):
    if transport == "ipc":
        os.environ["DFLO_RANK_TRANSPORT"] = "ipc"
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import test_gpu_multi_large as T
    from dflo_amd.gloo_transport import make_callbacks
    mesh, prm, ic, programs = T.case(name)
    xf, af = make_callbacks("cuda:0")
    claw = dflo_amd.MultiConservationLaw.for_rank_custom(mesh, prm, 0, rank, world, xf, af, partitioner="slab")
    T.setup(claw, mesh, ic, programs, part=0)
    got = T.run(claw, prm.limiter == "TVB")
    own = claw.part_cells(0)[0]
    parts = [None] * world
    dist.all_gather_object(parts, (own, got["u"].reshape(mesh.n_cells, -1)[own], got["avg"][own]))
    if rank == 0:
        u, avg = np.empty((mesh.n_cells, mesh.ndof)), np.empty((mesh.n_cells, 4))
        for o, a, b in parts:
            u[o], avg[o] = a, b
        ref = T.reference(name)
        ret["dt"] = got["dt"] == ref["dt"]
        ret["t"] = got["t"] == ref["t"]
        ret["equal"] = bool(np.array_equal(u.reshape(-1), ref["u"]) and np.array_equal(avg, ref["avg"]))
        ret["err"] = float(rel(u.reshape(-1), ref["u"]))
    dist.barrier()
    claw.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world,transport", [("c2", 2, "callbacks"), ("c2", 3, "callbacks"), ("c4", 2, "callbacks"), ("c2", 3, "ipc"), ("c4", 2, "ipc")])
def test_ranks_on_large_meshes_match_the_single_engine(name, world, transport):
    import random
    import torch.multiprocessing as mp
    mgr = mp.get_context("spawn").Manager()   # (no fork of a process that holds a HIP runtime)
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 31000 + random.randint(0, 2000), name, ret, transport), nprocs=world, join=True)
    assert ret["dt"] and ret["t"], dict(ret)
    if name == "c2":
        assert ret["equal"], ret["err"]
    else:
        assert ret["err"] < 1e-8
