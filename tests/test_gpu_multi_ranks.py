"""GPU: the one-process-per-GPU schedule of the native multi-device driver (dflo_hip_multi_create_rank_*: rim on the comm
stream, pack, exchange, unpack, all-reduced time step, summed residual norms, agreed error status) run by 2 and 3 real
processes on ONE device.  RCCL refuses two ranks on one GPU, so the ranks move their bytes with the driver's
bring-your-own-transport entry (dflo_hip_multi_create_rank_custom, what a dflo built on MPI would use): the callbacks of
dflo_amd/gloo_transport.py stage the device buffers through gloo.  Everything except the ncclSend / ncclRecv / ncclAllReduce calls themselves is
the code the 8-GPU RCCL run executes; those calls are exercised by test_gpu_multi.py::test_rccl_loopback_transport.
transport "ipc" (DFLO_RANK_TRANSPORT=ipc): the per-stage path without any transport call -- the ranks map each other's receive
areas, time-step tables and sequence words with hipIpcGetMemHandle / hipIpcOpenMemHandle (real handles between real processes,
here on one device), the pack kernels deliver and signal, one-wavefront kernels wait."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, name, ret, transport="callbacks"):
    if transport == "ipc_coarse":   # ... with the exported window in plain device memory (default: fine-grained, hipExtMallocWithFlags)
        os.environ["DFLO_PEER_FINEGRAINED"] = "0"
        transport = "ipc"
    if transport in ("callbacks_one", "callbacks_two"):   # TVB stages: one exchange (the default with a transport library) | the reference's two
        os.environ["DFLO_TVB_ONE_EXCHANGE"] = "1" if transport == "callbacks_one" else "0"
        transport = "callbacks"
    if transport == "ipc":   # the per-stage path without a transport library: hipIpc-mapped receive areas + sequence words; the
        os.environ["DFLO_RANK_TRANSPORT"] = "ipc"   # callbacks only carry the handles at create and the host-side reductions
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import dflo_amd
    import test_gpu_multi as T
    mesh, prm, ic = T._case(name)
    limited = prm.limiter == "TVB"
    from dflo_amd.gloo_transport import make_callbacks
    _exchange, _allreduce = make_callbacks("cuda:0")
    claw = dflo_amd.MultiConservationLaw.for_rank_custom(mesh, prm, 0, rank, world, _exchange, _allreduce,
                                                          partitioner="rcb" if name == "c5" else "slab")
    assert claw.n_parts == world and claw.n_local == 1
    assert claw.comm_info()[2].startswith("IPC: ") == (transport == "ipc"), claw.comm_info()
    T._setup(claw, mesh, ic)
    got = T._run(claw, limited)
    own = claw.part_cells(0)[0]
    parts = [None] * world
    dist.all_gather_object(parts, (own, got["u"].reshape(mesh.n_cells, -1)[own], got["avg"][own]))
    if rank == 0:
        u, avg = np.empty((mesh.n_cells, mesh.ndof)), np.empty((mesh.n_cells, 4))
        for o, a, b in parts:
            u[o], avg[o] = a, b
        one = dflo_amd.ConservationLaw(mesh, prm)
        T._setup(one, mesh, ic)
        ref = T._run(one, limited)
        ret["dt"] = got["dt"] == ref["dt"]
        ret["t"] = got["t"] == ref["t"]
        ret["norms"] = max(abs(a0 - b0) / b0 + abs(a1 - b1) / b1 for (a0, a1), (b0, b1) in zip(got["norms"], ref["norms"]))
        ret["equal"] = bool(np.array_equal(u.reshape(-1), ref["u"]) and np.array_equal(avg, ref["avg"]))
        ret["err"] = float(np.abs(u.reshape(-1) - ref["u"]).max() / np.abs(ref["u"]).max())
        import hashlib
        ret["sha"] = hashlib.sha1(u.tobytes() + avg.tobytes()).hexdigest()
        ret["what"] = claw.comm_info()[2]
    dist.barrier()
    claw.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world,transport", [("c2", 2, "callbacks"), ("c2", 3, "callbacks"), ("c1", 2, "callbacks"), ("c4", 2, "callbacks"),
                                                  ("c4", 3, "callbacks"), ("c5", 2, "callbacks"), ("kxrcf", 2, "callbacks"),
                                                  ("c2", 2, "ipc"), ("c2", 3, "ipc"), ("c1", 3, "ipc"), ("c3", 2, "ipc"), ("c4", 3, "ipc"),
                                                  ("c5", 2, "ipc"), ("kxrcf", 2, "ipc"), ("c2", 3, "ipc_coarse"), ("c4", 2, "ipc_coarse")])
def test_ranks_on_one_device_match_the_single_engine(name, world, transport):
    import random
    mgr = mp.get_context("spawn").Manager()   # (no fork of a process that holds a HIP runtime)
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29500 + random.randint(0, 2000), name, ret, transport), nprocs=world, join=True)
    assert ret["dt"] and ret["t"], dict(ret)                 # all-reduced minima, host-driven and device-resident
    assert ret["norms"] < 1e-11
    if name in ("c1", "c2"):
        assert ret["equal"], ret["err"]                      # smooth data: bit-identical to the single engine
    else:
        assert ret["err"] < 1e-8


@pytest.mark.parametrize("name,world", [("c4", 2), ("c3", 3)])
def test_ranks_with_one_exchange_per_tvb_stage_carry_the_bits_of_two(name, world):
    """one process per GPU over a transport library (here: the callbacks): the cut cells travel unlimited with their neighbours'
    averages and every rank limits its ghost cells itself -- against the reference's two exchanges per stage (src_mpi/limiter.cc:232,
    src_mpi/claw.cc:793): the same bits on every rank"""
    import random
    out = []
    for transport in ("callbacks_one", "callbacks_two"):
        mgr = mp.get_context("spawn").Manager()
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, 29500 + random.randint(0, 2000), name, ret, transport), nprocs=world, join=True)
        assert ret["dt"] and ret["t"] and ret["err"] < 1e-8, dict(ret)
        out.append(dict(ret))
    assert "one exchange per stage" in out[0]["what"] and "two exchanges per stage" in out[1]["what"], (out[0]["what"], out[1]["what"])
    assert out[0]["sha"] == out[1]["sha"]


@pytest.mark.parametrize("transport", ["callbacks", "ipc"])
def test_random_configurations_on_two_ranks(transport):
    """tools/fuzz_ranks.py: two processes walk 40 random configurations together (one engine per process; gloo callbacks, or the IPC
    transport set up over them: handles of every receive area of every random mesh); nodal basis bit-identical to the single engine,
    modal basis to 1e-13, the ranks agree on every time step and stop."""
    import subprocess
    env = dict(os.environ, DFLO_RANK_TRANSPORT="ipc") if transport == "ipc" else dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_ranks.py"), "40", "61" if transport == "callbacks" else "62", "2"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "40 cases on 2 ranks, 0 failures" in r.stdout


def _lifecycle_worker(rank, world, port, ret):
    os.environ["DFLO_RANK_TRANSPORT"] = "ipc"
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import time
    import dflo_amd
    import test_gpu_multi as T
    from dflo_amd.gloo_transport import make_callbacks
    xf, af = make_callbacks("cuda:0")
    mesh, prm, ic = T._case("c2")
    make = lambda: dflo_amd.MultiConservationLaw.for_rank_custom(mesh, prm, 0, rank, world, xf, af, partitioner="slab")
    t0 = time.time()
    make().close()                       # nothing has been sent: nobody waits for anybody (a rank whose peers' create failed must be able to leave)
    ret["idle_close_%d" % rank] = time.time() - t0
    states = []
    for cycle in range(3):               # windows freed and allocated again while the neighbour is late: every run the same
        claw = make()
        T._setup(claw, mesh, ic)
        got = T._run(claw, False)
        own = claw.part_cells(0)[0]
        states.append(got["u"].reshape(mesh.n_cells, -1)[own].copy())
        if rank == (cycle & 1):
            time.sleep(0.5)              # ... this rank's last launches and its close come late
        claw.close()
    ret["same_%d" % rank] = bool(all(np.array_equal(s, states[0]) for s in states[1:]))
    dist.barrier()
    dist.destroy_process_group()


def test_driver_lifecycle_with_the_ipc_transport():
    """dflo_hip_multi_destroy with exported windows: a driver that never sent anything closes without its peers; one that has run
    meets its peers before it frees (a neighbour's last stores are awaited by nobody else), so that create / run / close cycles with a
    late neighbour give the same bits every time."""
    import random
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_lifecycle_worker, args=(2, 29500 + random.randint(0, 2000), ret), nprocs=2, join=True)
    assert ret["same_0"] and ret["same_1"], dict(ret)
    assert ret["idle_close_0"] < 60 and ret["idle_close_1"] < 60, dict(ret)


def _cycles_worker(rank, world, port, kind, cycles, ret):
    os.environ["DFLO_RANK_TRANSPORT"] = "ipc"
    os.environ["DFLO_PEER_FINEGRAINED"] = "1" if kind == "finegrained" else "0"
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import dflo_amd
    import test_gpu_multi as T
    from dflo_amd.gloo_transport import make_callbacks
    xf, af = make_callbacks("cuda:0")
    cases = [T._case(n) for n in ("c2", "c4")]

    def single(mesh, prm, ic, limited):
        one = dflo_amd.ConservationLaw(mesh, prm)
        T._setup(one, mesh, ic)
        r = T._run(one, limited)
        one.close()
        return r["u"]

    ref = [single(mesh, prm, ic, prm.limiter == "TVB") for mesh, prm, ic in cases]      # before any driver exists
    bad_after, bad_ranks = 0, 0
    for cycle in range(cycles):
        k = cycle & 1
        mesh, prm, ic = cases[k]
        limited = prm.limiter == "TVB"
        claw = dflo_amd.MultiConservationLaw.for_rank_custom(mesh, prm, 0, rank, world, xf, af, partitioner="slab")
        T._setup(claw, mesh, ic)
        got = T._run(claw, limited)
        own = claw.part_cells(0)[0]
        mine = got["u"].reshape(mesh.n_cells, -1)[own]
        claw.close()
        # the single engine created right AFTER the driver has gone takes the blocks the driver freed (LAB R5.15 / R6.3)
        after = single(mesh, prm, ic, limited)
        bad_after += int(not np.array_equal(after, ref[k]))
        want = ref[k].reshape(mesh.n_cells, -1)[own]
        bad_ranks += int(not (np.array_equal(mine, want) if k == 0 else np.abs(mine - want).max() <= 1e-8 * np.abs(want).max()))
    ret["after_%d" % rank], ret["ranks_%d" % rank] = bad_after, bad_ranks
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["finegrained", "plain"])
def test_two_hundred_create_run_destroy_cycles_each_followed_by_a_single_engine(kind):
    """VERDICT r5 item 3a.  dflo_hip_multi_destroy with exported windows, round 6's order: everybody's streams idle -> every importer
    closes its mappings -> only then the exporters free.  200 cycles of the two-rank IPC driver (C2- and C4-style in turn), the exported
    data window fine-grained (the default) or plain; after every cycle each rank creates a single engine -- which takes the blocks the
    driver has just freed -- and must reproduce, bit for bit, the engine that ran before any driver existed."""
    import random
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_cycles_worker, args=(2, 29500 + random.randint(0, 2000), kind, 200, ret), nprocs=2, join=True)
    assert ret["after_0"] == 0 and ret["after_1"] == 0, dict(ret)
    assert ret["ranks_0"] == 0 and ret["ranks_1"] == 0, dict(ret)
