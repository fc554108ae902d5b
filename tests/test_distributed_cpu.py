"""CPU, world_size 2 and 3, gloo: the partition (slabs and RCB) and the exchange lists the native multi-device driver
works from (dflo_mesh_partition_ex: send cells / send offsets / receive offsets per peer), driven with the oracle on each
rank's owned+ghost sub-mesh, must reproduce the single-process oracle run.  The HIP engines cannot run here; on the GPU
box dflo_amd/csrc/multi.hip moves the same records between the same cells with RCCL send/recv or peer copies
(tests/test_gpu_multi.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


class HaloExchange:
    """What multi.hip's post()/arrive() do with ncclSend/ncclRecv, with gloo point-to-point: to peer q go the cells
    send_cells[send_off[q]:send_off[q+1]], from q come the ghost cells [recv_off[q], recv_off[q+1]) of the ghost range."""

    def __init__(self, so, ro):
        self.so, self.ro = [int(v) for v in so], [int(v) for v in ro]
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.peers = [q for q in range(self.world) if q != self.rank and (self.so[q + 1] > self.so[q] or self.ro[q + 1] > self.ro[q])]

    def exchange(self, send, recv, width):
        ops = []
        for q in self.peers:
            if self.ro[q + 1] > self.ro[q]:
                ops.append(dist.P2POp(dist.irecv, recv[self.ro[q] * width:self.ro[q + 1] * width], q))
            if self.so[q + 1] > self.so[q]:
                ops.append(dist.P2POp(dist.isend, send[self.so[q] * width:self.so[q + 1] * width], q))
        for w in dist.batch_isend_irecv(ops) if ops else []:
            w.wait()


def _worker(rank, world, port, case, ret, method="slab"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dflo_amd
    from dflo_amd import problems
    import oracle_lib as O

    nx, ny, degree, flux, limiter, pos, side_bc, bnd = case
    if side_bc == "unstructured":   # Delaunay triangles cut into quads: irregular connectivity, flipped faces, q1 mapping
        from dflo_amd import gmsh
        verts, quads, bed, bid = gmsh.unstructured_quads(nx, seed=5)
        mesh = dflo_amd.Mesh.from_quads(verts, quads, bed, bid, degree)
    else:
        mesh = dflo_amd.Mesh.cartesian(nx, ny, 0.0, 0.0, 1.0 / nx, side_bc, degree)
    prm = dflo_amd.Parameters(flux=flux, limiter=limiter, pos_lim=pos, boundary=bnd, beta=2.0)
    ic = (lambda x, y: problems.smooth_perturbation(x, y, L=1.0)) if limiter == "none" else problems.sod
    u0 = mesh.interpolate(ic)
    part = mesh.partition(world, rank, method)
    sc, so, ro = part.comm
    ndof = part.ndof
    gid = np.asarray(part.global_ids)
    ora = O.Oracle(part, prm)
    halo = HaloExchange(so, ro)
    n_ghost = part.n_cells - part.n_owned

    def bvals(o):
        cell, face, bid, xy = o.boundary_faces()
        if len(cell) == 0:
            return
        bv = np.stack(ic(xy[..., 0], xy[..., 1]), axis=-1)
        o.set_boundary_values(0, bv)
        o.set_boundary_values(1, bv)

    def exchange(width, get, put):
        full = get().reshape(part.n_cells, width)
        send = torch.from_numpy(np.ascontiguousarray(full[sc]).reshape(-1))
        recv = torch.empty(max(n_ghost, 1) * width, dtype=torch.float64)
        halo.exchange(send, recv, width)
        full[part.n_owned:] = recv.numpy()[: n_ghost * width].reshape(n_ghost, width)
        put(full.reshape(-1))

    ora.set_solution(u0.reshape(mesh.n_cells, ndof)[gid].reshape(-1))
    bvals(ora)
    dt = 0.04 / nx   # well inside the CFL limit so that round-off differences are not amplified
    for step in range(3):
        for rk in range(ora.n_rk):
            ora.set_dt(dt)
            ora.stage(rk)      # update + cell averages (+ limiter using ghost averages) on the owned cells
            exchange(ndof, ora.get_solution, ora.set_current_only)
            ora.compute_cell_average()
        ora.end_step()
    u = ora.get_solution().reshape(part.n_cells, ndof)[: part.n_owned]
    parts = [None] * world
    dist.all_gather_object(parts, (gid[: part.n_owned], u))
    if rank == 0:
        out = np.empty((mesh.n_cells, ndof))
        for g, v in parts:
            out[g] = v
        ref = O.Oracle(mesh, prm)
        ref.set_solution(u0)
        bvals(ref)
        for step in range(3):
            ref.step(dt)
        ret["err"] = float(np.abs(out.reshape(-1) - ref.get_solution()).max() / np.abs(ref.get_solution()).max())
    dist.destroy_process_group()


CASES = [
    (12, 6, 2, "hllc", "none", False, [-1, -1, -1, -1], None),
    (12, 6, 1, "lxf", "none", False, [-1, -1, 0, 0], {0: "slip"}),
    (16, 4, 1, "roe", "none", True, [2, 1, 0, 0], {0: "slip", 1: "outflow", 2: "inflow"}),
    (5, 5, 2, "hllc", "none", True, "unstructured", {0: "slip", 1: "outflow", 2: "slip", 3: "inflow"}),
]


@pytest.mark.parametrize("case", CASES)
def test_two_rank_halo_exchange_matches_single_process(case):
    import random
    port = 29500 + random.randint(0, 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, case, ret), nprocs=2, join=True)
    assert ret["err"] < 1e-12, ret["err"]


def test_rcb_blocks_and_three_ranks():
    """RCB blocks on the unstructured mesh (peers in two directions), and a 3-way partition (a middle slab with two
    different neighbours)."""
    import random
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(3, 29500 + random.randint(0, 2000), CASES[3], ret, "rcb"), nprocs=3, join=True)
    assert ret["err"] < 1e-12, ret["err"]
    ret2 = mgr.dict()
    mp.spawn(_worker, args=(3, 29500 + random.randint(0, 2000), CASES[2], ret2, "slab"), nprocs=3, join=True)
    assert ret2["err"] < 1e-12, ret2["err"]
