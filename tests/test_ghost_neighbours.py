"""CPU: what the shard plan of a part knows of its ghost cells' neighbours (Plan::ghost_lrbt, dflo_amd/csrc/plan.cc) -- the table the
limiter pass over the ghost shards reads when a TVB stage of a multi-device run makes ONE exchange instead of the reference's two
(src_mpi/limiter.cc:232 + src_mpi/claw.cc:793; tests/test_gpu_selfhalo.py, test_gpu_multi.py hold the bits on the device).  Host logic
only.  A ghost cell must find every face neighbour it has in the undivided mesh: as an owned cell of the part, as an entry of the record
its owner sends (the owner's own cells: their averages are the owner's to give), or not at all (a physical boundary)."""
import ctypes as C

import numpy as np
import pytest

import dflo_amd
from dflo_amd._lib import lib


def _table(sub):
    n_ghost = sub.n_cells - sub.n_owned
    t = np.full((max(n_ghost, 1), 4), -9, dtype=np.int32)
    assert lib.dflo_hip_plan_ghost_neighbours(sub._ptr, t.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    return t[:n_ghost]


def _check(mesh, n_parts, method):
    """-> number of (ghost, face) pairs whose neighbour belongs to a THIRD part (what forbids the one-exchange stage)"""
    owner = mesh.partition_owners(n_parts, method)
    nb = np.asarray(mesh.neighbors)
    third = 0
    for rank in range(n_parts):
        sub = mesh.partition(n_parts, rank, method)
        gid = np.asarray(sub.global_ids)
        t = _table(sub)
        assert t.shape == (sub.n_cells - sub.n_owned, 4)
        for g in range(len(t)):
            G = int(gid[sub.n_owned + g])
            assert owner[G] != rank
            for f in range(4):
                N = int(nb[G, f])
                if N < 0:
                    assert t[g, f] == -1                              # a physical boundary: the limiter takes the cell's own slope
                elif owner[N] == rank:
                    assert 0 <= t[g, f] < sub.n_owned and gid[t[g, f]] == N   # one of this part's cells: its average is here
                else:
                    assert t[g, f] == -2                              # from the record of the ghost's owner ...
                    third += owner[N] != owner[G]                     # ... who can only give it if the cell is its own
    return third


@pytest.mark.parametrize("periodic", [False, True])
@pytest.mark.parametrize("n_parts", [2, 3, 5])
def test_slabs_every_ghost_neighbour_is_here_or_with_the_owner(n_parts, periodic):
    mesh = dflo_amd.Mesh.cartesian(40, 12, 0.0, 0.0, 0.1, [-1, -1, 0, 0] if periodic else [2, 1, 0, 0], 2)
    assert _check(mesh, n_parts, "slab") == 0        # x-slabs: a cut cell borders on one other part -- the one-exchange stage is sound


def test_rcb_blocks_meet_in_corners():
    mesh = dflo_amd.Mesh.cartesian(32, 32, 0.0, 0.0, 1.0 / 32, [2, 1, 0, 0], 1)
    assert _check(mesh, 4, "rcb") > 0                # four blocks: the corner cells border on two other parts (the driver keeps two exchanges)


def test_self_halo_copies_see_the_owned_cells_across_the_cut_and_the_record_elsewhere():
    mesh = dflo_amd.Mesh.cartesian(24, 16, 0.0, 0.0, 1.0 / 24, [2, 1, 0, 0], 2)
    sub = mesh.partition_self(2, "slab")
    nb = np.asarray(mesh.neighbors)
    owner = mesh.partition_owners(2, "slab")
    gid = np.asarray(sub.global_ids)
    t = _table(sub)
    assert len(t) == 2 * 16                           # both sides of the one cut
    for g in range(len(t)):
        G = int(gid[sub.n_owned + g])
        for f in range(4):
            N = int(nb[G, f])
            if N < 0:
                assert t[g, f] == -1
            elif owner[N] != owner[G]:
                assert t[g, f] == N                   # across the cut: the owned original (the part owns every cell)
            else:
                assert t[g, f] == -2                  # on the copy's own side: what its "owner" sends along
