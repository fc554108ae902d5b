"""CPU: the self-halo partition (dflo_mesh_partition_self, dflo_amd/csrc/mesh.cc) -- one part that owns every cell and is its
own neighbour across a virtual cut.  Host logic only; the schedule that runs over it is tested on the GPU
(tests/test_gpu_selfhalo.py).  What it stands in for: the owned + ghost view of one rank of
parallel::distributed::Triangulation (src_mpi/claw.h:220) whose neighbours hold the same cells."""
import numpy as np
import pytest

import dflo_amd
from dflo_amd._lib import DfloError


def _check_structure(mesh, sub, cut_faces):
    """`cut_faces`: set of (cell, face) of the undivided mesh that lie on the virtual cut."""
    n = mesh.n_cells
    nb, nf = np.asarray(mesh.neighbors), np.asarray(mesh.neighbor_faces)
    snb, snf = np.asarray(sub.neighbors), np.asarray(sub.neighbor_faces)
    sc, so, ro = sub.comm
    cut_cells = sorted({c for c, _ in cut_faces})
    assert sub.n_owned == n and sub.n_cells == n + len(cut_cells)
    assert list(so) == [0, len(cut_cells)] and list(ro) == [0, len(cut_cells)]
    assert list(sc) == cut_cells                                   # send list = the cells on the cut, in global order
    assert list(sub.global_ids[n:]) == cut_cells                   # the ghost copies, sorted by the id of their original
    copy_of = {c: n + k for k, c in enumerate(cut_cells)}
    assert np.array_equal(sub.vertices[n:], mesh.vertices[cut_cells])
    assert np.array_equal(snf[:n], nf) and np.array_equal(snf[n:], nf[cut_cells])
    for c in range(n):
        for f in range(4):
            if (c, f) in cut_faces:
                assert snb[c, f] == copy_of[nb[c, f]]              # across the cut: the copy of the neighbour
            else:
                assert snb[c, f] == nb[c, f]
    none = min(int(snb.min()), -1)
    for k, c in enumerate(cut_cells):
        for f in range(4):
            if (c, f) in cut_faces:
                assert snb[n + k, f] == nb[c, f]                   # a copy sees the owned cells across its cut faces ...
            elif nb[c, f] >= 0:
                assert snb[n + k, f] < 0 and snb[n + k, f] == none   # ... and nothing else (DFLO_NBR_NONE)
            else:
                assert snb[n + k, f] == nb[c, f]                   # boundary faces stay what they are
    # the cut is symmetric: every cut face is seen from both sides
    for c, f in cut_faces:
        assert (int(nb[c, f]), int(nf[c, f]) & 3) in cut_faces


def test_periodic_seam_is_the_cut_of_a_single_virtual_part():
    mesh = dflo_amd.Mesh.cartesian(8, 6, 0.0, 0.0, 0.125, [-1, -1, -1, -1], 1)
    sub = mesh.partition_self(1)
    cut = {(j * 8, 0) for j in range(6)} | {(j * 8 + 7, 1) for j in range(6)}
    _check_structure(mesh, sub, cut)


@pytest.mark.parametrize("method", ["slab", "rcb"])
def test_cut_through_the_middle_of_a_bounded_mesh(method):
    mesh = dflo_amd.Mesh.cartesian(10, 4, 0.0, 0.0, 0.1, [2, 1, 0, 0], 2)
    sub = mesh.partition_self(2, method)
    own = mesh.partition_owners(2, method)
    nb = np.asarray(mesh.neighbors)
    cut = {(c, f) for c in range(mesh.n_cells) for f in range(4) if nb[c, f] >= 0 and own[nb[c, f]] != own[c]}
    assert len(cut) == 8     # one column of four faces, seen from both sides
    _check_structure(mesh, sub, cut)


def test_virtual_parts_leave_periodic_faces_alone():
    mesh = dflo_amd.Mesh.cartesian(8, 4, 0.0, 0.0, 0.125, [-1, -1, -1, -1], 1)
    sub = mesh.partition_self(2, "slab")
    own = mesh.partition_owners(2, "slab")
    nb, nf = np.asarray(mesh.neighbors), np.asarray(mesh.neighbor_faces)
    cut = {(c, f) for c in range(mesh.n_cells) for f in range(4) if own[nb[c, f]] != own[c] and not (nf[c, f] & 8)}
    assert len(cut) == 8     # the middle cut only: the seam between column 7 and column 0 is periodic and stays inside the part
    _check_structure(mesh, sub, cut)


def test_unstructured_quads_three_virtual_parts():
    from dflo_amd import gmsh
    verts, quads, bed, bid = gmsh.forward_step_quads(cl=0.2, seed=1)
    mesh = dflo_amd.Mesh.from_quads(verts, quads, bed, bid, 1)
    sub = mesh.partition_self(3, "rcb")
    own = mesh.partition_owners(3, "rcb")
    nb = np.asarray(mesh.neighbors)
    cut = {(c, f) for c in range(mesh.n_cells) for f in range(4) if nb[c, f] >= 0 and own[nb[c, f]] != own[c]}
    _check_structure(mesh, sub, cut)


def test_refusals():
    bounded = dflo_amd.Mesh.cartesian(4, 4, 0.0, 0.0, 0.25, [0, 0, 0, 0], 1)
    with pytest.raises(DfloError, match="no periodic faces"):
        bounded.partition_self(1)
    part = bounded.partition(2, 0)
    with pytest.raises(DfloError, match="already partitioned"):
        part.partition_self(2)
