"""CPU: the C-ABI library loads, exports every symbol the headers of include/ declare, the host-side mesh
builders work, and the engine refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import dflo_amd
from dflo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ("dflo_hip.h", "dflo_mesh.h", "dflo_hip_transport.h", "dflo_hip_diag.h")


def declared_symbols(headers=HEADERS):
    out = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(dflo_(?:hip|mesh)_\w+)\s*\(", text))
    return sorted(out)


def test_library_exports_every_declared_symbol():
    """the four headers of include/ -- the contract, the host-side mesh helpers, the seams below the contract, the diagnostics --
    together declare exactly what the library exports and what the Python binding lists; no symbol is declared twice"""
    syms = declared_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(_lib.lib, s), "libdflo_hip.so does not export %s" % s
    # and the python binding lists the same set
    assert sorted(_lib.SYMBOLS) == syms
    assert sum(len(declared_symbols((h,))) for h in HEADERS) == len(syms)
    # what the library exports under the ABI's prefixes and no header declares would be an undocumented seam
    out = os.popen("nm -D --defined-only %s" % _lib.LIB_PATH).read()
    exported = set(re.findall(r"\bT (dflo_(?:hip|mesh)_\w+)$", out, flags=re.M))
    assert exported - set(syms) <= {"dflo_hip_create_with_cell_size", "dflo_hip_run"}, sorted(exported - set(syms))


def test_the_contract_header_is_small_and_stands_alone():
    """include/dflo_hip.h is what a dflo maintainer reads: the drop-in surface of SURVEY 8b and the multi-device driver, <= 45
    functions, no transport plumbing and no debug hooks; the deal.II adaptor needs nothing else, the stand-alone C++ driver nothing
    but it and the mesh helpers; every header compiles on its own as C99 and as C++."""
    import subprocess
    pub = declared_symbols(("dflo_hip.h",))
    assert len(pub) <= 45, len(pub)
    for must in ("dflo_hip_create", "dflo_hip_destroy", "dflo_hip_last_error", "dflo_hip_set_solution", "dflo_hip_get_solution",
                 "dflo_hip_get_cell_average", "dflo_hip_set_boundary_values", "dflo_hip_residual", "dflo_hip_compute_dt", "dflo_hip_step",
                 "dflo_hip_advance", "dflo_hip_apply_limiter", "dflo_hip_apply_positivity_limiter", "dflo_hip_multi_create",
                 "dflo_hip_multi_create_rank", "dflo_hip_multi_advance", "dflo_hip_multi_destroy"):   # SURVEY 8b + INTEGRATION.md
        assert must in pub, must
    for s in pub:
        assert not any(w in s for w in ("debug", "deliver", "arrival", "pack_", "unpack_", "_dt_table", "_dt_slot", "attach_event", "timing")), s
    txt = open(os.path.join(ROOT, "include", "dflo_hip.h")).read()
    assert "#include \"dflo_" not in txt     # the contract includes none of the others
    adaptor = open(os.path.join(ROOT, "include", "dflo_hip_dealii.hpp")).read()
    assert set(re.findall(r'#include "(dflo_\w+\.h)"', adaptor)) == {"dflo_hip.h"}
    assert set(re.findall(r"\b(dflo_(?:hip|mesh)_\w+)\s*\(", re.sub(r"//[^\n]*", "", adaptor))) <= set(pub)
    run_cc = "".join(open(os.path.join(ROOT, "dflo_amd", "csrc", f)).read() for f in ("dflo_run.cc", "frontend.cc"))
    used = set(re.findall(r"\b(dflo_(?:hip|mesh)_\w+)\s*\(", re.sub(r"//[^\n]*", "", run_cc))) - {"dflo_hip_run"}
    assert used <= set(declared_symbols(("dflo_hip.h", "dflo_mesh.h"))), sorted(used - set(declared_symbols(("dflo_hip.h", "dflo_mesh.h"))))
    for h in HEADERS:
        for lang in (["gcc", "-x", "c", "-std=c99"], ["g++", "-x", "c++"]):
            r = subprocess.run(lang + ["-fsyntax-only", "-Wall", "-Werror", os.path.join(ROOT, "include", h)], capture_output=True, text=True)
            assert r.returncode == 0, (h, r.stderr[-400:])


def test_product_does_not_touch_the_oracle():
    """Nothing under dflo_amd/ may import, link or open anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dflo_amd")):
        for f in files:
            if f.endswith((".py", ".cc", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower(), os.path.join(dirpath, f)
    # the shared library does not depend on the oracle library
    out = os.popen("ldd %s" % _lib.LIB_PATH).read()
    assert "oracle" not in out


def test_variant_patches_still_apply():
    """tools/variants/*.patch: measured-and-lost kernel variants kept as patches -- they must keep applying to the tree they
    sit next to, or their evidence cannot be reproduced (only where the tree is a git checkout: the GPU box gets a snapshot)."""
    import glob
    import subprocess
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("not a git checkout")
    patches = sorted(glob.glob(os.path.join(ROOT, "tools", "variants", "*.patch")))
    assert patches
    hand_written = {"lds_pad.patch"}   # a description with a one-line hunk, not a `git diff`
    for f in patches:
        if os.path.basename(f) in hand_written:
            continue
        base = re.search(r"^# base-commit: ([0-9a-f]{7,40})", open(f).read(), re.M)
        if base:   # a record pinned to the tree it was measured on: that commit has to stay reachable
            r = subprocess.run(["git", "cat-file", "-e", base.group(1) + "^{commit}"], cwd=ROOT, capture_output=True, text=True)
            if r.returncode != 0 and subprocess.run(["git", "rev-parse", "--is-shallow-repository"], cwd=ROOT, capture_output=True,
                                                     text=True).stdout.strip() == "true":
                continue
            assert r.returncode == 0, "%s: base commit %s is gone" % (os.path.basename(f), base.group(1))
            continue
        r = subprocess.run(["git", "apply", "--check", f], cwd=ROOT, capture_output=True, text=True)
        assert r.returncode == 0, "%s: %s" % (os.path.basename(f), r.stderr[-400:])


def test_cartesian_mesh_builder():
    m = dflo_amd.Mesh.cartesian(4, 3, 1.0, 2.0, 0.5, [7, -1 if False else 8, -1, -1], 2)
    assert m.n_cells == 12 and m.n_owned == 12 and m.degree == 2
    v, nb, nf = m.vertices, m.neighbors, m.neighbor_faces
    assert np.allclose(v[5], [[1.5, 2.5], [2.0, 2.5], [1.5, 3.0], [2.0, 3.0]])
    assert nb[0, 0] == -1 - 7 and nb[3, 1] == -1 - 8          # boundary ids
    assert nb[0, 2] == 8 and nf[0, 2] == (3 | 8)              # periodic in y: bottom row <-> top row
    assert nb[5, 0] == 4 and nf[5, 0] == 1 and nb[5, 3] == 9 and nf[5, 3] == 2
    with pytest.raises(dflo_amd.DfloError):
        dflo_amd.Mesh.cartesian(4, 3, 0, 0, 0.5, [-1, 1, -1, -1], 1)   # periodic sides must pair up


def test_quad_soup_builder_orientation_and_flips():
    # two quads sharing an edge, the second one given clockwise and starting elsewhere
    verts = np.array([[0, 0], [1, 0], [1, 1], [0, 1], [2, 0], [2, 1]], dtype=float)
    quads = [[0, 1, 2, 3], [2, 1, 4, 5][::-1]]
    m = dflo_amd.Mesh.from_quads(verts, quads, [[0, 1], [1, 4]], [3, 4], degree=1)
    v = m.vertices
    for c in range(2):   # lexicographic vertex order with positive jacobian
        e1, e2 = v[c, 1] - v[c, 0], v[c, 2] - v[c, 0]
        assert e1[0] * e2[1] - e1[1] * e2[0] > 0
    assert m.neighbors[0, 1] == 1 and m.neighbors[1, 0] == 0
    assert m.neighbor_faces[0, 1] & 3 == 0 and m.neighbor_faces[0, 1] & 4 == 0   # same direction, no flip
    assert m.neighbors[0, 2] == -1 - 3 and m.neighbors[1, 2] == -1 - 4
    assert m.neighbors[0, 0] == -1 - 0    # unlabelled boundary edges get id 0


def test_gmsh_reader(tmp_path):
    msh = tmp_path / "two.msh"
    msh.write_text("""$MeshFormat
2.2 0 8
$EndMeshFormat
$Nodes
6
1 0 0 0
2 1 0 0
3 1 1 0
4 0 1 0
5 2 0 0
6 2 1 0
$EndNodes
$Elements
4
1 1 2 2 1 1 2
2 1 2 1 2 5 6
3 3 2 100 1 1 2 3 4
4 3 2 100 1 2 5 6 3
$EndElements
""")
    m = dflo_amd.Mesh.read_gmsh(msh, degree=2, mapping="cartesian")
    assert m.n_cells == 2 and m.struct.mapping == _lib.MAPPING["cartesian"]
    assert m.neighbors[0, 2] == -1 - 2 and m.neighbors[1, 1] == -1 - 1     # Physical Line id -> boundary id
    assert m.neighbors[0, 1] == 1


def test_support_points_are_gauss_points():
    m = dflo_amd.Mesh.cartesian(2, 1, 0.0, 0.0, 2.0, [0, 0, 0, 0], 2)
    xy = m.support_points()
    g = 0.5 - np.sqrt(15) / 10
    assert np.allclose(xy[0, 0], [2 * g, 2 * g]) and np.allclose(xy[1, 4], [3.0, 1.0])
    assert np.allclose(xy[0, 1], [1.0, 2 * g])          # x runs fastest


def test_partition_owned_plus_ghost_layer():
    m = dflo_amd.Mesh.cartesian(8, 4, 0.0, 0.0, 1.0, [-1, -1, 0, 0], 1)
    parts = [m.partition(2, r) for r in range(2)]
    assert sum(p.n_owned for p in parts) == m.n_cells
    for r, p in enumerate(parts):
        gid = p.global_ids
        assert len(set(gid)) == p.n_cells                      # no duplicates
        own = gid[: p.n_owned]
        assert ((own % 8 < 4) == (r == 0)).all()               # x-slabs
        assert p.n_cells - p.n_owned == 8                       # one ghost column on each side (periodic)
        sc, so, ro = p.comm
        other = 1 - r
        assert so[other + 1] - so[other] == 8 and ro[other + 1] - ro[other] == 8
        # what I send is what the other side expects to receive, in the same order
        q = parts[other]
        sent_gids = gid[sc[so[other]:so[other + 1]]]
        recv_gids = q.global_ids[q.n_owned + q.comm[2][r]: q.n_owned + q.comm[2][r + 1]]
        assert (sent_gids == recv_gids).all()
        # neighbours of owned cells all resolve locally
        nb = p.neighbors[: p.n_owned]
        assert (nb != _lib.NBR_NONE).all()


def test_engine_fails_loudly_without_gpu():
    if torch.cuda.is_available() or os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    m = dflo_amd.Mesh.cartesian(4, 4, 0.0, 0.0, 0.25, [-1] * 4, 1)
    with pytest.raises(dflo_amd.DfloError) as e:
        dflo_amd.ConservationLaw(m, dflo_amd.Parameters())
    assert e.value.code == -5 and "no CPU fallback" in str(e.value)


def test_create_rejects_bad_parameters():
    m = dflo_amd.Mesh.cartesian(4, 4, 0.0, 0.0, 0.25, [-1] * 4, 1)
    p = dflo_amd.Parameters().struct()
    h = C.c_void_p()
    p.flux_type = 9
    assert _lib.lib.dflo_hip_create(m._ptr, C.byref(p), 0, C.byref(h)) == -1
    assert b"flux" in _lib.lib.dflo_hip_last_error(None)


@pytest.mark.parametrize("seed", range(12))
def test_partition_invariants_on_random_meshes(seed):
    """dflo_mesh_partition on random meshes (lattices with or without periodic pairs, unstructured quadrilaterals) and
    2-5 ranks: every cell owned exactly once, the ghost layer is exactly the face neighbours of the owned cells that live
    elsewhere, every neighbour of an owned cell resolves locally to the right global cell (with its face and flip code),
    and what rank a sends to rank b is, cell for cell and in order, what b expects from a."""
    from dflo_amd import gmsh
    rng = np.random.default_rng(100 + seed)
    if seed % 3 == 2:
        verts, quads, bed, bid = gmsh.unstructured_quads(int(rng.integers(4, 9)), seed=seed)
        m = dflo_amd.Mesh.from_quads(verts, quads, bed, bid, 1)
    else:
        nx, ny = int(rng.integers(3, 30)), int(rng.integers(1, 12))
        side = [-1, -1, 0, 0] if seed % 3 == 0 else [1, 2, -1, -1] if rng.random() < 0.5 else [0, 1, 2, 3]
        m = dflo_amd.Mesh.cartesian(nx, ny, 0.0, 0.0, 0.1, side, 1)
    world = int(rng.integers(2, 6))
    if world > m.n_cells:
        world = 2
    method = "rcb" if seed % 2 else "slab"
    parts = [m.partition(world, r, method) for r in range(world)]
    gnb, gnf = m.neighbors, m.neighbor_faces
    owner = np.full(m.n_cells, -1)
    for r, p in enumerate(parts):
        gid = np.asarray(p.global_ids)
        assert len(set(gid.tolist())) == p.n_cells
        own = gid[: p.n_owned]
        assert (owner[own] == -1).all()
        owner[own] = r
    assert (owner >= 0).all() and sum(p.n_owned for p in parts) == m.n_cells
    assert (owner == m.partition_owners(world, method)).all()
    counts = np.bincount(owner, minlength=world)
    assert counts.max() - counts.min() <= (1 if method == "slab" else world)        # balanced
    for r, p in enumerate(parts):
        gid = np.asarray(p.global_ids)
        own, ghost = gid[: p.n_owned], gid[p.n_owned:]
        nbr = gnb[own]
        want = set(int(g) for g in nbr[nbr >= 0].reshape(-1) if owner[g] != r)
        assert set(ghost.tolist()) == want                                   # exactly one layer
        lnb, lnf = p.neighbors[: p.n_owned], p.neighbor_faces[: p.n_owned]
        assert (lnb != _lib.NBR_NONE).all()
        inner = nbr >= 0
        assert (gid[lnb[inner]] == nbr[inner]).all() and (lnb[~inner] == nbr[~inner]).all()
        assert (lnf[inner] == gnf[own][inner]).all()
        sc, so, ro = p.comm
        for q in range(world):
            if q == r:
                assert so[q + 1] == so[q] and ro[q + 1] == ro[q]
                continue
            pq = parts[q]
            sent = gid[sc[so[q]:so[q + 1]]]
            assert (owner[sent] == r).all()
            expect = np.asarray(pq.global_ids)[pq.n_owned + pq.comm[2][r]: pq.n_owned + pq.comm[2][r + 1]]
            assert (sent == expect).all()
        assert ro[world] == p.n_cells - p.n_owned                               # every ghost cell is received from someone


def test_rcb_cuts_compact_blocks():
    """Recursive coordinate bisection (DFLO_PART_RCB): a 16 x 16 lattice on 4 ranks falls into four 8 x 8 blocks (cut
    length 2 x 16 faces, where four slabs cut 3 x 16), 3 ranks get 85/86 cells, and on an unstructured mesh the cut is
    shorter than the slab partition's."""
    from dflo_amd import gmsh
    m = dflo_amd.Mesh.cartesian(16, 16, 0.0, 0.0, 1.0 / 16, [0, 0, 0, 0], 1)

    def cut(mesh, owner):
        nb = mesh.neighbors
        inner = nb >= 0
        return int((owner[:, None] != owner[np.where(inner, nb, 0)])[inner].sum()) // 2

    o = m.partition_owners(4, "rcb").reshape(16, 16)
    for r in range(4):
        jj, ii = np.nonzero(o == r)
        assert len(ii) == 64 and ii.max() - ii.min() == 7 and jj.max() - jj.min() == 7
    assert cut(m, o.reshape(-1)) == 32 and cut(m, m.partition_owners(4, "slab")) == 48
    c3 = np.bincount(m.partition_owners(3, "rcb"), minlength=3)
    assert sorted(c3.tolist()) == [85, 85, 86]
    verts, quads, bed, bid = gmsh.unstructured_quads(12, seed=3)
    u = dflo_amd.Mesh.from_quads(verts, quads, bed, bid, 1)
    assert cut(u, u.partition_owners(8, "rcb")) < cut(u, u.partition_owners(8, "slab"))


def _independent_face_table(mesh):
    """Face neighbours re-derived from nothing but the flattened cells' vertex coordinates: face f of a cell joins the vertices
    (0,2), (1,3), (0,1), (2,3) of its lexicographic vertex list (deal.II's reference cell); two cells are neighbours
    across a face when they hold the same two points; the face points run the opposite way on the two sides when the shared
    edge is traversed in opposite directions (increasing free coordinate on either side)."""
    v = np.asarray(mesh.vertices)
    nc = len(v)
    ends = np.array([[0, 2], [1, 3], [0, 1], [2, 3]])
    a = v[:, ends[:, 0], :]                      # [cell][face][xy] first end (free coordinate 0)
    b = v[:, ends[:, 1], :]
    # a key per undirected edge: the two end points sorted lexicographically, as exact doubles
    swap = (a[..., 0] > b[..., 0]) | ((a[..., 0] == b[..., 0]) & (a[..., 1] > b[..., 1]))
    lo = np.where(swap[..., None], b, a).reshape(-1, 2)
    hi = np.where(swap[..., None], a, b).reshape(-1, 2)
    key = np.concatenate([lo, hi], axis=1)
    order = np.lexsort(key.T[::-1])
    ks = key[order]
    same = (ks[1:] == ks[:-1]).all(axis=1)
    nbr = np.full(nc * 4, -1, dtype=np.int64)
    nbf = np.zeros(nc * 4, dtype=np.int64)
    first = np.nonzero(same)[0]
    assert not (same[1:] & same[:-1]).any()      # no edge shared by three cells
    i0, i1 = order[first], order[first + 1]
    nbr[i0], nbr[i1] = i1 // 4, i0 // 4
    flip = swap.reshape(-1)[i0] != swap.reshape(-1)[i1]
    nbf[i0] = (i1 % 4) + 4 * flip
    nbf[i1] = (i0 % 4) + 4 * flip
    return nbr.reshape(nc, 4), nbf.reshape(nc, 4)


@pytest.mark.parametrize("kind", ["tunnel", "square", "lattice"])
def test_mesh_flattening_against_an_independent_derivation(kind):
    """The flat mesh the engine AND the oracle consume comes out of dflo_amd/csrc/mesh.cc; this derives its face tables a second
    time, from the cells' vertex coordinates alone (numpy, edge matching by coordinates), at a size where hand-made fixtures end:
    neighbours, neighbour faces, flip flags, boundary ids, cell orientation."""
    from dflo_amd import gmsh
    if kind == "tunnel":
        verts, quads, bed, bid = gmsh.forward_step_quads(cl=0.2 / 16, seed=3)      # ~97 000 unstructured quadrilaterals
        mesh = dflo_amd.Mesh.from_quads(verts, quads, bed, bid, 1)
    elif kind == "square":
        verts, quads, bed, bid = gmsh.unstructured_quads(60, seed=5)
        mesh = dflo_amd.Mesh.from_quads(verts, quads, bed, bid, 2)
    else:
        mesh = dflo_amd.Mesh.cartesian(37, 23, -1.0, 2.0, 0.125, [5, 6, 7, 8], 1)
        verts = quads = None
    nbr, nbf = _independent_face_table(mesh)
    got, gotf = np.asarray(mesh.neighbors).astype(np.int64), np.asarray(mesh.neighbor_faces).astype(np.int64)
    inner = nbr >= 0
    assert (got[inner] == nbr[inner]).all()
    assert ((gotf[inner] & 7) == nbf[inner]).all()                 # face seen from the neighbour + 4 * flip
    assert (got[~inner] < 0).all()                                 # what has no partner is a boundary face
    v = np.asarray(mesh.vertices)
    area2 = (v[:, 1, 0] - v[:, 0, 0]) * (v[:, 2, 1] - v[:, 0, 1]) - (v[:, 1, 1] - v[:, 0, 1]) * (v[:, 2, 0] - v[:, 0, 0])
    assert (area2 > 0).all()                                       # lexicographic vertices of counter-clockwise cells
    if quads is not None:
        # the cells are the input's quadrilaterals (as vertex sets), and every boundary face carries the id of the input edge it is
        want = np.sort(np.asarray(verts)[np.asarray(quads)].reshape(len(quads), -1), axis=1)
        have = np.sort(v.reshape(len(v), -1), axis=1)
        assert np.array_equal(have[np.lexsort(have.T[::-1])], want[np.lexsort(want.T[::-1])])
        ends = np.array([[0, 2], [1, 3], [0, 1], [2, 3]])
        mid_of = {}
        for (p, q), b in zip(np.asarray(bed), np.asarray(bid)):
            m = 0.5 * (np.asarray(verts)[p] + np.asarray(verts)[q])
            mid_of[(round(m[0], 12), round(m[1], 12))] = int(b)
        cb, fb = np.nonzero(~inner)
        assert len(cb) == len(bed)
        for c, f in zip(cb, fb):
            m = 0.5 * (v[c, ends[f, 0]] + v[c, ends[f, 1]])
            assert -1 - got[c, f] == mid_of[(round(m[0], 12), round(m[1], 12))]
    else:
        # lattice: ids of (x-min, x-max, y-min, y-max), cells numbered x fastest
        nx, ny = 37, 23
        c = np.arange(nx * ny).reshape(ny, nx)
        assert (-1 - got[c[:, 0], 0] == 5).all() and (-1 - got[c[:, -1], 1] == 6).all()
        assert (-1 - got[c[0, :], 2] == 7).all() and (-1 - got[c[-1, :], 3] == 8).all()
        assert (got[c[:, 1:], 0] == c[:, :-1]).all() and (got[c[1:, :], 2] == c[:-1, :]).all()


def test_dealii_adaptor_calls_match_the_c_abi():
    """include/dflo_hip_dealii.hpp cannot be compiled here (it needs deal.II), so at least every dflo_* entry point it calls has
    to exist in include/dflo_hip.h with the number of arguments it is called with -- the header moves on, the adaptor must follow."""
    h = open(os.path.join(ROOT, "include", "dflo_hip.h")).read()     # (the contract alone: the adaptor includes nothing else)
    a = open(os.path.join(ROOT, "include", "dflo_hip_dealii.hpp")).read()
    a = re.sub(r"//[^\n]*", "", a)
    a = re.sub(r"/\*.*?\*/", "", a, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|int32_t|const char \*|void|double)\s*(dflo_(?:hip|mesh)_\w+)\s*\(([^;{]*?)\)\s*;", h, re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1

    def n_args(txt):
        depth, n = 0, 1 if txt.strip() else 0
        for ch in txt:
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == "," and depth == 0:
                n += 1
        return n

    calls = 0
    for m in re.finditer(r"\b(dflo_(?:hip|mesh)_\w+)\s*\(", a):
        i, depth = m.end(), 1
        while depth and i < len(a):
            depth += {"(": 1, ")": -1}.get(a[i], 0)
            i += 1
        assert m.group(1) in protos, "%s is not declared in dflo_hip.h" % m.group(1)
        assert protos[m.group(1)] == n_args(a[m.end():i - 1]), "%s: %d argument(s) declared" % (m.group(1), protos[m.group(1)])
        calls += 1
    assert calls >= 30


def test_developer_tools_compile():
    """tools/*.py (fuzzers, probes, profile digests) are not imported by any CPU test: at least they have to parse."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    assert len(files) > 10
    for f in files:
        compile(open(f).read(), f, "exec")
