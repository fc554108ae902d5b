"""Flat mesh description (dflo_mesh_t) -- what Triangulation + DoFHandler hold in dflo
(src/claw.h:178-185, src/claw.cc:957-967)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, DfloError


class Mesh:
    """Owns a dflo_mesh_t built by the C++ host library."""

    def __init__(self, ptr, comm=None, owner=None):
        self._ptr = ptr
        self.comm = comm  # (send_cells, send_offsets, recv_offsets) for partitioned meshes
        self._owner = owner  # borrowed view: `owner` keeps the C mesh alive and frees it

    def __del__(self):
        try:
            if self._ptr and self._owner is None:
                lib.dflo_mesh_free(self._ptr)
                self._ptr = None
        except Exception:
            pass

    # ---- builders
    @staticmethod
    def cartesian(nx, ny, x0, y0, h, side_bc, degree):
        """nx x ny squares; side_bc = boundary ids of (x-min, x-max, y-min, y-max), -1 = periodic."""
        out = C.POINTER(_lib.MeshStruct)()
        bc = np.asarray(side_bc, dtype=np.int32)
        rc = lib.dflo_mesh_cartesian(nx, ny, float(x0), float(y0), float(h), _lib.iptr(bc), degree, C.byref(out))
        if rc:
            raise DfloError(rc, lib.dflo_mesh_last_error().decode())
        return Mesh(out)

    @staticmethod
    def from_quads(vertices, quads, bedges=None, bedge_id=None, degree=1):
        v = np.ascontiguousarray(vertices, dtype=np.float64)
        q = np.ascontiguousarray(quads, dtype=np.int32)
        be = np.ascontiguousarray(bedges if bedges is not None else np.zeros((0, 2)), dtype=np.int32)
        bi = np.ascontiguousarray(bedge_id if bedge_id is not None else np.zeros(len(be)), dtype=np.int32)
        out = C.POINTER(_lib.MeshStruct)()
        rc = lib.dflo_mesh_from_quads(len(v), _lib.dptr(v), len(q), _lib.iptr(q), len(be), _lib.iptr(be),
                                      _lib.iptr(bi), degree, C.byref(out))
        if rc:
            raise DfloError(rc, lib.dflo_mesh_last_error().decode())
        return Mesh(out)

    @staticmethod
    def read_gmsh(path, degree, mapping="cartesian"):
        out = C.POINTER(_lib.MeshStruct)()
        rc = lib.dflo_mesh_read_gmsh(str(path).encode(), degree, _lib.MAPPING[mapping], C.byref(out))
        if rc:
            raise DfloError(rc, lib.dflo_mesh_last_error().decode())
        return Mesh(out)

    def make_periodic(self, id_first, id_second, direction):
        """Turn the boundaries id_first / id_second (offset along direction "x" | "y") into periodic neighbours."""
        rc = lib.dflo_mesh_make_periodic(self._ptr, id_first, id_second, {"x": 0, "y": 1}[direction])
        if rc:
            raise DfloError(rc, lib.dflo_mesh_last_error().decode())

    def partition_owners(self, n_ranks, method="slab"):
        """Owner rank of every cell under the partitioner `method` ("slab" | "rcb")."""
        o = np.empty(self.n_cells, dtype=np.int32)
        rc = lib.dflo_mesh_partition_owners(self._ptr, n_ranks, _lib.PARTITIONER[method], _lib.iptr(o))
        if rc:
            raise DfloError(rc, lib.dflo_mesh_last_error().decode())
        return o

    def partition(self, n_ranks, rank, method="slab"):
        """Owned + one ghost layer sub-mesh of `rank` (replaces parallel::distributed::Triangulation)."""
        out = C.POINTER(_lib.MeshStruct)()
        sc, so, ro = (C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)())
        rc = lib.dflo_mesh_partition_ex(self._ptr, n_ranks, rank, _lib.PARTITIONER[method], C.byref(out), C.byref(sc), C.byref(so),
                                        C.byref(ro))
        if rc:
            raise DfloError(rc, lib.dflo_mesh_last_error().decode())
        so_a = np.ctypeslib.as_array(so, shape=(n_ranks + 1,)).copy()
        ro_a = np.ctypeslib.as_array(ro, shape=(n_ranks + 1,)).copy()
        n_send = int(so_a[-1])
        sc_a = np.ctypeslib.as_array(sc, shape=(n_send,)).copy() if n_send > 0 else np.zeros(0, dtype=np.int32)
        return Mesh(out, comm=(sc_a, so_a, ro_a))

    def partition_self(self, n_virtual=1, method="slab"):
        """Self-halo view (dflo_mesh_partition_self): every cell owned, the cells on the virtual cut copied as ghost cells."""
        out = C.POINTER(_lib.MeshStruct)()
        sc, so, ro = (C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)())
        rc = lib.dflo_mesh_partition_self(self._ptr, n_virtual, _lib.PARTITIONER[method], C.byref(out), C.byref(sc), C.byref(so), C.byref(ro))
        if rc:
            raise DfloError(rc, lib.dflo_mesh_last_error().decode())
        so_a = np.ctypeslib.as_array(so, shape=(2,)).copy()
        ro_a = np.ctypeslib.as_array(ro, shape=(2,)).copy()
        n_send = int(so_a[-1])
        sc_a = np.ctypeslib.as_array(sc, shape=(n_send,)).copy() if n_send > 0 else np.zeros(0, dtype=np.int32)
        return Mesh(out, comm=(sc_a, so_a, ro_a))

    # ---- views
    @property
    def struct(self):
        return self._ptr.contents

    @property
    def n_cells(self):
        return self.struct.n_cells

    @property
    def n_owned(self):
        return self.struct.n_owned_cells

    @property
    def degree(self):
        return self.struct.degree

    @property
    def basis(self):
        return "Pk" if self.struct.basis == _lib.BASIS["Pk"] else "Qk"

    @property
    def n_s(self):
        n = self.degree + 1
        return n * (n + 1) // 2 if self.basis == "Pk" else n * n

    @property
    def ndof(self):
        return 4 * self.n_s

    @property
    def vertices(self):
        return np.ctypeslib.as_array(self.struct.cell_vertices, shape=(self.n_cells, 4, 2))

    @property
    def neighbors(self):
        return np.ctypeslib.as_array(self.struct.cell_face_neighbor, shape=(self.n_cells, 4))

    @property
    def neighbor_faces(self):
        return np.ctypeslib.as_array(self.struct.cell_face_neighbor_face, shape=(self.n_cells, 4))

    @property
    def global_ids(self):
        if not self.struct.cell_global_id:
            return np.arange(self.n_cells, dtype=np.int64)
        return np.ctypeslib.as_array(self.struct.cell_global_id, shape=(self.n_cells,))

    def set_mapping(self, mapping):
        self.struct.mapping = _lib.MAPPING[mapping]

    def set_basis(self, basis):
        """"Qk" (FE_DGQArbitraryNodes at Gauss points) or "Pk" (FE_DGP, orthonormal Legendre modes), src/main.cc:36-45."""
        self.struct.basis = _lib.BASIS[basis]

    def support_points(self):
        """Real-space support points of the Qk DoFs = points of QGauss<2>(k+1), [n_cells, (k+1)^2, 2] (src/ic.cc:104-121)."""
        xy = np.empty((self.n_cells, (self.degree + 1) ** 2, 2))
        rc = lib.dflo_mesh_support_points(self._ptr, _lib.dptr(xy))
        if rc:
            raise DfloError(rc, lib.dflo_mesh_last_error().decode())
        return xy

    def interpolate(self, fn):
        """VectorTools::interpolate for Qk (src/ic.cc:104-121): fn(x, y) -> [mx, my, rho, E] arrays.
        Returns the state vector in dflo's DoF order [cell][comp][node]."""
        if self.basis == "Pk":
            return self.project(fn)
        xy = self.support_points()
        w = fn(xy[..., 0], xy[..., 1])  # 4 arrays [n_cells, n_s]
        u = np.stack([np.broadcast_to(np.asarray(c, dtype=np.float64), xy.shape[:2]) for c in w], axis=1)
        return np.ascontiguousarray(u).reshape(-1)

    def modal_matrix(self):
        """T[j][m] = psi_m at Gauss node j = a + N b of the unit cell; psi_m = Pt_i(xi) Pt_j(eta) with
        Pt_n(x) = sqrt(2n+1) P_n(2x-1), modes ordered "for j: for i <= k-j" (FE_DGP; src/claw.cc:107-113).
        Also returns the tensor Gauss weights."""
        N = self.degree + 1
        t, w = np.polynomial.legendre.leggauss(N)
        w = 0.5 * w
        P = np.stack([np.sqrt(2 * n + 1) * np.polynomial.legendre.Legendre.basis(n)(t) for n in range(N)], axis=1)  # [q][n]
        modes = [(i, j) for j in range(N) for i in range(N - j)]
        T = np.empty((N * N, len(modes)))
        for m, (i, j) in enumerate(modes):
            T[:, m] = np.outer(P[:, j], P[:, i]).reshape(-1)  # node index a + N b: b (eta) slow
        return T, np.outer(w, w).reshape(-1)

    def project(self, fn):
        """L2 projection on the Pk modes with QGauss(k+1) (set_initial_condition_Pk, src/ic.cc:128-164):
        u_m = sum_q f(x_q) psi_m(x_q) w_q (the mass matrix of the orthonormal modes is |K| I)."""
        xy = self.support_points()
        T, ww = self.modal_matrix()
        w = fn(xy[..., 0], xy[..., 1])
        f = np.stack([np.broadcast_to(np.asarray(c, dtype=np.float64), xy.shape[:2]) for c in w], axis=1)  # [cell][4][q]
        u = np.einsum("ncq,q,qm->ncm", f, ww, T)
        return np.ascontiguousarray(u).reshape(-1)

    def angular_momentum(self, solution):
        """Total angular momentum  int x m_y - y m_x  with QGauss(k+1) (compute_angular_momentum, src/claw.cc:602-635;
        a printed diagnostic of run(), every `compute angular momentum` iterations)."""
        N = self.degree + 1
        xy = self.support_points()[: self.n_owned]                   # the points of QGauss<2>(k+1)
        T, ww = self.modal_matrix()
        u = np.asarray(solution).reshape(self.n_cells, 4, -1)[: self.n_owned]
        m = u[:, :2] if self.basis == "Qk" else np.einsum("ncm,qm->ncq", u[:, :2], T)
        v = np.asarray(self.vertices)[: self.n_owned]
        t, _ = np.polynomial.legendre.leggauss(N)
        g = 0.5 * (t + 1.0)
        xi, eta = np.tile(g, N), np.repeat(g, N)
        # |det J| of the bilinear map at the quadrature points (|K| for squares)
        xxi = (1 - eta) * (v[:, 1, 0] - v[:, 0, 0])[:, None] + eta * (v[:, 3, 0] - v[:, 2, 0])[:, None]
        yxi = (1 - eta) * (v[:, 1, 1] - v[:, 0, 1])[:, None] + eta * (v[:, 3, 1] - v[:, 2, 1])[:, None]
        xet = (1 - xi) * (v[:, 2, 0] - v[:, 0, 0])[:, None] + xi * (v[:, 3, 0] - v[:, 1, 0])[:, None]
        yet = (1 - xi) * (v[:, 2, 1] - v[:, 0, 1])[:, None] + xi * (v[:, 3, 1] - v[:, 1, 1])[:, None]
        jxw = np.abs(xxi * yet - xet * yxi) * ww[None, :]
        return float(((xy[..., 0] * m[:, 1] - xy[..., 1] * m[:, 0]) * jxw).sum())
