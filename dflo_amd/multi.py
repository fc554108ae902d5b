"""Host-side mirror of ConservationLaw on several devices: a thin ctypes wrapper of the native multi-device
driver (dflo_hip_multi_* in include/dflo_hip.h, dflo_amd/csrc/multi.hip).  No Python sits inside the step loop:
advance() is one C call for any number of steps, stages, halo exchanges and time-step reductions.

  MultiConservationLaw(mesh, parameters, devices=[0, 1, ...])       one process, one engine per listed device
  MultiConservationLaw.for_rank(mesh, parameters, device, rank, world, unique_id)    one process per GPU (torchrun)
  MultiConservationLaw.for_self(mesh, parameters, device)           one part, its own neighbour: the whole schedule on one GPU

State, boundary data and results use the numbering of the undivided mesh, as ConservationLaw does (src/claw.h:95-128;
what the MPI variant spreads over ranks, src_mpi/claw.cc:793, :579).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, DfloError


def comm_unique_id():
    """DFLO_COMM_ID_BYTES bytes naming a new RCCL communicator: call on one rank, hand to the others."""
    buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
    rc = lib.dflo_hip_comm_unique_id(buf)
    if rc:
        raise DfloError(rc, lib.dflo_hip_multi_last_error(None).decode())
    return buf.raw


class MultiConservationLaw:
    def __init__(self, mesh, parameters, devices=(0,), partitioner="slab", _rank=None):
        self.mesh = mesh
        self.parameters = parameters
        self._h = C.c_void_p()
        p = parameters.struct()
        if _rank is None:
            dev = (C.c_int * len(devices))(*devices)
            rc = lib.dflo_hip_multi_create(mesh._ptr, C.byref(p), len(devices), dev, _lib.PARTITIONER[partitioner], C.byref(self._h))
        elif len(_rank) == 3:      # self-halo: (device, n_virtual, transport)
            device, n_virtual, transport = _rank
            rc = lib.dflo_hip_multi_create_self(mesh._ptr, C.byref(p), device, n_virtual, _lib.PARTITIONER[partitioner],
                                                _lib.SELF_TRANSPORT[transport], C.byref(self._h))
        elif len(_rank) == 5:      # the host program's own transport: (device, rank, world, exchange, allreduce)
            device, rank, world, xf, af = _rank
            self._callbacks = (_lib.EXCHANGE_FN(xf), _lib.ALLREDUCE_FN(af))     # keep the trampolines alive
            rc = lib.dflo_hip_multi_create_rank_custom(mesh._ptr, C.byref(p), device, rank, world, self._callbacks[0], self._callbacks[1],
                                                       None, _lib.PARTITIONER[partitioner], C.byref(self._h))
        else:
            device, rank, world, uid = _rank
            rc = lib.dflo_hip_multi_create_rank(mesh._ptr, C.byref(p), device, rank, world, uid, _lib.PARTITIONER[partitioner],
                                                C.byref(self._h))
        if rc:
            self._h = C.c_void_p()
            raise DfloError(rc, lib.dflo_hip_multi_last_error(None).decode())
        self.n_dofs = lib.dflo_hip_multi_n_dofs(self._h)
        self.n_owned_dofs = lib.dflo_hip_multi_n_owned_dofs(self._h)
        self.dofs_per_cell = mesh.ndof
        self.n_rk = lib.dflo_hip_multi_n_rk(self._h)
        self.n_parts = lib.dflo_hip_multi_n_parts(self._h)
        self.n_local = lib.dflo_hip_multi_n_local(self._h)
        self.elapsed_time = 0.0
        self.global_dt = 0.0

    @classmethod
    def for_rank(cls, mesh, parameters, device, rank, world, unique_id, partitioner="slab"):
        return cls(mesh, parameters, partitioner=partitioner, _rank=(device, rank, world, unique_id))

    @classmethod
    def for_self(cls, mesh, parameters, device=0, n_virtual=None, transport="rccl", partitioner="slab"):
        """Self-halo (dflo_hip_multi_create_self): one part that is its own neighbour across a virtual cut, driven through the
        whole multi-device schedule on one GPU.  n_virtual None: 1 (the periodic seam in x) when the mesh has one, else 2 (a cut
        through the middle)."""
        if n_virtual is None:
            nf = np.asarray(mesh.neighbor_faces)
            n_virtual = 1 if bool(((nf[:, :2] & 8) != 0).any()) else 2
        return cls(mesh, parameters, partitioner=partitioner, _rank=(device, n_virtual, transport))

    @classmethod
    def for_rank_custom(cls, mesh, parameters, device, rank, world, exchange, allreduce, partitioner="slab"):
        """One process per GPU with the caller's transport: exchange(user, n_peers, peer, send_ptr, send_bytes, recv_ptr,
        recv_bytes, stream) and allreduce(user, values, n, op, stream) get device pointers (see include/dflo_hip.h)."""
        return cls(mesh, parameters, partitioner=partitioner, _rank=(device, rank, world, exchange, allreduce))

    def close(self):
        if self._h:
            lib.dflo_hip_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise DfloError(rc, lib.dflo_hip_multi_last_error(self._h).decode())

    # ---- parts
    def part_cells(self, i):
        """(global ids of the owned cells, global ids of the ghost cells) of the i-th local part."""
        no, ng, ids = C.c_int32(), C.c_int32(), C.POINTER(C.c_int64)()
        self._chk(lib.dflo_hip_multi_part_cells(self._h, i, C.byref(no), C.byref(ng), C.byref(ids)))
        g = np.ctypeslib.as_array(ids, shape=(no.value + ng.value,)).copy()
        return g[: no.value], g[no.value:]

    def part_mesh(self, i):
        """The i-th local part as a Mesh (owned cells first, then its ghost cells); borrowed from the handle."""
        from .mesh import Mesh
        ptr = lib.dflo_hip_multi_part_mesh(self._h, i)
        return Mesh(ptr, owner=self)

    def set_part_initial_condition(self, i, u_part):
        """Initial state of one local part in its own numbering (every rank evaluates the data on its cells only)."""
        u = np.ascontiguousarray(u_part, dtype=np.float64)
        self._chk(lib.dflo_hip_multi_set_part_solution(self._h, i, _lib.dptr(u)))

    def owned_cells(self):
        return np.concatenate([self.part_cells(i)[0] for i in range(self.n_local)])

    # ---- state
    def set_initial_condition(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert u.size == self.n_dofs
        self._chk(lib.dflo_hip_multi_set_solution(self._h, _lib.dptr(u)))

    @property
    def current_solution(self):
        """Global vector; with one process per GPU only this rank's owned cells are filled (the rest is NaN)."""
        u = np.full(self.n_dofs, np.nan)
        self._chk(lib.dflo_hip_multi_get_solution(self._h, _lib.dptr(u)))
        return u

    @property
    def cell_average(self):
        a = np.full((self.mesh.n_cells, 4), np.nan)
        self._chk(lib.dflo_hip_multi_get_cell_average(self._h, _lib.dptr(a)))
        return a

    # ---- boundary data
    def boundary_faces(self):
        n = lib.dflo_hip_multi_n_boundary_faces(self._h)
        N = self.mesh.degree + 1
        cell = np.zeros(max(n, 1), dtype=np.int32)
        face = np.zeros(max(n, 1), dtype=np.int32)
        bid = np.zeros(max(n, 1), dtype=np.int32)
        xy = np.zeros((max(n, 1), N, 2))
        self._chk(lib.dflo_hip_multi_boundary_faces(self._h, _lib.iptr(cell), _lib.iptr(face), _lib.iptr(bid), _lib.dptr(xy)))
        return cell[:n], face[:n], bid[:n], xy[:n]

    def set_boundary_values(self, which, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._chk(lib.dflo_hip_multi_set_boundary_values(self._h, which, _lib.dptr(v)))

    def set_boundary_function(self, boundary_id, expressions):
        from .expr import compile_program
        for c, e in enumerate(expressions):
            if e is None:
                ops, consts = np.zeros((0, 2), dtype=np.int32), np.zeros(0)
            else:
                ops, consts = compile_program(e, ("x", "y", "t"))
            ops = np.ascontiguousarray(ops, dtype=np.int32)
            consts = np.ascontiguousarray(consts, dtype=np.float64)
            self._chk(lib.dflo_hip_multi_set_boundary_program(self._h, boundary_id, c, len(ops), _lib.iptr(ops), len(consts),
                                                              _lib.dptr(consts)))

    # ---- the hot path
    def assemble_system(self, which=0):
        r = np.full(self.n_dofs, np.nan)
        self._chk(lib.dflo_hip_multi_residual(self._h, which, _lib.dptr(r)))
        return r

    def compute_time_step(self):
        dt = C.c_double()
        self._chk(lib.dflo_hip_multi_compute_dt(self._h, self.elapsed_time, C.byref(dt)))
        self.global_dt = dt.value
        return dt.value

    def iterate_explicit(self, dt=None):
        if dt is None:
            dt = self.global_dt
        r0, r1 = C.c_double(), C.c_double()
        self._chk(lib.dflo_hip_multi_step(self._h, dt, C.byref(r0), C.byref(r1)))
        self.elapsed_time += dt
        return r0.value, r1.value

    def advance(self, n_steps):
        t = C.c_double(self.elapsed_time)
        rc = lib.dflo_hip_multi_advance(self._h, n_steps, C.byref(t))
        self.elapsed_time = t.value
        self._chk(rc)
        return t.value

    def apply_limiter(self):
        self._chk(lib.dflo_hip_multi_apply_limiter(self._h))

    def apply_positivity_limiter(self):
        self._chk(lib.dflo_hip_multi_apply_positivity_limiter(self._h))

    def positivity_stats(self, reset=False):
        """Sum over the local engines of ConservationLaw.positivity_stats."""
        tot = [0, 0]
        for i in range(self.n_local):
            v = (C.c_int64 * 2)()
            rc = lib.dflo_hip_positivity_stats(lib.dflo_hip_multi_engine(self._h, i), v, int(reset))
            if rc:
                raise DfloError(rc, "positivity_stats")
            tot[0] += int(v[0])
            tot[1] += int(v[1])
        return tuple(tot)

    def synchronize(self):
        self._chk(lib.dflo_hip_multi_synchronize(self._h))

    def stage_timing(self, enable=True):
        ms, n = C.c_double(), C.c_int64()
        self._chk(lib.dflo_hip_multi_stage_timing(self._h, int(enable), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    @property
    def uses_mfma(self):
        return bool(lib.dflo_hip_uses_mfma(lib.dflo_hip_multi_engine(self._h, 0)))

    def exchange_timing(self, enable=True):
        """(average microseconds the comm stream spent per sampled halo exchange, number of samples) since the last call."""
        us, n = C.c_double(), C.c_int64()
        self._chk(lib.dflo_hip_multi_exchange_timing(self._h, int(enable), C.byref(us), C.byref(n)))
        return us.value, n.value

    def comm_info(self):
        """(ranks, own rank) as the transport itself reports them -- ncclCommCount / ncclCommUserRank for RCCL -- and a description."""
        cnt, rk = C.c_int32(), C.c_int32()
        buf = C.create_string_buffer(640)
        self._chk(lib.dflo_hip_multi_comm_info(self._h, C.byref(cnt), C.byref(rk), buf, 640))
        return cnt.value, rk.value, buf.value.decode()
