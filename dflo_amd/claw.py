"""Host-side mirror of the part of dflo's ConservationLaw<2> that lies on the explicit path
(src/claw.h:95-128): same method names and argument meaning, every call goes through the C ABI
(include/dflo_hip.h) into the HIP engine."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, DfloError


class ConservationLaw:
    def __init__(self, mesh, parameters, device=0):
        self.mesh = mesh
        self.parameters = parameters
        self._h = C.c_void_p()
        p = parameters.struct()
        rc = lib.dflo_hip_create(mesh._ptr, C.byref(p), device, C.byref(self._h))
        if rc:
            self._h = C.c_void_p()
            raise DfloError(rc, lib.dflo_hip_last_error(None).decode())
        self.n_dofs = lib.dflo_hip_n_dofs(self._h)
        self.dofs_per_cell = lib.dflo_hip_dofs_per_cell(self._h)
        self.n_rk = lib.dflo_hip_n_rk(self._h)
        self.elapsed_time = 0.0
        self.global_dt = 0.0

    def close(self):
        if self._h:
            lib.dflo_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise DfloError(rc, lib.dflo_hip_last_error(self._h).decode())

    # ---- state
    def set_initial_condition(self, u):
        """current_solution = old_solution = u, then the cell average (src/ic.cc:118-120, src/claw.cc:997)."""
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert u.size == self.n_dofs
        self._chk(lib.dflo_hip_set_solution(self._h, _lib.dptr(u)))

    @property
    def current_solution(self):
        u = np.empty(self.n_dofs)
        self._chk(lib.dflo_hip_get_solution(self._h, _lib.dptr(u)))
        return u

    @property
    def cell_average(self):
        a = np.empty((self.mesh.n_cells, 4))
        self._chk(lib.dflo_hip_get_cell_average(self._h, _lib.dptr(a)))
        return a

    # ---- boundary data
    def boundary_faces(self):
        n = lib.dflo_hip_n_boundary_faces(self._h)
        N = self.mesh.degree + 1
        cell = np.zeros(max(n, 1), dtype=np.int32)
        face = np.zeros(max(n, 1), dtype=np.int32)
        bid = np.zeros(max(n, 1), dtype=np.int32)
        xy = np.zeros((max(n, 1), N, 2))
        self._chk(lib.dflo_hip_boundary_faces(self._h, _lib.iptr(cell), _lib.iptr(face), _lib.iptr(bid), _lib.dptr(xy)))
        return cell[:n], face[:n], bid[:n], xy[:n]

    def set_boundary_values(self, which, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._chk(lib.dflo_hip_set_boundary_values(self._h, which, _lib.dptr(v)))

    def set_boundary_function(self, boundary_id, expressions):
        """Evaluate the four `w_i value` expressions of boundary `boundary_id` on the device at t / t + dt of every
        step (replaces the per-step set_boundary_values for time-dependent boundary data). None removes a program."""
        from .expr import compile_program
        for c, e in enumerate(expressions):
            if e is None:
                ops, consts = np.zeros((0, 2), dtype=np.int32), np.zeros(0)
            else:
                ops, consts = compile_program(e, ("x", "y", "t"))
            ops = np.ascontiguousarray(ops, dtype=np.int32)
            consts = np.ascontiguousarray(consts, dtype=np.float64)
            self._chk(lib.dflo_hip_set_boundary_program(self._h, boundary_id, c, len(ops), _lib.iptr(ops), len(consts), _lib.dptr(consts)))

    def get_boundary_values(self, which):
        n = lib.dflo_hip_n_boundary_faces(self._h)
        v = np.empty((n, self.mesh.degree + 1, 4))
        self._chk(lib.dflo_hip_get_boundary_values(self._h, which, _lib.dptr(v)))
        return v

    # ---- the hot path
    def assemble_system(self, which=0):
        """right_hand_side of the current solution (src/assemble_explicit.cc:433-452)."""
        r = np.empty(self.n_dofs)
        self._chk(lib.dflo_hip_residual(self._h, which, _lib.dptr(r)))
        return r

    def compute_time_step(self):
        dt = C.c_double()
        self._chk(lib.dflo_hip_compute_dt(self._h, self.elapsed_time, C.byref(dt)))
        self.global_dt = dt.value
        return dt.value

    def iterate_explicit(self, dt=None):
        """All RK stages of one step (src/claw.cc:726-772); returns (res_norm0, res_norm)."""
        if dt is None:
            dt = self.global_dt
        r0, r1 = C.c_double(), C.c_double()
        self._chk(lib.dflo_hip_step(self._h, dt, C.byref(r0), C.byref(r1)))
        self.elapsed_time += dt
        return r0.value, r1.value

    def stage(self, rk, dt=-1.0):
        self._chk(lib.dflo_hip_stage(self._h, rk, dt))

    def end_step(self):
        self._chk(lib.dflo_hip_end_step(self._h))

    def advance(self, n_steps):
        """n_steps x {compute_time_step; iterate_explicit} with dt resident on the device."""
        t = C.c_double(self.elapsed_time)
        self._chk(lib.dflo_hip_advance(self._h, n_steps, C.byref(t)))
        self.elapsed_time = t.value
        return t.value

    def compute_cell_average(self):
        self._chk(lib.dflo_hip_compute_cell_average(self._h))

    def apply_limiter(self):
        self._chk(lib.dflo_hip_apply_limiter(self._h))

    def apply_positivity_limiter(self):
        self._chk(lib.dflo_hip_apply_positivity_limiter(self._h))

    def compute_shock_indicator(self):
        """KXRCF indicator of the current solution (src/indicator.cc:17-198); returns shock_indicator[n_cells]."""
        self._chk(lib.dflo_hip_compute_shock_indicator(self._h))
        return self.shock_indicator

    @property
    def shock_indicator(self):
        a = np.empty(self.mesh.n_cells)
        self._chk(lib.dflo_hip_get_shock_indicator(self._h, _lib.dptr(a)))
        return a

    def positivity_stats(self, reset=False):
        """(cell-stages through the positivity limiter proper, cell-stages it changed) since the last reset -- the limiter
        applied inside the stage kernel (positivity without TVB on Qk)."""
        v = (C.c_int64 * 2)()
        self._chk(lib.dflo_hip_positivity_stats(self._h, v, int(reset)))
        return int(v[0]), int(v[1])

    def failure_step(self):
        st = C.c_int64()
        self._chk(lib.dflo_hip_failure_step(self._h, C.byref(st)))
        return st.value

    def synchronize(self):
        self._chk(lib.dflo_hip_synchronize(self._h))

    def stage_timing(self, enable=True):
        ms, n = C.c_double(), C.c_int64()
        self._chk(lib.dflo_hip_stage_timing(self._h, int(enable), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    @property
    def uses_mfma(self):
        """does the stage kernel form its per-element contractions with matrix instructions (degree 3, DFLO_MFMA=1)?"""
        return bool(lib.dflo_hip_uses_mfma(self._h))

    def set_stream(self, stream_ptr):
        self._chk(lib.dflo_hip_set_stream(self._h, C.c_void_p(stream_ptr)))
