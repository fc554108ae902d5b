"""ctypes wrapper of the CPU oracle (oracle/libdflo_oracle.so).  TEST INFRASTRUCTURE: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = C.CDLL(os.path.join(ROOT, "oracle", "libdflo_oracle.so"))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    return a.ctypes.data_as(_dp)


_lib.dflo_oracle_create.restype = C.c_void_p
_lib.dflo_oracle_create.argtypes = [C.c_void_p, C.c_void_p]
for name, res, args in [
    ("dflo_oracle_destroy", None, [C.c_void_p]),
    ("dflo_oracle_error", C.c_int, [C.c_void_p]),
    ("dflo_oracle_message", C.c_char_p, [C.c_void_p]),
    ("dflo_oracle_set_threads", None, [C.c_void_p, C.c_int]),
    ("dflo_oracle_n_rk", C.c_int, [C.c_void_p]),
    ("dflo_oracle_dofs_per_cell", C.c_int, [C.c_void_p]),
    ("dflo_oracle_n_dofs", C.c_longlong, [C.c_void_p]),
    ("dflo_oracle_n_boundary_faces", C.c_int, [C.c_void_p]),
    ("dflo_oracle_set_solution", None, [C.c_void_p, _dp]),
    ("dflo_oracle_set_current_only", None, [C.c_void_p, _dp]),
    ("dflo_oracle_get_solution", None, [C.c_void_p, _dp]),
    ("dflo_oracle_get_cell_average", None, [C.c_void_p, _dp]),
    ("dflo_oracle_set_cell_average", None, [C.c_void_p, _dp]),
    ("dflo_oracle_get_inv_mass", None, [C.c_void_p, _dp]),
    ("dflo_oracle_boundary_faces", None, [C.c_void_p, _ip, _ip, _ip, _dp]),
    ("dflo_oracle_set_boundary_values", None, [C.c_void_p, C.c_int, _dp]),
    ("dflo_oracle_support_points", None, [C.c_void_p, _dp]),
    ("dflo_oracle_cell_quadrature", None, [C.c_void_p, _dp, _dp]),
    ("dflo_oracle_cell_shape", None, [C.c_void_p, _dp]),
    ("dflo_oracle_assemble", None, [C.c_void_p, C.c_int, _dp]),
    ("dflo_oracle_compute_cell_average", None, [C.c_void_p]),
    ("dflo_oracle_compute_time_step", C.c_double, [C.c_void_p, C.c_double]),
    ("dflo_oracle_apply_limiter", None, [C.c_void_p]),
    ("dflo_oracle_compute_shock_indicator", None, [C.c_void_p, _dp]),
    ("dflo_oracle_get_shock_indicator", None, [C.c_void_p, _dp]),
    ("dflo_oracle_apply_positivity_limiter", C.c_int, [C.c_void_p]),
    ("dflo_oracle_set_dt", None, [C.c_void_p, C.c_double]),
    ("dflo_oracle_stage", C.c_int, [C.c_void_p, C.c_int, _dp]),
    ("dflo_oracle_end_step", None, [C.c_void_p]),
    ("dflo_oracle_step", C.c_int, [C.c_void_p, C.c_double, _dp, _dp]),
    ("dflo_oracle_twin_supported", C.c_int, [C.c_void_p]),
    ("dflo_oracle_twin_advance", C.c_int, [C.c_void_p, C.c_int, C.c_double, _dp, _dp]),
    ("dflo_oracle_numerical_flux", None, [C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]),
    ("dflo_oracle_normal_flux", None, [_dp, _dp, _dp]),
    ("dflo_oracle_flux_matrix", None, [_dp, _dp]),
    ("dflo_oracle_compute_Wminus", None, [C.c_int, _dp, _dp, _dp, _dp]),
    ("dflo_oracle_eigen", None, [_dp, _dp, _dp, _dp, _dp]),
    ("dflo_oracle_minmod", C.c_double, [C.c_double] * 4),
    ("dflo_oracle_erf", C.c_double, [C.c_double]),
    ("dflo_oracle_gauss", None, [C.c_int, _dp, _dp]),
    ("dflo_oracle_gauss_lobatto", None, [C.c_int, _dp, _dp]),
]:
    f = getattr(_lib, name)
    f.restype = res
    f.argtypes = args


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        self.code = code
        super().__init__("oracle error %d: %s" % (code, msg))


class Oracle:
    """CPU restatement of ConservationLaw's explicit path driven with the same mesh / parameters."""

    def __init__(self, mesh, parameters, threads=1):
        self.mesh = mesh
        p = parameters.struct()
        self._p = p
        self._h = _lib.dflo_oracle_create(C.cast(mesh._ptr, C.c_void_p), C.cast(C.byref(p), C.c_void_p))
        err = _lib.dflo_oracle_error(self._h)
        if err:
            raise OracleError(err, _lib.dflo_oracle_message(self._h).decode())
        _lib.dflo_oracle_set_threads(self._h, threads)
        self.n_dofs = _lib.dflo_oracle_n_dofs(self._h)
        self.ndof = _lib.dflo_oracle_dofs_per_cell(self._h)
        self.n_rk = _lib.dflo_oracle_n_rk(self._h)
        self.N = mesh.degree + 1

    def __del__(self):
        try:
            if self._h:
                _lib.dflo_oracle_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_solution(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        _lib.dflo_oracle_set_solution(self._h, _d(u))

    def set_current_only(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        _lib.dflo_oracle_set_current_only(self._h, _d(u))

    def get_solution(self):
        u = np.empty(self.n_dofs)
        _lib.dflo_oracle_get_solution(self._h, _d(u))
        return u

    def get_cell_average(self):
        a = np.empty((self.mesh.n_cells, 4))
        _lib.dflo_oracle_get_cell_average(self._h, _d(a))
        return a

    def set_cell_average(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        _lib.dflo_oracle_set_cell_average(self._h, _d(a))

    def inv_mass(self):
        m = np.empty(self.n_dofs)
        _lib.dflo_oracle_get_inv_mass(self._h, _d(m))
        return m

    def boundary_faces(self):
        n = _lib.dflo_oracle_n_boundary_faces(self._h)
        cell = np.zeros(max(n, 1), dtype=np.int32)
        face = np.zeros(max(n, 1), dtype=np.int32)
        bid = np.zeros(max(n, 1), dtype=np.int32)
        xy = np.zeros((max(n, 1), self.N, 2))
        _lib.dflo_oracle_boundary_faces(self._h, cell.ctypes.data_as(_ip), face.ctypes.data_as(_ip),
                                        bid.ctypes.data_as(_ip), _d(xy))
        return cell[:n], face[:n], bid[:n], xy[:n]

    def set_boundary_values(self, which, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        _lib.dflo_oracle_set_boundary_values(self._h, which, _d(v))

    def support_points(self):
        xy = np.empty((self.mesh.n_cells, self.ndof // 4, 2))
        _lib.dflo_oracle_support_points(self._h, _d(xy))
        return xy

    def cell_quadrature(self):
        nq = self.N * self.N
        xy = np.empty((self.mesh.n_cells, nq, 2))
        jxw = np.empty((self.mesh.n_cells, nq))
        _lib.dflo_oracle_cell_quadrature(self._h, _d(xy), _d(jxw))
        return xy, jxw

    def cell_shape(self):
        v = np.empty((self.ndof // 4, self.N * self.N))
        _lib.dflo_oracle_cell_shape(self._h, _d(v))
        return v

    def assemble(self, which=0):
        r = np.empty(self.n_dofs)
        _lib.dflo_oracle_assemble(self._h, which, _d(r))
        return r

    def compute_cell_average(self):
        _lib.dflo_oracle_compute_cell_average(self._h)

    @property
    def shock_indicator(self):
        s = np.empty(self.mesh.n_cells)
        _lib.dflo_oracle_get_shock_indicator(self._h, _d(s))
        return s

    def compute_shock_indicator(self):
        s = np.empty(self.mesh.n_cells)
        _lib.dflo_oracle_compute_shock_indicator(self._h, _d(s))
        return s

    def compute_time_step(self, elapsed):
        return _lib.dflo_oracle_compute_time_step(self._h, elapsed)

    def apply_limiter(self):
        _lib.dflo_oracle_apply_limiter(self._h)

    def apply_positivity_limiter(self):
        rc = _lib.dflo_oracle_apply_positivity_limiter(self._h)
        if rc:
            raise OracleError(rc, _lib.dflo_oracle_message(self._h).decode())

    def set_dt(self, dt):
        _lib.dflo_oracle_set_dt(self._h, dt)

    def stage(self, rk):
        r = C.c_double()
        rc = _lib.dflo_oracle_stage(self._h, rk, C.byref(r))
        if rc:
            raise OracleError(rc, _lib.dflo_oracle_message(self._h).decode())
        return r.value

    def end_step(self):
        _lib.dflo_oracle_end_step(self._h)

    def step(self, dt):
        r0, r1 = C.c_double(), C.c_double()
        rc = _lib.dflo_oracle_step(self._h, dt, C.byref(r0), C.byref(r1))
        if rc:
            raise OracleError(rc, _lib.dflo_oracle_message(self._h).decode())
        return r0.value, r1.value

    # ---- the optimised CPU twin (not the parity oracle; see oracle/dflo_oracle.cc)
    @property
    def twin_supported(self):
        return bool(_lib.dflo_oracle_twin_supported(self._h))

    def twin_advance(self, n_steps, dt=-1.0):
        """n_steps fused steps; dt < 0: CFL step from the cell averages.  Returns (elapsed time, last residual norm)."""
        t, r = C.c_double(0.0), C.c_double(0.0)
        rc = _lib.dflo_oracle_twin_advance(self._h, n_steps, dt, C.byref(t), C.byref(r))
        if rc:
            raise OracleError(rc, "the optimised twin covers axis-aligned Qk cells without limiters only")
        return t.value, r.value


# ---- pointwise functions
def numerical_flux(flux_type, n, Wp, Wm, Ap=None, Am=None):
    n = np.ascontiguousarray(n, dtype=np.float64)
    Wp = np.ascontiguousarray(Wp, dtype=np.float64)
    Wm = np.ascontiguousarray(Wm, dtype=np.float64)
    Ap = Wp if Ap is None else np.ascontiguousarray(Ap, dtype=np.float64)
    Am = Wm if Am is None else np.ascontiguousarray(Am, dtype=np.float64)
    F = np.empty(4)
    _lib.dflo_oracle_numerical_flux(flux_type, _d(n), _d(Wp), _d(Wm), _d(Ap), _d(Am), _d(F))
    return F


def normal_flux(W, n):
    W = np.ascontiguousarray(W, dtype=np.float64)
    n = np.ascontiguousarray(n, dtype=np.float64)
    F = np.empty(4)
    _lib.dflo_oracle_normal_flux(_d(W), _d(n), _d(F))
    return F


def flux_matrix(W):
    W = np.ascontiguousarray(W, dtype=np.float64)
    F = np.empty((4, 2))
    _lib.dflo_oracle_flux_matrix(_d(W), _d(F))
    return F


def compute_Wminus(kind, n, Wp, bv):
    n = np.ascontiguousarray(n, dtype=np.float64)
    Wp = np.ascontiguousarray(Wp, dtype=np.float64)
    bv = np.ascontiguousarray(bv, dtype=np.float64)
    Wm = np.empty(4)
    _lib.dflo_oracle_compute_Wminus(kind, _d(n), _d(Wp), _d(bv), _d(Wm))
    return Wm


def eigen(W):
    W = np.ascontiguousarray(W, dtype=np.float64)
    Rx, Lx, Ry, Ly = (np.empty((4, 4)) for _ in range(4))
    _lib.dflo_oracle_eigen(_d(W), _d(Rx), _d(Lx), _d(Ry), _d(Ly))
    return Rx, Lx, Ry, Ly


def minmod(a, b, c, Mdx2):
    return _lib.dflo_oracle_minmod(a, b, c, Mdx2)


def erf(x):
    return _lib.dflo_oracle_erf(x)


def gauss(n):
    x, w = np.empty(n), np.empty(n)
    _lib.dflo_oracle_gauss(n, _d(x), _d(w))
    return x, w


def gauss_lobatto(n):
    x, w = np.empty(n), np.empty(n)
    _lib.dflo_oracle_gauss_lobatto(n, _d(x), _d(w))
    return x, w
