"""Derivation of tests/golden/residual_fixture.json: known answers for the ASSEMBLED path -- right-hand side, cell
averages, CFL time step and the state after one SSP-RK step -- on small Cartesian meshes, computed here in 60-digit decimal
arithmetic from the weak form itself (i.e. src/assemble_explicit.cc:30-427, src/claw.cc:141-159,
444-511, 562-597, 694-760 in mathematical form), independently of oracle/ and of the HIP kernels: its own Gauss rule
(Newton on the Legendre polynomial), its own Lagrange basis, a loop over cells and faces with the numerical fluxes of
make_closed_forms.py (the 60-digit restatement of src/equation.h).  The oracle (tests/test_oracle_assembly.py) and the
device (tests/test_gpu_golden.py) are both held to it.

  R_i = int_K F(W):grad(phi_i) - sum_faces int_f F^(W+, W-, n) phi_i,     M_ii = w_a w_b h^2   (collocated Gauss nodes)
  stage: U <- ark U_n + (1 - ark)(U + dt R / M),  ark = (0, 1/2) for k = 1 and (0, 3/4, 1/3) for k >= 2
  dt = cfl / sum_d((c + |u_d|) / h) / (2k + 1) from the cell averages, minimum over cells

Like closed_forms.json this pins formulas, not the reference binary (deal.II is not available): parity stays "partial".
Usage: python tests/golden/make_residual_fixture.py   (rewrites residual_fixture.json; deterministic, ~1 min)
"""
import json
import math
import os
import sys
from decimal import Decimal as D, getcontext

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_closed_forms as cf   # noqa: E402  (sets the precision to 60 digits)

getcontext().prec = 60
G = cf.G


def legendre(n, x):
    p0, p1 = D(1), x
    if n == 0:
        return D(1), D(0)
    for k in range(2, n + 1):
        p0, p1 = p1, ((2 * k - 1) * x * p1 - (k - 1) * p0) / k
    return p1, n * (x * p1 - p0) / (x * x - 1)


def gauss01(n):
    """nodes and weights of Gauss-Legendre on [0, 1], ascending (QGauss<1>(n))"""
    xs, ws = [], []
    for i in range(n):
        x = D(repr(-math.cos(math.pi * (i + 0.75) / (n + 0.5))))    # a crude guess in floats; Newton does the rest
        for _ in range(100):
            p, dp = legendre(n, x)
            dx = p / dp
            x -= dx
            if abs(dx) < D("1e-58"):
                break
        p, dp = legendre(n, x)
        xs.append((1 + x) / 2)
        ws.append(1 / ((1 - x * x) * dp * dp))
    return xs, ws


def lagrange(xs, a, t):
    v = D(1)
    for m, xm in enumerate(xs):
        if m != a:
            v *= (t - xm) / (xs[a] - xm)
    return v


def dlagrange(xs, a, t):
    s = D(0)
    for j, xj in enumerate(xs):
        if j == a:
            continue
        v = 1 / (xs[a] - xj)
        for m, xm in enumerate(xs):
            if m != a and m != j:
                v *= (t - xm) / (xs[a] - xm)
        s += v
    return s


def flux_xy(W):
    rho, u, v, p = cf.prim(W)
    return ([W[0] * u + p, W[1] * u, rho * u, (W[3] + p) * u], [W[0] * v, W[1] * v + p, rho * v, (W[3] + p) * v])


def numerical_flux(name, n, Wl, Wr, Al, Ar):
    if name == "lxf":
        return cf.lxf(n, Wl, Wr, Al, Ar)
    if name == "sw":
        return cf.steger_warming(n, Wl, Wr)
    if name == "kfvs":
        return cf.kfvs(n, Wl, Wr)
    if name == "roe":
        return cf.roe(n, Wl, Wr)[0]
    return cf.hllc(n, Wl, Wr)[0]


class Case:
    """nx x ny squares of size h from (x0, y0); side = boundary id of the sides x-min, x-max, y-min, y-max or -1 (periodic)"""

    def __init__(self, name, nx, ny, h, k, flux, side, kinds, cfl, field, x0=D(0), y0=D(0)):
        self.name, self.nx, self.ny, self.h, self.k, self.N = name, nx, ny, D(h), k, k + 1
        self.flux, self.side, self.kinds, self.cfl, self.field = flux, side, kinds, D(cfl), field
        self.x0, self.y0 = D(x0), D(y0)
        self.xs, self.ws = gauss01(self.N)
        N = self.N
        self.Dm = [[dlagrange(self.xs, a, self.xs[q]) for a in range(N)] for q in range(N)]   # D[q][a] = l_a'(x_q)
        self.L0 = [lagrange(self.xs, a, D(0)) for a in range(N)]
        self.L1 = [lagrange(self.xs, a, D(1)) for a in range(N)]

    # state U[cell][comp][node], node = a + N b (x fastest) -- dflo's DoF order
    def initial(self):
        N, U = self.N, []
        for j in range(self.ny):
            for i in range(self.nx):
                cell = [[None] * (N * N) for _ in range(4)]
                for b in range(N):
                    for a in range(N):
                        W = self.field(self.x0 + (i + self.xs[a]) * self.h, self.y0 + (j + self.xs[b]) * self.h)
                        for c in range(4):
                            cell[c][a + N * b] = W[c]
                U.append(cell)
        return U

    def averages(self, U):
        N = self.N
        return [[sum(self.ws[a] * self.ws[b] * U[c][comp][a + N * b] for a in range(N) for b in range(N)) for comp in range(4)]
                for c in range(len(U))]

    def neighbour(self, i, j, f):
        """(cell index, None) across face f, or (None, boundary id)"""
        di, dj = [(-1, 0), (1, 0), (0, -1), (0, 1)][f]
        ii, jj = i + di, j + dj
        if 0 <= ii < self.nx and 0 <= jj < self.ny:
            return ii + self.nx * jj, None
        s = self.side[f]
        if s < 0:
            return (ii % self.nx) + self.nx * (jj % self.ny), None
        return None, s

    def trace(self, Uc, f, q):
        N = self.N
        if f < 2:
            L = self.L0 if f == 0 else self.L1
            return [sum(L[m] * Uc[c][m + N * q] for m in range(N)) for c in range(4)]
        L = self.L0 if f == 2 else self.L1
        return [sum(L[m] * Uc[c][q + N * m] for m in range(N)) for c in range(4)]

    def face_point(self, i, j, f, q):
        s = self.xs[q]
        xi = D(0) if f == 0 else (D(1) if f == 1 else s)
        eta = D(0) if f == 2 else (D(1) if f == 3 else s)
        return self.x0 + (i + xi) * self.h, self.y0 + (j + eta) * self.h

    def residual(self, U, bc_time=None):
        N, h, A = self.N, self.h, self.averages(U)
        R = [[[D(0)] * (N * N) for _ in range(4)] for _ in U]
        normals = [[D(-1), D(0)], [D(1), D(0)], [D(0), D(-1)], [D(0), D(1)]]
        for j in range(self.ny):
            for i in range(self.nx):
                c = i + self.nx * j
                Uc = U[c]
                # volume term: sum_q w_q w_b h D[q][a] Fx(W_(q,b)) + w_a w_q h D[q][b] Fy(W_(a,q))
                F = [[flux_xy([Uc[comp][a + N * b] for comp in range(4)]) for a in range(N)] for b in range(N)]   # F[b][a] = (Fx, Fy)
                for b in range(N):
                    for a in range(N):
                        for comp in range(4):
                            s = D(0)
                            for q in range(N):
                                s += self.ws[q] * self.ws[b] * h * self.Dm[q][a] * F[b][q][0][comp]
                                s += self.ws[a] * self.ws[q] * h * self.Dm[q][b] * F[q][a][1][comp]
                            R[c][comp][a + N * b] += s
                        g = getattr(self, "gravity", D(0))   # forcing (src/equation.h:831-850): (0, -rho, 0, -my) g in (mx, my, rho, E)
                        if g != 0:
                            jxw = self.ws[a] * self.ws[b] * h * h
                            R[c][1][a + N * b] += g * (-Uc[2][a + N * b]) * jxw
                            R[c][3][a + N * b] += g * (-Uc[1][a + N * b]) * jxw
                # faces: every cell subtracts the flux through its own faces with its own outward normal (the flux is
                # conservative: F^(W+, W-, n) = -F^(W-, W+, -n) for every scheme of the reference, so this is the one-flux-per-face
                # assembly of MeshWorker written per cell)
                for f in range(4):
                    nb, bid = self.neighbour(i, j, f)
                    n = normals[f]
                    for q in range(N):
                        Wp = self.trace(Uc, f, q)
                        if nb is not None:
                            Wm = self.trace(U[nb], f ^ 1, q)
                            Fh = numerical_flux(self.flux, n, Wp, Wm, A[c], A[nb])
                        else:
                            x, y = self.face_point(i, j, f, q)
                            bv = self.field_t(x, y, bc_time) if getattr(self, "field_t", None) else self.field(x, y)
                            Wm = cf.compute_Wminus(self.kinds[bid], n, Wp, bv)
                            Fh = numerical_flux(self.flux, n, Wp, Wm, A[c], A[c])   # both averages the interior cell's, src/assemble_explicit.cc:200-205
                        L = (self.L0 if f in (0, 2) else self.L1)
                        for m in range(N):
                            node = (m + N * q) if f < 2 else (q + N * m)
                            for comp in range(4):
                                R[c][comp][node] -= Fh[comp] * L[m] * self.ws[q] * h
        return R

    def dt(self, U):
        best = None
        for Ac in self.averages(U):
            rho, u, v, p = cf.prim(Ac)
            c = (G * p / rho).sqrt()
            d = self.cfl / ((c + abs(u)) / self.h + (c + abs(v)) / self.h) / (2 * self.k + 1)
            best = d if best is None or d < best else best
        return best

    def cell_dts(self, U):
        """time step type = local: dt(c) of compute_time_step_cartesian (src/claw.cc:486-511), kept per cell"""
        out = []
        for Ac in self.averages(U):
            rho, u, v, p = cf.prim(Ac)
            c = (G * p / rho).sqrt()
            out.append(self.cfl / ((c + abs(u)) / self.h + (c + abs(v)) / self.h) / (2 * self.k + 1))
        return out

    def step(self, U, dt):
        """dt: one number, or a list with the time step of every cell (local time stepping)"""
        N = self.N
        ark = [D(0), D(1) / 2] if self.k == 1 else ([D(0)] if self.k == 0 else [D(0), D(3) / 4, D(1) / 3])
        Un, Uc = U, U
        dts = dt if isinstance(dt, list) else [dt] * len(U)
        for rk, a_rk in enumerate(ark):
            # boundary functions at t in the first stage, at t + dt in the later ones (src/claw.cc:733-745)
            R = self.residual(Uc, bc_time=D(0) if rk == 0 else min(dts))
            new = []
            for c in range(len(Uc)):
                dt = dts[c]
                cell = [[None] * (N * N) for _ in range(4)]
                for comp in range(4):
                    for b in range(N):
                        for a in range(N):
                            i = a + N * b
                            u = Uc[c][comp][i] + dt * R[c][comp][i] / (self.ws[a] * self.ws[b] * self.h * self.h)
                            cell[comp][i] = a_rk * Un[c][comp][i] + (1 - a_rk) * u
                new.append(cell)
            Uc = new
        return Uc


# ---------------------------------------------------------------- limiters
def minmod(a, b, c, Mdx2):
    """src/limiter.cc:15-30"""
    if abs(a) < Mdx2:
        return a
    if a * b > 0 and b * c > 0:
        return (1 if a > 0 else -1) * min(abs(a), abs(b), abs(c))
    return D(0)


def eigen_matrices(A):
    """compute_eigen_matrix (src/equation.h:226-263) at the cell average; rows/columns in the order (rho, mx, my, E)"""
    g1 = G - 1
    rho, u, v, p = cf.prim(A)
    q2, c2 = u * u + v * v, G * p / rho
    c, beta, phi2, h = c2.sqrt(), 1 / (2 * c2), g1 * (u * u + v * v) / 2, c2 / g1 + (u * u + v * v) / 2
    Rx = [[1, 0, 1, 1], [u, 0, u + c, u - c], [v, -1, v, v], [q2 / 2, -v, h + c * u, h - c * u]]
    Ry = [[1, 0, 1, 1], [u, 1, u, u], [v, 0, v + c, v - c], [q2 / 2, u, h + c * v, h - c * v]]
    Lx = [[1 - phi2 / c2, g1 * u / c2, g1 * v / c2, -g1 / c2], [v, 0, -1, 0],
          [beta * (phi2 - c * u), beta * (c - g1 * u), -beta * g1 * v, beta * g1],
          [beta * (phi2 + c * u), -beta * (c + g1 * u), -beta * g1 * v, beta * g1]]
    Ly = [[1 - phi2 / c2, g1 * u / c2, g1 * v / c2, -g1 / c2], [-u, 1, 0, 0],
          [beta * (phi2 - c * v), -beta * g1 * u, beta * (c - g1 * v), beta * g1],
          [beta * (phi2 + c * v), -beta * g1 * u, -beta * (c + g1 * v), beta * g1]]
    return Rx, Lx, Ry, Ly


def to_char(L, W):      # W in dflo's order (mx, my, rho, E) -> characteristic variables, src/equation.h:268-285
    V = [W[2], W[0], W[1], W[3]]
    return [sum(D(L[i][j]) * V[j] for j in range(4)) for i in range(4)]


def to_con(R, W):       # characteristic -> (mx, my, rho, E), src/equation.h:290-305
    V = [sum(D(R[i][j]) * W[j] for j in range(4)) for i in range(4)]
    return [V[1], V[2], V[0], V[3]]


def tvb_qk(cs, U, M, beta, char_lim, marked=None):
    """apply_limiter_TVB_Qk (src/limiter.cc:225-370) on squares; marked: the cells the shock indicator offers to the limiter
    (indicator > 1, :263), None = every cell (shock indicator = limiter)"""
    N, h = cs.N, cs.h
    A = cs.averages(U)
    out = [[list(comp) for comp in cell] for cell in U]
    for j in range(cs.ny):
        for i in range(cs.nx):
            c = i + cs.nx * j
            if marked is not None and not marked[c]:
                continue
            dx, Mdx2 = h, D(M) * h * h
            # dx * average gradient: (1/|K|) sum_q grad u(x_q) JxW_q with the Gauss rule of the element
            Dx, Dy = [], []
            for comp in range(4):
                gx = sum(cs.ws[a] * cs.ws[b] * sum(cs.Dm[a][m] * U[c][comp][m + N * b] for m in range(N)) for a in range(N) for b in range(N)) / h
                gy = sum(cs.ws[a] * cs.ws[b] * sum(cs.Dm[b][m] * U[c][comp][a + N * m] for m in range(N)) for a in range(N) for b in range(N)) / h
                Dx.append(dx * gx)
                Dy.append(dx * gy)

            def diff(f, own):     # backward / forward difference of the cell averages, or the own slope at a boundary
                nb, bid = cs.neighbour(i, j, f)
                if nb is None:
                    return list(own)
                return [A[c][k] - A[nb][k] for k in range(4)] if f in (0, 2) else [A[nb][k] - A[c][k] for k in range(4)]
            dbx, dfx, dby, dfy = diff(0, Dx), diff(1, Dx), diff(2, Dy), diff(3, Dy)
            if char_lim:
                Rx, Lx, Ry, Ly = eigen_matrices(A[c])
                dbx, dfx, Dx = to_char(Lx, dbx), to_char(Lx, dfx), to_char(Lx, Dx)
                dby, dfy, Dy = to_char(Ly, dby), to_char(Ly, dfy), to_char(Ly, Dy)
            Dxn = [minmod(Dx[k], D(beta) * dbx[k], D(beta) * dfx[k], Mdx2) for k in range(4)]
            Dyn = [minmod(Dy[k], D(beta) * dby[k], D(beta) * dfy[k], Mdx2) for k in range(4)]
            change = sum(abs(Dxn[k] - Dx[k]) for k in range(4)) / 4 + sum(abs(Dyn[k] - Dy[k]) for k in range(4)) / 4
            if change > D("1e-10"):
                Dxn, Dyn = [v / dx for v in Dxn], [v / dx for v in Dyn]
                if char_lim:
                    Dxn, Dyn = to_con(Rx, Dxn), to_con(Ry, Dyn)
                for comp in range(4):
                    for b in range(N):
                        for a in range(N):
                            out[c][comp][a + N * b] = A[c][comp] + (cs.xs[a] - D(1) / 2) * h * Dxn[comp] + (cs.xs[b] - D(1) / 2) * h * Dyn[comp]
    return out


def positivity(cs, U):
    """apply_positivity_limiter (src/positivity.cc:17-208), Qk; returns (limited state, [theta1, theta2] per cell)"""
    N, eps = cs.N, D("1e-13")
    Ng = (cs.k + 3) // 2 if (cs.k + 3) % 2 == 0 else (cs.k + 4) // 2
    gll = {2: [D(0), D(1)], 3: [D(0), D(1) / 2, D(1)]}[Ng]
    A = cs.averages(U)
    out = [[list(comp) for comp in cell] for cell in U]
    thetas = []
    for c in range(len(U)):
        assert min(A[c][2], cf.prim(A[c])[3]) >= eps
        def values(comp, d, state):   # the component at GLL(Ng) x Gauss(N) (d = 0) or Gauss(N) x GLL(Ng) (d = 1)
            vals = []
            for l in range(N):
                for g in gll:
                    if d == 0:
                        vals.append(sum(lagrange(cs.xs, m, g) * state[c][comp][m + N * l] for m in range(N)))
                    else:
                        vals.append(sum(lagrange(cs.xs, m, g) * state[c][comp][l + N * m] for m in range(N)))
            return vals
        rho_min = min(values(2, 0, out) + values(2, 1, out))
        th1 = min(abs(A[c][2] - eps) / (abs(A[c][2] - rho_min) + D("1e-13")), D(1))
        if th1 < 1:
            out[c][2] = [th1 * v + (1 - th1) * A[c][2] for v in out[c][2]]
        th2 = D(1)
        for d in range(2):
            rho, mx, my, en = values(2, d, out), values(0, d, out), values(1, d, out), values(3, d, out)
            for q in range(len(rho)):
                pre = (G - 1) * (en[q] - (mx[q] * mx[q] + my[q] * my[q]) / (2 * rho[q]))
                if pre < eps:
                    drho, dmx, dmy, dE = rho[q] - A[c][2], mx[q] - A[c][0], my[q] - A[c][1], en[q] - A[c][3]
                    a1 = 2 * drho * dE - (dmx * dmx + dmy * dmy)
                    b1 = (2 * drho * (A[c][3] - eps / (G - 1)) + 2 * A[c][2] * dE - 2 * (A[c][0] * dmx + A[c][1] * dmy)) / a1
                    c1 = (2 * A[c][2] * A[c][3] - (A[c][0] ** 2 + A[c][1] ** 2) - 2 * eps * A[c][2] / (G - 1)) / a1
                    Dq = abs(b1 * b1 - 4 * c1).sqrt()
                    t1, t2 = (-b1 - Dq) / 2, (-b1 + Dq) / 2
                    lo, hi = D("-1e-12"), 1 + D("1e-12")
                    t = t1 if lo < t1 < hi else t2
                    assert lo < t < hi
                    t = max(D(0), min(D(1), t))
                    if abs(1 - t) < D("1e-14"):
                        t = D(0)
                    th2 = min(th2, t)
        if th2 < 1:
            out[c] = [[th2 * v + (1 - th2) * A[c][comp] for v in out[c][comp]] for comp in range(4)]
        thetas.append([th1, th2])
    return out, thetas


def jump(x, y):
    """a Sod-like jump across an oblique line, smooth variation on either side"""
    left = x + y / 4 < D("0.55")
    rho = (1 + y / 5) if left else (D("0.125") + x / 10)
    p = (1 + x / 7) if left else (D("0.1") + y / 20)
    return cf.cons(rho, D("0.1") + y / 6, D("-0.05") + x / 8, p)


# ---------------------------------------------------------------- bilinear (MappingQ1) cells, SURVEY A.3
class BilinearCase(Case):
    """the same logical nx x ny grid with its interior vertices moved: every cell is a general quadrilateral.  Metric terms,
    normals, face and cell JxW from the bilinear map; lumped mass M_j = w_j |J_j| (src/claw.cc:223-227); time step from
    compute_time_step_q (src/claw.cc:520-557)."""

    def __init__(self, name, nx, ny, h, k, flux, side, kinds, cfl, field, shift):
        super().__init__(name, nx, ny, h, k, flux, side, kinds, cfl, field)
        self.V = {}
        for j in range(ny + 1):
            for i in range(nx + 1):
                sx, sy = shift(i, j) if (0 < i < nx and 0 < j < ny) else (D(0), D(0))
                self.V[(i, j)] = (i * self.h + sx * self.h, j * self.h + sy * self.h)

    def corners(self, i, j):   # deal.II order v0 v1 v2 v3
        return [self.V[(i, j)], self.V[(i + 1, j)], self.V[(i, j + 1)], self.V[(i + 1, j + 1)]]

    def xmap(self, i, j, xi, eta):
        v = self.corners(i, j)
        w = [(1 - xi) * (1 - eta), xi * (1 - eta), (1 - xi) * eta, xi * eta]
        return sum(w[k] * v[k][0] for k in range(4)), sum(w[k] * v[k][1] for k in range(4))

    def jac(self, i, j, xi, eta):
        v = self.corners(i, j)
        xxi = (1 - eta) * (v[1][0] - v[0][0]) + eta * (v[3][0] - v[2][0])
        yxi = (1 - eta) * (v[1][1] - v[0][1]) + eta * (v[3][1] - v[2][1])
        xet = (1 - xi) * (v[2][0] - v[0][0]) + xi * (v[3][0] - v[1][0])
        yet = (1 - xi) * (v[2][1] - v[0][1]) + xi * (v[3][1] - v[1][1])
        return xxi, yxi, xet, yet

    def initial(self):
        N, U = self.N, []
        for j in range(self.ny):
            for i in range(self.nx):
                cell = [[None] * (N * N) for _ in range(4)]
                for b in range(N):
                    for a in range(N):
                        W = self.field(*self.xmap(i, j, self.xs[a], self.xs[b]))
                        for c in range(4):
                            cell[c][a + N * b] = W[c]
                U.append(cell)
        return U

    def jxw(self, i, j, a, b):
        xxi, yxi, xet, yet = self.jac(i, j, self.xs[a], self.xs[b])
        return self.ws[a] * self.ws[b] * (xxi * yet - xet * yxi)

    def averages(self, U):
        N, out = self.N, []
        for j in range(self.ny):
            for i in range(self.nx):
                c = i + self.nx * j
                area = sum(self.jxw(i, j, a, b) for a in range(N) for b in range(N))
                out.append([sum(self.jxw(i, j, a, b) * U[c][comp][a + N * b] for a in range(N) for b in range(N)) / area for comp in range(4)])
        return out

    def face_geometry(self, i, j, f, q):
        """(outward unit normal, length element |dx/dt|) at face point q"""
        s = self.xs[q]
        xi = D(0) if f == 0 else (D(1) if f == 1 else s)
        eta = D(0) if f == 2 else (D(1) if f == 3 else s)
        xxi, yxi, xet, yet = self.jac(i, j, xi, eta)
        if f < 2:
            tx, ty = xet, yet
            n = (yet, -xet)          # the tangent turned clockwise points towards +xi
            sign = 1 if f == 1 else -1
        else:
            tx, ty = xxi, yxi
            n = (-yxi, xxi)          # the tangent turned counter-clockwise points towards +eta
            sign = 1 if f == 3 else -1
        ln = (tx * tx + ty * ty).sqrt()
        return [sign * n[0] / ln, sign * n[1] / ln], ln

    def face_point(self, i, j, f, q):
        s = self.xs[q]
        xi = D(0) if f == 0 else (D(1) if f == 1 else s)
        eta = D(0) if f == 2 else (D(1) if f == 3 else s)
        return self.xmap(i, j, xi, eta)

    def residual(self, U, bc_time=None):
        N, A = self.N, self.averages(U)
        R = [[[D(0)] * (N * N) for _ in range(4)] for _ in U]
        for j in range(self.ny):
            for i in range(self.nx):
                c = i + self.nx * j
                Uc = U[c]
                # volume: sum_q w_q [ (y_eta Fx - x_eta Fy) dphi/dxi + (-y_xi Fx + x_xi Fy) dphi/deta ]   (|J| cancels)
                for b in range(N):
                    for a in range(N):
                        for comp in range(4):
                            sacc = D(0)
                            for q in range(N):
                                xxi, yxi, xet, yet = self.jac(i, j, self.xs[q], self.xs[b])      # node (q, b): dphi_(a,b)/dxi = D[q][a]
                                Fx, Fy = flux_xy([Uc[cc][q + N * b] for cc in range(4)])
                                sacc += self.ws[q] * self.ws[b] * (yet * Fx[comp] - xet * Fy[comp]) * self.Dm[q][a]
                                xxi, yxi, xet, yet = self.jac(i, j, self.xs[a], self.xs[q])      # node (a, q): dphi_(a,b)/deta = D[q][b]
                                Fx, Fy = flux_xy([Uc[cc][a + N * q] for cc in range(4)])
                                sacc += self.ws[a] * self.ws[q] * (-yxi * Fx[comp] + xxi * Fy[comp]) * self.Dm[q][b]
                            R[c][comp][a + N * b] += sacc
                for f in range(4):
                    nb, bid = self.neighbour(i, j, f)
                    for q in range(N):
                        n, ln = self.face_geometry(i, j, f, q)
                        Wp = self.trace(Uc, f, q)
                        if nb is not None:
                            Wm = self.trace(U[nb], f ^ 1, q)
                            Fh = numerical_flux(self.flux, n, Wp, Wm, A[c], A[nb])
                        else:
                            bv = self.field(*self.face_point(i, j, f, q))
                            Wm = cf.compute_Wminus(self.kinds[bid], n, Wp, bv)
                            Fh = numerical_flux(self.flux, n, Wp, Wm, A[c], A[c])
                        L = (self.L0 if f in (0, 2) else self.L1)
                        for m in range(N):
                            node = (m + N * q) if f < 2 else (q + N * m)
                            for comp in range(4):
                                R[c][comp][node] -= Fh[comp] * L[m] * self.ws[q] * ln
        return R

    def dt(self, U):
        """compute_time_step_q: max of |v| + c over the 4 x 4 equispaced points, h = longest diagonal / sqrt(2)"""
        N, best = self.N, None
        for j in range(self.ny):
            for i in range(self.nx):
                c = i + self.nx * j
                v = self.corners(i, j)
                d1 = ((v[3][0] - v[0][0]) ** 2 + (v[3][1] - v[0][1]) ** 2).sqrt()
                d2 = ((v[2][0] - v[1][0]) ** 2 + (v[2][1] - v[1][1]) ** 2).sqrt()
                hh = max(d1, d2) / D(2).sqrt()
                lam = D(0)
                for pb in range(4):
                    for pa in range(4):
                        xi, eta = D(pa) / 3, D(pb) / 3
                        W = [sum(lagrange(self.xs, a, xi) * lagrange(self.xs, b, eta) * U[c][comp][a + N * b] for a in range(N) for b in range(N))
                             for comp in range(4)]
                        rho, uu, vv, p = cf.prim(W)
                        lam = max(lam, (uu * uu + vv * vv).sqrt() + (G * p / rho).sqrt())
                d = self.cfl * hh / lam / (2 * self.k + 1)
                best = d if best is None or d < best else best
        return best

    def step(self, U, dt):
        N = self.N
        ark = [D(0), D(1) / 2] if self.k == 1 else [D(0), D(3) / 4, D(1) / 3]
        Un, Uc = U, U
        for a_rk in ark:
            R = self.residual(Uc)
            new = []
            for j in range(self.ny):
                for i in range(self.nx):
                    c = i + self.nx * j
                    cell = [[None] * (N * N) for _ in range(4)]
                    for comp in range(4):
                        for b in range(N):
                            for a in range(N):
                                k = a + N * b
                                u = Uc[c][comp][k] + dt * R[c][comp][k] / self.jxw(i, j, a, b)
                                cell[comp][k] = a_rk * Un[c][comp][k] + (1 - a_rk) * u
                    new.append(cell)
            Uc = new
        return Uc


# ---------------------------------------------------------------- Pk (FE_DGP) basis, SURVEY A.10
def pt_legendre(n, x):
    """orthonormal Legendre polynomial on [0, 1] and its derivative: sqrt(2n+1) P_n(2x-1)"""
    t = 2 * x - 1
    if n == 0:
        return D(1), D(0)
    p0, p1, d0, d1 = D(1), t, D(0), D(1)
    for k in range(2, n + 1):
        p0, p1, d0, d1 = p1, ((2 * k - 1) * t * p1 - (k - 1) * p0) / k, d1, ((2 * k - 1) * (p1 + t * d1) - (k - 1) * d0) / k
    s = D(2 * n + 1).sqrt()
    return s * p1, 2 * s * d1


class PkCase(Case):
    """modal DoFs U[cell][comp][m], psi_m = Pt_i(xi) Pt_j(eta), modes ordered "for j: for i <= k - j" (src/claw.cc:107-113);
    mass matrix |K| I; initial state by L2 projection with QGauss(k+1) (src/ic.cc:128-164)"""

    def __init__(self, *args):
        super().__init__(*args)
        self.modes = [(i, j) for j in range(self.N) for i in range(self.N - j)]

    def psi(self, m, xi, eta):
        i, j = self.modes[m]
        pi, dpi = pt_legendre(i, xi)
        pj, dpj = pt_legendre(j, eta)
        return pi * pj, dpi * pj, pi * dpj

    def value(self, Uc, xi, eta):
        return [sum(Uc[comp][m] * self.psi(m, xi, eta)[0] for m in range(len(self.modes))) for comp in range(4)]

    def initial(self):
        N, U = self.N, []
        for j in range(self.ny):
            for i in range(self.nx):
                cell = [[D(0)] * len(self.modes) for _ in range(4)]
                for b in range(N):
                    for a in range(N):
                        W = self.field(self.x0 + (i + self.xs[a]) * self.h, self.y0 + (j + self.xs[b]) * self.h)
                        for m in range(len(self.modes)):
                            ps = self.psi(m, self.xs[a], self.xs[b])[0]
                            for comp in range(4):
                                cell[comp][m] += W[comp] * ps * self.ws[a] * self.ws[b]
                U.append(cell)
        return U

    def averages(self, U):
        return [[U[c][comp][0] for comp in range(4)] for c in range(len(U))]

    def trace(self, Uc, f, q):
        s = self.xs[q]
        xi = D(0) if f == 0 else (D(1) if f == 1 else s)
        eta = D(0) if f == 2 else (D(1) if f == 3 else s)
        return self.value(Uc, xi, eta)

    def residual(self, U, bc_time=None):
        N, h, A, nm = self.N, self.h, self.averages(U), len(self.modes)
        R = [[[D(0)] * nm for _ in range(4)] for _ in U]
        normals = [[D(-1), D(0)], [D(1), D(0)], [D(0), D(-1)], [D(0), D(1)]]
        for j in range(self.ny):
            for i in range(self.nx):
                c = i + self.nx * j
                for b in range(N):
                    for a in range(N):
                        Fx, Fy = flux_xy(self.value(U[c], self.xs[a], self.xs[b]))
                        jxw = self.ws[a] * self.ws[b] * h * h
                        for m in range(nm):
                            _, px, py = self.psi(m, self.xs[a], self.xs[b])
                            for comp in range(4):
                                R[c][comp][m] += (Fx[comp] * px / h + Fy[comp] * py / h) * jxw
                for f in range(4):
                    nb, bid = self.neighbour(i, j, f)
                    n = normals[f]
                    for q in range(N):
                        Wp = self.trace(U[c], f, q)
                        if nb is not None:
                            Fh = numerical_flux(self.flux, n, Wp, self.trace(U[nb], f ^ 1, q), A[c], A[nb])
                        else:
                            Wm = cf.compute_Wminus(self.kinds[bid], n, Wp, self.field(*self.face_point(i, j, f, q)))
                            Fh = numerical_flux(self.flux, n, Wp, Wm, A[c], A[c])
                        s = self.xs[q]
                        xi = D(0) if f == 0 else (D(1) if f == 1 else s)
                        eta = D(0) if f == 2 else (D(1) if f == 3 else s)
                        for m in range(nm):
                            ps = self.psi(m, xi, eta)[0]
                            for comp in range(4):
                                R[c][comp][m] -= Fh[comp] * ps * self.ws[q] * h
        return R

    def step(self, U, dt):
        ark = [D(0), D(1) / 2] if self.k == 1 else [D(0), D(3) / 4, D(1) / 3]
        Un, Uc = U, U
        for a_rk in ark:
            R = self.residual(Uc)
            Uc = [[[a_rk * Un[c][comp][m] + (1 - a_rk) * (Uc[c][comp][m] + dt * R[c][comp][m] / (self.h * self.h)) for m in range(len(self.modes))]
                   for comp in range(4)] for c in range(len(Uc))]
        return Uc


def tvb_pk(cs, U, M, beta, char_lim, conserve_ang_mom=False):
    """apply_limiter_TVB_Pk (src/limiter.cc:377-516), every cell marked: slopes from the modes (1,0) and (0,1) of the orthonormal
    basis (times sqrt 3), beta / 2, the other higher modes zeroed when a cell is limited"""
    N, h, sq3, half_beta = cs.N, cs.h, D(3).sqrt(), D(beta) / 2
    A = cs.averages(U)
    out = [[list(comp) for comp in cell] for cell in U]
    for j in range(cs.ny):
        for i in range(cs.nx):
            c = i + cs.nx * j
            Mdx2 = D(M) * h * h
            Dx = [U[c][comp][1] * sq3 for comp in range(4)]
            Dy = [U[c][comp][N] * sq3 for comp in range(4)]
            ang = Dx[1] - Dy[0]

            def diff(f, own):
                nb, bid = cs.neighbour(i, j, f)
                if nb is None:
                    return list(own)
                return [A[c][k] - A[nb][k] for k in range(4)] if f in (0, 2) else [A[nb][k] - A[c][k] for k in range(4)]
            dbx, dfx, dby, dfy = diff(0, Dx), diff(1, Dx), diff(2, Dy), diff(3, Dy)
            if char_lim:
                Rx, Lx, Ry, Ly = eigen_matrices(A[c])
                dbx, dfx, Dx = to_char(Lx, dbx), to_char(Lx, dfx), to_char(Lx, Dx)
                dby, dfy, Dy = to_char(Ly, dby), to_char(Ly, dfy), to_char(Ly, Dy)
            Dxn = [minmod(Dx[k], half_beta * dbx[k], half_beta * dfx[k], Mdx2) for k in range(4)]
            Dyn = [minmod(Dy[k], half_beta * dby[k], half_beta * dfy[k], Mdx2) for k in range(4)]
            change = sum(abs(Dxn[k] - Dx[k]) for k in range(4)) / 4 + sum(abs(Dyn[k] - Dy[k]) for k in range(4)) / 4
            if change > D("1e-10"):
                if char_lim:
                    Dxn, Dyn = to_con(Rx, Dxn), to_con(Ry, Dyn)
                if conserve_ang_mom:
                    Dyn[0] = (Dyn[0] - (ang - Dxn[1])) / 2
                    Dxn[1] = ang + Dyn[0]
                for comp in range(4):
                    for m in range(1, len(cs.modes)):
                        out[c][comp][m] = Dxn[comp] / sq3 if m == 1 else (Dyn[comp] / sq3 if m == N else D(0))
    return out


def positivity_pk(cs, U):
    """apply_positivity_limiter, Pk branch (src/positivity.cc:17-208 with :100-109, :197-205): the same points (Gauss-Lobatto x
    Gauss lines), the modal polynomial evaluated there, the modes >= 1 scaled"""
    N, eps = cs.N, D("1e-13")
    Ng = (cs.k + 3) // 2 if (cs.k + 3) % 2 == 0 else (cs.k + 4) // 2
    gll = {2: [D(0), D(1)], 3: [D(0), D(1) / 2, D(1)]}[Ng]
    A = cs.averages(U)
    out = [[list(comp) for comp in cell] for cell in U]
    thetas = []
    for c in range(len(U)):
        assert min(A[c][2], cf.prim(A[c])[3]) >= eps
        pts = [(g, cs.xs[l]) for l in range(N) for g in gll], [(cs.xs[l], g) for l in range(N) for g in gll]
        rho_min = min(cs.value(out[c], xi, eta)[2] for d in range(2) for xi, eta in pts[d])
        th1 = min(abs(A[c][2] - eps) / (abs(A[c][2] - rho_min) + D("1e-13")), D(1))
        if th1 < 1:
            out[c][2] = [out[c][2][0]] + [th1 * v for v in out[c][2][1:]]
        th2 = D(1)
        for d in range(2):
            for xi, eta in pts[d]:
                mx, my, rho, en = cs.value(out[c], xi, eta)
                pre = (G - 1) * (en - (mx * mx + my * my) / (2 * rho))
                if pre < eps:
                    drho, dmx, dmy, dE = rho - A[c][2], mx - A[c][0], my - A[c][1], en - A[c][3]
                    a1 = 2 * drho * dE - (dmx * dmx + dmy * dmy)
                    b1 = (2 * drho * (A[c][3] - eps / (G - 1)) + 2 * A[c][2] * dE - 2 * (A[c][0] * dmx + A[c][1] * dmy)) / a1
                    c1 = (2 * A[c][2] * A[c][3] - (A[c][0] ** 2 + A[c][1] ** 2) - 2 * eps * A[c][2] / (G - 1)) / a1
                    Dq = abs(b1 * b1 - 4 * c1).sqrt()
                    t1, t2 = (-b1 - Dq) / 2, (-b1 + Dq) / 2
                    lo, hi = D("-1e-12"), 1 + D("1e-12")
                    t = t1 if lo < t1 < hi else t2
                    assert lo < t < hi
                    t = max(D(0), min(D(1), t))
                    if abs(1 - t) < D("1e-14"):
                        t = D(0)
                    th2 = min(th2, t)
        if th2 < 1:
            out[c] = [[out[c][comp][0]] + [th2 * v for v in out[c][comp][1:]] for comp in range(4)]
        thetas.append([th1, th2])
    return out, thetas


def kxrcf(cs, U, component):
    """compute_shock_indicator_kxrcf (src/indicator.cc:51-198) on squares, Qk: jump of the indicator variable over the inflow
    part of the cell boundary (inflow judged by the cell-average velocity), boundary faces skipped"""
    N, h, A = cs.N, cs.h, cs.averages(U)
    out = []
    normals = [[D(-1), D(0)], [D(1), D(0)], [D(0), D(-1)], [D(0), D(1)]]
    for j in range(cs.ny):
        for i in range(cs.nx):
            c = i + cs.nx * j
            vel = [A[c][0] / A[c][2], A[c][1] / A[c][2]]
            ind, measure = D(0), D(0)
            for f in range(4):
                nb, bid = cs.neighbour(i, j, f)
                if nb is None:
                    continue
                inflow = 1 if vel[0] * normals[f][0] + vel[1] * normals[f][1] < 0 else 0
                for q in range(N):
                    jxw = cs.ws[q] * h
                    ind += inflow * (cs.trace(U[c], f, q)[component] - cs.trace(U[nb], f ^ 1, q)[component]) * jxw
                    measure += inflow * jxw
            diam = h * D(2).sqrt()
            den = (diam ** (D(cs.k + 1) / 2)) * measure * A[c][component]
            out.append(abs(ind) / den if den != 0 else None)
    return out


def smooth(x, y):
    """a smooth subsonic state with all gradients alive (polynomials: exact in decimal arithmetic)"""
    rho = 1 + (x * (1 - x) + y * y / 2) / 4
    u = D("0.4") + x * y / 3 - y / 5
    v = D("-0.25") + x / 4 + y * (1 - y) / 3
    p = 1 + (x - y) / 5 + x * x / 6
    return cf.cons(rho, u, v, p)


def periodic(x, y):
    """periodic on the unit square to rounding of the polynomials below (period 1 in x and y)"""
    sx, sy = x * (1 - x) * (1 - 2 * x), y * (1 - y) * (1 - 2 * y)     # zero mean, matching values and slopes at 0 and 1
    rho = 1 + sx / 2 + sy / 3
    u = D("0.5") + sy - sx / 2
    v = D("0.3") + sx
    p = 1 + sx * sy * 4 + sy / 2
    return cf.cons(rho, u, v, p)


def flat(U):
    return [format(v, ".25e") for cell in U for comp in cell for v in comp]


def main():
    third, quarter = D(1) / 3, D(1) / 4
    cases = [
        Case("4x3 periodic Q2 HLLC", 4, 3, quarter, 2, "hllc", [-1, -1, -1, -1], {}, "0.9", lambda x, y: periodic(x, y * 4 / 3), D(0), D(0)),
        Case("3x3 Q1 LxF, inflow / outflow / slip / farfield walls", 3, 3, third, 1, "lxf", [2, 1, 0, 3],
             {0: "slip", 1: "outflow", 2: "inflow", 3: "farfield"}, "0.8", smooth),
        Case("2x2 periodic Q3 KFVS", 2, 2, D(1) / 2, 3, "kfvs", [-1, -1, -1, -1], {}, "0.5", periodic),
        Case("3x2 Q2 Roe, pressure outlet and slip walls", 3, 2, third, 2, "roe", [2, 4, 0, 0],
             {0: "slip", 2: "inflow", 4: "pressure"}, "0.7", smooth),
        Case("4x4 periodic Q1 Steger-Warming", 4, 4, quarter, 1, "sw", [-1, -1, -1, -1], {}, "0.9", periodic),
    ]
    out = {"gamma": "1.4", "layout": "U[cell = i + nx j][component mx, my, rho, E][node a + N b]", "cases": []}
    for cs in cases:
        U0 = cs.initial()
        R = cs.residual(U0)
        dt = cs.dt(U0)
        U1 = cs.step(U0, dt)
        # the boundary values the tests hand to the oracle / the engine: the field at the face quadrature points, MeshWorker order
        bfaces = []
        for j in range(cs.ny):
            for i in range(cs.nx):
                for f in range(4):
                    nb, bid = cs.neighbour(i, j, f)
                    if nb is None:
                        bfaces.append({"cell": i + cs.nx * j, "face": f, "id": bid,
                                       "values": [[format(v, ".25e") for v in cs.field(*cs.face_point(i, j, f, q))] for q in range(cs.N)]})
        out["cases"].append({
            "name": cs.name, "nx": cs.nx, "ny": cs.ny, "h": format(cs.h, ".25e"), "degree": cs.k, "flux": cs.flux,
            "side": cs.side, "kinds": {str(k): v for k, v in cs.kinds.items()}, "cfl": str(cs.cfl),
            "U0": flat(U0), "residual": flat(R), "cell_average": [format(v, ".25e") for a in cs.averages(U0) for v in a],
            "dt": format(dt, ".25e"), "U1": flat(U1), "boundary_faces": bfaces,
        })
        print(cs.name, "dt", format(dt, ".6e"), flush=True)
    # ---- bilinear cells (the C5 path): interior vertices moved by up to a fifth of a cell
    out["bilinear_cases"] = []
    shift = lambda i, j: (D((7 * i + 3 * j) % 5 - 2) / 10, D((3 * i + 11 * j) % 5 - 2) / 10)
    for name, k, flux, side, kinds, cfl in [
            ("3x3 bilinear Q2 HLLC, inflow / outflow / slip / farfield walls", 2, "hllc", [2, 1, 0, 3], {0: "slip", 1: "outflow", 2: "inflow", 3: "farfield"}, "0.6"),
            ("3x3 bilinear Q3 KFVS, slip and outflow walls", 3, "kfvs", [2, 1, 0, 0], {0: "slip", 1: "outflow", 2: "inflow"}, "0.5"),
            ("4x3 bilinear Q1 LxF", 1, "lxf", [2, 1, 0, 3], {0: "slip", 1: "outflow", 2: "inflow", 3: "farfield"}, "0.7")]:
        nx = 4 if name.startswith("4x3") else 3
        cs = BilinearCase(name, nx, 3, D(1) / 3, k, flux, side, kinds, cfl, smooth, shift)
        U0 = cs.initial()
        R, dt = cs.residual(U0), cs.dt(U0)
        U1 = cs.step(U0, dt)
        bfaces = []
        for j in range(cs.ny):
            for i in range(cs.nx):
                for f in range(4):
                    nb, bid = cs.neighbour(i, j, f)
                    if nb is None:
                        bfaces.append({"cell": i + cs.nx * j, "face": f, "id": bid,
                                       "values": [[format(v, ".25e") for v in cs.field(*cs.face_point(i, j, f, q))] for q in range(cs.N)]})
        verts = [[format(cs.V[(i, j)][0], ".25e"), format(cs.V[(i, j)][1], ".25e")] for j in range(cs.ny + 1) for i in range(cs.nx + 1)]
        nodes = [[format(c2, ".25e") for c2 in cs.xmap(i, j, cs.xs[a], cs.xs[b])] for j in range(cs.ny) for i in range(cs.nx)
                 for b in range(cs.N) for a in range(cs.N)]
        out["bilinear_cases"].append({
            "name": name, "nx": cs.nx, "ny": cs.ny, "degree": k, "flux": flux, "side": side, "kinds": {str(a): b for a, b in kinds.items()},
            "cfl": cfl, "vertices": verts, "nodes": nodes, "U0": flat(U0), "residual": flat(R),
            "cell_average": [format(v, ".25e") for a in cs.averages(U0) for v in a], "dt": format(dt, ".25e"), "U1": flat(U1), "boundary_faces": bfaces})
        print(name, "dt", format(dt, ".6e"), flush=True)
    # ---- Pk basis (modal DoFs): projection, residual, time step, one step
    out["pk_cases"] = []
    for name, nx, ny, k, flux, side, kinds, cfl, field in [
            ("4x3 periodic P2 HLLC", 4, 3, 2, "hllc", [-1, -1, -1, -1], {}, "0.9", lambda x, y: periodic(x, y * 4 / 3)),
            ("3x3 P1 LxF with walls", 3, 3, 1, "lxf", [2, 1, 0, 3], {0: "slip", 1: "outflow", 2: "inflow", 3: "farfield"}, "0.8", smooth),
            ("2x2 periodic P3 Roe", 2, 2, 3, "roe", [-1, -1, -1, -1], {}, "0.5", periodic)]:
        cs = PkCase(name, nx, ny, D(1) / (4 if nx == 4 else nx), k, flux, side, kinds, cfl, field)
        U0 = cs.initial()
        R, dt = cs.residual(U0), cs.dt(U0)
        U1 = cs.step(U0, dt)
        bfaces = []
        for j in range(cs.ny):
            for i in range(cs.nx):
                for f in range(4):
                    nb, bid = cs.neighbour(i, j, f)
                    if nb is None:
                        bfaces.append({"cell": i + cs.nx * j, "face": f, "id": bid,
                                       "values": [[format(v, ".25e") for v in cs.field(*cs.face_point(i, j, f, q))] for q in range(cs.N)]})
        out["pk_cases"].append({"name": name, "nx": nx, "ny": ny, "h": format(cs.h, ".25e"), "degree": k, "flux": flux, "side": side,
                                "kinds": {str(a): b for a, b in kinds.items()}, "cfl": cfl, "U0": flat(U0), "residual": flat(R),
                                "cell_average": [format(v, ".25e") for a in cs.averages(U0) for v in a], "dt": format(dt, ".25e"),
                                "U1": flat(U1), "boundary_faces": bfaces})
        print(name, "dt", format(dt, ".6e"), flush=True)
    # ---- KXRCF troubled-cell indicator
    out["kxrcf_cases"] = []
    for name, k, comp, cname in [("6x4 Q1 KXRCF density", 1, 2, "density"), ("6x4 Q2 KXRCF energy", 2, 3, "energy"), ("6x4 Q3 KXRCF density", 3, 2, "density")]:
        cs = Case(name, 6, 4, D(1) / 6, k, "hllc", [0, 0, 0, 0], {0: "outflow"}, "0.5", jump)
        U0 = cs.initial()
        ind = kxrcf(cs, U0, comp)
        out["kxrcf_cases"].append({"name": name, "nx": 6, "ny": 4, "h": format(cs.h, ".25e"), "degree": k, "variable": cname, "U0": flat(U0),
                                   "indicator": [format(v, ".25e") if v is not None else "nan" for v in ind]})
        print(name, "cells over 1:", sum(1 for v in ind if v is not None and v > 1), flush=True)
    # ---- degree 0 (the one-stage finite-volume limit, src/claw.cc:141-145) and whole steps with the limiters in the stage loop
    out["extra_cases"] = []
    for name, nx, ny, k, flux, side, kinds, cfl, field, lim in [
            ("5x4 periodic Q0 HLLC", 5, 4, 0, "hllc", [-1, -1, -1, -1], {}, "0.8", lambda x, y: periodic(x, y * 5 / 4), None),
            ("4x3 Q0 Roe with walls", 4, 3, 0, "roe", [2, 1, 0, 3], {0: "slip", 1: "outflow", 2: "inflow", 3: "farfield"}, "0.9", smooth, None),
            ("6x4 Q1 Roe, TVB (char, M = 0) + positivity in every stage", 6, 4, 1, "roe", [0, 0, 0, 0], {0: "outflow"}, "0.6", jump,
             {"M": "0", "beta": "2", "char_lim": True}),
            ("6x4 Q2 HLLC, TVB (component-wise, M = 20) + positivity in every stage", 6, 4, 2, "hllc", [0, 0, 0, 0], {0: "outflow"}, "0.5", jump,
             {"M": "20", "beta": "1.5", "char_lim": False}),
            ("6x4 Q2 HLLC, KXRCF (density) gating TVB (char, M = 0) + positivity in every stage", 6, 4, 2, "hllc", [0, 0, 0, 0], {0: "outflow"}, "0.5", jump,
             {"M": "0", "beta": "1.5", "char_lim": True, "indicator": "density"})]:
        cs = Case(name, nx, ny, D(1) / max(nx, 5) if k == 0 and nx == 5 else D(1) / nx, k, flux, side, kinds, cfl, field)
        U0 = cs.initial()
        def limit(V):   # compute_shock_indicator, apply_limiter, apply_positivity_limiter (src/claw.cc:762-766)
            marked = None
            if lim.get("indicator"):
                ind = kxrcf(cs, V, 2 if lim["indicator"] == "density" else 3)
                marked = [v is None or v > 1 for v in ind]     # 0/0 = NaN at cells whose inflow faces are all boundary faces: NaN > 1 is false
                marked = [False if v is None else m for v, m in zip(ind, marked)]
            return positivity(cs, tvb_qk(cs, V, lim["M"], lim["beta"], lim["char_lim"], marked))[0]
        if lim:   # run() limits the initial state first (src/claw.cc:997-1001)
            U0 = limit(U0)
        dt = cs.dt(U0)
        if lim:
            # iterate_explicit (src/claw.cc:726-772): every stage = update, then the TVB limiter, then the positivity limiter
            ark = [D(0), D(1) / 2] if k == 1 else [D(0), D(3) / 4, D(1) / 3]
            Uc = U0
            for a_rk in ark:
                Rr = cs.residual(Uc)
                new = []
                for c in range(len(Uc)):
                    cell = [[None] * (cs.N * cs.N) for _ in range(4)]
                    for comp in range(4):
                        for b in range(cs.N):
                            for a in range(cs.N):
                                i = a + cs.N * b
                                u = Uc[c][comp][i] + dt * Rr[c][comp][i] / (cs.ws[a] * cs.ws[b] * cs.h * cs.h)
                                cell[comp][i] = a_rk * U0[c][comp][i] + (1 - a_rk) * u
                    new.append(cell)
                Uc = limit(new)
            U1 = Uc
        else:
            U1 = cs.step(U0, dt)
        bfaces = []
        for j in range(cs.ny):
            for i in range(cs.nx):
                for f in range(4):
                    nb, bid = cs.neighbour(i, j, f)
                    if nb is None:
                        bfaces.append({"cell": i + cs.nx * j, "face": f, "id": bid,
                                       "values": [[format(v, ".25e") for v in cs.field(*cs.face_point(i, j, f, q))] for q in range(cs.N)]})
        rec = {"name": name, "nx": nx, "ny": ny, "h": format(cs.h, ".25e"), "degree": k, "flux": flux, "side": side,
               "kinds": {str(a): b for a, b in kinds.items()}, "cfl": cfl, "U0": flat(U0), "residual": flat(cs.residual(U0)),
               "cell_average": [format(v, ".25e") for a in cs.averages(U0) for v in a], "dt": format(dt, ".25e"), "U1": flat(U1),
               "boundary_faces": bfaces}
        if lim:
            rec["limiter"] = lim
        out["extra_cases"].append(rec)
        print(name, "dt", format(dt, ".6e"), flush=True)
    # ---- boundary values that move in time: the table of t in the first stage, of t + dt in the later ones
    out["moving_bc_cases"] = []
    for name, nx, ny, k, flux, side, kinds, cfl in [("4x3 Q2 HLLC, inflow and farfield states moving in time", 4, 3, 2, "hllc", [2, 1, 0, 3],
                                                     {0: "slip", 1: "outflow", 2: "inflow", 3: "farfield"}, "0.8"),
                                                    ("3x3 Q1 LxF, pressure and inflow states moving in time", 3, 3, 1, "lxf", [2, 1, 1, 2],
                                                     {1: "pressure", 2: "inflow"}, "0.7")]:
        cs = Case(name, nx, ny, D(1) / nx, k, flux, side, kinds, cfl, smooth)
        def moving(x, y, t):
            W = smooth(x + 3 * t, y - 2 * t)
            return [W[0] * (1 + 4 * t), W[1], W[2] * (1 + 2 * t), W[3] * (1 + 5 * t)]
        cs.field_t = moving
        U0 = cs.initial()
        dt = cs.dt(U0)
        R0, R1 = cs.residual(U0, bc_time=D(0)), cs.residual(U0, bc_time=dt)
        U1 = cs.step(U0, dt)
        bfaces = []
        for j in range(cs.ny):
            for i in range(cs.nx):
                for f in range(4):
                    nb, bid = cs.neighbour(i, j, f)
                    if nb is None:
                        pts = [cs.face_point(i, j, f, q) for q in range(cs.N)]
                        bfaces.append({"cell": i + cs.nx * j, "face": f, "id": bid,
                                       "values": [[format(v, ".25e") for v in moving(x, y, D(0))] for x, y in pts],
                                       "values_later": [[format(v, ".25e") for v in moving(x, y, dt)] for x, y in pts]})
        out["moving_bc_cases"].append({"name": name, "nx": nx, "ny": ny, "h": format(cs.h, ".25e"), "degree": k, "flux": flux, "side": side,
                                       "kinds": {str(a): b for a, b in kinds.items()}, "cfl": cfl, "U0": flat(U0), "residual": flat(R0),
                                       "residual_later": flat(R1), "cell_average": [format(v, ".25e") for a in cs.averages(U0) for v in a],
                                       "dt": format(dt, ".25e"), "U1": flat(U1), "boundary_faces": bfaces})
        print(name, "dt", format(dt, ".6e"), flush=True)
    # ---- a whole step on the modal basis with TVB-Pk + positivity after every stage
    out["pk_step_cases"] = []
    for name, k, flux, cfl, lim in [("6x4 P2 HLLC, TVB-Pk (char, M = 0) + positivity in every stage", 2, "hllc", "0.5", {"M": "0", "beta": "2", "char_lim": True}),
                                    ("6x4 P1 LxF, TVB-Pk (component-wise, M = 10) + positivity in every stage", 1, "lxf", "0.7", {"M": "10", "beta": "1.5", "char_lim": False})]:
        cs = PkCase(name, 6, 4, D(1) / 6, k, flux, [0, 0, 0, 0], {0: "outflow"}, cfl, jump)
        limit = lambda V: positivity_pk(cs, tvb_pk(cs, V, lim["M"], lim["beta"], lim["char_lim"]))[0]
        U0 = limit(cs.initial())
        dt = cs.dt(U0)
        ark = [D(0), D(1) / 2] if k == 1 else [D(0), D(3) / 4, D(1) / 3]
        Uc = U0
        for a_rk in ark:
            Rr = cs.residual(Uc)
            new = [[[a_rk * U0[c][comp][m] + (1 - a_rk) * (Uc[c][comp][m] + dt * Rr[c][comp][m] / (cs.h * cs.h)) for m in range(len(cs.modes))]
                    for comp in range(4)] for c in range(len(Uc))]
            Uc = limit(new)
        bfaces = []
        for j in range(cs.ny):
            for i in range(cs.nx):
                for f in range(4):
                    nb, bid = cs.neighbour(i, j, f)
                    if nb is None:
                        bfaces.append({"cell": i + cs.nx * j, "face": f, "id": bid,
                                       "values": [[format(v, ".25e") for v in cs.field(*cs.face_point(i, j, f, q))] for q in range(cs.N)]})
        out["pk_step_cases"].append({"name": name, "nx": 6, "ny": 4, "h": format(cs.h, ".25e"), "degree": k, "flux": flux, "side": [0, 0, 0, 0],
                                     "kinds": {"0": "outflow"}, "cfl": cfl, "limiter": lim, "U0": flat(U0), "residual": flat(cs.residual(U0)),
                                     "cell_average": [format(v, ".25e") for a in cs.averages(U0) for v in a], "dt": format(dt, ".25e"),
                                     "U1": flat(Uc), "boundary_faces": bfaces})
        print(name, "dt", format(dt, ".6e"), flush=True)
    # ---- gravity forcing and local time stepping (time step type = local)
    out["forcing_cases"] = []
    for name, nx, ny, k, flux, side, kinds, cfl, grav, local in [
            ("4x3 Q2 HLLC, gravity 0.4, walls", 4, 3, 2, "hllc", [2, 1, 0, 0], {0: "slip", 1: "outflow", 2: "inflow"}, "0.8", "0.4", False),
            ("3x3 Q1 Roe, gravity 0.25, local time steps", 3, 3, 1, "roe", [0, 0, 0, 0], {0: "slip"}, "0.7", "0.25", True),
            ("4x4 periodic Q3 LxF, local time steps", 4, 4, 3, "lxf", [-1, -1, -1, -1], {}, "0.6", "0", True)]:
        cs = Case(name, nx, ny, D(1) / nx, k, flux, side, kinds, cfl, periodic if side[0] < 0 else smooth)
        cs.gravity = D(grav)
        U0 = cs.initial()
        R = cs.residual(U0)
        dts = cs.cell_dts(U0)
        U1 = cs.step(U0, dts if local else min(dts))
        bfaces = []
        for j in range(cs.ny):
            for i in range(cs.nx):
                for f in range(4):
                    nb, bid = cs.neighbour(i, j, f)
                    if nb is None:
                        bfaces.append({"cell": i + cs.nx * j, "face": f, "id": bid,
                                       "values": [[format(v, ".25e") for v in cs.field(*cs.face_point(i, j, f, q))] for q in range(cs.N)]})
        out["forcing_cases"].append({"name": name, "nx": nx, "ny": ny, "h": format(cs.h, ".25e"), "degree": k, "flux": flux, "side": side,
                                     "kinds": {str(a): b for a, b in kinds.items()}, "cfl": cfl, "gravity": grav, "local": local,
                                     "U0": flat(U0), "residual": flat(R), "cell_average": [format(v, ".25e") for a in cs.averages(U0) for v in a],
                                     "dt": format(min(dts), ".25e"), "cell_dt": [format(v, ".25e") for v in dts], "U1": flat(U1),
                                     "boundary_faces": bfaces})
        print(name, "dt", format(min(dts), ".6e"), flush=True)
    # ---- limiters: TVB (characteristic and component-wise, M = 0 and M > 0) and the positivity limiter
    out["limiter_cases"] = []
    wall = [0, 0, 0, 0]
    for name, k, M, beta, char in [("6x4 Q1 TVB characteristic, M = 0", 1, "0", "2", True), ("6x4 Q2 TVB component-wise, M = 30", 2, "30", "1.5", False),
                                    ("6x4 Q3 TVB characteristic, M = 5", 3, "5", "1", True)]:
        cs = Case(name, 6, 4, D(1) / 6, k, "hllc", wall, {0: "outflow"}, "0.5", jump)
        U0 = cs.initial()
        U1 = tvb_qk(cs, U0, M, beta, char)
        changed = sum(1 for c in range(len(U0)) if U0[c] != U1[c])
        out["limiter_cases"].append({"name": name, "kind": "tvb", "nx": 6, "ny": 4, "h": format(cs.h, ".25e"), "degree": k, "side": wall,
                                     "M": M, "beta": beta, "char_lim": char, "U0": flat(U0), "U1": flat(U1), "cells_changed": changed})
        print(name, "cells changed", changed, flush=True)
    for k in (1, 2, 3):
        cs = Case("3x2 Q%d positivity" % k, 3, 2, D(1) / 3, k, "hllc", wall, {0: "outflow"}, "0.5", smooth)
        U0 = cs.initial()
        N = cs.N
        # cell 1: density dips below zero at a Gauss-Lobatto point (theta1 < 1); cell 4: pressure does (theta2 root in (0, 1));
        # cell 5: both; the others are left alone
        for c, (dr, de) in {1: ("1.2", "0"), 4: ("0", "0.995"), 5: ("1.1", "0.85")}.items():
            for b in range(N):
                for a in range(N):
                    wgt = (1 - cs.xs[a]) * (1 - cs.xs[b]) * 4 - 1          # zero mean over the cell: the average stays put
                    U0[c][2][a + N * b] *= 1 + D(dr) * wgt
                    U0[c][3][a + N * b] *= 1 - D(de) * (-wgt)
        U1, th = positivity(cs, U0)
        out["limiter_cases"].append({"name": cs.name, "kind": "positivity", "nx": 3, "ny": 2, "h": format(cs.h, ".25e"), "degree": k, "side": wall,
                                     "U0": flat(U0), "U1": flat(U1), "theta": [[format(t, ".20e") for t in pair] for pair in th]})
        print(cs.name, "theta", [[float(t) for t in pair] for pair in th], flush=True)
    # ---- the same limiters on the modal (Pk) basis
    out["pk_limiter_cases"] = []
    for name, k, M, beta, char, ang in [("6x4 P1 TVB characteristic, M = 0", 1, "0", "2", True, False),
                                         ("6x4 P2 TVB component-wise, M = 30, angular momentum kept", 2, "30", "1.5", False, True),
                                         ("6x4 P3 TVB characteristic, M = 5", 3, "5", "1", True, False)]:
        cs = PkCase(name, 6, 4, D(1) / 6, k, "hllc", wall, {0: "outflow"}, "0.5", jump)
        U0 = cs.initial()
        U1 = tvb_pk(cs, U0, M, beta, char, ang)
        changed = sum(1 for c in range(len(U0)) if U0[c] != U1[c])
        out["pk_limiter_cases"].append({"name": name, "kind": "tvb", "nx": 6, "ny": 4, "h": format(cs.h, ".25e"), "degree": k, "side": wall,
                                        "M": M, "beta": beta, "char_lim": char, "conserve_angular_momentum": ang, "U0": flat(U0), "U1": flat(U1),
                                        "cells_changed": changed})
        print(name, "cells changed", changed, flush=True)
    for k in (1, 2, 3):
        cs = PkCase("3x2 P%d positivity" % k, 3, 2, D(1) / 3, k, "hllc", wall, {0: "outflow"}, "0.5", smooth)
        U0 = cs.initial()
        # cell 1: the density dips below zero towards a corner (theta1 < 1); cell 4: the pressure does (theta2 root in (0, 1));
        # cell 5: both -- through the linear modes, which leave the average alone
        for c, (dr, de) in {1: ("0.75", "0"), 4: ("0", "0.62"), 5: ("0.7", "0.5")}.items():
            for m in (1, cs.N):
                U0[c][2][m] -= D(dr) * U0[c][2][0] / D(3).sqrt()
                U0[c][3][m] -= D(de) * U0[c][3][0] / D(3).sqrt()
        U1, th = positivity_pk(cs, U0)
        out["pk_limiter_cases"].append({"name": cs.name, "kind": "positivity", "nx": 3, "ny": 2, "h": format(cs.h, ".25e"), "degree": k, "side": wall,
                                        "U0": flat(U0), "U1": flat(U1), "theta": [[format(t, ".20e") for t in pair] for pair in th]})
        print(cs.name, "theta", [[float(t) for t in pair] for pair in th], flush=True)
    json.dump(out, open(os.path.join(HERE, "residual_fixture.json"), "w"), indent=0)
    print("residual_fixture.json written")


if __name__ == "__main__":
    main()
