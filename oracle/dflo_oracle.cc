// dflo_oracle.cc -- CPU restatement of dflo's explicit DG residual + SSP-RK path.
//
// TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load this library; nothing under dflo_amd/ links,
// imports or calls it.  It restates, loop for loop, the algorithm of the reference
// (paths relative to the reference root, cited at every function) on the same flat
// mesh description the C ABI takes (include/dflo_hip.h), so the HIP engine and this
// oracle can be driven with identical inputs.
//
// PINNING STATUS (see DESIGN.md "Oracle"):
//   * pointwise numerical fluxes: pinned against outputs of the reference's own
//     src/equation.h recorded in SURVEY.md section 8c (tests/golden/flux_reference.json)
//     and against analytic identities (consistency H(W,W,n) = F(W).n, conservation).
//   * assembled residual / RK solution: deal.II (un-vendored, unpinned version,
//     src/CMakeLists.txt:26) owns the assembly machinery and is absent here, the
//     reference has no tests or fixtures => PARITY UNPINNED at that level; it is
//     anchored on the mathematical statement of the reference loops only (exactness
//     on the steady isentropic vortex src/ic.cc:44-61, free-stream preservation,
//     conservation).
//
// The last section ("optimised CPU twin") is a second CPU implementation of the same stage, fused and threaded the
// way a CPU wants it; it is checked against the restatement by tests/test_oracle_assembly.py and timed by bench.py
// as cpu_baseline.optimised_twin.  It is not the parity checker.
//
// deal.II conventions relied upon (third party, cannot be cited in the reference
// tree): QGauss on [0,1] ascending; tensor points x fastest; FESystem of one DG base
// element x4 is component-major; faces 0:x=0 1:x=1 2:y=0 3:y=1; MeshWorker::loop
// integrates an interior face once, from the cell with the smaller index.

#include "../include/dflo_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

const int NC = 4;            // n_components = dim+2, src/equation.h:26
const int DENS = 2;          // density_component = dim, src/equation.h:27
const int ENER = 3;          // energy_component = dim+1, src/equation.h:28
const double gas_gamma = 1.4;  // src/equation.cc:33

// --------------------------------------------------------------------------
// 1-D rules on [0,1]
// --------------------------------------------------------------------------
struct Rule {
  std::vector<double> x, w;
};

void legendre(int n, double t, double &p, double &dp) {
  // P_n(t), P_n'(t) on [-1,1]
  double p0 = 1.0, p1 = t;
  if (n == 0) { p = 1.0; dp = 0.0; return; }
  for (int k = 2; k <= n; ++k) {
    double pk = ((2.0 * k - 1.0) * t * p1 - (k - 1.0) * p0) / k;
    p0 = p1;
    p1 = pk;
  }
  p = p1;
  dp = n * (t * p1 - p0) / (t * t - 1.0);
}

Rule gauss_rule(int n) {  // QGauss<1>(n)
  Rule r;
  r.x.resize(n);
  r.w.resize(n);
  for (int i = 0; i < n; ++i) {
    double t = -std::cos(M_PI * (i + 0.75) / (n + 0.5));
    for (int it = 0; it < 100; ++it) {
      double p, dp;
      legendre(n, t, p, dp);
      double dt = p / dp;
      t -= dt;
      if (std::fabs(dt) < 1e-16) break;
    }
    double p, dp;
    legendre(n, t, p, dp);
    r.x[i] = 0.5 * (1.0 + t);
    r.w[i] = 1.0 / ((1.0 - t * t) * dp * dp);
  }
  return r;
}

Rule gauss_lobatto_rule(int n) {  // QGaussLobatto<1>(n), n>=2
  Rule r;
  r.x.assign(n, 0.0);
  r.w.assign(n, 0.0);
  const int m = n - 1;  // interior points are roots of P_m'
  r.x[0] = 0.0;
  r.x[n - 1] = 1.0;
  for (int i = 1; i < n - 1; ++i) {
    double t = -std::cos(M_PI * i / m);
    for (int it = 0; it < 100; ++it) {
      double p, dp;
      legendre(m, t, p, dp);
      // P_m'' from the Legendre ODE: (1-t^2) P'' - 2 t P' + m(m+1) P = 0
      double ddp = (2.0 * t * dp - m * (m + 1.0) * p) / (1.0 - t * t);
      double dt = dp / ddp;
      t -= dt;
      if (std::fabs(dt) < 1e-16) break;
    }
    r.x[i] = 0.5 * (1.0 + t);
  }
  for (int i = 0; i < n; ++i) {
    double t = 2.0 * r.x[i] - 1.0, p, dp;
    if (i == 0) t = -1.0;
    if (i == n - 1) t = 1.0;
    if (std::fabs(std::fabs(t) - 1.0) < 1e-15) {
      p = (t > 0 || m % 2 == 0) ? 1.0 : -1.0;
    } else {
      legendre(m, t, p, dp);
    }
    r.w[i] = 1.0 / (n * (n - 1.0) * p * p);
  }
  return r;
}

// --------------------------------------------------------------------------
// Scalar finite element on the unit square
// --------------------------------------------------------------------------
struct ScalarFE {
  int degree, basis, N, ns;
  std::vector<double> nodes;       // Gauss nodes (Qk)
  std::vector<int> pk_i, pk_j;     // Pk: modal index -> (i,j), ordered as src/claw.cc:107-113

  void init(int k, int b) {
    degree = k;
    basis = b;
    N = k + 1;
    if (basis == DFLO_BASIS_QK) {
      ns = N * N;
      nodes = gauss_rule(N).x;  // FE_DGQArbitraryNodes(QGauss<1>(k+1)), src/main.cc:40
    } else {
      ns = (k + 1) * (k + 2) / 2;  // FE_DGP<2>(k), src/main.cc:46
      for (int j = 0; j <= k; ++j)
        for (int i = 0; i <= k - j; ++i) {
          pk_i.push_back(i);
          pk_j.push_back(j);
        }
    }
  }
  double lag(int a, double x) const {
    double v = 1.0;
    for (int m = 0; m < N; ++m)
      if (m != a) v *= (x - nodes[m]) / (nodes[a] - nodes[m]);
    return v;
  }
  double dlag(int a, double x) const {
    double s = 0.0;
    for (int j = 0; j < N; ++j) {
      if (j == a) continue;
      double v = 1.0 / (nodes[a] - nodes[j]);
      for (int m = 0; m < N; ++m)
        if (m != a && m != j) v *= (x - nodes[m]) / (nodes[a] - nodes[m]);
      s += v;
    }
    return s;
  }
  // orthonormal Legendre on [0,1]: sqrt(2n+1) P_n(2x-1)
  static void leg01(int n, double x, double &v, double &dv) {
    double t = 2.0 * x - 1.0, p0 = 1.0, p1 = t, d0 = 0.0, d1 = 1.0;
    if (n == 0) { v = 1.0; dv = 0.0; return; }
    for (int k = 2; k <= n; ++k) {
      double pk = ((2.0 * k - 1.0) * t * p1 - (k - 1.0) * p0) / k;
      double dk = d0 + (2.0 * k - 1.0) * p1;
      p0 = p1; p1 = pk; d0 = d1; d1 = dk;
    }
    double s = std::sqrt(2.0 * n + 1.0);
    v = s * p1;
    dv = s * d1 * 2.0;
  }
  void eval(int j, double xi, double eta, double &v, double &gx, double &gy) const {
    if (basis == DFLO_BASIS_QK) {
      int a = j % N, b = j / N;
      double la = lag(a, xi), lb = lag(b, eta);
      v = la * lb;
      gx = dlag(a, xi) * lb;
      gy = la * dlag(b, eta);
    } else {
      double pi, dpi, pj, dpj;
      leg01(pk_i[j], xi, pi, dpi);
      leg01(pk_j[j], eta, pj, dpj);
      v = pi * pj;
      gx = dpi * pj;
      gy = pi * dpj;
    }
  }
};

// Table of scalar shape values / reference gradients on a point set.
struct ShapeTable {
  int np = 0;
  std::vector<double> xi, eta, w;     // reference points, weights
  std::vector<double> v, gx, gy;      // [ns][np]
  void build(const ScalarFE &fe) {
    v.resize(fe.ns * np);
    gx.resize(fe.ns * np);
    gy.resize(fe.ns * np);
    for (int j = 0; j < fe.ns; ++j)
      for (int p = 0; p < np; ++p) fe.eval(j, xi[p], eta[p], v[j * np + p], gx[j * np + p], gy[j * np + p]);
  }
};

ShapeTable tensor_table(const ScalarFE &fe, const Rule &rx, const Rule &ry) {
  ShapeTable t;
  t.np = (int)(rx.x.size() * ry.x.size());
  for (size_t b = 0; b < ry.x.size(); ++b)
    for (size_t a = 0; a < rx.x.size(); ++a) {  // x fastest
      t.xi.push_back(rx.x[a]);
      t.eta.push_back(ry.x[b]);
      t.w.push_back(rx.w[a] * ry.w[b]);
    }
  t.build(fe);
  return t;
}

ShapeTable face_table(const ScalarFE &fe, const Rule &r, int face) {
  ShapeTable t;
  t.np = (int)r.x.size();
  for (int q = 0; q < t.np; ++q) {
    double s = r.x[q];
    switch (face) {
      case 0: t.xi.push_back(0.0); t.eta.push_back(s); break;
      case 1: t.xi.push_back(1.0); t.eta.push_back(s); break;
      case 2: t.xi.push_back(s); t.eta.push_back(0.0); break;
      default: t.xi.push_back(s); t.eta.push_back(1.0); break;
    }
    t.w.push_back(r.w[q]);
  }
  t.build(fe);
  return t;
}

// --------------------------------------------------------------------------
// Pointwise physics: src/equation.h
// --------------------------------------------------------------------------
double kinetic_energy(const double *W) {  // src/equation.h:70-79
  double ke = 0;
  for (int d = 0; d < 2; ++d) ke += W[d] * W[d];
  ke *= 0.5 / W[DENS];
  return ke;
}
double pressure(const double *W) {  // src/equation.h:87-92
  return (gas_gamma - 1.0) * (W[ENER] - kinetic_energy(W));
}
double max_eigenvalue(const double *W) {  // src/equation.h:100-114
  const double p = pressure(W);
  double vel = 0;
  for (int d = 0; d < 2; ++d) vel += W[d] * W[d];
  vel = std::sqrt(vel) / W[DENS];
  return vel + std::sqrt(gas_gamma * p / W[DENS]);
}
double max_eigenvalue_n(const double *W, const double *n) {  // src/equation.h:122-137
  const double p = pressure(W);
  const double sonic = std::sqrt(gas_gamma * p / W[DENS]);
  double vel = 0;
  for (int d = 0; d < 2; ++d) vel += W[d] * n[d];
  vel /= W[DENS];
  return std::fabs(vel) + sonic;
}
double sound_speed(const double *W) {  // src/equation.h:145-152
  return std::sqrt(gas_gamma * pressure(W) / W[DENS]);
}
void flux_matrix(const double *W, double (&flux)[NC][2]) {  // src/equation.h:160-193
  const double p = pressure(W);
  for (int d = 0; d < 2; ++d) {
    for (int e = 0; e < 2; ++e) flux[d][e] = W[d] * W[e] / W[DENS];
    flux[d][d] += p;
  }
  for (int d = 0; d < 2; ++d) flux[DENS][d] = W[d];
  for (int d = 0; d < 2; ++d) flux[ENER][d] = W[d] / W[DENS] * (W[ENER] + p);
}
void forcing_vector(const double *W, double (&f)[NC]) {  // src/equation.h:831-850
  const double gravity = -1.0;
  for (int c = 0; c < NC; ++c) {
    if (c == 1) f[c] = gravity * W[DENS];        // case dim-1
    else if (c == ENER) f[c] = gravity * W[1];   // case energy_component: W[dim-1]
    else f[c] = 0;
  }
}

void lxf_flux(const double *n, const double *Wp, const double *Wm, const double *Ap, const double *Am,
              double *F) {  // src/equation.h:326-377
  double vp = 0, vm = 0;
  for (int d = 0; d < 2; ++d) {
    vp += Wp[d] * n[d];
    vm += Wm[d] * n[d];
  }
  vp /= Wp[DENS];
  vm /= Wm[DENS];
  const double pp = pressure(Wp), pm = pressure(Wm);
  const double lp = max_eigenvalue_n(Ap, n), lm = max_eigenvalue_n(Am, n);
  const double lambda = std::max(lp, lm);
  for (int d = 0; d < 2; ++d) F[d] = 0.5 * (pp * n[d] + Wp[d] * vp + pm * n[d] + Wm[d] * vm);
  F[DENS] = 0.5 * (Wp[DENS] * vp + Wm[DENS] * vm);
  F[ENER] = 0.5 * ((Wp[ENER] + pp) * vp + (Wm[ENER] + pm) * vm);
  for (int c = 0; c < NC; ++c) F[c] += 0.5 * lambda * (Wp[c] - Wm[c]);
}

void steger_warming_flux(const double *n, const double *Wp, const double *Wm, double *F) {  // src/equation.h:384-464
  double pf[NC], mf[NC];
  double vp = 0, vm = 0, q2p = 0, q2m = 0;
  for (int d = 0; d < 2; ++d) {
    vp += Wp[d] * n[d];
    vm += Wm[d] * n[d];
    q2p += Wp[d] * Wp[d];
    q2m += Wm[d] * Wm[d];
  }
  vp /= Wp[DENS];
  vm /= Wm[DENS];
  q2p /= Wp[DENS] * Wp[DENS];
  q2m /= Wm[DENS] * Wm[DENS];
  const double pp = pressure(Wp), pm = pressure(Wm);
  const double cp = std::sqrt(gas_gamma * pp / Wp[DENS]);
  const double cm = std::sqrt(gas_gamma * pm / Wm[DENS]);
  double l1p = std::max(vp, 0.0), l2p = std::max(vp + cp, 0.0), l3p = std::max(vp - cp, 0.0);
  double ap = 2.0 * (gas_gamma - 1.0) * l1p + l2p + l3p;
  double fp = 0.5 * Wp[DENS] / gas_gamma;
  for (int d = 0; d < 2; ++d) pf[d] = ap * Wp[d] / Wp[DENS] + cp * (l2p - l3p) * n[d];
  pf[DENS] = ap;
  pf[ENER] = 0.5 * ap * q2p + cp * vp * (l2p - l3p) + cp * cp * (l2p + l3p) / (gas_gamma - 1.0);
  double l1m = std::min(vm, 0.0), l2m = std::min(vm + cm, 0.0), l3m = std::min(vm - cm, 0.0);
  double am = 2.0 * (gas_gamma - 1.0) * l1m + l2m + l3m;
  double fm = 0.5 * Wm[DENS] / gas_gamma;
  for (int d = 0; d < 2; ++d) mf[d] = am * Wm[d] / Wm[DENS] + cm * (l2m - l3m) * n[d];
  mf[DENS] = am;
  mf[ENER] = 0.5 * am * q2m + cm * vm * (l2m - l3m) + cm * cm * (l2m + l3m) / (gas_gamma - 1.0);
  for (int c = 0; c < NC; ++c) F[c] = fp * pf[c] + fm * mf[c];
}

void roe_flux(const double *n, const double *Wl, const double *Wr, double *F) {  // src/equation.h:471-556
  double rls = std::sqrt(Wl[DENS]), rrs = std::sqrt(Wr[DENS]);
  double fl = rls / (rls + rrs), fr = 1.0 - fl;
  double vl[2], vr[2], vel[2], dv[2];
  double v2l = 0, v2r = 0, vln = 0, vrn = 0, veln = 0, v2 = 0, vdv = 0;
  for (int d = 0; d < 2; ++d) {
    vl[d] = Wl[d] / Wl[DENS];
    vr[d] = Wr[d] / Wr[DENS];
    v2l += vl[d] * vl[d];
    v2r += vr[d] * vr[d];
    vln += vl[d] * n[d];
    vrn += vr[d] * n[d];
    vel[d] = vl[d] * fl + vr[d] * fr;
    veln += vel[d] * n[d];
    v2 += vel[d] * vel[d];
    dv[d] = vr[d] - vl[d];
    vdv += vel[d] * dv[d];
  }
  double pl = (gas_gamma - 1) * (Wl[ENER] - 0.5 * Wl[DENS] * v2l);
  double pr = (gas_gamma - 1) * (Wr[ENER] - 0.5 * Wr[DENS] * v2r);
  double hl = gas_gamma * pl / Wl[DENS] / (gas_gamma - 1) + 0.5 * v2l;
  double hr = gas_gamma * pr / Wr[DENS] / (gas_gamma - 1) + 0.5 * v2r;
  double rho = rls * rrs;
  double h = hl * fl + hr * fr;
  double c = std::sqrt((gas_gamma - 1.0) * (h - 0.5 * v2));
  double drho = Wr[DENS] - Wl[DENS];
  double dp = pr - pl;
  double dvn = vrn - vln;
  double a1 = (dp - rho * c * dvn) / (2.0 * c * c);
  double a2 = drho - dp / (c * c);
  double a3 = (dp + rho * c * dvn) / (2.0 * c * c);
  double l1 = std::fabs(veln - c), l2 = std::fabs(veln), l3 = std::fabs(veln + c);
  double delta = 0.1 * c;  // entropy fix, src/equation.h:529-531
  if (l1 < delta) l1 = 0.5 * (l1 * l1 / delta + delta);
  if (l3 < delta) l3 = 0.5 * (l3 * l3 / delta + delta);
  double D[NC];
  D[DENS] = l1 * a1 + l2 * a2 + l3 * a3;
  D[ENER] = l1 * a1 * (h - c * veln) + l2 * a2 * 0.5 * v2 + l2 * rho * (vdv - veln * dvn) + l3 * a3 * (h + c * veln);
  F[DENS] = 0.5 * (Wl[DENS] * vln + Wr[DENS] * vrn - D[DENS]);
  F[ENER] = 0.5 * (Wl[DENS] * hl * vln + Wr[DENS] * hr * vrn - D[ENER]);
  double pavg = 0.5 * (pl + pr);
  for (int d = 0; d < 2; ++d) {
    D[d] = (vel[d] - n[d] * c) * l1 * a1 + vel[d] * l2 * a2 + (dv[d] - n[d] * dvn) * l2 * rho + (vel[d] + n[d] * c) * l3 * a3;
    F[d] = n[d] * pavg + 0.5 * (Wl[d] * vln + Wr[d] * vrn) - 0.5 * D[d];
  }
}

void hllc_flux(const double *n, const double *Wl, const double *Wr, double *F) {  // src/equation.h:565-681
  double rls = std::sqrt(Wl[DENS]), rrs = std::sqrt(Wr[DENS]);
  double fl = rls / (rls + rrs), fr = 1.0 - fl;
  double vl[2], vr[2], vel[2];
  double v2l = 0, v2r = 0, vln = 0, vrn = 0, veln = 0, v2 = 0;
  for (int d = 0; d < 2; ++d) {
    vl[d] = Wl[d] / Wl[DENS];
    vr[d] = Wr[d] / Wr[DENS];
    v2l += vl[d] * vl[d];
    v2r += vr[d] * vr[d];
    vln += vl[d] * n[d];
    vrn += vr[d] * n[d];
    vel[d] = vl[d] * fl + vr[d] * fr;
    veln += vel[d] * n[d];
    v2 += vel[d] * vel[d];
  }
  double pl = (gas_gamma - 1) * (Wl[ENER] - 0.5 * Wl[DENS] * v2l);
  double pr = (gas_gamma - 1) * (Wr[ENER] - 0.5 * Wr[DENS] * v2r);
  double hl = (Wl[ENER] + pl) / Wl[DENS];
  double hr = (Wr[ENER] + pr) / Wr[DENS];
  double cl = std::sqrt(gas_gamma * pl / Wl[DENS]);
  double cr = std::sqrt(gas_gamma * pr / Wr[DENS]);
  double el = Wl[ENER] / Wl[DENS];
  double er = Wr[ENER] / Wr[DENS];
  double h = hl * fl + hr * fr;
  double c = std::sqrt((gas_gamma - 1.0) * (h - 0.5 * v2));
  double sl = std::min(veln - c, vln - cl);
  double sr = std::max(veln + c, vrn + cr);
  double sm = (pl - pr - Wl[DENS] * vln * (sl - vln) + Wr[DENS] * vrn * (sr - vrn)) /
              (Wr[DENS] * (sr - vrn) - Wl[DENS] * (sl - vln));
  double pstar = Wr[DENS] * (vrn - sr) * (vrn - sm) + pr;
  if (sm >= 0.0) {
    if (sl > 0.0) {
      F[DENS] = Wl[DENS] * vln;
      for (int d = 0; d < 2; ++d) F[d] = Wl[DENS] * vl[d] * vln + pl * n[d];
      F[ENER] = el * Wl[DENS] * vln + pl * vln;
    } else {
      double inv = 1.0 / (sl - sm);
      double slmul = sl - vln;
      double rhosl = Wl[DENS] * slmul * inv;
      double rhousl[2];
      for (int d = 0; d < 2; ++d) rhousl[d] = (Wl[DENS] * vl[d] * slmul + (pstar - pl) * n[d]) * inv;
      double esl = (slmul * el * Wl[DENS] - pl * vln + pstar * sm) * inv;
      F[DENS] = rhosl * sm;
      for (int d = 0; d < 2; ++d) F[d] = rhousl[d] * sm + pstar * n[d];
      F[ENER] = (esl + pstar) * sm;
    }
  } else {
    if (sr >= 0.0) {
      double inv = 1.0 / (sr - sm);
      double srmur = sr - vrn;
      double rhosr = Wr[DENS] * srmur * inv;
      double rhousr[2];
      for (int d = 0; d < 2; ++d) rhousr[d] = (Wr[DENS] * vr[d] * srmur + (pstar - pr) * n[d]) * inv;
      double esr = (srmur * er * Wr[DENS] - pr * vrn + pstar * sm) * inv;
      F[DENS] = rhosr * sm;
      for (int d = 0; d < 2; ++d) F[d] = rhousr[d] * sm + pstar * n[d];
      F[ENER] = (esr + pstar) * sm;
    } else {
      F[DENS] = Wr[DENS] * vrn;
      for (int d = 0; d < 2; ++d) F[d] = Wr[DENS] * vr[d] * vrn + pr * n[d];
      F[ENER] = er * Wr[DENS] * vrn + pr * vrn;
    }
  }
}

double ERF(double xarg) {  // src/equation.h:688-709, Abramowitz-Stegun 7.1.26
  const double a1 = 0.254829592, a2 = -0.284496736, a3 = 1.421413741, a4 = -1.453152027, a5 = 1.061405429;
  const double p = 0.3275911;
  int sign = 1;
  if (xarg < 0) sign = -1;
  double x = std::fabs(xarg);
  double t = 1.0 / (1.0 + p * x);
  double y = 1.0 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * std::exp(-x * x);
  return sign * y;
}
void kinetic_split_flux(int sign, const double *n, const double *W, double *F) {  // src/equation.h:716-751
  double vdotn = 0;
  for (int d = 0; d < 2; ++d) vdotn += W[d] * n[d];
  vdotn /= W[DENS];
  double p = pressure(W);
  double beta = 0.5 * W[DENS] / p;
  double s = vdotn * std::sqrt(beta);
  double A = 0.5 * (1.0 + sign * ERF(s));
  double B = 0.5 * sign * std::exp(-s * s) / std::sqrt(M_PI * beta);
  double ufact = vdotn * A + B;
  for (int d = 0; d < 2; ++d) F[d] = p * n[d] * A + W[d] * ufact;
  F[DENS] = W[DENS] * ufact;
  F[ENER] = (W[ENER] + p) * vdotn * A + (W[ENER] + 0.5 * p) * B;
}
void kfvs_flux(const double *n, const double *Wp, const double *Wm, double *F) {  // src/equation.h:758-782
  double pf[NC], mf[NC];
  kinetic_split_flux(+1, n, Wp, pf);
  kinetic_split_flux(-1, n, Wm, mf);
  for (int c = 0; c < NC; ++c) F[c] = pf[c] + mf[c];
}

// flux dispatcher, src/claw.h:271-325
void numerical_normal_flux(int flux_type, const double *n, const double *Wp, const double *Wm, const double *Ap,
                           const double *Am, double *F) {
  switch (flux_type) {
    case DFLO_FLUX_LXF: lxf_flux(n, Wp, Wm, Ap, Am, F); break;
    case DFLO_FLUX_SW: steger_warming_flux(n, Wp, Wm, F); break;
    case DFLO_FLUX_KFVS: kfvs_flux(n, Wp, Wm, F); break;
    case DFLO_FLUX_ROE: roe_flux(n, Wp, Wm, F); break;
    case DFLO_FLUX_HLLC: hllc_flux(n, Wp, Wm, F); break;
    default: for (int c = 0; c < NC; ++c) F[c] = NAN;
  }
}

void compute_Wminus(int kind, const double *n, const double *Wp, const double *bv, double *Wm) {  // src/equation.h:942-1033
  switch (kind) {
    case DFLO_BC_INFLOW:
      for (int c = 0; c < NC; ++c) Wm[c] = bv[c];
      break;
    case DFLO_BC_OUTFLOW:
      for (int c = 0; c < NC; ++c) Wm[c] = Wp[c];
      break;
    case DFLO_BC_PRESSURE: {
      const double rho = Wp[DENS];
      double ke = 0;
      for (int d = 0; d < 2; ++d) ke += Wp[d] * Wp[d];
      ke *= 0.5 / rho;
      for (int c = 0; c < 2; ++c) Wm[c] = Wp[c];
      Wm[DENS] = rho;
      Wm[ENER] = bv[ENER] / (gas_gamma - 1.0) + ke;  // w_3 read as pressure, src/equation.h:992
      break;
    }
    case DFLO_BC_SLIP: {
      double vdotn = 0;
      for (int d = 0; d < 2; ++d) vdotn += Wp[d] * n[d];
      for (int c = 0; c < 2; ++c) Wm[c] = Wp[c] - 2.0 * vdotn * n[c];
      Wm[DENS] = Wp[DENS];
      Wm[ENER] = Wp[ENER];
      break;
    }
    case DFLO_BC_FARFIELD:
      for (int c = 0; c < NC; ++c) Wm[c] = bv[c];
      break;
    default:
      for (int c = 0; c < NC; ++c) Wm[c] = NAN;
  }
}

// eigenvector matrices at a state, src/equation.h:226-265
void compute_eigen_matrix(const double *W, double (&Rx)[NC][NC], double (&Lx)[NC][NC], double (&Ry)[NC][NC],
                          double (&Ly)[NC][NC]) {
  double g1 = gas_gamma - 1.0;
  double rho = W[DENS], E = W[ENER];
  double u = W[0] / rho, v = W[1] / rho;
  double q2 = u * u + v * v;
  double p = g1 * (E - 0.5 * rho * q2);
  double c2 = gas_gamma * p / rho;
  double c = std::sqrt(c2);
  double beta = 0.5 / c2;
  double phi2 = 0.5 * g1 * q2;
  double h = c2 / g1 + 0.5 * q2;
  double rx[NC][NC] = {{1, 0, 1, 1}, {u, 0, u + c, u - c}, {v, -1, v, v}, {0.5 * q2, -v, h + c * u, h - c * u}};
  double ry[NC][NC] = {{1, 0, 1, 1}, {u, 1, u, u}, {v, 0, v + c, v - c}, {0.5 * q2, u, h + c * v, h - c * v}};
  double lx[NC][NC] = {{1 - phi2 / c2, g1 * u / c2, g1 * v / c2, -g1 / c2},
                       {v, 0, -1, 0},
                       {beta * (phi2 - c * u), beta * (c - g1 * u), -beta * g1 * v, beta * g1},
                       {beta * (phi2 + c * u), -beta * (c + g1 * u), -beta * g1 * v, beta * g1}};
  double ly[NC][NC] = {{1 - phi2 / c2, g1 * u / c2, g1 * v / c2, -g1 / c2},
                       {-u, 1, 0, 0},
                       {beta * (phi2 - c * v), -beta * g1 * u, beta * (c - g1 * v), beta * g1},
                       {beta * (phi2 + c * v), -beta * g1 * u, -beta * (c + g1 * v), beta * g1}};
  std::memcpy(Rx, rx, sizeof(rx));
  std::memcpy(Ry, ry, sizeof(ry));
  std::memcpy(Lx, lx, sizeof(lx));
  std::memcpy(Ly, ly, sizeof(ly));
}
void transform_to_char(const double (&L)[NC][NC], double *W) {  // src/equation.h:271-285
  double V[NC];
  V[0] = W[DENS];
  V[NC - 1] = W[ENER];
  for (int d = 0; d < 2; ++d) V[d + 1] = W[d];
  for (int i = 0; i < NC; ++i) W[i] = 0;
  for (int i = 0; i < NC; ++i)
    for (int j = 0; j < NC; ++j) W[i] += L[i][j] * V[j];
}
void transform_to_con(const double (&R)[NC][NC], double *W) {  // src/equation.h:291-306
  double V[NC] = {0, 0, 0, 0};
  for (int i = 0; i < NC; ++i)
    for (int j = 0; j < NC; ++j) V[i] += R[i][j] * W[j];
  W[DENS] = V[0];
  W[ENER] = V[NC - 1];
  for (int d = 0; d < 2; ++d) W[d] = V[d + 1];
}

double minmod(double a, double b, double c, double Mdx2) {  // src/limiter.cc:15-30
  double aa = std::fabs(a);
  if (aa < Mdx2) return a;
  if (a * b > 0 && b * c > 0) {
    double s = (a > 0) ? 1.0 : -1.0;
    return s * std::min(aa, std::min(std::fabs(b), std::fabs(c)));
  } else
    return 0;
}

// --------------------------------------------------------------------------
// The "ConservationLaw" state
// --------------------------------------------------------------------------
struct BFace {
  int cell, face, id;
};

struct Oracle {
  // mesh
  int n_cells = 0, n_owned = 0, degree = 1, basis = 0, mapping = DFLO_MAP_CARTESIAN;
  std::vector<double> vert;
  std::vector<int> nbr, nbrf;
  std::vector<long long> gid;
  dflo_params_t prm;
  // FE
  ScalarFE fe;
  int N = 2, ns = 4, ndof = 16;
  Rule g1d;
  ShapeTable cellq;        // QGauss<2>(k+1): cell quadrature (src/claw.cc:419-422)
  ShapeTable faceq[4];     // face quadrature seen from each local face
  ShapeTable posx, posy;   // positivity point sets (src/positivity.cc:43-47)
  ShapeTable trap;         // QIterated(QTrapez,3) (src/claw.cc:523)
  ShapeTable support;      // unit support points (Qk) (src/limiter.cc:234)
  // state
  std::vector<double> cur, old, rhs, upd, avg, invM, dtc, shock;
  std::vector<int> lcell, rcell, bcell, tcell;
  std::vector<BFace> bfaces;
  std::vector<int> bface_of;   // [n_cells*4] -> bface index or -1
  std::vector<double> bval[2];
  double ark[3];
  int n_rk = 1;
  double global_dt = 0;
  int error = 0;
  int nthreads = 1;
  std::string msg;

  const double *V(int c, int v) const { return &vert[(size_t)c * 8 + v * 2]; }

  // bilinear (Q1) map: position and jacobian; MappingCartesian for cartesian cells.
  void map(int c, double xi, double eta, double &x, double &y, double (&J)[2][2]) const {
    const double *v0 = V(c, 0), *v1 = V(c, 1), *v2 = V(c, 2), *v3 = V(c, 3);
    if (mapping == DFLO_MAP_CARTESIAN) {
      double hx = v1[0] - v0[0], hy = v2[1] - v0[1];
      x = v0[0] + hx * xi;
      y = v0[1] + hy * eta;
      J[0][0] = hx; J[0][1] = 0; J[1][0] = 0; J[1][1] = hy;
      return;
    }
    double s0 = (1 - xi) * (1 - eta), s1 = xi * (1 - eta), s2 = (1 - xi) * eta, s3 = xi * eta;
    x = s0 * v0[0] + s1 * v1[0] + s2 * v2[0] + s3 * v3[0];
    y = s0 * v0[1] + s1 * v1[1] + s2 * v2[1] + s3 * v3[1];
    for (int d = 0; d < 2; ++d) {
      J[d][0] = (1 - eta) * (v1[d] - v0[d]) + eta * (v3[d] - v2[d]);
      J[d][1] = (1 - xi) * (v2[d] - v0[d]) + xi * (v3[d] - v1[d]);
    }
  }
  double diameter(int c) const {  // longest diagonal
    const double *v0 = V(c, 0), *v1 = V(c, 1), *v2 = V(c, 2), *v3 = V(c, 3);
    double d1 = std::hypot(v3[0] - v0[0], v3[1] - v0[1]);
    double d2 = std::hypot(v2[0] - v1[0], v2[1] - v1[1]);
    return std::max(d1, d2);
  }
  double measure(int c) const {
    const double *v0 = V(c, 0), *v1 = V(c, 1), *v2 = V(c, 2), *v3 = V(c, 3);
    // shoelace over v0,v1,v3,v2
    return 0.5 * std::fabs((v0[0] * v1[1] - v1[0] * v0[1]) + (v1[0] * v3[1] - v3[0] * v1[1]) +
                           (v3[0] * v2[1] - v2[0] * v3[1]) + (v2[0] * v0[1] - v0[0] * v2[1]));
  }
  void center(int c, double &x, double &y) const {
    x = y = 0;
    for (int v = 0; v < 4; ++v) { x += 0.25 * V(c, v)[0]; y += 0.25 * V(c, v)[1]; }
  }
  void face_center(int c, int f, double &x, double &y) const {
    static const int fv[4][2] = {{0, 2}, {1, 3}, {0, 1}, {2, 3}};
    x = 0.5 * (V(c, fv[f][0])[0] + V(c, fv[f][1])[0]);
    y = 0.5 * (V(c, fv[f][0])[1] + V(c, fv[f][1])[1]);
  }
  // face geometry at reference point: JxW factor (|tangent|) and outward unit normal
  void face_geom(int c, int f, double xi, double eta, double &jac, double (&n)[2]) const {
    double x, y, J[2][2];
    map(c, xi, eta, x, y, J);
    double tx, ty;
    if (f < 2) { tx = J[0][1]; ty = J[1][1]; } else { tx = J[0][0]; ty = J[1][0]; }
    jac = std::hypot(tx, ty);
    // outward normal ~ J^{-T} nhat
    double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    double nh[2] = {0, 0};
    if (f == 0) nh[0] = -1; else if (f == 1) nh[0] = 1; else if (f == 2) nh[1] = -1; else nh[1] = 1;
    // J^{-T} = 1/det [[J11, -J10],[-J01, J00]]
    double nx = (J[1][1] * nh[0] - J[1][0] * nh[1]) / det;
    double ny = (-J[0][1] * nh[0] + J[0][0] * nh[1]) / det;
    double nn = std::hypot(nx, ny);
    n[0] = nx / nn;
    n[1] = ny / nn;
  }
};

// ------------------------------------------------------------- setup_system
int setup(Oracle &o) {
  o.N = o.degree + 1;
  o.fe.init(o.degree, o.basis);
  o.ns = o.fe.ns;
  o.ndof = NC * o.ns;
  o.g1d = gauss_rule(o.N);
  o.cellq = tensor_table(o.fe, o.g1d, o.g1d);
  for (int f = 0; f < 4; ++f) o.faceq[f] = face_table(o.fe, o.g1d, f);
  {  // src/positivity.cc:43-47
    unsigned k = o.degree;
    unsigned Ng = (k + 3) % 2 == 0 ? (k + 3) / 2 : (k + 4) / 2;
    Rule gll = gauss_lobatto_rule((int)Ng);
    o.posx = tensor_table(o.fe, gll, o.g1d);
    o.posy = tensor_table(o.fe, o.g1d, gll);
  }
  {  // QIterated(QTrapez<1>(),3): 4 equispaced points
    Rule t;
    t.x = {0.0, 1.0 / 3.0, 2.0 / 3.0, 1.0};
    t.w = {1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0};
    o.trap = tensor_table(o.fe, t, t);
  }
  if (o.basis == DFLO_BASIS_QK) {
    Rule nodes;
    nodes.x = o.fe.nodes;
    nodes.w.assign(o.N, 0.0);
    o.support = tensor_table(o.fe, nodes, nodes);
  }
  // SSP-RK coefficients by degree, src/claw.cc:141-159
  if (o.degree == 0) { o.ark[0] = 0.0; o.n_rk = 1; }
  else if (o.degree == 1) { o.ark[0] = 0.0; o.ark[1] = 1.0 / 2.0; o.n_rk = 2; }
  else { o.ark[0] = 0.0; o.ark[1] = 3.0 / 4.0; o.ark[2] = 1.0 / 3.0; o.n_rk = 3; }
  if (o.prm.n_rk > 0) {
    o.n_rk = o.prm.n_rk;
    if (o.n_rk == 1) o.ark[0] = 0;
    if (o.n_rk == 2) { o.ark[0] = 0; o.ark[1] = 0.5; }
    if (o.n_rk == 3) { o.ark[0] = 0; o.ark[1] = 0.75; o.ark[2] = 1.0 / 3.0; }
  }
  size_t nd = (size_t)o.n_cells * o.ndof;
  o.cur.assign(nd, 0);
  o.old.assign(nd, 0);
  o.rhs.assign(nd, 0);
  o.upd.assign(nd, 0);
  o.avg.assign((size_t)o.n_cells * NC, 0);
  o.dtc.assign(o.n_cells, 0);
  // boundary faces in MeshWorker order
  o.bface_of.assign((size_t)o.n_cells * 4, -1);
  for (int c = 0; c < o.n_owned; ++c)
    for (int f = 0; f < 4; ++f) {
      int nb = o.nbr[c * 4 + f];
      if (nb < 0 && nb != DFLO_NBR_NONE) {
        o.bface_of[c * 4 + f] = (int)o.bfaces.size();
        o.bfaces.push_back({c, f, DFLO_NBR_BOUNDARY_ID(nb)});
      }
    }
  for (int w = 0; w < 2; ++w) o.bval[w].assign(o.bfaces.size() * o.N * NC, 0.0);

  // compute_cartesian_mesh_size, src/claw.cc:197-221
  if (o.mapping == DFLO_MAP_CARTESIAN) {
    const double geom_tol = 1.0e-12;
    for (int c = 0; c < o.n_cells; ++c) {
      double xmin = 1e20, xmax = -1e20, ymin = 1e20, ymax = -1e20;
      for (int f = 0; f < 4; ++f) {
        double x, y;
        o.face_center(c, f, x, y);
        xmin = std::min(xmin, x); xmax = std::max(xmax, x);
        ymin = std::min(ymin, y); ymax = std::max(ymax, y);
      }
      if (!(std::fabs((xmax - xmin) - (ymax - ymin)) < geom_tol)) {
        o.msg = "Cell is not square";
        return DFLO_ERR_NONSQUARE_CELL;
      }
    }
  }
  // compute_inv_mass_matrix, src/claw.cc:229-258 (diagonal only)
  o.invM.assign(nd, 0);
  for (int c = 0; c < o.n_cells; ++c) {
    for (int i = 0; i < o.ndof; ++i) {
      int j = i % o.ns;
      double m = 0;
      for (int q = 0; q < o.cellq.np; ++q) {
        double x, y, J[2][2];
        o.map(c, o.cellq.xi[q], o.cellq.eta[q], x, y, J);
        double JxW = std::fabs(J[0][0] * J[1][1] - J[0][1] * J[1][0]) * o.cellq.w[q];
        double s = o.cellq.v[j * o.cellq.np + q];
        m += s * s * JxW;
      }
      o.invM[(size_t)c * o.ndof + i] = 1.0 / m;
    }
  }
  // l/r/b/t neighbours, src/claw.cc:336-380 (+ periodic: src_mpi/claw.cc:417-465)
  o.lcell.assign(o.n_cells, -1);
  o.rcell.assign(o.n_cells, -1);
  o.bcell.assign(o.n_cells, -1);
  o.tcell.assign(o.n_cells, -1);
  if (o.mapping == DFLO_MAP_CARTESIAN) {
    for (int c = 0; c < o.n_cells; ++c) {
      double dx = o.diameter(c) / std::sqrt(2.0);
      double cx, cy;
      o.center(c, cx, cy);
      for (int f = 0; f < 4; ++f) {
        int nb = o.nbr[c * 4 + f];
        if (nb < 0) continue;
        bool periodic = (o.nbrf[c * 4 + f] & 8) != 0;
        double drx, dry;
        if (!periodic) {
          double nx, ny;
          o.center(nb, nx, ny);
          drx = nx - cx; dry = ny - cy;
          if (drx < -0.5 * dx) o.lcell[c] = nb;
          else if (drx > 0.5 * dx) o.rcell[c] = nb;
          else if (dry < -0.5 * dx) o.bcell[c] = nb;
          else if (dry > 0.5 * dx) o.tcell[c] = nb;
          else { o.msg = "Did not find all neighbours"; return DFLO_ERR_BAD_PARAM; }
        } else {
          double fx, fy;
          o.face_center(c, f, fx, fy);
          drx = fx - cx; dry = fy - cy;
          if (drx < -0.2 * dx) o.lcell[c] = nb;
          else if (drx > 0.2 * dx) o.rcell[c] = nb;
          else if (dry < -0.2 * dx) o.bcell[c] = nb;
          else if (dry > 0.2 * dx) o.tcell[c] = nb;
          else { o.msg = "Did not find all neighbours"; return DFLO_ERR_BAD_PARAM; }
        }
      }
    }
  }
  return DFLO_OK;
}

// ----------------------------------------------- assemble_system callbacks
// "shape_value_component(i,q,c)" of the FESystem: scalar shape of node i%ns if c == i/ns.

// src/assemble_explicit.cc:30-120
void integrate_cell_term_explicit(const Oracle &o, int cell, std::vector<double> &local) {
  const int nq = o.cellq.np, ndof = o.ndof, ns = o.ns;
  std::vector<double> W((size_t)nq * NC);
  typedef double FluxMatrix[NC][2];
  typedef double ForcingVector[NC];
  FluxMatrix *flux = new FluxMatrix[nq];
  ForcingVector *forcing = new ForcingVector[nq];
  std::vector<double> JxW(nq), gxs((size_t)ns * nq), gys((size_t)ns * nq);
  for (int q = 0; q < nq; ++q) {  // FEValues::reinit(cell)
    double x, y, J[2][2];
    o.map(cell, o.cellq.xi[q], o.cellq.eta[q], x, y, J);
    double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    JxW[q] = std::fabs(det) * o.cellq.w[q];
    for (int j = 0; j < ns; ++j) {
      double gx = o.cellq.gx[j * nq + q], gy = o.cellq.gy[j * nq + q];
      // grad = J^{-T} gradhat
      gxs[j * nq + q] = (J[1][1] * gx - J[1][0] * gy) / det;
      gys[j * nq + q] = (-J[0][1] * gx + J[0][0] * gy) / det;
    }
  }
  const double *u = &o.cur[(size_t)cell * ndof];
  for (int q = 0; q < nq; ++q) {
    for (int c = 0; c < NC; ++c) W[q * NC + c] = 0.0;
    for (int i = 0; i < ndof; ++i) {
      const int c = i / ns;
      W[q * NC + c] += u[i] * o.cellq.v[(i % ns) * nq + q];
    }
    flux_matrix(&W[q * NC], flux[q]);
    forcing_vector(&W[q * NC], forcing[q]);
  }
  for (int i = 0; i < ndof; ++i) {
    double F_i = 0;
    const int ci = i / ns, j = i % ns;
    for (int p = 0; p < nq; ++p) {
      F_i -= flux[p][ci][0] * gxs[j * nq + p] * JxW[p];
      F_i -= flux[p][ci][1] * gys[j * nq + p] * JxW[p];
      F_i -= o.prm.gravity * forcing[p][ci] * o.cellq.v[j * nq + p] * JxW[p];
    }
    local[i] -= F_i;
  }
  delete[] forcing;
  delete[] flux;
}

// trace of the discrete solution of `cell` on its local face f at the face points
void face_values(const Oracle &o, int cell, int f, std::vector<double> &W) {
  const ShapeTable &t = o.faceq[f];
  const int nq = t.np, ns = o.ns;
  const double *u = &o.cur[(size_t)cell * o.ndof];
  for (int q = 0; q < nq; ++q) {
    for (int c = 0; c < NC; ++c) W[q * NC + c] = 0.0;
    for (int i = 0; i < o.ndof; ++i) {
      const int c = i / ns;
      W[q * NC + c] += u[i] * t.v[(i % ns) * nq + q];
    }
  }
}

// src/assemble_explicit.cc:127-248
void integrate_boundary_term_explicit(const Oracle &o, int cell, int f, int which, std::vector<double> &local) {
  const ShapeTable &t = o.faceq[f];
  const int nq = t.np, ns = o.ns;
  const int bf = o.bface_of[cell * 4 + f];
  const int kind = o.prm.bc_kind[o.bfaces[bf].id];
  std::vector<double> Wp((size_t)nq * NC), Wm((size_t)nq * NC);
  typedef double NormalFlux[NC];
  NormalFlux *nf = new NormalFlux[nq];
  std::vector<double> JxW(nq);
  face_values(o, cell, f, Wp);
  for (int q = 0; q < nq; ++q) {
    double jac, n[2];
    o.face_geom(cell, f, t.xi[q], t.eta[q], jac, n);
    JxW[q] = jac * t.w[q];
    const double *bv = &o.bval[which][((size_t)bf * nq + q) * NC];
    compute_Wminus(kind, n, &Wp[q * NC], bv, &Wm[q * NC]);
    // both LxF averages are the interior cell's, src/assemble_explicit.cc:200-205
    numerical_normal_flux(o.prm.flux_type, n, &Wp[q * NC], &Wm[q * NC], &o.avg[(size_t)cell * NC],
                          &o.avg[(size_t)cell * NC], nf[q]);
  }
  for (int i = 0; i < o.ndof; ++i) {
    double F_i = 0;
    for (int p = 0; p < nq; ++p) {
      const int ci = i / ns;
      F_i += nf[p][ci] * t.v[(i % ns) * nq + p] * JxW[p];
    }
    local[i] -= F_i;
  }
  delete[] nf;
}

// src/assemble_explicit.cc:256-427.  If one_sided, only the visiting cell receives its
// contribution (periodic faces of the MPI variant are integrated from both sides separately,
// src_mpi/assemble_explicit.cc:186-260).
void integrate_face_term_explicit(const Oracle &o, int cell, int f, int ncell, int nface, bool flip, bool one_sided,
                                  std::vector<double> &local, std::vector<double> &local_n) {
  const ShapeTable &t = o.faceq[f], &tn = o.faceq[nface];
  const int nq = t.np, ns = o.ns;
  std::vector<double> Wp((size_t)nq * NC), Wm((size_t)nq * NC), Wmq((size_t)nq * NC);
  typedef double NormalFlux[NC];
  NormalFlux *nf = new NormalFlux[nq];
  std::vector<double> JxW(nq), JxWn(nq);
  face_values(o, cell, f, Wp);
  face_values(o, ncell, nface, Wmq);
  for (int q = 0; q < nq; ++q) {
    const int qn = flip ? nq - 1 - q : q;
    for (int c = 0; c < NC; ++c) Wm[q * NC + c] = Wmq[qn * NC + c];
    double jac, n[2], jn, nn[2];
    o.face_geom(cell, f, t.xi[q], t.eta[q], jac, n);
    JxW[q] = jac * t.w[q];
    o.face_geom(ncell, nface, tn.xi[qn], tn.eta[qn], jn, nn);
    JxWn[q] = jn * tn.w[qn];
    numerical_normal_flux(o.prm.flux_type, n, &Wp[q * NC], &Wm[q * NC], &o.avg[(size_t)cell * NC],
                          &o.avg[(size_t)ncell * NC], nf[q]);
  }
  for (int i = 0; i < o.ndof; ++i) {
    double F_i = 0;
    for (int p = 0; p < nq; ++p) {
      const int ci = i / ns;
      F_i += nf[p][ci] * t.v[(i % ns) * nq + p] * JxW[p];
    }
    local[i] -= F_i;
  }
  if (!one_sided)
    for (int i = 0; i < o.ndof; ++i) {
      double F_i = 0;
      for (int p = 0; p < nq; ++p) {
        const int ci = i / ns;
        const int pn = flip ? nq - 1 - p : p;
        F_i -= nf[p][ci] * tn.v[(i % ns) * nq + pn] * JxWn[p];
      }
      local_n[i] -= F_i;
    }
  delete[] nf;
}

long long gid_of(const Oracle &o, int c) { return o.gid.empty() ? c : o.gid[c]; }

// src/assemble_explicit.cc:433-452 (MeshWorker::loop + ResidualSimple assembler).
// Like MeshWorker on TBB, the cell/face integrals are computed in parallel into per-cell local
// vectors and then copied into the global vector by one thread, in cell order (deterministic).
void assemble_system(Oracle &o, int which) {
  std::fill(o.rhs.begin(), o.rhs.end(), 0.0);  // right_hand_side = 0
  const int ndof = o.ndof;
  const int chunk = std::max(4096, o.nthreads * 128);
  std::vector<double> loc((size_t)chunk * ndof), locn((size_t)chunk * 4 * ndof);
  std::vector<int> tgt((size_t)chunk * 4);
  for (int c0 = 0; c0 < o.n_cells; c0 += chunk) {
    const int c1 = std::min(o.n_cells, c0 + chunk);
#pragma omp parallel for schedule(dynamic, 8) num_threads(o.nthreads) if (o.nthreads > 1)
    for (int cell = c0; cell < c1; ++cell) {
      const bool owned = cell < o.n_owned;
      std::vector<double> local(ndof, 0.0), local_n(ndof, 0.0);
      for (int f = 0; f < 4; ++f) tgt[(size_t)(cell - c0) * 4 + f] = -1;
      if (owned) integrate_cell_term_explicit(o, cell, local);
      for (int f = 0; f < 4; ++f) {
        const int nb = o.nbr[cell * 4 + f];
        if (nb == DFLO_NBR_NONE) continue;
        if (nb < 0) {
          if (owned) integrate_boundary_term_explicit(o, cell, f, which, local);
          continue;
        }
        const int code = o.nbrf[cell * 4 + f];
        const bool periodic = (code & 8) != 0, flip = (code & 4) != 0;
        const int nface = code & 3;
        if (periodic) {  // boundary callback of the MPI variant: each side integrates its own flux
          if (owned) integrate_face_term_explicit(o, cell, f, nb, nface, flip, true, local, local_n);
          continue;
        }
        // interior face: integrated once, from the cell with the smaller index
        if (!(gid_of(o, cell) < gid_of(o, nb))) continue;
        if (!owned && !(nb < o.n_owned)) continue;
        std::fill(local_n.begin(), local_n.end(), 0.0);
        if (owned) {
          integrate_face_term_explicit(o, cell, f, nb, nface, flip, false, local, local_n);
        } else {  // ghost cell visiting a face shared with an owned cell: only the owned side is kept
          std::vector<double> dummy(ndof, 0.0);
          integrate_face_term_explicit(o, cell, f, nb, nface, flip, false, dummy, local_n);
        }
        if (nb < o.n_owned) {
          tgt[(size_t)(cell - c0) * 4 + f] = nb;
          std::copy(local_n.begin(), local_n.end(), &locn[((size_t)(cell - c0) * 4 + f) * ndof]);
        }
      }
      std::copy(local.begin(), local.end(), &loc[(size_t)(cell - c0) * ndof]);
    }
    for (int cell = c0; cell < c1; ++cell) {  // the "copier"
      if (cell < o.n_owned)
        for (int i = 0; i < ndof; ++i) o.rhs[(size_t)cell * ndof + i] += loc[(size_t)(cell - c0) * ndof + i];
      for (int f = 0; f < 4; ++f) {
        const int nb = tgt[(size_t)(cell - c0) * 4 + f];
        if (nb >= 0)
          for (int i = 0; i < ndof; ++i) o.rhs[(size_t)nb * ndof + i] += locn[((size_t)(cell - c0) * 4 + f) * ndof + i];
      }
    }
  }
}

// src/claw.cc:562-597
void compute_cell_average(Oracle &o) {
  const int nq = o.cellq.np, ns = o.ns;
  for (int cell = 0; cell < o.n_cells; ++cell) {
    double a[NC] = {0, 0, 0, 0};
    const double *u = &o.cur[(size_t)cell * o.ndof];
    for (int q = 0; q < nq; ++q) {
      double x, y, J[2][2];
      o.map(cell, o.cellq.xi[q], o.cellq.eta[q], x, y, J);
      double JxW = std::fabs(J[0][0] * J[1][1] - J[0][1] * J[1][0]) * o.cellq.w[q];
      for (int c = 0; c < NC; ++c) {
        double v = 0;  // get_function_values
        for (int j = 0; j < ns; ++j) v += u[c * ns + j] * o.cellq.v[j * nq + q];
        a[c] += v * JxW;
      }
    }
    double m = o.measure(cell);
    for (int c = 0; c < NC; ++c) o.avg[(size_t)cell * NC + c] = a[c] / m;
  }
}

// src/claw.cc:444-557
double compute_time_step(Oracle &o, double elapsed_time) {
  const int dim = 2;
  if (o.prm.global_time_step && o.prm.cfl <= 0.0) {
    std::fill(o.dtc.begin(), o.dtc.end(), o.prm.time_step);
    o.global_dt = o.prm.time_step;
    return o.global_dt;
  }
  o.global_dt = 1.0e20;
  if (o.mapping == DFLO_MAP_CARTESIAN) {  // compute_time_step_cartesian
    for (int c = 0; c < o.n_owned; ++c) {
      const double h = o.diameter(c) / std::sqrt(1.0 * dim);
      const double *A = &o.avg[(size_t)c * NC];
      const double sonic = sound_speed(A);
      const double density = A[DENS];
      double maxeig = 0.0;
      for (int d = 0; d < dim; ++d) maxeig += (sonic + std::fabs(A[d] / density)) / h;
      o.dtc[c] = o.prm.cfl / maxeig / (2.0 * o.degree + 1.0);
      o.global_dt = std::min(o.global_dt, o.dtc[c]);
    }
  } else {  // compute_time_step_q
    const int nq = o.trap.np, ns = o.ns;
    for (int c = 0; c < o.n_owned; ++c) {
      const double *u = &o.cur[(size_t)c * o.ndof];
      double maxeig = 0.0;
      for (int q = 0; q < nq; ++q) {
        double W[NC];
        for (int k = 0; k < NC; ++k) {
          W[k] = 0;
          for (int j = 0; j < ns; ++j) W[k] += u[k * ns + j] * o.trap.v[j * nq + q];
        }
        maxeig = std::max(maxeig, max_eigenvalue(W));
      }
      const double h = o.diameter(c) / std::sqrt(1.0 * dim);
      o.dtc[c] = o.prm.cfl * h / maxeig / (2.0 * o.degree + 1.0);
      o.global_dt = std::min(o.global_dt, o.dtc[c]);
    }
  }
  if (o.prm.global_time_step) {
    if (o.global_dt > 0 && o.prm.time_step > 0) o.global_dt = std::min(o.global_dt, o.prm.time_step);
    if (elapsed_time + o.global_dt > o.prm.final_time) o.global_dt = o.prm.final_time - elapsed_time;
    std::fill(o.dtc.begin(), o.dtc.end(), o.global_dt);
  }
  return o.global_dt;
}

// compute_shock_indicator (src/indicator.cc:17-30) and compute_shock_indicator_kxrcf (src/indicator.cc:51-198).
// Only the same-level branch (:111-132) exists here: the meshes are not adapted.  Faces with a periodic
// neighbour count as boundary faces, as cell->at_boundary(f) does for them in src_mpi/indicator.cc:94.
// The jump indicator computed alongside (:126-128, :188-195) is written nowhere but its own min/max/avg; omitted.
void compute_shock_indicator(Oracle &o) {
  o.shock.assign(o.n_cells, 0.0);
  if (o.prm.shock_indicator == DFLO_IND_LIMITER) {  // :19-22: mark all cells
    std::fill(o.shock.begin(), o.shock.end(), 1.0e20);
    return;
  }
  const int component = o.prm.shock_indicator == DFLO_IND_DENSITY ? DENS : ENER;   // :70-82
  const int nq = o.N;
  std::vector<double> W((size_t)nq * NC), Wn((size_t)nq * NC);
  for (int c = 0; c < o.n_owned; ++c) {
    double cell_shock_ind = 0, inflow_measure = 0;
    const double *A = &o.avg[(size_t)c * NC];
    const double vel[2] = {A[0] / A[DENS], A[1] / A[DENS]};   // :106-108 velocity of the cell average
    for (int f = 0; f < 4; ++f) {
      const int nb = o.nbr[(size_t)c * 4 + f], code = o.nbrf[(size_t)c * 4 + f];
      if (nb < 0 || (code & 8)) continue;   // boundary (or periodic) face: nothing, :169-174
      const int nface = code & 3;
      const bool flip = (code & 4) != 0;
      const ShapeTable &t = o.faceq[f];
      face_values(o, c, f, W);
      face_values(o, nb, nface, Wn);
      for (int q = 0; q < nq; ++q) {
        const int qn = flip ? nq - 1 - q : q;
        double jac, n[2];
        o.face_geom(c, f, t.xi[q], t.eta[q], jac, n);
        const double JxW = jac * t.w[q];
        const int inflow_status = (vel[0] * n[0] + vel[1] * n[1] < 0);
        cell_shock_ind += inflow_status * (W[q * NC + component] - Wn[qn * NC + component]) * JxW;
        inflow_measure += inflow_status * JxW;
      }
    }
    const double cell_norm = A[component];
    const double denominator = std::pow(o.diameter(c), 0.5 * (o.degree + 1)) * inflow_measure * cell_norm;   // :179-181
    o.shock[c] = std::fabs(cell_shock_ind) / denominator;
  }
}

// src/limiter.cc:225-370
void apply_limiter_TVB_Qk(Oracle &o) {
  if (o.degree == 0) return;
  const int nq = o.cellq.np, ns = o.ns;
  const double beta = o.prm.beta;
  for (int c = 0; c < o.n_owned; ++c) {
    if (!(o.shock[c] > 1.0)) continue;   // src/limiter.cc:263
    const double dx = o.diameter(c) / std::sqrt(2.0);
    const double Mdx2 = o.prm.M * dx * dx;
    double *u = &o.cur[(size_t)c * o.ndof];
    double Dx[NC], Dy[NC], dbx[NC], dfx[NC], dby[NC], dfy[NC], Dxn[NC], Dyn[NC];
    {
      double ag[NC][2];
      for (int i = 0; i < NC; ++i) ag[i][0] = ag[i][1] = 0;
      for (int q = 0; q < nq; ++q) {
        double x, y, J[2][2];
        o.map(c, o.cellq.xi[q], o.cellq.eta[q], x, y, J);
        double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
        double JxW = std::fabs(det) * o.cellq.w[q];
        for (int i = 0; i < NC; ++i) {
          double gx = 0, gy = 0;
          for (int j = 0; j < ns; ++j) {
            double rx = o.cellq.gx[j * nq + q], ry = o.cellq.gy[j * nq + q];
            gx += u[i * ns + j] * (J[1][1] * rx - J[1][0] * ry) / det;
            gy += u[i * ns + j] * (-J[0][1] * rx + J[0][0] * ry) / det;
          }
          ag[i][0] += gx * JxW;
          ag[i][1] += gy * JxW;
        }
      }
      double m = o.measure(c);
      for (int i = 0; i < NC; ++i) {
        Dx[i] = dx * (ag[i][0] / m);
        Dy[i] = dx * (ag[i][1] / m);
      }
    }
    const double *A = &o.avg[(size_t)c * NC];
    for (int i = 0; i < NC; ++i) { dbx[i] = Dx[i]; dfx[i] = Dx[i]; dby[i] = Dy[i]; dfy[i] = Dy[i]; }
    if (o.lcell[c] >= 0) for (int i = 0; i < NC; ++i) dbx[i] = A[i] - o.avg[(size_t)o.lcell[c] * NC + i];
    if (o.rcell[c] >= 0) for (int i = 0; i < NC; ++i) dfx[i] = o.avg[(size_t)o.rcell[c] * NC + i] - A[i];
    if (o.bcell[c] >= 0) for (int i = 0; i < NC; ++i) dby[i] = A[i] - o.avg[(size_t)o.bcell[c] * NC + i];
    if (o.tcell[c] >= 0) for (int i = 0; i < NC; ++i) dfy[i] = o.avg[(size_t)o.tcell[c] * NC + i] - A[i];
    double Rx[NC][NC], Lx[NC][NC], Ry[NC][NC], Ly[NC][NC];
    if (o.prm.char_lim) {
      compute_eigen_matrix(A, Rx, Lx, Ry, Ly);
      transform_to_char(Lx, dbx);
      transform_to_char(Lx, dfx);
      transform_to_char(Ly, dby);
      transform_to_char(Ly, dfy);
      transform_to_char(Lx, Dx);
      transform_to_char(Ly, Dy);
    }
    double change_x = 0, change_y = 0;
    for (int i = 0; i < NC; ++i) {
      Dxn[i] = minmod(Dx[i], beta * dbx[i], beta * dfx[i], Mdx2);
      Dyn[i] = minmod(Dy[i], beta * dby[i], beta * dfy[i], Mdx2);
      change_x += std::fabs(Dxn[i] - Dx[i]);
      change_y += std::fabs(Dyn[i] - Dy[i]);
    }
    change_x /= NC;
    change_y /= NC;
    if (change_x + change_y > 1.0e-10) {
      for (int i = 0; i < NC; ++i) { Dxn[i] /= dx; Dyn[i] /= dx; }
      if (o.prm.char_lim) {
        transform_to_con(Rx, Dxn);
        transform_to_con(Ry, Dyn);
      }
      double cx, cy;
      o.center(c, cx, cy);
      for (int i = 0; i < o.ndof; ++i) {
        int ci = i / ns, j = i % ns;
        double x, y, J[2][2];
        o.map(c, o.support.xi[j], o.support.eta[j], x, y, J);
        u[i] = A[ci] + (x - cx) * Dxn[ci] + (y - cy) * Dyn[ci];
      }
    }
  }
}

// src/limiter.cc:377-516
void apply_limiter_TVB_Pk(Oracle &o) {
  if (o.degree == 0) return;
  const int ns = o.ns;
  static const double sqrt_3 = std::sqrt(3.0);
  const double beta = 0.5 * o.prm.beta;
  for (int c = 0; c < o.n_owned; ++c) {
    if (!(o.shock[c] > 1.0)) continue;   // src/limiter.cc:406
    const double dx = o.diameter(c) / std::sqrt(2.0);
    const double Mdx2 = o.prm.M * dx * dx;
    double *u = &o.cur[(size_t)c * o.ndof];
    double Dx[NC], Dy[NC], dbx[NC], dfx[NC], dby[NC], dfy[NC], Dxn[NC], Dyn[NC];
    for (int i = 0; i < NC; ++i) {
      Dx[i] = u[i * ns + 1] * sqrt_3;
      Dy[i] = u[i * ns + o.degree + 1] * sqrt_3;
    }
    const double ang_mom = Dx[1] - Dy[0];   // angular momentum for square cells = v_x - u_y, src/limiter.cc:453
    const double *A = &o.avg[(size_t)c * NC];
    for (int i = 0; i < NC; ++i) { dbx[i] = Dx[i]; dfx[i] = Dx[i]; dby[i] = Dy[i]; dfy[i] = Dy[i]; }
    if (o.lcell[c] >= 0) for (int i = 0; i < NC; ++i) dbx[i] = A[i] - o.avg[(size_t)o.lcell[c] * NC + i];
    if (o.rcell[c] >= 0) for (int i = 0; i < NC; ++i) dfx[i] = o.avg[(size_t)o.rcell[c] * NC + i] - A[i];
    if (o.bcell[c] >= 0) for (int i = 0; i < NC; ++i) dby[i] = A[i] - o.avg[(size_t)o.bcell[c] * NC + i];
    if (o.tcell[c] >= 0) for (int i = 0; i < NC; ++i) dfy[i] = o.avg[(size_t)o.tcell[c] * NC + i] - A[i];
    double Rx[NC][NC], Lx[NC][NC], Ry[NC][NC], Ly[NC][NC];
    if (o.prm.char_lim) {
      compute_eigen_matrix(A, Rx, Lx, Ry, Ly);
      transform_to_char(Lx, dbx);
      transform_to_char(Lx, dfx);
      transform_to_char(Ly, dby);
      transform_to_char(Ly, dfy);
      transform_to_char(Lx, Dx);
      transform_to_char(Ly, Dy);
    }
    double change_x = 0, change_y = 0;
    for (int i = 0; i < NC; ++i) {
      Dxn[i] = minmod(Dx[i], beta * dbx[i], beta * dfx[i], Mdx2);
      Dyn[i] = minmod(Dy[i], beta * dby[i], beta * dfy[i], Mdx2);
      change_x += std::fabs(Dxn[i] - Dx[i]);
      change_y += std::fabs(Dyn[i] - Dy[i]);
    }
    change_x /= NC;
    change_y /= NC;
    if (change_x + change_y > 1.0e-10) {
      if (o.prm.char_lim) {
        transform_to_con(Rx, Dxn);
        transform_to_con(Ry, Dyn);
      }
      if (o.prm.conserve_angular_momentum) {   // src/limiter.cc:496-500
        Dyn[0] = 0.5 * (Dyn[0] - (ang_mom - Dxn[1]));
        Dxn[1] = ang_mom + Dyn[0];
      }
      for (int i = 0; i < o.ndof; ++i) {
        int ci = i / ns, bi = i % ns;
        if (bi == 1) u[i] = Dxn[ci] / sqrt_3;
        else if (bi == o.degree + 1) u[i] = Dyn[ci] / sqrt_3;
        else if (bi != 0) u[i] = 0.0;
      }
    }
  }
}

void apply_limiter(Oracle &o) {  // src/limiter.cc:36-65
  if (o.prm.limiter_type == DFLO_LIMITER_NONE) return;
  if ((int)o.shock.size() != o.n_cells) compute_shock_indicator(o);   // run() computes it before limiting the IC, src/claw.cc:1000
  if (o.basis == DFLO_BASIS_QK) apply_limiter_TVB_Qk(o);
  else apply_limiter_TVB_Pk(o);
}

// src/positivity.cc:17-208
int apply_positivity_limiter(Oracle &o) {
  if (o.degree == 0) return DFLO_OK;
  const double eps = 1.0e-13;
  for (int c = 0; c < o.n_owned; ++c) {
    const double *A = &o.avg[(size_t)c * NC];
    double eps1 = std::min(A[DENS], pressure(A));
    if (eps1 < eps) {
      o.msg = "Fatal: Negative states";
      return DFLO_ERR_NEGATIVE_MEAN_STATE;
    }
  }
  const int ns = o.ns, np = o.posx.np;
  std::vector<double> rho(np), mx(np), my(np), en(np);
  for (int c = 0; c < o.n_owned; ++c) {
    double *u = &o.cur[(size_t)c * o.ndof];
    const double *A = &o.avg[(size_t)c * NC];
    double rho_min = 1.0e20;
    for (int pass = 0; pass < 2; ++pass) {
      const ShapeTable &t = pass == 0 ? o.posx : o.posy;
      for (int q = 0; q < np; ++q) {
        double v = 0;
        for (int j = 0; j < ns; ++j) v += u[DENS * ns + j] * t.v[j * np + q];
        rho_min = std::min(rho_min, v);
      }
    }
    double density_average = A[DENS];
    double rat = std::fabs(density_average - eps) / (std::fabs(density_average - rho_min) + 1.0e-13);
    double theta1 = std::min(rat, 1.0);
    if (theta1 < 1.0) {
      if (o.basis == DFLO_BASIS_QK) {
        for (int j = 0; j < ns; ++j) u[DENS * ns + j] = theta1 * u[DENS * ns + j] + (1.0 - theta1) * density_average;
      } else {
        for (int j = 1; j < ns; ++j) u[DENS * ns + j] *= theta1;
      }
    }
    double energy_average = A[ENER];
    double mavg[2] = {A[0], A[1]};
    double theta2 = 1.0;
    for (int d = 0; d < 2; ++d) {
      const ShapeTable &t = d == 0 ? o.posx : o.posy;
      for (int q = 0; q < np; ++q) {
        rho[q] = mx[q] = my[q] = en[q] = 0;
        for (int j = 0; j < ns; ++j) {
          double s = t.v[j * np + q];
          rho[q] += u[DENS * ns + j] * s;
          mx[q] += u[0 * ns + j] * s;
          my[q] += u[1 * ns + j] * s;
          en[q] += u[ENER * ns + j] * s;
        }
      }
      for (int q = 0; q < np; ++q) {
        double pre = (gas_gamma - 1.0) * (en[q] - 0.5 * (mx[q] * mx[q] + my[q] * my[q]) / rho[q]);
        if (pre < eps) {
          double drho = rho[q] - density_average;
          double dm[2] = {mx[q] - mavg[0], my[q] - mavg[1]};
          double dE = en[q] - energy_average;
          double a1 = 2.0 * drho * dE - (dm[0] * dm[0] + dm[1] * dm[1]);
          double b1 = 2.0 * drho * (energy_average - eps / (gas_gamma - 1.0)) + 2.0 * density_average * dE -
                      2.0 * (mavg[0] * dm[0] + mavg[1] * dm[1]);
          double c1 = 2.0 * density_average * energy_average - (mavg[0] * mavg[0] + mavg[1] * mavg[1]) -
                      2.0 * eps * density_average / (gas_gamma - 1.0);
          b1 /= a1;
          c1 /= a1;
          double D = std::sqrt(std::fabs(b1 * b1 - 4.0 * c1));
          double t1 = 0.5 * (-b1 - D), t2 = 0.5 * (-b1 + D), tt;
          if (t1 > -1.0e-12 && t1 < 1.0 + 1.0e-12) tt = t1;
          else if (t2 > -1.0e-12 && t2 < 1.0 + 1.0e-12) tt = t2;
          else {
            o.msg = "Problem in positivity limiter";
            return DFLO_ERR_POSITIVITY_NO_ROOT;
          }
          tt = std::min(1.0, tt);
          tt = std::max(0.0, tt);
          if (std::fabs(1.0 - tt) < 1.0e-14) tt = 0.0;
          theta2 = std::min(theta2, tt);
        }
      }
    }
    if (theta2 < 1.0) {
      if (o.basis == DFLO_BASIS_QK) {
        for (int i = 0; i < o.ndof; ++i) u[i] = theta2 * u[i] + (1.0 - theta2) * A[i / ns];
      } else {
        for (int i = 0; i < o.ndof; ++i)
          if (i % ns > 0) u[i] *= theta2;
      }
    }
  }
  return DFLO_OK;
}

// solve(), rk3 branch: src/claw.cc:694-713
void solve_rk(Oracle &o) {
  for (int c = 0; c < o.n_owned; ++c)
    for (int i = 0; i < o.ndof; ++i) {
      size_t k = (size_t)c * o.ndof + i;
      o.upd[k] = o.dtc[c] * o.rhs[k] * o.invM[k];
    }
}

double l2_norm_owned(const Oracle &o) {
  double s = 0;
  size_t n = (size_t)o.n_owned * o.ndof;
  for (size_t k = 0; k < n; ++k) s += o.rhs[k] * o.rhs[k];
  return std::sqrt(s);
}

// one stage of iterate_explicit, src/claw.cc:732-771
int stage(Oracle &o, int rk, double *res_norm) {
  assemble_system(o, rk == 0 ? 0 : 1);
  if (res_norm) *res_norm = l2_norm_owned(o);
  solve_rk(o);
  size_t n = (size_t)o.n_owned * o.ndof;
  for (size_t k = 0; k < n; ++k) o.cur[k] += o.upd[k];                                          // :757
  for (size_t k = 0; k < n; ++k) o.cur[k] = (1.0 - o.ark[rk]) * o.cur[k] + o.ark[rk] * o.old[k];  // sadd :760
  compute_cell_average(o);
  compute_shock_indicator(o);   // :763
  apply_limiter(o);
  if (o.prm.pos_lim) {
    int e = apply_positivity_limiter(o);
    if (e) return e;
  }
  return DFLO_OK;
}


// ==========================================================================
// Optimised CPU twin ("also report the optimised CPU twin for honesty").
// NOT the parity oracle: the same stage as stage() above for axis-aligned Qk cells without limiters, written the
// way a CPU wants it -- collocation (W_q = U_q, sum-factorised gradients), one fused pass per stage (volume term,
// the four face terms, M^-1, SSP combine and the new cell average, 24 B per DoF of memory traffic), no
// temporaries on the heap, every cell owned by one thread (both cells of a face evaluate its flux, nothing is
// scattered).  The pointwise physics is the oracle's own.  tests/test_oracle_assembly.py checks it against
// stage(); bench.py reports it next to the reference-style figure.
// ==========================================================================
template <int N>
struct Twin1D {
  double x[N], w[N], D[N][N], L[2][N];   // D[q][m] = l_m'(x_q), L[s][m] = l_m(s)
};
template <int N>
Twin1D<N> twin_tables() {
  Twin1D<N> t;
  const Rule g = gauss_rule(N);
  double bw[N];
  for (int m = 0; m < N; ++m) {
    t.x[m] = g.x[m];
    t.w[m] = g.w[m];
    bw[m] = 1.0;
    for (int j = 0; j < N; ++j)
      if (j != m) bw[m] /= (g.x[m] - g.x[j]);
  }
  for (int q = 0; q < N; ++q)
    for (int m = 0; m < N; ++m) {
      if (q != m) t.D[q][m] = (bw[m] / bw[q]) / (g.x[q] - g.x[m]);
      else {
        double d = 0;
        for (int j = 0; j < N; ++j)
          if (j != m) d += 1.0 / (g.x[m] - g.x[j]);
        t.D[q][m] = d;
      }
    }
  for (int s = 0; s < 2; ++s)
    for (int m = 0; m < N; ++m) {
      double v = 1.0;
      for (int j = 0; j < N; ++j)
        if (j != m) v *= ((double)s - g.x[j]) / (g.x[m] - g.x[j]);
      t.L[s][m] = v;
    }
  return t;
}

bool twin_supported(const Oracle &o) {
  return o.mapping == DFLO_MAP_CARTESIAN && o.basis == DFLO_BASIS_QK && o.degree >= 1 && o.degree <= 3 &&
         o.n_owned == o.n_cells && o.prm.limiter_type == DFLO_LIMITER_NONE && !o.prm.pos_lim;
}

// node of the line of N nodes that carries face point q of local face f, m-th along the line
template <int N>
inline int twin_node(int f, int m, int q) { return f < 2 ? m + N * q : q + N * m; }

template <int N>
inline void twin_trace(const double *u, const Twin1D<N> &T, int f, int q, double *W) {
  const double *L = T.L[f & 1];
  for (int k = 0; k < NC; ++k) {
    double v = 0;
    for (int m = 0; m < N; ++m) v += L[m] * u[k * N * N + twin_node<N>(f, m, q)];
    W[k] = v;
  }
}

template <int N>
void twin_stage(Oracle &o, int rk, const double *src, const double *old, double *dst, double *res_norm) {
  static const Twin1D<N> T = twin_tables<N>();
  constexpr int NS = N * N, ND = NC * NS;
  const double ark = o.ark[rk];
  const int which = rk == 0 ? 0 : 1, flux_type = o.prm.flux_type;
  const double gravity = o.prm.gravity;
  const double *avg = o.avg.data();
  double *avg2 = o.rhs.data();   // rhs is free here: its first n_cells*4 entries take the new averages
  double res = 0;
#pragma omp parallel for schedule(static) reduction(+ : res) num_threads(o.nthreads) if (o.nthreads > 1)
  for (int c = 0; c < o.n_cells; ++c) {
    const double *u = src + (size_t)c * ND;
    const double h = o.V(c, 1)[0] - o.V(c, 0)[0];
    double R[NC][NS];
    for (int k = 0; k < NC; ++k)
      for (int j = 0; j < NS; ++j) R[k][j] = 0.0;
    // volume term: grad phi_(m,b) at node (a,b) is D[a][m]/h e_x, grad phi_(a,m) at (a,b) is D[b][m]/h e_y, JxW = w_a w_b h^2
    for (int b = 0; b < N; ++b)
      for (int a = 0; a < N; ++a) {
        double W[NC], F[NC][2];
        for (int k = 0; k < NC; ++k) W[k] = u[k * NS + a + N * b];
        flux_matrix(W, F);
        const double wh = T.w[a] * T.w[b] * h;
        for (int k = 0; k < NC; ++k) {
          const double fx = F[k][0] * wh, gy = F[k][1] * wh;
          for (int m = 0; m < N; ++m) {
            R[k][m + N * b] += fx * T.D[a][m];
            R[k][a + N * m] += gy * T.D[b][m];
          }
        }
        if (gravity != 0.0) {
          double S[NC];
          forcing_vector(W, S);
          for (int k = 0; k < NC; ++k) R[k][a + N * b] += gravity * S[k] * wh * h;
        }
      }
    // face terms
    for (int f = 0; f < 4; ++f) {
      const int nb = o.nbr[c * 4 + f];
      if (nb == DFLO_NBR_NONE) continue;
      const double n[2] = {f == 0 ? -1.0 : (f == 1 ? 1.0 : 0.0), f == 2 ? -1.0 : (f == 3 ? 1.0 : 0.0)};
      const double *L = T.L[f & 1];
      for (int q = 0; q < N; ++q) {
        double Wp[NC], Wm[NC], F[NC];
        twin_trace<N>(u, T, f, q, Wp);
        double sgn = -1.0;
        if (nb >= 0) {
          const int code = o.nbrf[c * 4 + f], nf = code & 3;
          const bool periodic = (code & 8) != 0, flip = (code & 4) != 0;
          twin_trace<N>(src + (size_t)nb * ND, T, nf, flip ? N - 1 - q : q, Wm);
          if (periodic || gid_of(o, c) < gid_of(o, nb)) {
            numerical_normal_flux(flux_type, n, Wp, Wm, avg + (size_t)c * NC, avg + (size_t)nb * NC, F);
          } else {  // the face is integrated from the other cell (smaller index): its normal, its order of the states
            const double nn[2] = {-n[0], -n[1]};
            numerical_normal_flux(flux_type, nn, Wm, Wp, avg + (size_t)nb * NC, avg + (size_t)c * NC, F);
            sgn = 1.0;
          }
        } else {
          const int bf = o.bface_of[c * 4 + f];
          compute_Wminus(o.prm.bc_kind[o.bfaces[bf].id], n, Wp, &o.bval[which][((size_t)bf * N + q) * NC], Wm);
          numerical_normal_flux(flux_type, n, Wp, Wm, avg + (size_t)c * NC, avg + (size_t)c * NC, F);
        }
        const double jxw = sgn * T.w[q] * h;
        for (int k = 0; k < NC; ++k) {
          const double fq = F[k] * jxw;
          for (int m = 0; m < N; ++m) R[k][twin_node<N>(f, m, q)] += fq * L[m];
        }
      }
    }
    // M^-1, SSP combine (src/claw.cc:708-710, 757-760), new cell average (:562-597)
    const double dt = o.dtc[c], ih2 = 1.0 / (h * h);
    double *v = dst + (size_t)c * ND;
    const double *uo = old + (size_t)c * ND;
    for (int k = 0; k < NC; ++k) {
      double a = 0;
      for (int b = 0; b < N; ++b)
        for (int aa = 0; aa < N; ++aa) {
          const int j = aa + N * b;
          const double ww = T.w[aa] * T.w[b], r = R[k][j];
          res += r * r;
          double un = u[k * NS + j] + dt * r * (ih2 / ww);
          un = (1.0 - ark) * un + ark * uo[k * NS + j];
          v[k * NS + j] = un;
          a += ww * un;
        }
      avg2[(size_t)c * NC + k] = a;
    }
  }
  std::copy(avg2, avg2 + (size_t)o.n_cells * NC, o.avg.begin());
  if (res_norm) *res_norm = std::sqrt(res);
}

double twin_time_step(Oracle &o) {   // compute_time_step_cartesian (src/claw.cc:476-513), threaded
  double dtmin = 1.0e20;
#pragma omp parallel for schedule(static) reduction(min : dtmin) num_threads(o.nthreads) if (o.nthreads > 1)
  for (int c = 0; c < o.n_cells; ++c) {
    const double h = o.V(c, 1)[0] - o.V(c, 0)[0];
    const double *A = &o.avg[(size_t)c * NC];
    const double sonic = sound_speed(A);
    double maxeig = 0.0;
    for (int d = 0; d < 2; ++d) maxeig += (sonic + std::fabs(A[d] / A[DENS])) / h;
    const double dt = o.prm.cfl / maxeig / (2.0 * o.degree + 1.0);
    o.dtc[c] = dt;
    dtmin = std::min(dtmin, dt);
  }
  if (o.prm.global_time_step) {
    if (o.prm.time_step > 0) dtmin = std::min(dtmin, o.prm.time_step);
    std::fill(o.dtc.begin(), o.dtc.end(), dtmin);
  }
  o.global_dt = dtmin;
  return dtmin;
}

}  // namespace

// ==========================================================================
// C interface (ctypes)
// ==========================================================================
extern "C" {

void *dflo_oracle_create(const dflo_mesh_t *m, const dflo_params_t *p) {
  Oracle *o = new Oracle;
  o->n_cells = m->n_cells;
  o->n_owned = m->n_owned_cells > 0 ? m->n_owned_cells : m->n_cells;
  o->degree = m->degree;
  o->basis = m->basis;
  o->mapping = m->mapping;
  o->vert.assign(m->cell_vertices, m->cell_vertices + (size_t)m->n_cells * 8);
  o->nbr.assign(m->cell_face_neighbor, m->cell_face_neighbor + (size_t)m->n_cells * 4);
  o->nbrf.assign(m->cell_face_neighbor_face, m->cell_face_neighbor_face + (size_t)m->n_cells * 4);
  if (m->cell_global_id) o->gid.assign(m->cell_global_id, m->cell_global_id + m->n_cells);
  o->prm = *p;
  o->error = setup(*o);
  return o;
}
void dflo_oracle_destroy(void *h) { delete (Oracle *)h; }
int dflo_oracle_error(void *h) { return ((Oracle *)h)->error; }
const char *dflo_oracle_message(void *h) { return ((Oracle *)h)->msg.c_str(); }
void dflo_oracle_set_threads(void *h, int n) { ((Oracle *)h)->nthreads = n > 0 ? n : 1; }
int dflo_oracle_n_rk(void *h) { return ((Oracle *)h)->n_rk; }
int dflo_oracle_dofs_per_cell(void *h) { return ((Oracle *)h)->ndof; }
long long dflo_oracle_n_dofs(void *h) { return (long long)((Oracle *)h)->n_cells * ((Oracle *)h)->ndof; }
int dflo_oracle_n_boundary_faces(void *h) { return (int)((Oracle *)h)->bfaces.size(); }

// current_solution = old_solution = u (src/ic.cc:118-120) and the post-IC cell average (src/claw.cc:997)
void dflo_oracle_set_solution(void *h, const double *u) {
  Oracle &o = *(Oracle *)h;
  std::copy(u, u + o.cur.size(), o.cur.begin());
  o.old = o.cur;
  compute_cell_average(o);
}
void dflo_oracle_set_current_only(void *h, const double *u) {
  Oracle &o = *(Oracle *)h;
  std::copy(u, u + o.cur.size(), o.cur.begin());
}
void dflo_oracle_get_solution(void *h, double *u) {
  Oracle &o = *(Oracle *)h;
  std::copy(o.cur.begin(), o.cur.end(), u);
}
void dflo_oracle_get_cell_average(void *h, double *a) {
  Oracle &o = *(Oracle *)h;
  std::copy(o.avg.begin(), o.avg.end(), a);
}
void dflo_oracle_set_cell_average(void *h, const double *a) {
  Oracle &o = *(Oracle *)h;
  std::copy(a, a + o.avg.size(), o.avg.begin());
}
void dflo_oracle_get_inv_mass(void *h, double *m) {
  Oracle &o = *(Oracle *)h;
  std::copy(o.invM.begin(), o.invM.end(), m);
}
void dflo_oracle_boundary_faces(void *h, int *cell, int *face, int *id, double *xy) {
  Oracle &o = *(Oracle *)h;
  for (size_t b = 0; b < o.bfaces.size(); ++b) {
    if (cell) cell[b] = o.bfaces[b].cell;
    if (face) face[b] = o.bfaces[b].face;
    if (id) id[b] = o.bfaces[b].id;
    if (xy) {
      const ShapeTable &t = o.faceq[o.bfaces[b].face];
      for (int q = 0; q < o.N; ++q) {
        double x, y, J[2][2];
        o.map(o.bfaces[b].cell, t.xi[q], t.eta[q], x, y, J);
        xy[(b * o.N + q) * 2 + 0] = x;
        xy[(b * o.N + q) * 2 + 1] = y;
      }
    }
  }
}
void dflo_oracle_set_boundary_values(void *h, int which, const double *v) {
  Oracle &o = *(Oracle *)h;
  std::copy(v, v + o.bval[which].size(), o.bval[which].begin());
}
// support points of the Qk DoFs (VectorTools::interpolate target points, src/ic.cc:104-121)
void dflo_oracle_support_points(void *h, double *xy) {
  Oracle &o = *(Oracle *)h;
  for (int c = 0; c < o.n_cells; ++c)
    for (int j = 0; j < o.ns; ++j) {
      double x, y, J[2][2];
      o.map(c, o.support.xi[j], o.support.eta[j], x, y, J);
      xy[((size_t)c * o.ns + j) * 2 + 0] = x;
      xy[((size_t)c * o.ns + j) * 2 + 1] = y;
    }
}
// cell quadrature points and JxW (for L2 errors / Pk projection in tests)
void dflo_oracle_cell_quadrature(void *h, double *xy, double *jxw) {
  Oracle &o = *(Oracle *)h;
  const int nq = o.cellq.np;
  for (int c = 0; c < o.n_cells; ++c)
    for (int q = 0; q < nq; ++q) {
      double x, y, J[2][2];
      o.map(c, o.cellq.xi[q], o.cellq.eta[q], x, y, J);
      xy[((size_t)c * nq + q) * 2 + 0] = x;
      xy[((size_t)c * nq + q) * 2 + 1] = y;
      jxw[(size_t)c * nq + q] = std::fabs(J[0][0] * J[1][1] - J[0][1] * J[1][0]) * o.cellq.w[q];
    }
}
// shape values of scalar function j at cell quadrature points: [ns][nq]
void dflo_oracle_cell_shape(void *h, double *v) {
  Oracle &o = *(Oracle *)h;
  std::copy(o.cellq.v.begin(), o.cellq.v.end(), v);
}

void dflo_oracle_assemble(void *h, int which, double *rhs_out) {
  Oracle &o = *(Oracle *)h;
  assemble_system(o, which);
  if (rhs_out) std::copy(o.rhs.begin(), o.rhs.end(), rhs_out);
}
void dflo_oracle_compute_cell_average(void *h) { compute_cell_average(*(Oracle *)h); }
double dflo_oracle_compute_time_step(void *h, double elapsed) { return compute_time_step(*(Oracle *)h, elapsed); }
void dflo_oracle_apply_limiter(void *h) {   // run(): compute_shock_indicator(); apply_limiter(); old_solution =
  Oracle &o = *(Oracle *)h;                  // current_solution;  src/claw.cc:1000-1002
  compute_shock_indicator(o);
  apply_limiter(o);
  o.old = o.cur;
}
void dflo_oracle_get_shock_indicator(void *h, double *out) {   // as left by the last stage / apply_limiter
  Oracle &o = *(Oracle *)h;
  std::copy(o.shock.begin(), o.shock.end(), out);
}
void dflo_oracle_compute_shock_indicator(void *h, double *out) {
  Oracle &o = *(Oracle *)h;
  compute_shock_indicator(o);
  if (out) std::copy(o.shock.begin(), o.shock.end(), out);
}
int dflo_oracle_apply_positivity_limiter(void *h) { return apply_positivity_limiter(*(Oracle *)h); }

void dflo_oracle_set_dt(void *h, double dt) {
  Oracle &o = *(Oracle *)h;
  o.global_dt = dt;
  std::fill(o.dtc.begin(), o.dtc.end(), dt);
}
int dflo_oracle_stage(void *h, int rk, double *res_norm) { return stage(*(Oracle *)h, rk, res_norm); }
void dflo_oracle_end_step(void *h) {
  Oracle &o = *(Oracle *)h;
  o.old = o.cur;  // src/claw.cc:1110
}
// iterate_explicit (src/claw.cc:726-772) with the dt set before, then old = current
int dflo_oracle_step(void *h, double dt, double *res_norm0, double *res_norm) {
  Oracle &o = *(Oracle *)h;
  if (dt >= 0) dflo_oracle_set_dt(h, dt);
  double r = 0;
  for (int rk = 0; rk < o.n_rk; ++rk) {
    int e = stage(o, rk, &r);
    if (e) return e;
    if (rk == 0 && res_norm0) *res_norm0 = r;
  }
  if (res_norm) *res_norm = r;
  o.old = o.cur;
  return DFLO_OK;
}

// ---- the optimised CPU twin (bench.py's second CPU figure; see the section above)
int dflo_oracle_twin_supported(void *h) { return twin_supported(*(Oracle *)h) ? 1 : 0; }
// n_steps time steps: dt < 0 -> the CFL step from the cell averages each step; the three state vectors rotate, no
// copy of the state inside the loop (old_solution = current_solution becomes a pointer swap)
int dflo_oracle_twin_advance(void *h, int n_steps, double dt, double *elapsed, double *res_norm) {
  Oracle &o = *(Oracle *)h;
  if (!twin_supported(o)) return DFLO_ERR_BAD_PARAM;
  double *A = o.old.data(), *B = o.cur.data(), *C = o.upd.data();   // A holds u^n
  for (int step = 0; step < n_steps; ++step) {
    if (dt >= 0) dflo_oracle_set_dt(h, dt);
    else twin_time_step(o);
    if (elapsed) *elapsed += o.global_dt;
    for (int rk = 0; rk < o.n_rk; ++rk) {
      const double *src = rk == 0 ? A : B;
      if (o.N == 2) twin_stage<2>(o, rk, src, A, C, res_norm);
      else if (o.N == 3) twin_stage<3>(o, rk, src, A, C, res_norm);
      else twin_stage<4>(o, rk, src, A, C, res_norm);
      std::swap(B, C);   // B: newest stage
    }
    std::swap(A, B);     // u^(n+1)
  }
  const size_t n = o.cur.size();
  if (A != o.cur.data()) std::copy(A, A + n, o.cur.begin());
  if (A != o.old.data()) std::copy(A, A + n, o.old.begin());
  return DFLO_OK;
}

// ---- pointwise functions for the golden-vector tests
void dflo_oracle_numerical_flux(int flux_type, const double *n, const double *Wp, const double *Wm, const double *Ap,
                                const double *Am, double *F) {
  numerical_normal_flux(flux_type, n, Wp, Wm, Ap, Am, F);
}
void dflo_oracle_normal_flux(const double *W, const double *n, double *F) {  // src/equation.h:200-215
  const double p = pressure(W);
  double vdotn = 0;
  for (int d = 0; d < 2; ++d) vdotn += W[d] * n[d];
  vdotn /= W[DENS];
  F[DENS] = W[DENS] * vdotn;
  F[ENER] = (W[ENER] + p) * vdotn;
  for (int d = 0; d < 2; ++d) F[d] = p * n[d] + W[d] * vdotn;
}
void dflo_oracle_flux_matrix(const double *W, double *F /*[4][2]*/) {
  double f[NC][2];
  flux_matrix(W, f);
  std::memcpy(F, f, sizeof(f));
}
void dflo_oracle_compute_Wminus(int kind, const double *n, const double *Wp, const double *bv, double *Wm) {
  compute_Wminus(kind, n, Wp, bv, Wm);
}
void dflo_oracle_eigen(const double *W, double *Rx, double *Lx, double *Ry, double *Ly) {
  double rx[NC][NC], lx[NC][NC], ry[NC][NC], ly[NC][NC];
  compute_eigen_matrix(W, rx, lx, ry, ly);
  std::memcpy(Rx, rx, sizeof(rx));
  std::memcpy(Lx, lx, sizeof(lx));
  std::memcpy(Ry, ry, sizeof(ry));
  std::memcpy(Ly, ly, sizeof(ly));
}
double dflo_oracle_minmod(double a, double b, double c, double Mdx2) { return minmod(a, b, c, Mdx2); }
double dflo_oracle_erf(double x) { return ERF(x); }
void dflo_oracle_gauss(int n, double *x, double *w) {
  Rule r = gauss_rule(n);
  std::copy(r.x.begin(), r.x.end(), x);
  std::copy(r.w.begin(), r.w.end(), w);
}
void dflo_oracle_gauss_lobatto(int n, double *x, double *w) {
  Rule r = gauss_lobatto_rule(n);
  std::copy(r.x.begin(), r.x.end(), x);
  std::copy(r.w.begin(), r.w.end(), w);
}

}  // extern "C"
