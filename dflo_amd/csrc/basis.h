// basis.h -- 1-D tables of the collocated Qk element dflo uses:
// Lagrange basis at the N=k+1 Gauss points with Gauss(N) quadrature
// (FE_DGQArbitraryNodes(QGauss<1>(k+1)), src/main.cc:40; quadrature src/claw.cc:419-422).
// Everything the kernels need is a handful of N x N matrices; they are computed once on the
// host in long double and uploaded as kernel arguments.
#pragma once
#include <cmath>
#include <vector>

namespace dflo {

constexpr int kMaxN = 6;     // degree <= 5 (DFLO_MAX_DEGREE; the reference takes any degree, src/main.cc:40, its examples use 1..3)
constexpr int kMaxGLL = 4;   // positivity point set: N_g = 2 (k=1), 3 (k=2,3), 4 (k=4,5)  src/positivity.cc:43
constexpr int kTrap = 4;     // QIterated(QTrapez,3): 4 points per direction, src/claw.cc:523

struct BasisTables {
  int N;
  int Ng;
  double x[kMaxN];            // Gauss nodes on [0,1]
  double w[kMaxN];            // Gauss weights
  double D[kMaxN][kMaxN];     // D[q][a] = l_a'(x_q)
  double L0[kMaxN];           // l_a(0)
  double L1[kMaxN];           // l_a(1)
  double Pg[kMaxGLL][kMaxN];  // l_a(x_g) at the Gauss-Lobatto points
  double Pt[kTrap][kMaxN];    // l_a at 0, 1/3, 2/3, 1
};

inline void legendre_ld(int n, long double t, long double &p, long double &dp) {
  long double p0 = 1.0L, p1 = t;
  if (n == 0) { p = 1.0L; dp = 0.0L; return; }
  for (int k = 2; k <= n; ++k) {
    long double pk = ((2.0L * k - 1.0L) * t * p1 - (k - 1.0L) * p0) / k;
    p0 = p1;
    p1 = pk;
  }
  p = p1;
  dp = n * (t * p1 - p0) / (t * t - 1.0L);
}

inline void gauss01(int n, long double *x, long double *w) {
  const long double pi = 3.14159265358979323846264338327950288L;
  for (int i = 0; i < n; ++i) {
    long double t = -cosl(pi * (i + 0.75L) / (n + 0.5L));
    for (int it = 0; it < 200; ++it) {
      long double p, dp;
      legendre_ld(n, t, p, dp);
      long double dt = p / dp;
      t -= dt;
      if (fabsl(dt) < 1e-19L) break;
    }
    long double p, dp;
    legendre_ld(n, t, p, dp);
    x[i] = 0.5L * (1.0L + t);
    w[i] = 1.0L / ((1.0L - t * t) * dp * dp);
  }
}

inline void gauss_lobatto01(int n, long double *x) {
  const long double pi = 3.14159265358979323846264338327950288L;
  const int m = n - 1;
  x[0] = 0.0L;
  x[n - 1] = 1.0L;
  for (int i = 1; i < n - 1; ++i) {
    long double t = -cosl(pi * i / m);
    for (int it = 0; it < 200; ++it) {
      long double p, dp;
      legendre_ld(m, t, p, dp);
      long double ddp = (2.0L * t * dp - m * (m + 1.0L) * p) / (1.0L - t * t);
      long double dt = dp / ddp;
      t -= dt;
      if (fabsl(dt) < 1e-19L) break;
    }
    x[i] = 0.5L * (1.0L + t);
  }
}

inline long double lagrange_ld(int N, const long double *nodes, int a, long double x) {
  long double v = 1.0L;
  for (int m = 0; m < N; ++m)
    if (m != a) v *= (x - nodes[m]) / (nodes[a] - nodes[m]);
  return v;
}
inline long double dlagrange_ld(int N, const long double *nodes, int a, long double x) {
  long double s = 0.0L;
  for (int j = 0; j < N; ++j) {
    if (j == a) continue;
    long double v = 1.0L / (nodes[a] - nodes[j]);
    for (int m = 0; m < N; ++m)
      if (m != a && m != j) v *= (x - nodes[m]) / (nodes[a] - nodes[m]);
    s += v;
  }
  return s;
}

inline BasisTables make_basis(int degree) {
  BasisTables b{};
  const int N = degree + 1;
  b.N = N;
  b.Ng = ((degree + 3) % 2 == 0) ? (degree + 3) / 2 : (degree + 4) / 2;  // src/positivity.cc:43
  long double x[kMaxN], w[kMaxN], g[kMaxGLL + 1];
  gauss01(N, x, w);
  gauss_lobatto01(b.Ng, g);
  for (int a = 0; a < N; ++a) {
    b.x[a] = (double)x[a];
    b.w[a] = (double)w[a];
    b.L0[a] = (double)lagrange_ld(N, x, a, 0.0L);
    b.L1[a] = (double)lagrange_ld(N, x, a, 1.0L);
    for (int q = 0; q < N; ++q) b.D[q][a] = (double)dlagrange_ld(N, x, a, x[q]);
    for (int p = 0; p < b.Ng; ++p) b.Pg[p][a] = (double)lagrange_ld(N, x, a, g[p]);
    for (int p = 0; p < kTrap; ++p) b.Pt[p][a] = (double)lagrange_ld(N, x, a, (long double)p / 3.0L);
  }
  return b;
}


// ---- compile-time tables for the device kernels: with the 1-D tables as constants the compiler
// materialises them with SALU moves instead of re-loading kernel arguments through the scalar cache
// (each such reload is an s_waitcnt lgkmcnt stall in the hot loops).
template <int N> struct GaussLit;
template <> struct GaussLit<1> {   // degree 0: one midpoint node, the finite-volume limit (src/claw.cc:141-145: one RK stage)
  static constexpr double x[1] = {0.5};
  static constexpr double w[1] = {1.0};
};
template <> struct GaussLit<2> {
  static constexpr double x[2] = {0.2113248654051871177454, 0.7886751345948128822545};
  static constexpr double w[2] = {0.5, 0.5};
};
template <> struct GaussLit<3> {
  static constexpr double x[3] = {0.1127016653792583114820, 0.5, 0.8872983346207416885179};
  static constexpr double w[3] = {0.2777777777777777777777, 0.4444444444444444444444, 0.2777777777777777777777};
};
template <> struct GaussLit<4> {
  static constexpr double x[4] = {0.0694318442029737123880, 0.3300094782075718675986, 0.6699905217924281324013,
                                  0.9305681557970262876119};
  static constexpr double w[4] = {0.1739274225687269286865, 0.3260725774312730713134, 0.3260725774312730713134,
                                  0.1739274225687269286865};
};

template <> struct GaussLit<5> {
  static constexpr double x[5] = {0.04691007703066800360118656, 0.2307653449471584544818428, 0.5, 0.7692346550528415455181572,
                                  0.9530899229693319963988134};
  static constexpr double w[5] = {0.118463442528094543757132, 0.2393143352496832340206458, 0.2844444444444444444444444,
                                  0.2393143352496832340206458, 0.118463442528094543757132};
};
template <> struct GaussLit<6> {
  static constexpr double x[6] = {0.03376524289842398609384922, 0.1693953067668677431693002, 0.3806904069584015456847491,
                                  0.6193095930415984543152509, 0.8306046932331322568306998, 0.9662347571015760139061508};
  static constexpr double w[6] = {0.08566224618958517252014807, 0.1803807865240693037849168, 0.2339569672863455236949352,
                                  0.2339569672863455236949352, 0.1803807865240693037849168, 0.08566224618958517252014807};
};

template <int N>
struct CBTable {
  double x[N], w[N], iw[N], L0[N], L1[N], D[N][N], DW[N][N];
};
template <int N>
constexpr double cb_lag(int a, double t) {
  double v = 1.0;
  for (int m = 0; m < N; ++m)
    if (m != a) v *= (t - GaussLit<N>::x[m]) / (GaussLit<N>::x[a] - GaussLit<N>::x[m]);
  return v;
}
template <int N>
constexpr double cb_dlag(int a, double t) {
  double s = 0.0;
  for (int j = 0; j < N; ++j) {
    if (j == a) continue;
    double v = 1.0 / (GaussLit<N>::x[a] - GaussLit<N>::x[j]);
    for (int m = 0; m < N; ++m)
      if (m != a && m != j) v *= (t - GaussLit<N>::x[m]) / (GaussLit<N>::x[a] - GaussLit<N>::x[m]);
    s += v;
  }
  return s;
}
template <int N>
constexpr CBTable<N> make_cb() {
  CBTable<N> t{};
  for (int a = 0; a < N; ++a) {
    t.x[a] = GaussLit<N>::x[a];
    t.w[a] = GaussLit<N>::w[a];
    t.iw[a] = 1.0 / GaussLit<N>::w[a];
    t.L0[a] = cb_lag<N>(a, 0.0);
    t.L1[a] = cb_lag<N>(a, 1.0);
    for (int q = 0; q < N; ++q) {
      t.D[q][a] = cb_dlag<N>(a, GaussLit<N>::x[q]);     // l_a'(x_q)
      t.DW[q][a] = t.D[q][a] * GaussLit<N>::w[q];
    }
  }
  return t;
}
template <int N>
struct CB {
  static constexpr CBTable<N> t = make_cb<N>();
};


// ---- Pk (FE_DGP) basis: psi_m(xi, eta) = Pt_i(xi) Pt_j(eta), Pt_n(x) = sqrt(2n+1) P_n(2x-1), orthonormal on
// the unit square, modes ordered "for j: for i <= k-j" (src/claw.cc:107-113).  Since P_k is a subspace of
// Q_k, a P_k function is represented exactly by its values at the N x N Gauss nodes and
//   psi_m = sum_j psi_m(x_j) phi_j      (phi_j: the collocated Q_k Lagrange functions),
// so the modal residual is T^T times the nodal (Q_k) residual with T[j][m] = psi_m(x_j).
constexpr double kSqrtOdd[6] = {1.0, 1.7320508075688772935, 2.2360679774997896964, 2.6457513110645905905, 3.0, 3.3166247903553998491};
constexpr double legendre01(int n, double x) {
  const double t = 2.0 * x - 1.0;
  double p0 = 1.0, p1 = t;
  if (n == 0) return 1.0;
  for (int k = 2; k <= n; ++k) {
    const double pk = ((2.0 * k - 1.0) * t * p1 - (k - 1.0) * p0) / k;
    p0 = p1;
    p1 = pk;
  }
  return kSqrtOdd[n] * p1;
}
template <int N>
struct PBTable {
  static constexpr int NS = N * N, NM = N * (N + 1) / 2;
  int mi[NM], mj[NM];
  double T[NS][NM];     // psi_m at the Gauss node j = a + N b
  double P0[N], P1[N];  // Pt_n(0), Pt_n(1)
  double Px[N][N];      // Px[q][n] = Pt_n(x_q)
};
template <int N>
constexpr PBTable<N> make_pb() {
  PBTable<N> t{};
  int m = 0;
  for (int j = 0; j < N; ++j)
    for (int i = 0; i < N - j; ++i) {
      t.mi[m] = i;
      t.mj[m] = j;
      ++m;
    }
  for (int n = 0; n < N; ++n) {
    t.P0[n] = legendre01(n, 0.0);
    t.P1[n] = legendre01(n, 1.0);
    for (int q = 0; q < N; ++q) t.Px[q][n] = legendre01(n, GaussLit<N>::x[q]);
  }
  for (int b = 0; b < N; ++b)
    for (int a = 0; a < N; ++a)
      for (int mm = 0; mm < PBTable<N>::NM; ++mm) t.T[a + N * b][mm] = t.Px[a][t.mi[mm]] * t.Px[b][t.mj[mm]];
  return t;
}
template <int N>
struct PB {
  static constexpr PBTable<N> t = make_pb<N>();
};

}  // namespace dflo
