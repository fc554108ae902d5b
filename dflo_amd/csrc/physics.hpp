// physics.hpp -- pointwise Euler physics on the device (HIP, gfx950).
//
// Device restatement of EulerEquations<2> (src/equation.h): state W = [mx, my, rho, E]
// (src/equation.h:25-28), gamma = 1.4 (src/equation.cc:33).  Each function cites the
// reference function it replaces.  Branches of the reference that are data dependent
// (HLLC wave pattern, Roe entropy fix) are kept as selects so that a wavefront does not
// diverge.
#pragma once
#include <hip/hip_runtime.h>

namespace dflo {

constexpr double kGamma = 1.4;
constexpr double kG1 = kGamma - 1.0;
constexpr int MX = 0, MY = 1, RHO = 2, EN = 3;

// ---- fp64 reciprocal / square root without the IEEE corner-case scaffolding.
// hipcc expands a/b into v_div_scale x2 + v_rcp + 5 fma + v_div_fmas + v_div_fixup (10 issue slots of
// 4 cycles) and sqrt into 17; states here are finite, positive and far from the exponent limits, so
// v_rcp_f64 / v_rsq_f64 (good to ~2^-24) plus the short refinements below do: they return the correctly rounded
// value on every sample of the accuracy probe (tests/test_gpu_parity.py asserts <= 2 ulp).
__device__ __forceinline__ double frcp(double x) {
  // v_rcp_f64 is good to ~2^-24; one third-order step r (1 + e + e^2), e = 1 - x r, lands on the correctly
  // rounded quotient for every sample of the accuracy probe (tests/test_gpu_parity.py)
  const double r = __builtin_amdgcn_rcp(x);
  const double e = __builtin_fma(-x, r, 1.0);
  const double p = __builtin_fma(e, e, e);
  return __builtin_fma(r, p, r);
}
__device__ __forceinline__ double fsqrt(double x) {  // x > 0 (NaN for x < 0, as std::sqrt)
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  const double h = 0.5 * y;
  double d = __builtin_fma(-g, g, x);   // two corrections g += (x - g^2) / (2 sqrt(x)) with the raw h
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}
__device__ __forceinline__ double fsqrt0(double x) { return x > 0.0 ? fsqrt(x) : 0.0; }
// sqrt(x) and 1/x from one v_rsq_f64: g -> sqrt(|x|), h -> 1/(2 sqrt(|x|)), 1/|x| = 4 h^2 refined once.
// For x < 0 the reference's `W/rho` is an ordinary (negative) quotient while its sqrt(rho) is NaN -- both are
// reproduced: the reciprocal carries the sign of x, the root becomes NaN.
__device__ __forceinline__ void fsqrt_rcp(double x, double &sq, double &rc) {
  const double ax = fabs(x);
  const double y = __builtin_amdgcn_rsq(ax);
  double g = ax * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, ax);
  g = __builtin_fma(d, h, g);
  sq = x < 0.0 ? __builtin_nan("") : g;
  double q = 4.0 * h * h;                       // ~ 1/|x|
  const double e = __builtin_fma(-ax, q, 1.0);  // one Newton step
  rc = __builtin_copysign(__builtin_fma(q, e, q), x);
}

__device__ __forceinline__ double pressure(const double *W) {  // src/equation.h:87-92
  const double ke = (W[MX] * W[MX] + W[MY] * W[MY]) * (0.5 * frcp(W[RHO]));
  return kG1 * (W[EN] - ke);
}

// F(W): x- and y- flux columns, src/equation.h:160-193
__device__ __forceinline__ void flux_xy(const double *W, double *Fx, double *Gy) {
  const double ri = frcp(W[RHO]);
  const double u = W[MX] * ri, v = W[MY] * ri;
  const double p = kG1 * (W[EN] - 0.5 * (W[MX] * u + W[MY] * v));
  Fx[MX] = W[MX] * u + p;
  Fx[MY] = W[MY] * u;
  Gy[MX] = W[MX] * v;
  Gy[MY] = W[MY] * v + p;
  Fx[RHO] = W[MX];
  Gy[RHO] = W[MY];
  Fx[EN] = u * (W[EN] + p);
  Gy[EN] = v * (W[EN] + p);
}
// y column only
__device__ __forceinline__ void flux_y(const double *W, double *Gy) {
  const double ri = frcp(W[RHO]);
  const double u = W[MX] * ri, v = W[MY] * ri;
  const double p = kG1 * (W[EN] - 0.5 * (W[MX] * u + W[MY] * v));
  Gy[MX] = W[MX] * v;
  Gy[MY] = W[MY] * v + p;
  Gy[RHO] = W[MY];
  Gy[EN] = v * (W[EN] + p);
}

// std::min / std::max exactly as the reference calls them, (b < a) ? b : a and (a < b) ? b : a -- not fmin / fmax.
// With a NaN operand (the square root of a negative pressure or density at a face point of an under-resolved
// strong shock) the result depends on the argument order, every later comparison is false, and the branch the
// reference then falls into (e.g. the one-sided supersonic flux of hllc_flux) has to be reproduced.
__device__ __forceinline__ double smin(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ double smax(double a, double b) { return (a < b) ? b : a; }

// |v.n| + c of a (cell average) state, src/equation.h:122-137
__device__ __forceinline__ double max_eigenvalue_n(const double *W, double nx, double ny) {
  const double ri = frcp(W[RHO]);
  const double p = kG1 * (W[EN] - 0.5 * (W[MX] * W[MX] + W[MY] * W[MY]) * ri);
  const double sonic = fsqrt(kGamma * p * ri);
  const double vel = (W[MX] * nx + W[MY] * ny) * ri;
  return fabs(vel) + sonic;
}
// |v| + c, src/equation.h:100-114
__device__ __forceinline__ double max_eigenvalue(const double *W) {
  const double ri = frcp(W[RHO]);
  const double q2 = W[MX] * W[MX] + W[MY] * W[MY];
  const double p = kG1 * (W[EN] - 0.5 * q2 * ri);
  return fsqrt0(q2) * ri + fsqrt(kGamma * p * ri);
}

// (u, v, c) of a cell average: what max_eigenvalue(W, n) = |v.n| + c (src/equation.h:122-137) needs
__device__ __forceinline__ void wave_speed_uvc(const double *W, double *uvc) {
  const double ri = frcp(W[RHO]);
  const double p = kG1 * (W[EN] - 0.5 * (W[MX] * W[MX] + W[MY] * W[MY]) * ri);
  uvc[0] = W[MX] * ri;
  uvc[1] = W[MY] * ri;
  uvc[2] = fsqrt(kGamma * p * ri);
}

// src/equation.h:326-377; lambda comes from the two CELL AVERAGES (src/equation.h:357-359), handed over as
// Ap, Am = (u, v, c) of the averages
__device__ __forceinline__ void lxf_flux(double nx, double ny, const double *Wp, const double *Wm, const double *Ap,
                                         const double *Am, double *F) {
  const double rp = frcp(Wp[RHO]), rm = frcp(Wm[RHO]);
  const double vp = (Wp[MX] * nx + Wp[MY] * ny) * rp;
  const double vm = (Wm[MX] * nx + Wm[MY] * ny) * rm;
  const double pp = kG1 * (Wp[EN] - 0.5 * (Wp[MX] * Wp[MX] + Wp[MY] * Wp[MY]) * rp);
  const double pm = kG1 * (Wm[EN] - 0.5 * (Wm[MX] * Wm[MX] + Wm[MY] * Wm[MY]) * rm);
  const double lambda = smax(fabs(Ap[0] * nx + Ap[1] * ny) + Ap[2], fabs(Am[0] * nx + Am[1] * ny) + Am[2]);
  F[MX] = 0.5 * (pp * nx + Wp[MX] * vp + pm * nx + Wm[MX] * vm);
  F[MY] = 0.5 * (pp * ny + Wp[MY] * vp + pm * ny + Wm[MY] * vm);
  F[RHO] = 0.5 * (Wp[RHO] * vp + Wm[RHO] * vm);
  F[EN] = 0.5 * ((Wp[EN] + pp) * vp + (Wm[EN] + pm) * vm);
#pragma unroll
  for (int c = 0; c < 4; ++c) F[c] += 0.5 * lambda * (Wp[c] - Wm[c]);
}

// src/equation.h:384-464
__device__ __forceinline__ void steger_warming_flux(double nx, double ny, const double *Wp, const double *Wm, double *F) {
  const double n[2] = {nx, ny};
  const double rp = frcp(Wp[RHO]), rm = frcp(Wm[RHO]);
  const double vp = (Wp[MX] * nx + Wp[MY] * ny) * rp;
  const double vm = (Wm[MX] * nx + Wm[MY] * ny) * rm;
  const double q2p = (Wp[MX] * Wp[MX] + Wp[MY] * Wp[MY]) * (rp * rp);
  const double q2m = (Wm[MX] * Wm[MX] + Wm[MY] * Wm[MY]) * (rm * rm);
  const double pp = kG1 * (Wp[EN] - 0.5 * Wp[RHO] * q2p), pm = kG1 * (Wm[EN] - 0.5 * Wm[RHO] * q2m);
  const double cp = fsqrt(kGamma * pp * rp), cm = fsqrt(kGamma * pm * rm);
  const double l1p = smax(vp, 0.0), l2p = smax(vp + cp, 0.0), l3p = smax(vp - cp, 0.0);
  const double ap = 2.0 * kG1 * l1p + l2p + l3p;
  const double fp = Wp[RHO] * (0.5 / kGamma);
  const double l1m = smin(vm, 0.0), l2m = smin(vm + cm, 0.0), l3m = smin(vm - cm, 0.0);
  const double am = 2.0 * kG1 * l1m + l2m + l3m;
  const double fm = Wm[RHO] * (0.5 / kGamma);
  double pf[4], mf[4];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    pf[d] = ap * Wp[d] * rp + cp * (l2p - l3p) * n[d];
    mf[d] = am * Wm[d] * rm + cm * (l2m - l3m) * n[d];
  }
  pf[RHO] = ap;
  pf[EN] = 0.5 * ap * q2p + cp * vp * (l2p - l3p) + cp * cp * (l2p + l3p) * (1.0 / kG1);
  mf[RHO] = am;
  mf[EN] = 0.5 * am * q2m + cm * vm * (l2m - l3m) + cm * cm * (l2m + l3m) * (1.0 / kG1);
#pragma unroll
  for (int c = 0; c < 4; ++c) F[c] = fp * pf[c] + fm * mf[c];
}

// src/equation.h:471-556 (Harten entropy fix delta = 0.1 c, :529-531)
__device__ __forceinline__ void roe_flux(double nx, double ny, const double *Wl, const double *Wr, double *F) {
  const double n[2] = {nx, ny};
  double rls, rrs, ril, rir;
  fsqrt_rcp(Wl[RHO], rls, ril);
  fsqrt_rcp(Wr[RHO], rrs, rir);
  const double fl = rls * frcp(rls + rrs), fr = 1.0 - fl;
  double vl[2], vr[2], vel[2], dv[2];
  double v2l = 0, v2r = 0, vln = 0, vrn = 0, veln = 0, v2 = 0, vdv = 0;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    vl[d] = Wl[d] * ril;
    vr[d] = Wr[d] * rir;
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
  const double pl = kG1 * (Wl[EN] - 0.5 * Wl[RHO] * v2l);
  const double pr = kG1 * (Wr[EN] - 0.5 * Wr[RHO] * v2r);
  const double hl = (kGamma / kG1) * pl * ril + 0.5 * v2l;
  const double hr = (kGamma / kG1) * pr * rir + 0.5 * v2r;
  const double rho = rls * rrs;
  const double h = hl * fl + hr * fr;
  const double c2 = kG1 * (h - 0.5 * v2);
  const double c = fsqrt(c2);
  const double ic2 = frcp(c2);
  const double drho = Wr[RHO] - Wl[RHO];
  const double dp = pr - pl;
  const double dvn = vrn - vln;
  const double a1 = (dp - rho * c * dvn) * (0.5 * ic2);
  const double a2 = drho - dp * ic2;
  const double a3 = (dp + rho * c * dvn) * (0.5 * ic2);
  double l1 = fabs(veln - c);
  const double l2 = fabs(veln);
  double l3 = fabs(veln + c);
  const double delta = 0.1 * c, idelta = 10.0 * frcp(c);
  l1 = (l1 < delta) ? 0.5 * (l1 * l1 * idelta + delta) : l1;
  l3 = (l3 < delta) ? 0.5 * (l3 * l3 * idelta + delta) : l3;
  const double Drho = l1 * a1 + l2 * a2 + l3 * a3;
  const double Den = l1 * a1 * (h - c * veln) + l2 * a2 * 0.5 * v2 + l2 * rho * (vdv - veln * dvn) + l3 * a3 * (h + c * veln);
  F[RHO] = 0.5 * (Wl[RHO] * vln + Wr[RHO] * vrn - Drho);
  F[EN] = 0.5 * (Wl[RHO] * hl * vln + Wr[RHO] * hr * vrn - Den);
  const double pavg = 0.5 * (pl + pr);
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const double Dd = (vel[d] - n[d] * c) * l1 * a1 + vel[d] * l2 * a2 + (dv[d] - n[d] * dvn) * l2 * rho + (vel[d] + n[d] * c) * l3 * a3;
    F[d] = n[d] * pavg + 0.5 * (Wl[d] * vln + Wr[d] * vrn) - 0.5 * Dd;
  }
}

// src/equation.h:565-681.  The four branches (:631-679) are evaluated as: supersonic state and star
// state of the side picked by the sign of s_m, then selected (no wavefront divergence).
__device__ __forceinline__ void hllc_flux(double nx, double ny, const double *Wl, const double *Wr, double *F) {
  const double n[2] = {nx, ny};
  double rls, rrs, ril, rir;
  fsqrt_rcp(Wl[RHO], rls, ril);
  fsqrt_rcp(Wr[RHO], rrs, rir);
  const double fl = rls * frcp(rls + rrs), fr = 1.0 - fl;
  double vl[2], vr[2];
  double v2l = 0, v2r = 0, vln = 0, vrn = 0, veln = 0, v2 = 0;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    vl[d] = Wl[d] * ril;
    vr[d] = Wr[d] * rir;
    v2l += vl[d] * vl[d];
    v2r += vr[d] * vr[d];
    vln += vl[d] * n[d];
    vrn += vr[d] * n[d];
    const double ve = vl[d] * fl + vr[d] * fr;
    veln += ve * n[d];
    v2 += ve * ve;
  }
  const double pl = kG1 * (Wl[EN] - 0.5 * Wl[RHO] * v2l);
  const double pr = kG1 * (Wr[EN] - 0.5 * Wr[RHO] * v2r);
  const double hl = (Wl[EN] + pl) * ril;
  const double hr = (Wr[EN] + pr) * rir;
  const double cl = fsqrt(kGamma * pl * ril);
  const double cr = fsqrt(kGamma * pr * rir);
  const double el = Wl[EN] * ril;
  const double er = Wr[EN] * rir;
  const double h = hl * fl + hr * fr;
  const double c = fsqrt(kG1 * (h - 0.5 * v2));
  const double sl = smin(veln - c, vln - cl);
  const double sr = smax(veln + c, vrn + cr);
  const double sm = (pl - pr - Wl[RHO] * vln * (sl - vln) + Wr[RHO] * vrn * (sr - vrn)) *
                    frcp(Wr[RHO] * (sr - vrn) - Wl[RHO] * (sl - vln));
  const double pstar = Wr[RHO] * (vrn - sr) * (vrn - sm) + pr;
  const bool left = sm >= 0.0;  // which side of the contact the face lies on
  const bool supersonic = left ? (sl > 0.0) : !(sr >= 0.0);
  const double rK = left ? Wl[RHO] : Wr[RHO];
  const double vKn = left ? vln : vrn;
  const double pK = left ? pl : pr;
  const double eK = left ? el : er;
  const double sK = left ? sl : sr;
  const double vK0 = left ? vl[0] : vr[0], vK1 = left ? vl[1] : vr[1];
  // supersonic flux of that side (:634-637 / :674-677)
  const double Fs_rho = rK * vKn;
  const double Fs_0 = rK * vK0 * vKn + pK * n[0];
  const double Fs_1 = rK * vK1 * vKn + pK * n[1];
  const double Fs_en = eK * rK * vKn + pK * vKn;
  // star state flux (:641-652 / :659-670)
  const double dS = sK - sm;
  const double inv = frcp(supersonic ? 1.0 : dS);
  const double sKmu = sK - vKn;
  const double rhoS = rK * sKmu * inv;
  const double ru0 = (rK * vK0 * sKmu + (pstar - pK) * n[0]) * inv;
  const double ru1 = (rK * vK1 * sKmu + (pstar - pK) * n[1]) * inv;
  const double eS = (sKmu * eK * rK - pK * vKn + pstar * sm) * inv;
  F[RHO] = supersonic ? Fs_rho : rhoS * sm;
  F[MX] = supersonic ? Fs_0 : ru0 * sm + pstar * n[0];
  F[MY] = supersonic ? Fs_1 : ru1 * sm + pstar * n[1];
  F[EN] = supersonic ? Fs_en : (eS + pstar) * sm;
}

// exp(x) for x <= 0 (the Gaussians of the kinetic split fluxes).  The arithmetic of the device library's exp() -- x = k ln 2 + r
// with ln 2 in two pieces, its degree-11 polynomial in r (coefficients in hexadecimal below), 2^k by v_ldexp_f64 -- so the value
// is the library's bit for bit; what goes is the re-creation of the ten coefficients in vector registers at every call and the
// overflow cases that cannot occur here (half of the routine's instructions).
__device__ __forceinline__ double fexp_neg(double x) {
  x = x < -1000.0 ? -1000.0 : x;   // exp(-inf) = 0 (2^-1443 underflows to 0 in v_ldexp_f64); a NaN stays a NaN
  const double k = __builtin_rint(x * 0x1.71547652b82fep+0);
  double r = __builtin_fma(-0x1.62e42fefa39efp-1, k, x);
  r = __builtin_fma(-0x1.abc9e3b39803fp-56, k, r);
  double p = __builtin_fma(0x1.ade156a5dcb37p-26, r, 0x1.28af3fca7ab0cp-22);
  p = __builtin_fma(r, p, 0x1.71dee623fde64p-19);
  p = __builtin_fma(r, p, 0x1.a01997c89e6b0p-16);
  p = __builtin_fma(r, p, 0x1.a01a014761f6ep-13);
  p = __builtin_fma(r, p, 0x1.6c16c1852b7b0p-10);
  p = __builtin_fma(r, p, 0x1.1111111122322p-7);
  p = __builtin_fma(r, p, 0x1.55555555502a1p-5);
  p = __builtin_fma(r, p, 0x1.5555555555511p-3);
  p = __builtin_fma(r, p, 0x1.000000000000bp-1);
  p = __builtin_fma(r, p, 1.0);
  p = __builtin_fma(r, p, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)k);
}
// Abramowitz-Stegun 7.1.26 exactly as the reference uses it (src/equation.h:688-709) -- NOT erf()
__device__ __forceinline__ double ERF(double xarg, double gauss /* exp(-xarg^2) */) {
  const double a1 = 0.254829592, a2 = -0.284496736, a3 = 1.421413741, a4 = -1.453152027, a5 = 1.061405429;
  const double p = 0.3275911;
  const double sign = (xarg < 0) ? -1.0 : 1.0;
  const double x = fabs(xarg);
  const double t = frcp(1.0 + p * x);
  const double y = 1.0 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * gauss;
  return sign * y;
}
// src/equation.h:716-751
__device__ __forceinline__ void kinetic_split_flux(double sign, double nx, double ny, const double *W, double *F) {
  const double ri = frcp(W[RHO]);
  const double vdotn = (W[MX] * nx + W[MY] * ny) * ri;
  const double p = kG1 * (W[EN] - 0.5 * (W[MX] * W[MX] + W[MY] * W[MY]) * ri);
  const double beta = 0.5 * W[RHO] * frcp(p);
  const double sb = fsqrt(beta);
  const double s = vdotn * sb;
  const double gauss = fexp_neg(-s * s);
  const double A = 0.5 * (1.0 + sign * ERF(s, gauss));
  const double B = 0.5 * sign * gauss * frcp(1.7724538509055160273 * sb);  // sqrt(pi*beta)
  const double ufact = vdotn * A + B;
  F[MX] = p * nx * A + W[MX] * ufact;
  F[MY] = p * ny * A + W[MY] * ufact;
  F[RHO] = W[RHO] * ufact;
  F[EN] = (W[EN] + p) * vdotn * A + (W[EN] + 0.5 * p) * B;
}
// src/equation.h:758-782
__device__ __forceinline__ void kfvs_flux(double nx, double ny, const double *Wp, const double *Wm, double *F) {
  double pf[4], mf[4];
  kinetic_split_flux(+1.0, nx, ny, Wp, pf);
  kinetic_split_flux(-1.0, nx, ny, Wm, mf);
#pragma unroll
  for (int c = 0; c < 4; ++c) F[c] = pf[c] + mf[c];
}

// flux dispatcher (src/claw.h:271-325) resolved at compile time
template <int FLUX>
__device__ __forceinline__ void numerical_normal_flux(double nx, double ny, const double *Wp, const double *Wm,
                                                      const double *Ap, const double *Am, double *F) {
  if constexpr (FLUX == DFLO_FLUX_LXF) lxf_flux(nx, ny, Wp, Wm, Ap, Am, F);
  else if constexpr (FLUX == DFLO_FLUX_SW) steger_warming_flux(nx, ny, Wp, Wm, F);
  else if constexpr (FLUX == DFLO_FLUX_KFVS) kfvs_flux(nx, ny, Wp, Wm, F);
  else if constexpr (FLUX == DFLO_FLUX_ROE) roe_flux(nx, ny, Wp, Wm, F);
  else hllc_flux(nx, ny, Wp, Wm, F);
}

// ghost state of a boundary face, src/equation.h:942-1033
__device__ __forceinline__ void compute_Wminus(int kind, double nx, double ny, const double *Wp, const double *bv, double *Wm) {
  if (kind == DFLO_BC_INFLOW || kind == DFLO_BC_FARFIELD) {
#pragma unroll
    for (int c = 0; c < 4; ++c) Wm[c] = bv[c];
  } else if (kind == DFLO_BC_OUTFLOW) {
#pragma unroll
    for (int c = 0; c < 4; ++c) Wm[c] = Wp[c];
  } else if (kind == DFLO_BC_PRESSURE) {
    const double ke = (Wp[MX] * Wp[MX] + Wp[MY] * Wp[MY]) * (0.5 * frcp(Wp[RHO]));
    Wm[MX] = Wp[MX];
    Wm[MY] = Wp[MY];
    Wm[RHO] = Wp[RHO];
    Wm[EN] = bv[EN] * (1.0 / kG1) + ke;  // w_3 is read as a pressure, src/equation.h:992
  } else {  // slip: reflect the normal momentum
    const double vdotn = Wp[MX] * nx + Wp[MY] * ny;
    Wm[MX] = Wp[MX] - 2.0 * vdotn * nx;
    Wm[MY] = Wp[MY] - 2.0 * vdotn * ny;
    Wm[RHO] = Wp[RHO];
    Wm[EN] = Wp[EN];
  }
}

// TVB minmod, src/limiter.cc:15-30
__device__ __forceinline__ double minmod(double a, double b, double c, double Mdx2) {
  const double aa = fabs(a);
  if (aa < Mdx2) return a;
  if (a * b > 0 && b * c > 0) {
    const double s = (a > 0) ? 1.0 : -1.0;
    return s * smin(aa, smin(fabs(b), fabs(c)));
  }
  return 0.0;
}

// characteristic projection at a cell mean.  The reference builds Rx,Lx,Ry,Ly
// (src/equation.h:226-265) on [rho,mx,my,E]-ordered vectors and reorders inside
// transform_to_char / transform_to_con (src/equation.h:271-306); here the matrices are
// applied in place to W = [mx,my,rho,E].
struct EigenXY {
  double u, v, c, q2, h, beta, phi2, c2, ic2;
};
__device__ __forceinline__ EigenXY eigen_at(const double *A) {
  EigenXY e;
  const double rho = A[RHO], ri = frcp(rho);
  e.u = A[MX] * ri;
  e.v = A[MY] * ri;
  e.q2 = e.u * e.u + e.v * e.v;
  const double p = kG1 * (A[EN] - 0.5 * rho * e.q2);
  e.c2 = kGamma * p * ri;
  e.c = fsqrt(e.c2);
  e.ic2 = frcp(e.c2);
  e.beta = 0.5 * e.ic2;
  e.phi2 = 0.5 * kG1 * e.q2;
  e.h = e.c2 * (1.0 / kG1) + 0.5 * e.q2;
  return e;
}
// W <- L * W   (dir 0: Lx, dir 1: Ly)
__device__ __forceinline__ void to_char(const EigenXY &e, int dir, double *W) {
  const double V0 = W[RHO], V1 = W[MX], V2 = W[MY], V3 = W[EN];
  const double un = dir == 0 ? e.u : e.v;
  double r0 = (1 - e.phi2 * e.ic2) * V0 + (kG1 * e.u * e.ic2) * V1 + (kG1 * e.v * e.ic2) * V2 + (-kG1 * e.ic2) * V3;
  double r1, r2, r3;
  if (dir == 0) {
    r1 = e.v * V0 + 0.0 * V1 + (-1.0) * V2 + 0.0 * V3;
    r2 = e.beta * (e.phi2 - e.c * un) * V0 + e.beta * (e.c - kG1 * e.u) * V1 + (-e.beta * kG1 * e.v) * V2 + e.beta * kG1 * V3;
    r3 = e.beta * (e.phi2 + e.c * un) * V0 + (-e.beta * (e.c + kG1 * e.u)) * V1 + (-e.beta * kG1 * e.v) * V2 + e.beta * kG1 * V3;
  } else {
    r1 = (-e.u) * V0 + 1.0 * V1 + 0.0 * V2 + 0.0 * V3;
    r2 = e.beta * (e.phi2 - e.c * un) * V0 + (-e.beta * kG1 * e.u) * V1 + e.beta * (e.c - kG1 * e.v) * V2 + e.beta * kG1 * V3;
    r3 = e.beta * (e.phi2 + e.c * un) * V0 + (-e.beta * kG1 * e.u) * V1 + (-e.beta * (e.c + kG1 * e.v)) * V2 + e.beta * kG1 * V3;
  }
  W[0] = r0; W[1] = r1; W[2] = r2; W[3] = r3;  // characteristic variables, reference index order
}
// W <- R * W and back to [mx,my,rho,E]
__device__ __forceinline__ void to_con(const EigenXY &e, int dir, double *W) {
  const double a0 = W[0], a1 = W[1], a2 = W[2], a3 = W[3];
  double V0, V1, V2, V3;
  V0 = a0 + 0.0 * a1 + a2 + a3;
  if (dir == 0) {
    V1 = e.u * a0 + 0.0 * a1 + (e.u + e.c) * a2 + (e.u - e.c) * a3;
    V2 = e.v * a0 + (-1.0) * a1 + e.v * a2 + e.v * a3;
    V3 = 0.5 * e.q2 * a0 + (-e.v) * a1 + (e.h + e.c * e.u) * a2 + (e.h - e.c * e.u) * a3;
  } else {
    V1 = e.u * a0 + 1.0 * a1 + e.u * a2 + e.u * a3;
    V2 = e.v * a0 + 0.0 * a1 + (e.v + e.c) * a2 + (e.v - e.c) * a3;
    V3 = 0.5 * e.q2 * a0 + e.u * a1 + (e.h + e.c * e.v) * a2 + (e.h - e.c * e.v) * a3;
  }
  W[RHO] = V0; W[MX] = V1; W[MY] = V2; W[EN] = V3;
}

}  // namespace dflo
