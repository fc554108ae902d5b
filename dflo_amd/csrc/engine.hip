// engine.hip -- MI355X (gfx950) explicit DG residual + SSP-RK engine behind include/dflo_hip.h.
//
// One RK stage of dflo's iterate_explicit (src/claw.cc:732-771) is ONE kernel launch:
//   assemble_system (volume + boundary + interior faces, src/assemble_explicit.cc:30-452)
//   -> dt * M^-1 (src/claw.cc:702-711) -> SSP combine (src/claw.cc:757-760)
//   -> cell average (src/claw.cc:562-597) -> CFL partial minimum (src/claw.cc:486-511)
// followed, when enabled, by one limiter launch (TVB src/limiter.cc:225-370 + positivity
// src/positivity.cc:17-208).
//
// Data layout in HBM: cells are grouped in shards of 64; a shard stores its DoFs
// structure-of-arrays, U[(shard*ndof + dof)*64 + lane], so that one wavefront (lane = cell)
// moves every DoF with a single fully coalesced 512-byte access and the per-cell arithmetic
// (collocated Qk: W at a quadrature point IS a DoF, src/main.cc:40 + src/claw.cc:419-422)
// needs no gather.  Algorithmic HBM traffic per DoF and stage: read u(s), read u(n), write
// u(s+1) = 24 bytes (+16 when a limiter pass runs).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dflo_hip.h"
#include "basis.h"
#include "physics.hpp"
#include "plan.h"

namespace dflo {

// ------------------------------------------------------------------ kernel arguments
struct KBasis {       // 1-D tables, see basis.h
  double w[kMaxN];
  double iw[kMaxN];  // 1 / w
  double x[kMaxN];
  double L0[kMaxN], L1[kMaxN];
  double D[kMaxN][kMaxN];   // D[q][a] = l_a'(x_q)
  double DW[kMaxN][kMaxN];  // D[q][a] * w[q]
  double Pg[kMaxGLL][kMaxN];
  double Pt[kTrap][kMaxN];
  double PLg[kMaxGLL][kMaxN];  // Pk: orthonormal Legendre Pt_n at the Gauss-Lobatto points
  double PLx[kMaxN][kMaxN];    // Pk: Pt_n at the Gauss points
  double pg_neg;               // max over the Gauss-Lobatto points of the sum of the negative weights in Pg
  int Ng;
};

#ifdef DFLO_PHASE_TIMING
#define PHASE_MARK(i) do { if (lane == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define PHASE_MARK(i) do { } while (0)
#endif

struct StageArgs {
  unsigned long long *phase_cycles;  // [grid][4 waves][8], only with DFLO_PHASE_TIMING
  const double *Ucur, *Uold;
  double *Unew;
  const double *avg_cur;
  double *avg_new;
  double *rhs_out;  // parity hook: write the assembled rhs instead of updating
  const int32_t *shard_count;
  const int4 *shard_hdr;      // {cells, faces, halo cells, 0}
  const int32_t *halo_pad;    // [n_shards][halo_pitch]: internal cell slot | local face << 28
  int halo_pitch, halo_stride;
  const uint32_t *faces_pad;  // [n_shards][face_pitch] packed face records (pface_*)
  const int32_t *bnd_pad;     // [n_shards][bnd_pitch] boundary-face index of the shard's l-th boundary face
  int bnd_pitch;
  int face_pitch;
  const uint16_t *cell_face;
  const double *cell_h;
  const double *cell_vert;    // GEO 1: [8][n_slots]
  const double *fgeom_pad;    // GEO 1: [n_shards][3][face_pitch] (nx, ny, length) of each face record
  int n_slots;
  const double *bval;
  const int32_t *bface_kind;
  const double *dt_dev;   // device-resident global dt (used when dt_host < 0)
  const double *dt_cell;  // local time stepping: per internal slot, else null
  double *shard_res, *shard_dtmin;
  double dt_host, ark, gravity, cfl, h_uniform;
  int n_shards, max_fp, max_faces, max_bnd, uniform_h, want_dt, degree, prefetch_ahead;
  const int32_t *shard_list;  // null: all shards; else the n_list shards of this launch (rim / interior)
  int n_list;
  int *flags;   // POS 1: [0] negative mean state, [1] positivity root failure (as LimArgs::flags)
  unsigned long long *lim_mask;   // POS 2: [n_shards] bit = the limiter pass may have something to do in that cell
  double tvb_M;                   // POS 2: TVB constant M, < 0: the limiter pass has no TVB part
  int tvb_char, pos_check;        // POS 2: characteristic limiting; the positivity limiter runs in the pass
  KBasis kb;
};

// compute_time_step_cartesian for one cell, src/claw.cc:495-509
__device__ __forceinline__ double cfl_dt(const double *A, double h, double cfl, int degree) {
  const double sonic = sqrt(kGamma * pressure(A) / A[RHO]);
  const double maxeig = (sonic + fabs(A[MX] / A[RHO])) / h + (sonic + fabs(A[MY] / A[RHO])) / h;
  return cfl / maxeig / (2.0 * degree + 1.0);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}

// Wave-wide sum / minimum through DPP (row shifts inside the rows of 16 lanes, then row_bcast:15 / row_bcast:31): the
// total arrives in lane 63 after six dependent VALU steps, where the shuffle loops above take six round trips through the
// LDS crossbar.  Lanes without a source keep the identity (`old` operand, bound_ctrl off).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double ident, double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_lane63(double v) {
  v += dpp_f64<0x111, 0xf>(0.0, v);   // row_shr:1
  v += dpp_f64<0x112, 0xf>(0.0, v);   // row_shr:2
  v += dpp_f64<0x114, 0xf>(0.0, v);   // row_shr:4
  v += dpp_f64<0x118, 0xf>(0.0, v);   // row_shr:8  -> lane 15 of each row holds the row total
  v += dpp_f64<0x142, 0xa>(0.0, v);   // row_bcast:15 into rows 1 and 3
  v += dpp_f64<0x143, 0xc>(0.0, v);   // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ double wave_min_lane63(double v) {
  constexpr double big = 1.0e300;
  v = fmin(v, dpp_f64<0x111, 0xf>(big, v));
  v = fmin(v, dpp_f64<0x112, 0xf>(big, v));
  v = fmin(v, dpp_f64<0x114, 0xf>(big, v));
  v = fmin(v, dpp_f64<0x118, 0xf>(big, v));
  v = fmin(v, dpp_f64<0x142, 0xa>(big, v));
  v = fmin(v, dpp_f64<0x143, 0xc>(big, v));
  return v;
}

// blockIdx -> shard so that every XCD (block b runs on XCD b % 8) sweeps one contiguous run of
// the Morton-ordered shards: halo re-reads then hit that XCD's own L2.
__device__ __forceinline__ int shard_of_block(int b, int n_shards) {
  const int chunk = (n_shards + 7) >> 3;
  const int s = (b & 7) * chunk + (b >> 3);
  return (b >> 3) < chunk && s < n_shards ? s : -1;
}

// ------------------------------------------------------------------ positivity limiter, pointwise parts
// (shared by limiter_kernel and the stage kernels that apply the limiter on the way out, so that both round alike)
__device__ __forceinline__ double positivity_blend(double theta, double u, double avg) {   // src/positivity.cc:84-87, 196-199
  return fma(theta, u, (1.0 - theta) * avg);
}
// theta of one point W with pressure below eps: root of the pressure along the segment mean -> W (src/positivity.cc:138-178);
// 1 if the pressure is fine there
__device__ __forceinline__ double positivity_theta2(const double (&W)[4], const double (&A)[4], double eps, bool &fail) {
  const double pre = kG1 * (W[EN] - 0.5 * (W[MX] * W[MX] + W[MY] * W[MY]) * frcp(W[RHO]));
  if (!(pre < eps)) return 1.0;
  const double drho = W[RHO] - A[RHO], dmx = W[MX] - A[MX], dmy = W[MY] - A[MY], dE = W[EN] - A[EN];
  const double a1 = 2.0 * drho * dE - (dmx * dmx + dmy * dmy);
  double b1 = 2.0 * drho * (A[EN] - eps / kG1) + 2.0 * A[RHO] * dE - 2.0 * (A[MX] * dmx + A[MY] * dmy);
  double c1 = 2.0 * A[RHO] * A[EN] - (A[MX] * A[MX] + A[MY] * A[MY]) - 2.0 * eps * A[RHO] / kG1;
  b1 /= a1;
  c1 /= a1;
  const double D = sqrt(fabs(b1 * b1 - 4.0 * c1));
  const double t1 = 0.5 * (-b1 - D), t2 = 0.5 * (-b1 + D);
  double t;
  if (t1 > -1.0e-12 && t1 < 1.0 + 1.0e-12) t = t1;
  else if (t2 > -1.0e-12 && t2 < 1.0 + 1.0e-12) t = t2;
  else { fail = true; t = 0.0; }
  t = smin(1.0, t);
  t = smax(0.0, t);
  if (fabs(1.0 - t) < 1.0e-14) t = 0.0;
  return t;
}

// Row b of every cell (wave b) leaves the extremes of its new values in LDS, pb[(2 c + {0: min, 1: max}) N + b][64]; a NaN or
// Inf anywhere in the row turns the density minimum into a NaN.
template <int N, int B>
__device__ __forceinline__ void positivity_row_bounds(double *pb, int lane, const double (&unew)[4][N]) {
  double chk = 0.0, lo_[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double lo = unew[c][0], hi = unew[c][0];
    chk += unew[c][0];
#pragma unroll
    for (int m = 1; m < N; ++m) {
      lo = fmin(lo, unew[c][m]);
      hi = fmax(hi, unew[c][m]);
      chk += unew[c][m];
    }
    lo_[c] = lo;
    pb[((2 * c + 1) * N + B) * 64 + lane] = hi;
  }
  lo_[RHO] += chk - chk;   // 0, or NaN if anything in the row is not finite
#pragma unroll
  for (int c = 0; c < 4; ++c) pb[((2 * c) * N + B) * 64 + lane] = lo_[c];
}

// The cell's nodal box from the row extremes left by positivity_row_bounds, and the test on it: a point value on a line
// through Gauss nodes lies within [lo - d s, hi + d s] (d = hi - lo, s = sum of the negative Gauss-Lobatto interpolation
// weights); if the lowest density and pressure of that box are safely positive the positivity limiter has nothing to do.
template <int N>
__device__ __forceinline__ bool positivity_box_settled(const double *pb, int lane, double sn) {
  double lo[4], hi[4];
  bool fin = true;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    lo[c] = pb[((2 * c) * N) * 64 + lane];
    hi[c] = pb[((2 * c + 1) * N) * 64 + lane];
    if (c == RHO) fin = fin && lo[c] == lo[c];
#pragma unroll
    for (int b = 1; b < N; ++b) {
      const double l = pb[((2 * c) * N + b) * 64 + lane];
      if (c == RHO) fin = fin && l == l;
      lo[c] = fmin(lo[c], l);
      hi[c] = fmax(hi[c], pb[((2 * c + 1) * N + b) * 64 + lane]);
    }
  }
  const double rho_lo = lo[RHO] - (hi[RHO] - lo[RHO]) * sn, e_lo = lo[EN] - (hi[EN] - lo[EN]) * sn;
  const double dmx = (hi[MX] - lo[MX]) * sn, dmy = (hi[MY] - lo[MY]) * sn;
  const double mxa = fmax(fabs(lo[MX] - dmx), fabs(hi[MX] + dmx)), mya = fmax(fabs(lo[MY] - dmy), fabs(hi[MY] + dmy));
  const double p_lo = kG1 * (e_lo - 0.5 * (mxa * mxa + mya * mya) * frcp(rho_lo));
  return fin && rho_lo >= 1.0e-10 + 1.0e-8 * hi[RHO] && p_lo >= 1.0e-10 + 1.0e-8 * fabs(hi[EN]);
}

// ------------------------------------------------------------------ the stage kernel
// One workgroup of N wavefronts per shard: lane = cell, wavefront = node row b of the (k+1)^2
// collocation nodes, so control flow is wave-uniform and every global access is a coalesced
// 512-byte line.  LDS image: Us[ndof (+3: u, v, c of the cell average for LxF)][65] the own 64 cells,
// Th[4N (+3)][halo_stride] the traces of the halo cells on the shared faces, Fh[4][max_fp] the numerical
// fluxes at the shard's face points, the packed face records and the shard's boundary data.

// phase C for node row B of every cell of the shard (lane = cell)
template <int N, int B, int MODE, int POS>
__device__ __forceinline__ void row_update(const StageArgs &a, double *Us, const int S, const double *Fh,
                                           double *red, int shard, int lane, bool active, double h,
                                           const uint16_t (&cref)[4], const double (&uold)[4][N],
                                           const double (&Wrow)[N][4], double (&unew)[4][N], const double dt) {
  constexpr int NS = N * N;
  double R[4][N];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int m = 0; m < N; ++m) R[c][m] = 0.0;
  // volume term (integrate_cell_term_explicit :57-115); collocation: W_q = U_q,
  // grad phi_(m,B)(x_(aa,B)) = D[aa][m]/h e_x, grad phi_(aa,B)(x_(aa,q)) = D[q][B]/h e_y, JxW = w w h^2.
  // Every wave evaluates F and G once at the nodes of its own row (values still in registers), then
  // overwrites its own rows of the LDS image with G: after one barrier each wave reads the G of the
  // other rows instead of re-evaluating the flux there.
  double Gown[N][4];
#pragma unroll
  for (int aa = 0; aa < N; ++aa) {
    double Fx[4];
    flux_xy(Wrow[aa], Fx, Gown[aa]);
    const double wbh = CB<N>::t.w[B] * h;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double fx = Fx[c] * wbh;
#pragma unroll
      for (int m = 0; m < N; ++m) R[c][m] += fx * CB<N>::t.DW[aa][m];
      Us[(c * NS + aa + N * B) * S + lane] = Gown[aa][c];
    }
    if (a.gravity != 0.0) {  // forcing (src/equation.h:831-850): (0, -rho, 0, -my) * gravity
      const double jxw = CB<N>::t.w[aa] * CB<N>::t.w[B] * h * h;
      R[MY][aa] += a.gravity * (-1.0 * Wrow[aa][RHO]) * jxw;
      R[EN][aa] += a.gravity * (-1.0 * Wrow[aa][MY]) * jxw;
    }
  }
  __syncthreads();
#pragma unroll
  for (int aa = 0; aa < N; ++aa) {
    const double wah = CB<N>::t.w[aa] * h;
#pragma unroll
    for (int q = 0; q < N; ++q) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double gy = q == B ? Gown[aa][c] : Us[(c * NS + aa + N * q) * S + lane];
        R[c][aa] += gy * (wah * CB<N>::t.DW[q][B]);
      }
    }
  }
  // face terms (:209-244, :344-423): - flux * phi * JxW on the integrating side, + on the other
  if (active) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const uint16_t ref = cref[f];
      if (ref == kNoFace) continue;
      const int k = ref & 0x3FFF;
      const bool flip = (ref >> 14) & 1;
      const double sgn = (ref >> 15) ? 1.0 : -1.0;
      if (f < 2) {  // x faces: face point q = B lifts to the nodes (m, B)
        const int qq = flip ? N - 1 - B : B;
        const double jxw = sgn * CB<N>::t.w[B] * h;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double fq = Fh[c * a.max_fp + k * N + qq] * jxw;
#pragma unroll
          for (int m = 0; m < N; ++m) R[c][m] += fq * ((f & 1) ? CB<N>::t.L1[m] : CB<N>::t.L0[m]);
        }
      } else {  // y faces: face point q = aa lifts to the node (aa, B) with l_B(0|1)
        const double lw = (f & 1) ? CB<N>::t.L1[B] : CB<N>::t.L0[B];
#pragma unroll
        for (int q = 0; q < N; ++q) {
          const int qq = flip ? N - 1 - q : q;
          const double jxw = sgn * (CB<N>::t.w[q] * lw) * h;
#pragma unroll
          for (int c = 0; c < 4; ++c) R[c][q] += Fh[c * a.max_fp + k * N + qq] * jxw;
        }
      }
    }
  }
  double part[5] = {0, 0, 0, 0, 0};
  if (active) {
    if constexpr (MODE == 2) {
      double *rp = a.rhs_out + (size_t)shard * 4 * NS * 64 + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < N; ++m) rp[(c * NS + m + N * B) * 64] = R[c][m];
    } else {
      // solve() rk3 branch + SSP combine (src/claw.cc:708-710, 757-760)
      const double rh2 = frcp(h * h);
      double *np = a.Unew + (size_t)shard * 4 * NS * 64 + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < N; ++m) {
          const int d = c * NS + m + N * B;
          const double ww = CB<N>::t.w[m] * CB<N>::t.w[B];
          const double invM = rh2 * (CB<N>::t.iw[m] * CB<N>::t.iw[B]);
          part[4] += R[c][m] * R[c][m];
          double u = Wrow[m][c];
          u += dt * R[c][m] * invM;
          if constexpr (MODE == 1) u = (1.0 - a.ark) * u + a.ark * uold[c][m];
          np[d * 64] = u;
          if constexpr (POS) unew[c][m] = u;   // kept for the positivity step of the caller (which stores again if it scales)
          part[c] += ww * u;
        }
    }
  } else if constexpr (POS) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < N; ++m) unew[c][m] = Wrow[m][c];
  }
  // partial cell averages / residual of this row -> LDS (red aliases Fh, see the caller's barriers)
  if constexpr (MODE != 2) {
    __syncthreads();  // every wave is done reading Fh (and the G rows of Us)
#pragma unroll
    for (int c = 0; c < 5; ++c) red[(B * 5 + c) * 64 + lane] = part[c];
    if constexpr (POS == 1) positivity_row_bounds<N, B>(Us, lane, unew);   // the LDS image is free now
    if constexpr (POS == 2) {
      if (a.pos_check) positivity_row_bounds<N, B>(Us, lane, unew);
    }
    if constexpr (POS == 2) {   // x part of "dx * gradient of the cell average" (src/limiter.cc:283-289): l_m(1) - l_m(0) is
                                // antisymmetric in m, pairing the nodes makes the slope of a constant state exactly zero
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double g = 0.0;
#pragma unroll
        for (int m = 0; m < N / 2; ++m) g += (CB<N>::t.L1[m] - CB<N>::t.L0[m]) * (unew[c][m] - unew[c][N - 1 - m]);
        red[(5 * N + c * N + B) * 64 + lane] = CB<N>::t.w[B] * g;
      }
    }
  }
}

// phase C on bilinear (Q1-mapped) cells (the reference gets all of this from
// FEValues with MappingQ1): J = [x_xi x_eta; y_xi y_eta] varies inside the cell,
//   int F.grad(phi) = sum_q w_q [ d(phi)/d(xi) (y_eta F - x_eta G) + d(phi)/d(eta) (-y_xi F + x_xi G) ],
// lumped mass M_j = w_j det J_j (src/claw.cc:223-227), face JxW = w_q |edge|.
template <int N, int B, int MODE, int POS>
__device__ __forceinline__ void row_update_q1(const StageArgs &a, double *Us, const int S, const double *Fh,
                                              const double *Fg, double *red, int shard, int lane, bool active,
                                              const double (&vx)[8], const uint16_t (&cref)[4],
                                              const double (&uold)[4][N], const double (&Wrow)[N][4], double (&unew)[4][N],
                                              const double dt) {
  constexpr int NS = N * N;
  double R[4][N];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int m = 0; m < N; ++m) R[c][m] = 0.0;
  // metric terms of the bilinear map: x_xi depends on eta only, x_eta on xi only
  const double ax = vx[2] - vx[0], bx = (vx[6] - vx[4]) - ax;   // x_xi(eta) = ax + eta bx
  const double ay = vx[3] - vx[1], by = (vx[7] - vx[5]) - ay;
  const double cx = vx[4] - vx[0], dx = (vx[6] - vx[2]) - cx;   // x_eta(xi) = cx + xi dx
  const double cy = vx[5] - vx[1], dy = (vx[7] - vx[3]) - cy;
  // Like row_update: every wave evaluates the fluxes once, at the nodes of its own row (values still in registers),
  // lifts the xi part itself and leaves the eta part, (x_xi G - y_xi F) w w, in its rows of the LDS image; after one
  // barrier each wave reads the eta parts of the other rows instead of evaluating the fluxes there again.
  double Hown[N][4];
  {
    const double xxi = ax + CB<N>::t.x[B] * bx, yxi = ay + CB<N>::t.x[B] * by;
#pragma unroll
    for (int aa = 0; aa < N; ++aa) {
      const double xeta = cx + CB<N>::t.x[aa] * dx, yeta = cy + CB<N>::t.x[aa] * dy;
      double Fx[4], Gy[4];
      flux_xy(Wrow[aa], Fx, Gy);
      const double wq = CB<N>::t.w[aa] * CB<N>::t.w[B];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double f1 = (yeta * Fx[c] - xeta * Gy[c]) * wq;
#pragma unroll
        for (int m = 0; m < N; ++m) R[c][m] += f1 * CB<N>::t.D[aa][m];
        Hown[aa][c] = (xxi * Gy[c] - yxi * Fx[c]) * wq;
        Us[(c * NS + aa + N * B) * S + lane] = Hown[aa][c];
      }
      if (a.gravity != 0.0) {
        const double jxw = wq * (xxi * yeta - xeta * yxi);
        R[MY][aa] += a.gravity * (-1.0 * Wrow[aa][RHO]) * jxw;
        R[EN][aa] += a.gravity * (-1.0 * Wrow[aa][MY]) * jxw;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int aa = 0; aa < N; ++aa)
#pragma unroll
    for (int q = 0; q < N; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double hy = q == B ? Hown[aa][c] : Us[(c * NS + aa + N * q) * S + lane];
        R[c][aa] += hy * CB<N>::t.D[q][B];
      }
  if (active) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const uint16_t ref = cref[f];
      if (ref == kNoFace) continue;
      const int k = ref & 0x3FFF;
      const bool flip = (ref >> 14) & 1;
      const double sgn = (ref >> 15) ? 1.0 : -1.0;
      const double len = Fg[2 * a.max_faces + k];
      if (f < 2) {
        const int qq = flip ? N - 1 - B : B;
        const double jxw = sgn * CB<N>::t.w[B] * len;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double fq = Fh[c * a.max_fp + k * N + qq] * jxw;
#pragma unroll
          for (int m = 0; m < N; ++m) R[c][m] += fq * ((f & 1) ? CB<N>::t.L1[m] : CB<N>::t.L0[m]);
        }
      } else {
        const double lw = (f & 1) ? CB<N>::t.L1[B] : CB<N>::t.L0[B];
#pragma unroll
        for (int q = 0; q < N; ++q) {
          const int qq = flip ? N - 1 - q : q;
          const double jxw = sgn * (CB<N>::t.w[q] * lw) * len;
#pragma unroll
          for (int c = 0; c < 4; ++c) R[c][q] += Fh[c * a.max_fp + k * N + qq] * jxw;
        }
      }
    }
  }
  double part[5] = {0, 0, 0, 0, 0};
  if (active) {
    if constexpr (MODE == 2) {
      double *rp = a.rhs_out + (size_t)shard * 4 * NS * 64 + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < N; ++m) rp[(c * NS + m + N * B) * 64] = R[c][m];
    } else {
      double *np = a.Unew + (size_t)shard * 4 * NS * 64 + lane;
      const double xxi = ax + CB<N>::t.x[B] * bx, yxi = ay + CB<N>::t.x[B] * by;
#pragma unroll
      for (int m = 0; m < N; ++m) {
        const double det = xxi * (cy + CB<N>::t.x[m] * dy) - (cx + CB<N>::t.x[m] * dx) * yxi;
        const double wd = CB<N>::t.w[m] * CB<N>::t.w[B] * det;   // JxW of node (m, B) = its lumped mass
        const double invM = frcp(wd);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int d = c * NS + m + N * B;
          part[4] += R[c][m] * R[c][m];
          double u = Wrow[m][c];
          u += dt * R[c][m] * invM;
          if constexpr (MODE == 1) u = (1.0 - a.ark) * u + a.ark * uold[c][m];
          np[d * 64] = u;
          if constexpr (POS) unew[c][m] = u;
          part[c] += wd * u;
        }
      }
    }
  } else if constexpr (POS) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < N; ++m) unew[c][m] = Wrow[m][c];
  }
  if constexpr (MODE != 2) {
    __syncthreads();  // every wave is done reading Fh and Us
#pragma unroll
    for (int c = 0; c < 5; ++c) red[(B * 5 + c) * 64 + lane] = part[c];
    if constexpr (POS) positivity_row_bounds<N, B>(Us, lane, unew);
  }
}

// phase B: one numerical flux per face point of the shard (integrate_face_term_explicit :303-341,
// integrate_boundary_term_explicit :176-206).  Face-point index p = q * nf + k: neighbouring lanes take
// neighbouring faces at the same q -> same LDS rows, consecutive slots.  Reads LDS only.
// packed face record (4 bytes) of the device tables; plan.h's FaceRec is the host-side form
//   bits 0-8 slot of the integrating cell, 9-10 its local face, 11 boundary, 12 flip,
//   interior: 13-14 local face of the other cell, 15-23 its slot;  boundary: 13-22 index among the shard's boundary faces
__host__ __device__ __forceinline__ uint32_t pface_pack(const FaceRec &r) {
  const uint32_t slot = r.w0 & 0xFFFF, f = (r.w0 >> 16) & 3, bnd = (r.w0 >> 18) & 1, flip = (r.w0 >> 19) & 1;
  uint32_t w = slot | (f << 9) | (bnd << 11) | (flip << 12);
  if (bnd) w |= ((r.w0 >> 20) & 0x3FF) << 13;
  else w |= (((r.w0 >> 20) & 3) << 13) | ((uint32_t)r.w1 << 15);
  return w;
}
__device__ __forceinline__ int pface_slot(uint32_t w) { return w & 0x1FF; }
__device__ __forceinline__ int pface_face(uint32_t w) { return (w >> 9) & 3; }
__device__ __forceinline__ bool pface_bnd(uint32_t w) { return (w >> 11) & 1; }
__device__ __forceinline__ bool pface_flip(uint32_t w) { return (w >> 12) & 1; }
__device__ __forceinline__ int pface_other_face(uint32_t w) { return (w >> 13) & 3; }
__device__ __forceinline__ int pface_other_slot(uint32_t w) { return (w >> 15) & 0x1FF; }
__device__ __forceinline__ int pface_bnd_local(uint32_t w) { return (w >> 13) & 0x3FF; }

template <int N, int FLUX, int GEO>
__device__ __forceinline__ void flux_phase(const StageArgs &a, const double *Us, const double *Th, double *Fh,
                                           const uint32_t *Fr, const double *Bv, const int *Bk, const double *Fg,
                                           const int HS, const int nf, const int tid) {
  constexpr int NS = N * N, NDOF = 4 * NS, NT = 64 * N, S = 65;
  const int nfp = nf * N;
  for (int p = tid; p < nfp; p += NT) {
    const int k = p / N, q = p - k * N;   // the N points of a face sit in consecutive lanes; faces are sorted by kind
    const uint32_t r = Fr[k];
    const int slotL = pface_slot(r), fL = pface_face(r);
    const bool bnd = pface_bnd(r), flip = pface_flip(r);
    const int fR = pface_other_face(r);
    double Wp[4], Wm[4], Ap[4], Am[4], F[4];
    // trace of a cell on its local face f at face point qq: own cells from their DoFs,
    // W = sum_m l_m(0|1) U[m,qq] (x faces) or U[qq,m] (y faces); halo cells from the stored trace
    auto trace = [&](int slot, int f, int qq, double *W, double *A) {
      if (slot < 64) {
        // l_m(1) = l_(N-1-m)(0) (Gauss points are symmetric): walk the line of nodes backwards on the
        // faces at 1 and use the weights l_m(0) throughout -> no per-lane weight selects
        const int str0 = f < 2 ? 1 : N;
        const int base = (f < 2 ? N * qq : qq) + ((f & 1) ? (N - 1) * str0 : 0);
        const int str = (f & 1) ? -str0 : str0;
        const double *u0 = Us + base * S + slot;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double v = 0;
#pragma unroll
          for (int m = 0; m < N; ++m) v += CB<N>::t.L0[m] * u0[(c * NS + m * str) * S];
          W[c] = v;
        }
        if constexpr (FLUX == DFLO_FLUX_LXF) {
#pragma unroll
          for (int c = 0; c < 3; ++c) A[c] = Us[(NDOF + c) * S + slot];
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) W[c] = Th[(c * N + qq) * HS + slot - 64];
        if constexpr (FLUX == DFLO_FLUX_LXF) {
#pragma unroll
          for (int c = 0; c < 3; ++c) A[c] = Th[(4 * N + c) * HS + slot - 64];
        }
      }
    };
    trace(slotL, fL, q, Wp, Ap);
    double nx, ny;  // outward unit normal of the integrating cell
    if constexpr (GEO == 0) {
      nx = fL == 0 ? -1.0 : (fL == 1 ? 1.0 : 0.0);
      ny = fL == 2 ? -1.0 : (fL == 3 ? 1.0 : 0.0);
    } else {
      nx = Fg[k];
      ny = Fg[a.max_faces + k];
    }
    if (!bnd) {
      trace(pface_other_slot(r), fR, flip ? N - 1 - q : q, Wm, Am);
    } else {
      const int bl = pface_bnd_local(r);
      const double *bv = Bv + (bl * N + q) * 4;
      double bvv[4] = {bv[0], bv[1], bv[2], bv[3]};
      compute_Wminus(Bk[bl], nx, ny, Wp, bvv, Wm);
      if constexpr (FLUX == DFLO_FLUX_LXF) {  // both averages are the interior cell's, :200-205
#pragma unroll
        for (int c = 0; c < 3; ++c) Am[c] = Ap[c];
      }
    }
    numerical_normal_flux<FLUX>(nx, ny, Wp, Wm, Ap, Am, F);
#pragma unroll
    for (int c = 0; c < 4; ++c) Fh[c * a.max_fp + k * N + q] = F[c];
  }
}

// One workgroup of N wavefronts per shard.  Occupancy, not software prefetch, hides HBM latency:
// the kernel is kept under 168 VGPRs and ~40 KB of LDS so that 3 wavefronts per SIMD stay resident
// (measured on MI355X, C2: persistent workgroups that prefetch the next shard into registers need
// > 168 VGPRs, run at 2 waves/SIMD and reach 112 GDoF/s against 139 GDoF/s for this kernel).  All global loads of a shard are issued at the top, before
// anything waits.
//   MODE 0: first stage (ark = 0, u(n) not read)   1: later stages   2: residual only (parity hook)
//   GEO 0: axis-aligned squares (MappingCartesian)   1: bilinear cells (MappingQ1)
//   POS 1: apply_positivity_limiter (src/positivity.cc:17-208) on the way out, for runs without the TVB limiter
//   POS 2 (squares, TVB runs): one bit per cell goes out beside its average -- can the limiter pass (TVB, then positivity)
//          change anything in this cell? -- so that the pass reads the DoFs of the marked cells only
template <int N, int FLUX, int MODE, int GEO, int POS>
__global__ __launch_bounds__(64 * N, ((GEO == 1 && N != 3) || N == 4) ? 2 : 3) void stage_kernel(const StageArgs a) {
  constexpr int NS = N * N, NDOF = 4 * NS, NT = 64 * N;
  constexpr int ROWS = NDOF + (FLUX == DFLO_FLUX_LXF ? 3 : 0);   // LxF: (u, v, c) of the cell average ride along
  constexpr int TROWS = 4 * N + (FLUX == DFLO_FLUX_LXF ? 3 : 0); // halo image: face trace (+ the same three)
  constexpr int S = 65;                                          // own-cell row stride: 1 mod 32 doubles
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int sidx = shard_of_block(blockIdx.x, a.n_list);
  if (sidx < 0) return;
  const int shard = a.shard_list ? a.shard_list[sidx] : sidx;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int row = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HS = a.halo_stride;
  double *Us = lds;                                   // [ROWS][S] DoFs (and averages) of the own cells
  double *Th = Us + ROWS * S;                         // [TROWS][HS] traces of the halo cells on the shared face
  double *Fh = Th + TROWS * HS;                       // [4][max_fp] numerical fluxes
  uint32_t *Fr = (uint32_t *)(Fh + 4 * a.max_fp);     // [max_faces] (even count)
  double *Bv = (double *)(Fr + a.max_faces);          // [max_bnd][N][4] boundary values of the shard
  int *Bk = (int *)(Bv + a.max_bnd * 4 * N);          // [max_bnd] boundary kinds
  double *Fg = (double *)(Bk + ((a.max_bnd + 1) & ~1)); // GEO 1: [3][max_faces] unit normal and length of the faces

#ifdef DFLO_PHASE_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
  // ---- index data of a shard that a later workgroup of this XCD will take: touch it now so that its
  //      (dependent) index loads hit L2.  Issued first = oldest in the in-order vmcnt queue; the result
  //      is never used and never waited for.
  //      The destination registers stay reserved until the loads have landed (see the asm further down):
  //      a load issued through inline asm writes its register whenever the data arrives.
  int pf0 = 0, pf1 = 0, pf2 = 0;
  {
    const int ahead = min(shard + a.prefetch_ahead, a.n_shards - 1);
    const int32_t *p0 = a.halo_pad + (size_t)ahead * a.halo_pitch + (tid & 31);
    const uint32_t *p1 = a.faces_pad + (size_t)ahead * a.face_pitch + tid;
    const uint16_t *p2 = a.cell_face + (size_t)ahead * 4 * 64 + 2 * (tid & 127);
    asm volatile("global_load_dword %0, %1, off" : "=v"(pf0) : "v"(p0));
    asm volatile("global_load_dword %0, %1, off" : "=v"(pf1) : "v"(p1));
    asm volatile("global_load_dword %0, %1, off" : "=v"(pf2) : "v"(p2));
  }
  // ---- all loads of the shard, issued back to back; the halo entries first (the halo values depend on them)
  // halo entries: thread t works on entry (t & 31) + 32 b of every block b of 32 entries (8x8 lattice shards have one
  // block, unstructured shards two or three): load them all now, the gathers below then depend on nothing else
  constexpr int HB = 3;
  int hentb[HB];
#pragma unroll
  for (int b = 0; b < HB; ++b) hentb[b] = a.halo_pad[(size_t)shard * a.halo_pitch + min((tid & 31) + 32 * b, a.halo_pitch - 1)];
  const int4 hdr = a.shard_hdr[shard];                // {cells, faces, halo entries, boundary faces}
  const int nf = hdr.y, nh = hdr.z, nbnd = hdr.w;
  const bool active = lane < hdr.x;
  double urow[4][N];                                  // node row `row` of the own cells
  {
    const double *up = a.Ucur + (size_t)shard * NDOF * 64 + (size_t)(N * row) * 64 + lane;   // one base, constant offsets
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < N; ++m) urow[c][m] = up[(c * NS + m) * 64];
  }
  double uavg[4];
  if constexpr (FLUX == DFLO_FLUX_LXF) {
    if (row == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) uavg[c] = a.avg_cur[((size_t)shard * 4 + c) * 64 + lane];
    }
  }
  const uint32_t *fp = a.faces_pad + (size_t)shard * a.face_pitch;
  const uint32_t fr0 = fp[tid], fr1 = fp[tid + NT];
  double fg[3][2];   // GEO 1: unit normal and length of the faces tid and tid + NT (the table has the pitch of the face records)
  if constexpr (GEO == 1) {
    const double *gp = a.fgeom_pad + (size_t)shard * 3 * a.face_pitch;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      fg[j][0] = gp[j * a.face_pitch + tid];
      fg[j][1] = gp[j * a.face_pitch + tid + NT];
    }
  }
  uint16_t cref[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) cref[f] = a.cell_face[((size_t)shard * 4 + f) * 64 + lane];
  double h = 0.0, vx[8];
  if constexpr (GEO == 0) {
    h = a.uniform_h ? a.h_uniform : a.cell_h[(size_t)shard * 64 + lane];
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) vx[k] = a.cell_vert[(size_t)k * a.n_slots + (size_t)shard * 64 + lane];
  }
  // the time step of the update: fetched here with everything else (it used to be read in the middle of phase C, one more
  // trip to memory on every wave's critical path)
  double dt_step = 0.0;
  if constexpr (MODE != 2) dt_step = a.dt_cell ? a.dt_cell[(size_t)shard * 64 + lane] : (a.dt_host >= 0.0 ? a.dt_host : *a.dt_dev);
  double uold[4][N];
  if constexpr (MODE == 1) {
    const double *op = a.Uold + (size_t)shard * NDOF * 64 + (size_t)(N * row) * 64 + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < N; ++m) uold[c][m] = op[(c * NS + m) * 64];
  }

  PHASE_MARK(0);
  // ---- phase A: own rows -> LDS; halo: only the trace on the shared face is kept.
  //      halo item i -> (entry s = i % nh, q = (i / nh) % N, comp = i / (nh N)); an entry is
  //      (internal cell slot | local face << 28) of a face neighbour outside the shard
  //      A thread takes, of its entry in block b, the two (component, point) rows r = g and g + 2N (g = t >> 5):
  //      2N independent loads per block.
  {
    const int g = tid >> 5, l32 = tid & 31;
    for (int b = 0; b * 32 < nh; ++b) {
      const int sl = l32 + 32 * b;
      if (sl >= nh) continue;
      const int e = b == 0 ? hentb[0] : (b == 1 ? hentb[1] : (b == 2 ? hentb[2] : a.halo_pad[(size_t)shard * a.halo_pitch + sl]));
      const int ic = e & 0x0FFFFFFF, f = (e >> 28) & 3;
      const int str0 = f < 2 ? 1 : N, str = (f & 1) ? -str0 : str0;
      double val[2][N];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = g + 2 * N * j, q = r % N, c = r / N;
        const double *hp = a.Ucur + ((size_t)(ic >> 6) * NDOF + c * NS) * 64 + (ic & 63);
        const int base = (f < 2 ? N * q : q) + ((f & 1) ? (N - 1) * str0 : 0);
#pragma unroll
        for (int m = 0; m < N; ++m) val[j][m] = hp[(base + m * str) * 64];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = g + 2 * N * j, q = r % N, c = r / N;
        double v = 0.0;
#pragma unroll
        for (int m = 0; m < N; ++m) v += CB<N>::t.L0[m] * val[j][m];
        Th[(c * N + q) * HS + sl] = v;
      }
    }
  }
  if constexpr (FLUX == DFLO_FLUX_LXF) {  // lambda of the LxF flux comes from the cell averages (src/equation.h:357-359):
                                          // keep (u, v, c) of each average instead of the four components
    for (int sl = tid; sl < nh; sl += NT) {
      const int blk = sl >> 5;   // sl = tid + k NT: entry (tid & 31) + 32 blk is this thread's own preloaded one
      const int ic = (blk == 0 ? hentb[0] : (blk == 1 ? hentb[1] : (blk == 2 ? hentb[2] : a.halo_pad[(size_t)shard * a.halo_pitch + sl]))) & 0x0FFFFFFF;
      double A[4], uvc[3];
#pragma unroll
      for (int c = 0; c < 4; ++c) A[c] = a.avg_cur[((size_t)(ic >> 6) * 4 + c) * 64 + (ic & 63)];
      wave_speed_uvc(A, uvc);
#pragma unroll
      for (int c = 0; c < 3; ++c) Th[(4 * N + c) * HS + sl] = uvc[c];
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int m = 0; m < N; ++m) Us[(c * NS + m + N * row) * S + lane] = urow[c][m];
  asm volatile("" ::"v"(pf0), "v"(pf1), "v"(pf2));  // the touch loads (oldest in the queue) have landed by now
  if constexpr (FLUX == DFLO_FLUX_LXF) {
    if (row == 0) {
      double uvc[3];
      wave_speed_uvc(uavg, uvc);
#pragma unroll
      for (int c = 0; c < 3; ++c) Us[(NDOF + c) * S + lane] = uvc[c];
    }
  }
  if (tid < nf) Fr[tid] = fr0;
  if (tid + NT < nf) Fr[tid + NT] = fr1;
  for (int i = tid + 2 * NT; i < nf; i += NT) Fr[i] = fp[i];
  if constexpr (GEO == 1) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (tid < nf) Fg[j * a.max_faces + tid] = fg[j][0];
      if (tid + NT < nf) Fg[j * a.max_faces + tid + NT] = fg[j][1];
    }
  }
  if (nbnd > 0) {  // boundary values and kinds of this shard's boundary faces
    for (int i = tid; i < nbnd * 4 * N; i += NT) {
      const int bl = i / (4 * N), k2 = i - bl * 4 * N;
      const int bf = a.bnd_pad[(size_t)shard * a.bnd_pitch + bl];
      if (k2 == 0) Bk[bl] = a.bface_kind[bf];
      Bv[i] = a.bval[(size_t)bf * 4 * N + k2];
    }
  }
  PHASE_MARK(1);
  __syncthreads();
  PHASE_MARK(2);

  // ---- phase B
  flux_phase<N, FLUX, GEO>(a, Us, Th, Fh, Fr, Bv, Bk, Fg, HS, nf, tid);
  PHASE_MARK(3);
  __syncthreads();
  PHASE_MARK(4);

  // ---- phase C: volume + lifting + RK update of node row `row`
  double *red = Fh;  // reused after the barrier inside row_update
  double wrow[N][4];
  double unew[4][N];   // POS: the updated row, held back until the positivity step below
#pragma unroll
  for (int m = 0; m < N; ++m)
#pragma unroll
    for (int c = 0; c < 4; ++c) wrow[m][c] = urow[c][m];
#define DFLO_ROW(Bq)                                                                                     \
  do {                                                                                                   \
    if constexpr (GEO == 0) row_update<N, Bq, MODE, POS>(a, Us, S, Fh, red, shard, lane, active, h, cref, uold, wrow, unew, dt_step); \
    else row_update_q1<N, Bq, MODE, POS>(a, Us, S, Fh, Fg, red, shard, lane, active, vx, cref, uold, wrow, unew, dt_step); \
  } while (0)
  if constexpr (N == 2) {
    if (row == 0) DFLO_ROW(0); else DFLO_ROW(1);
  } else if constexpr (N == 3) {
    if (row == 0) DFLO_ROW(0); else if (row == 1) DFLO_ROW(1); else DFLO_ROW(2);
  } else {
    if (row == 0) DFLO_ROW(0); else if (row == 1) DFLO_ROW(1); else if (row == 2) DFLO_ROW(2); else DFLO_ROW(3);
  }
#undef DFLO_ROW
  PHASE_MARK(5);
  if constexpr (MODE == 2) return;
  __syncthreads();
  PHASE_MARK(6);
  if constexpr (POS == 1 && MODE != 2) {
    // ---- apply_positivity_limiter (src/positivity.cc:17-208) on the new state.
    //      First a bound that settles almost every cell: the limiter looks at the solution on lines through the Gauss nodes
    //      (Gauss-Lobatto points on them), and a point value on such a line lies within [lo - d s, hi + d s] of the cell's
    //      nodal extremes (d = hi - lo, s = sum of the negative interpolation weights).  If the lowest density and the lowest
    //      pressure possible in that box are safely positive, theta1 = theta2 = 1 and the mean is admissible (the pressure
    //      is concave): nothing to do.  Only wavefronts with a cell that fails the bound run the limiter proper.
    constexpr int NS2 = N * N;
    bool settled;
    {
      const bool ok = positivity_box_settled<N>(Us, lane, a.kb.pg_neg);
      settled = __all(ok || !active);   // the same in every wave of the workgroup: all of them see the same numbers
    }
    if (!settled) {
    // the limiter proper, the same arithmetic as limiter_kernel: wave b holds row b of every cell in registers and reads
    // column b from the LDS image, so it sees the points (GLL g, Gauss b) and (Gauss b, GLL g); the minima of the rows are
    // combined through LDS.  theta1, theta2 come out identical in every wave, which keeps the barriers uniform.
    __syncthreads();   // every wave has read the bounds
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < N; ++m) Us[(c * NS2 + m + N * row) * S + lane] = unew[c][m];
    __syncthreads();
    double A[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double v = 0;
#pragma unroll
      for (int b = 0; b < N; ++b) v += red[(b * 5 + c) * 64 + lane];
      A[c] = v;
    }
    if constexpr (GEO == 1) {
      const double area = 0.5 * fabs((vx[0] * vx[3] - vx[2] * vx[1]) + (vx[2] * vx[7] - vx[6] * vx[3]) +
                                     (vx[6] * vx[5] - vx[4] * vx[7]) + (vx[4] * vx[1] - vx[0] * vx[5]));
      const double ia = 1.0 / area;
#pragma unroll
      for (int c = 0; c < 4; ++c) A[c] *= ia;
    }
    const double eps = 1.0e-13;
    const bool bad = smin(A[RHO], pressure(A)) < eps;   // "Fatal: Negative states" :26-38
    if (bad && active && row == 0) atomicOr(&a.flags[0], 1);
    double *pm = red + 5 * N * 64;   // [3][N][64] minima of the rows: density, theta2 (speculative), theta2 (after theta1)
    // theta2 of this wave's points (:138-178) for the current unew / Us
    auto pressure_theta = [&](bool &fail) {
      double th = 1.0;
      for (int g = 0; g < a.kb.Ng; ++g)
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
          double W[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            double v = 0;
#pragma unroll
            for (int m = 0; m < N; ++m) v += a.kb.Pg[g][m] * (dir == 0 ? unew[c][m] : Us[(c * NS2 + row + N * m) * S + lane]);
            W[c] = v;
          }
          th = smin(th, positivity_theta2(W, A, eps, fail));
        }
      return th;
    };
    // first round: the density minimum and, on the guess theta1 = 1 (true almost everywhere), theta2 as well
    bool fail = false;
    {
      double rmin = 1.0e20;
      for (int g = 0; g < a.kb.Ng; ++g) {
        double px = 0, py = 0;
#pragma unroll
        for (int m = 0; m < N; ++m) {
          px += a.kb.Pg[g][m] * unew[RHO][m];
          py += a.kb.Pg[g][m] * Us[(RHO * NS2 + row + N * m) * S + lane];
        }
        rmin = smin(smin(rmin, px), py);
      }
      pm[row * 64 + lane] = rmin;
      pm[(N + row) * 64 + lane] = pressure_theta(fail);
    }
    __syncthreads();
    double rho_min = 1.0e20, theta2 = 1.0;
#pragma unroll
    for (int b = 0; b < N; ++b) {
      rho_min = smin(rho_min, pm[b * 64 + lane]);
      theta2 = smin(theta2, pm[(N + b) * 64 + lane]);
    }
    const double rat = fabs(A[RHO] - eps) * frcp(fabs(A[RHO] - rho_min) + 1.0e-13);
    const double theta1 = smin(rat, 1.0);
    const bool t1 = !bad && theta1 < 1.0;
    if (__any(t1)) {   // the same lanes in every wave: the density was scaled somewhere, theta2 has to be formed again
      if (t1) {
#pragma unroll
        for (int m = 0; m < N; ++m) {
          unew[RHO][m] = positivity_blend(theta1, unew[RHO][m], A[RHO]);
          Us[(RHO * NS2 + m + N * row) * S + lane] = unew[RHO][m];
        }
      }
      __syncthreads();
      fail = false;
      pm[(2 * N + row) * 64 + lane] = pressure_theta(fail);
      __syncthreads();
      theta2 = 1.0;
#pragma unroll
      for (int b = 0; b < N; ++b) theta2 = smin(theta2, pm[(2 * N + b) * 64 + lane]);
    }
    if (bad) theta2 = 1.0;
    else if (fail && active) atomicOr(&a.flags[1], 1);
    if (active && (t1 || theta2 < 1.0)) {   // rare: the rows stored by the update are replaced
      double *np = a.Unew + (size_t)shard * 4 * NS2 * 64 + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < N; ++m)
          np[(c * NS2 + m + N * row) * 64] = theta2 < 1.0 ? positivity_blend(theta2, unew[c][m], A[c]) : unew[c][m];
    }
    }   // !settled
  }
  if constexpr (POS == 2 && GEO == 0 && MODE != 2) {
    // Which cells can the limiter pass change?  TVB (src/limiter.cc:15-30): minmod hands back its first argument when it is
    // below M dx^2 or zero, so a cell whose (characteristic) slopes all are is left alone; wave 0 looks at the x slopes,
    // wave 1 at the y slopes, with a margin on the threshold so that the pass, which forms the slopes once more from the
    // DoFs, can never disagree in the other direction.  Positivity: the nodal box test.  The pass itself is
    // unchanged for the marked cells, so the results are those of the plain pass.
    if (row < 2) {
      double A[4], D[4];
      bool any = false;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double v = 0.0, g = 0.0;
#pragma unroll
        for (int b = 0; b < N; ++b) v += red[(b * 5 + c) * 64 + lane];
        if (row == 0) {
#pragma unroll
          for (int b = 0; b < N; ++b) g += red[(5 * N + c * N + b) * 64 + lane];
        } else {
#pragma unroll
          for (int m = 0; m < N / 2; ++m)   // the row sums of the average are w_m * (sum of the row): w_m = w_(N-1-m)
            g += (CB<N>::t.L1[m] - CB<N>::t.L0[m]) * CB<N>::t.iw[m] * (red[(m * 5 + c) * 64 + lane] - red[((N - 1 - m) * 5 + c) * 64 + lane]);
        }
        A[c] = v;
        D[c] = g;
        any = any || !(g == 0.0);
      }
      bool need = false;
      if (a.tvb_M >= 0.0) {
        if (a.tvb_char && __any(any)) {
          const EigenXY e = eigen_at(A);
          to_char(e, row, D);
        }
        // margin: relative on the threshold, and absolute against the rounding of the slopes (formed here from row
        // partials, in the pass from the DoFs; both errors are a few ulp of the state)
        const double thr = a.tvb_M * h * h * (1.0 - 1.0e-9) - 1.0e-11 * (fabs(A[0]) + fabs(A[1]) + fabs(A[2]) + fabs(A[3]));
#pragma unroll
        for (int i = 0; i < 4; ++i) need = need || !(fabs(D[i]) < thr || D[i] == 0.0);
      }
      if (row == 0 && a.pos_check) need = need || !positivity_box_settled<N>(Us, lane, a.kb.pg_neg);
      const unsigned long long m = __ballot(need && active);
      if (lane == 0 && m) atomicOr(&a.lim_mask[shard], m);
    }
  }
  if (row == N - 1) {  // cell averages (src/claw.cc:562-597), residual norm, CFL minimum of the shard; on the
                       // last wave: wave 0 carries the extra pass over the face points
    double avg[4], res = 0.0, dtmin = 1.0e20;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double v = 0;
#pragma unroll
      for (int b = 0; b < N; ++b) v += red[(b * 5 + c) * 64 + lane];
      avg[c] = v;
    }
#pragma unroll
    for (int b = 0; b < N; ++b) res += red[(b * 5 + 4) * 64 + lane];
    if constexpr (GEO == 1) {  // cell average = sum u JxW / |K| (src/claw.cc:589-593), |K| by the shoelace formula
      const double area = 0.5 * fabs((vx[0] * vx[3] - vx[2] * vx[1]) + (vx[2] * vx[7] - vx[6] * vx[3]) +
                                     (vx[6] * vx[5] - vx[4] * vx[7]) + (vx[4] * vx[1] - vx[0] * vx[5]));
      const double ia = 1.0 / area;
#pragma unroll
      for (int c = 0; c < 4; ++c) avg[c] *= ia;
    }
    if (active) {
#pragma unroll
      for (int c = 0; c < 4; ++c) a.avg_new[((size_t)shard * 4 + c) * 64 + lane] = avg[c];
      if constexpr (GEO == 0) {
        if (a.want_dt) dtmin = cfl_dt(avg, h, a.cfl, a.degree);
      }
    }
    res = wave_sum_lane63(res);
    if (GEO == 0 && a.want_dt) dtmin = wave_min_lane63(dtmin);
    if (lane == 63) {
      a.shard_res[shard] = res;
      if (GEO == 0 && a.want_dt) a.shard_dtmin[shard] = dtmin;
    }
  }
  PHASE_MARK(7);
#ifdef DFLO_PHASE_TIMING
  if (lane == 0 && a.phase_cycles)
    for (int i = 0; i < 8; ++i) a.phase_cycles[((size_t)blockIdx.x * 4 + row) * 8 + i] = tacc[i];
#endif
}

// compute_time_step_q for one cell (src/claw.cc:520-557): max of |v| + c over the 4 x 4 points of QIterated(QTrapez,3),
// dt = cfl h / lambda / (2k+1).  U: the cell's DoFs [4][N*N]; the interpolation is sum-factorised, one point row at a time.
template <int N>
__device__ __forceinline__ double dt_q_cell(const double *U, const KBasis &kb, double h, double cfl, int degree) {
  constexpr int NS = N * N;
  double maxeig = 0.0;
#pragma unroll
  for (int pa = 0; pa < kTrap; ++pa) {
    double v[4][N];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int b = 0; b < N; ++b) {
        double t = 0;
#pragma unroll
        for (int aa = 0; aa < N; ++aa) t += kb.Pt[pa][aa] * U[c * NS + aa + N * b];
        v[c][b] = t;
      }
#pragma unroll
    for (int pb = 0; pb < kTrap; ++pb) {
      double w[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double t = 0;
#pragma unroll
        for (int b = 0; b < N; ++b) t += kb.Pt[pb][b] * v[c][b];
        w[c] = t;
      }
      maxeig = fmax(maxeig, max_eigenvalue(w));
    }
  }
  return cfl * h / maxeig / (2.0 * degree + 1.0);
}

// ------------------------------------------------------------------ limiter kernel
struct LimArgs {
  double *U;
  const double *avg;
  const int32_t *shard_count;
  const int32_t *lrbt;
  const double *cell_h;
  int *flags;  // [0] negative mean state, [1] positivity root failure
  double h_uniform, M, beta;
  int n_shards, uniform_h, tvb, char_lim, pos_lim;
  const int32_t *shard_list;
  int n_list;
  const double *shock;  // KXRCF indicator per cell, or null: "shock indicator = limiter" marks every cell (1e20)
  unsigned long long *mask;   // [n_shards] from the stage kernel: the cells this pass can change (cleared here), or null: all cells
  // bilinear cells, last stage: the time step of the limited solution is formed here, while the cell is in registers
  double *shard_dtmin, *dt_cell;
  double cfl;
  int degree, dtq;
  KBasis kb;
};

// apply_limiter_TVB_Qk (src/limiter.cc:225-370) then apply_positivity_limiter
// (src/positivity.cc:17-208), lane = cell, all DoFs of the cell in registers.
template <int N>
__global__ __launch_bounds__(64) void limiter_kernel(const LimArgs a) {
  constexpr int NS = N * N, NDOF = 4 * NS;
  const int sidx = shard_of_block(blockIdx.x, a.n_list);
  if (sidx < 0) return;
  const int shard = a.shard_list ? a.shard_list[sidx] : sidx;
  const int lane = threadIdx.x;
  const bool active = lane < a.shard_count[shard];   // padding lanes hold a harmless state and run along
  const KBasis &kb = a.kb;
  double *up = a.U + (size_t)shard * NDOF * 64 + lane;
  double U[NDOF], A[4];
  // With the stage kernel's marks only the cells the limiters can change go through the pass (the others are provably left
  // as they are, see the stage kernel): most wavefronts return after one load.
  bool marked = true;
  if (a.mask) {
    const unsigned long long m = a.mask[shard];
    if (m == 0) return;
    if (lane == 0) a.mask[shard] = 0;   // consumed (the load above has returned: m was compared)
    marked = (m >> lane) & 1;
  }
  if (marked) {
#pragma unroll
    for (int d = 0; d < NDOF; ++d) U[d] = up[d * 64];
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) A[c] = a.avg[((size_t)shard * 4 + c) * 64 + lane];
  const double h = a.uniform_h ? a.h_uniform : a.cell_h[(size_t)shard * 64 + lane];
  bool changed = false;

  if (a.tvb && marked && (!a.shock || a.shock[(size_t)shard * 64 + lane] > 1.0)) {  // src/limiter.cc:263,406
    const double dx = h;  // diameter/sqrt(2) of a square
    const double Mdx2 = a.M * dx * dx;
    // one direction after the other (x: left/right neighbours, y: bottom/top), so that only one set of differences
    // is alive at a time
    EigenXY e;
    if (a.char_lim) e = eigen_at(A);
    double Dxn[4], Dyn[4], change_x = 0, change_y = 0;
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
      double D[4], db[4], df[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {   // dx * cell-average gradient, see above; l_m(1) - l_m(0) is antisymmetric in m, and
        double g = 0;                  // pairing the nodes makes the slope of a constant state exactly zero
#pragma unroll
        for (int b = 0; b < N; ++b)
#pragma unroll
          for (int m = 0; m < N / 2; ++m) {
            const int j0 = dir == 0 ? m + N * b : b + N * m, j1 = dir == 0 ? (N - 1 - m) + N * b : b + N * (N - 1 - m);
            g += CB<N>::t.w[b] * (CB<N>::t.L1[m] - CB<N>::t.L0[m]) * (U[c * NS + j0] - U[c * NS + j1]);
          }
        D[c] = g;
      }
      // the boundary case "no neighbour: difference = own slope" (:296-316) is resolved before the projection
      const double D0[4] = {D[0], D[1], D[2], D[3]};
      if (a.char_lim) to_char(e, dir, D);
      // minmod returns its first argument untouched when |a| < M dx^2 (src/limiter.cc:21): if that holds for
      // every component of every cell of the wavefront, the neighbour differences are not needed at all
      bool smooth = true;
#pragma unroll
      for (int i = 0; i < 4; ++i) smooth = smooth && (fabs(D[i]) < Mdx2 || D[i] == 0.0);   // minmod(0, b, c) = 0 as well
      if (__all(smooth)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (dir == 0) Dxn[i] = D[i];
          else Dyn[i] = D[i];
        }
        continue;
      }
      const int ib = a.lrbt[((size_t)shard * 4 + 2 * dir) * 64 + lane], ifw = a.lrbt[((size_t)shard * 4 + 2 * dir + 1) * 64 + lane];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        db[c] = ib >= 0 ? A[c] - a.avg[((size_t)(ib >> 6) * 4 + c) * 64 + (ib & 63)] : D0[c];
        df[c] = ifw >= 0 ? a.avg[((size_t)(ifw >> 6) * 4 + c) * 64 + (ifw & 63)] - A[c] : D0[c];
      }
      if (a.char_lim) {
        to_char(e, dir, db);
        to_char(e, dir, df);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double dn = minmod(D[i], a.beta * db[i], a.beta * df[i], Mdx2);
        if (dir == 0) { Dxn[i] = dn; change_x += fabs(dn - D[i]); }
        else { Dyn[i] = dn; change_y += fabs(dn - D[i]); }
      }
    }
    change_x *= 0.25;
    change_y *= 0.25;
    if (change_x + change_y > 1.0e-10) {  // :347 -- reduce to the limited linear polynomial
      if (a.char_lim) {
        to_con(e, 0, Dxn);
        to_con(e, 1, Dyn);
      }
      // u = A + (x - x_c) Dxn/dx + (y - y_c) Dyn/dx with x - x_c = dx (xi - 1/2): the division by dx (:349) and
      // the factor dx cancel
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < NS; ++j)
          U[c * NS + j] = A[c] + (CB<N>::t.x[j % N] - 0.5) * Dxn[c] + (CB<N>::t.x[j / N] - 0.5) * Dyn[c];
      changed = true;
    }
  }

  if (a.pos_lim && marked) {
    const double eps = 1.0e-13;
    if (smin(A[RHO], pressure(A)) < eps) {  // "Fatal: Negative states" :26-38
      if (active) atomicOr(&a.flags[0], 1);
    } else {
      // density at GLL(Ng) x Gauss(N) and Gauss(N) x GLL(Ng)  (:43-47, :72-78)
      double rho_min = 1.0e20;
#pragma unroll
      for (int l = 0; l < N; ++l)
        for (int g = 0; g < a.kb.Ng; ++g) {
          double vx = 0, vy = 0;
#pragma unroll
          for (int m = 0; m < N; ++m) {
            vx += kb.Pg[g][m] * U[RHO * NS + m + N * l];
            vy += kb.Pg[g][m] * U[RHO * NS + l + N * m];
          }
          rho_min = smin(smin(rho_min, vx), vy);
        }
      const double rat = fabs(A[RHO] - eps) * frcp(fabs(A[RHO] - rho_min) + 1.0e-13);
      const double theta1 = smin(rat, 1.0);
      if (theta1 < 1.0) {
#pragma unroll
        for (int j = 0; j < NS; ++j) U[RHO * NS + j] = positivity_blend(theta1, U[RHO * NS + j], A[RHO]);
        changed = true;
      }
      double theta2 = 1.0;
      bool fail = false;
#pragma unroll
      for (int dir = 0; dir < 2; ++dir)
#pragma unroll
        for (int l = 0; l < N; ++l)
          for (int g = 0; g < a.kb.Ng; ++g) {
            double W[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              double v = 0;
#pragma unroll
              for (int m = 0; m < N; ++m) v += kb.Pg[g][m] * (dir == 0 ? U[c * NS + m + N * l] : U[c * NS + l + N * m]);
              W[c] = v;
            }
            theta2 = smin(theta2, positivity_theta2(W, A, eps, fail));
          }
      if (fail && active) atomicOr(&a.flags[1], 1);
      if (theta2 < 1.0) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int j = 0; j < NS; ++j) U[c * NS + j] = positivity_blend(theta2, U[c * NS + j], A[c]);
        changed = true;
      }
    }
  }
  if (changed && active) {
#pragma unroll
    for (int d = 0; d < NDOF; ++d) up[d * 64] = U[d];
  }
  if (a.dtq) {   // wave-uniform
    double dtmin = 1.0e20;
    if (active) {
      dtmin = dt_q_cell<N>(U, kb, h, a.cfl, a.degree);
      if (a.dt_cell) a.dt_cell[(size_t)shard * 64 + lane] = dtmin;
    }
    dtmin = wave_min(dtmin);
    if (lane == 0) a.shard_dtmin[shard] = dtmin;
  }
}

// =====================================================================================================
// Pk (FE_DGP) basis: the same shard machinery on modal DoFs.  A P_k function is a Q_k function, so it is
// carried through phases A-C by its values at the Gauss nodes (exact), and only the two ends change:
//   load:   u(x_j) = sum_m psi_m(x_j) U_m                      (modal -> nodal, T)
//   store:  rhs_m  = sum_j psi_m(x_j) rhs_j, M = |K| I          (nodal residual -> modal, T^T; src/claw.cc:228-258
//           gives 1/|K| on the diagonal for the orthonormal basis), update and SSP combine on the modes.
// Cell average = mode 0 (psi_0 = 1).  Cartesian cells only.
// =====================================================================================================
template <int N, int B>
__device__ __forceinline__ void modal_to_row(const double (&um)[4][N * (N + 1) / 2], double (&urow)[4][N]) {
  constexpr int NM = N * (N + 1) / 2;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int aa = 0; aa < N; ++aa) {
      double v = 0.0;
#pragma unroll
      for (int m = 0; m < NM; ++m) v += PB<N>::t.T[aa + N * B][m] * um[c][m];
      urow[c][aa] = v;
    }
}

// phase C for node row B, then projection of the nodal residual on the modes and the modal update of the
// modes this wave owns (m = B, B+N, ...)
template <int N, int B, int MODE>
__device__ __forceinline__ void row_update_pk(const StageArgs &a, double *Us, const int S, const double *Fh, double *red,
                                              int shard, int lane, bool active, double h, const uint16_t (&cref)[4],
                                              const double (&Wrow)[N][4], const double (&ucur)[4][(N * (N + 1) / 2 + N - 1) / N],
                                              const double (&uold)[4][(N * (N + 1) / 2 + N - 1) / N], const double dt) {
  constexpr int NS = N * N, NM = N * (N + 1) / 2;
  double R[4][N];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int m = 0; m < N; ++m) R[c][m] = 0.0;
  double Gown[N][4];
#pragma unroll
  for (int aa = 0; aa < N; ++aa) {
    double Fx[4];
    flux_xy(Wrow[aa], Fx, Gown[aa]);
    const double wbh = CB<N>::t.w[B] * h;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double fx = Fx[c] * wbh;
#pragma unroll
      for (int m = 0; m < N; ++m) R[c][m] += fx * CB<N>::t.DW[aa][m];
      Us[(c * NS + aa + N * B) * S + lane] = Gown[aa][c];
    }
    if (a.gravity != 0.0) {
      const double jxw = CB<N>::t.w[aa] * CB<N>::t.w[B] * h * h;
      R[MY][aa] += a.gravity * (-1.0 * Wrow[aa][RHO]) * jxw;
      R[EN][aa] += a.gravity * (-1.0 * Wrow[aa][MY]) * jxw;
    }
  }
  __syncthreads();
#pragma unroll
  for (int aa = 0; aa < N; ++aa) {
    const double wah = CB<N>::t.w[aa] * h;
#pragma unroll
    for (int q = 0; q < N; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double gy = q == B ? Gown[aa][c] : Us[(c * NS + aa + N * q) * S + lane];
        R[c][aa] += gy * (wah * CB<N>::t.DW[q][B]);
      }
  }
  if (active) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const uint16_t ref = cref[f];
      if (ref == kNoFace) continue;
      const int k = ref & 0x3FFF;
      const bool flip = (ref >> 14) & 1;
      const double sgn = (ref >> 15) ? 1.0 : -1.0;
      if (f < 2) {
        const int qq = flip ? N - 1 - B : B;
        const double jxw = sgn * CB<N>::t.w[B] * h;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double fq = Fh[c * a.max_fp + k * N + qq] * jxw;
#pragma unroll
          for (int m = 0; m < N; ++m) R[c][m] += fq * ((f & 1) ? CB<N>::t.L1[m] : CB<N>::t.L0[m]);
        }
      } else {
        const double lw = (f & 1) ? CB<N>::t.L1[B] : CB<N>::t.L0[B];
#pragma unroll
        for (int q = 0; q < N; ++q) {
          const int qq = flip ? N - 1 - q : q;
          const double jxw = sgn * (CB<N>::t.w[q] * lw) * h;
#pragma unroll
          for (int c = 0; c < 4; ++c) R[c][q] += Fh[c * a.max_fp + k * N + qq] * jxw;
        }
      }
    }
  }
  // ---- nodal residual -> modal residual: rhs_m = sum over rows of sum_a psi_m(x_(a,B)) R[.][a]; the rows
  //      are added in a fixed order (row 0 first), through the now unused LDS image
  __syncthreads();  // every wave is done with the G exchange
  double *acc = Us;  // [4 NM][64]
#pragma unroll
  for (int w = 0; w < N; ++w) {
    if (B == w) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          double pr = 0.0;
#pragma unroll
          for (int aa = 0; aa < N; ++aa) pr += PB<N>::t.T[aa + N * B][m] * R[c][aa];
          if (w == 0) acc[(c * NM + m) * 64 + lane] = pr;
          else acc[(c * NM + m) * 64 + lane] += pr;
        }
    }
    __syncthreads();
  }
  double part[5] = {0, 0, 0, 0, 0};
  if (active) {
    const double rh2 = frcp(h * h);  // inverse mass of the orthonormal modes: 1/|K|
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int t = 0;
#pragma unroll
      for (int m = B; m < NM; m += N, ++t) {
        const double rm = acc[(c * NM + m) * 64 + lane];
        if constexpr (MODE == 2) {
          a.rhs_out[((size_t)shard * 4 * NM + c * NM + m) * 64 + lane] = rm;
        } else {
          part[4] += rm * rm;
          double u = ucur[c][t];
          u += dt * rm * rh2;
          if constexpr (MODE == 1) u = (1.0 - a.ark) * u + a.ark * uold[c][t];
          a.Unew[((size_t)shard * 4 * NM + c * NM + m) * 64 + lane] = u;
          if (m == 0) part[c] = u;  // the cell average is mode 0
        }
      }
    }
  }
  if constexpr (MODE != 2) {
    __syncthreads();  // every wave is done reading Fh
#pragma unroll
    for (int c = 0; c < 5; ++c) red[(B * 5 + c) * 64 + lane] = part[c];
  }
}

template <int N, int FLUX, int MODE>
__global__ __launch_bounds__(64 * N, N == 4 ? 2 : 3) void stage_kernel_pk(const StageArgs a) {
  constexpr int NS = N * N, NM = N * (N + 1) / 2, NDOFM = 4 * NM, NT = 64 * N, MS = (NM + N - 1) / N;
  constexpr int ROWS = 4 * NS + (FLUX == DFLO_FLUX_LXF ? 3 : 0);
  constexpr int TROWS = 4 * N + (FLUX == DFLO_FLUX_LXF ? 3 : 0);
  constexpr int S = 65;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int sidx = shard_of_block(blockIdx.x, a.n_list);
  if (sidx < 0) return;
  const int shard = a.shard_list ? a.shard_list[sidx] : sidx;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int row = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HS = a.halo_stride;
  double *Us = lds;
  double *Th = Us + ROWS * S;
  double *Fh = Th + TROWS * HS;
  uint32_t *Fr = (uint32_t *)(Fh + 4 * a.max_fp);
  double *Bv = (double *)(Fr + a.max_faces);
  int *Bk = (int *)(Bv + a.max_bnd * 4 * N);

  // ---- loads: every wave reads all modes of its cells (the rows need all of them); the modes a wave will
  //      update (m = row, row+N, ...) of u(s) and u(n) are requested separately and consumed at the end
  int hent[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) hent[t] = a.halo_pad[(size_t)shard * a.halo_pitch + ((tid + t * NT) & 31)];
  const int4 hdr = a.shard_hdr[shard];
  const int nf = hdr.y, nh = hdr.z, nbnd = hdr.w;
  const bool active = lane < hdr.x;
  double umode[4][NM];
  {
    const double *up = a.Ucur + (size_t)shard * NDOFM * 64 + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < NM; ++m) umode[c][m] = up[(c * NM + m) * 64];
  }
  double uavg[4];
  if constexpr (FLUX == DFLO_FLUX_LXF) {
    if (row == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) uavg[c] = a.avg_cur[((size_t)shard * 4 + c) * 64 + lane];
    }
  }
  const uint32_t *fp = a.faces_pad + (size_t)shard * a.face_pitch;
  const uint32_t fr0 = fp[tid], fr1 = fp[tid + NT];
  uint16_t cref[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) cref[f] = a.cell_face[((size_t)shard * 4 + f) * 64 + lane];
  const double h = a.uniform_h ? a.h_uniform : a.cell_h[(size_t)shard * 64 + lane];
  double dt_step = 0.0;   // fetched with the other loads, not in the middle of phase C
  if constexpr (MODE != 2) dt_step = a.dt_cell ? a.dt_cell[(size_t)shard * 64 + lane] : (a.dt_host >= 0.0 ? a.dt_host : *a.dt_dev);
  double ucur[4][MS], uold[4][MS];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < MS; ++t) {
      const int m = min(row + N * t, NM - 1);
      ucur[c][t] = a.Ucur[((size_t)shard * NDOFM + c * NM + m) * 64 + lane];
      if constexpr (MODE == 1) uold[c][t] = a.Uold[((size_t)shard * NDOFM + c * NM + m) * 64 + lane];
    }

  // ---- phase A
  double urow[4][N];
  if constexpr (N == 2) {
    if (row == 0) modal_to_row<N, 0>(umode, urow); else modal_to_row<N, 1>(umode, urow);
  } else if constexpr (N == 3) {
    if (row == 0) modal_to_row<N, 0>(umode, urow); else if (row == 1) modal_to_row<N, 1>(umode, urow); else modal_to_row<N, 2>(umode, urow);
  } else {
    if (row == 0) modal_to_row<N, 0>(umode, urow); else if (row == 1) modal_to_row<N, 1>(umode, urow);
    else if (row == 2) modal_to_row<N, 2>(umode, urow); else modal_to_row<N, 3>(umode, urow);
  }
  // halo: trace of the neighbour's modal expansion at the face point, psi_m = Pt_i(xi) Pt_j(eta) with
  // (xi, eta) on face f: xi in {0, 1, x_q}
  for (int i = tid; i < ((nh + 31) & ~31) * 4 * N; i += NT) {
    const int sl = (i & 31) + ((i >> 5) / (4 * N)) * 32, r = (i >> 5) % (4 * N), q = r % N, c = r / N;
    if (sl >= nh) continue;
    const int e = i == tid ? hent[0] : (i == tid + NT ? hent[1] : a.halo_pad[(size_t)shard * a.halo_pitch + sl]);
    const int ic = e & 0x0FFFFFFF, f = (e >> 28) & 3;
    const double *hp = a.Ucur + ((size_t)(ic >> 6) * NDOFM + c * NM) * 64 + (ic & 63);
    double pxi[N], peta[N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
      double pq = PB<N>::t.Px[0][n];
#pragma unroll
      for (int qq = 1; qq < N; ++qq) pq = q == qq ? PB<N>::t.Px[qq][n] : pq;
      pxi[n] = f == 0 ? PB<N>::t.P0[n] : (f == 1 ? PB<N>::t.P1[n] : pq);
      peta[n] = f == 2 ? PB<N>::t.P0[n] : (f == 3 ? PB<N>::t.P1[n] : pq);
    }
    double v = 0.0;
#pragma unroll
    for (int m = 0; m < NM; ++m) v += pxi[PB<N>::t.mi[m]] * peta[PB<N>::t.mj[m]] * hp[m * 64];
    Th[(c * N + q) * HS + sl] = v;
  }
  if constexpr (FLUX == DFLO_FLUX_LXF) {  // lambda of the LxF flux comes from the cell averages (src/equation.h:357-359):
                                          // keep (u, v, c) of each average instead of the four components
    for (int sl = tid; sl < nh; sl += NT) {
      const int ic = (sl < 32 ? hent[0] : a.halo_pad[(size_t)shard * a.halo_pitch + sl]) & 0x0FFFFFFF;
      double A[4], uvc[3];
#pragma unroll
      for (int c = 0; c < 4; ++c) A[c] = a.avg_cur[((size_t)(ic >> 6) * 4 + c) * 64 + (ic & 63)];
      wave_speed_uvc(A, uvc);
#pragma unroll
      for (int c = 0; c < 3; ++c) Th[(4 * N + c) * HS + sl] = uvc[c];
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int m = 0; m < N; ++m) Us[(c * NS + m + N * row) * S + lane] = urow[c][m];
  if constexpr (FLUX == DFLO_FLUX_LXF) {
    if (row == 0) {
      double uvc[3];
      wave_speed_uvc(uavg, uvc);
#pragma unroll
      for (int c = 0; c < 3; ++c) Us[(4 * NS + c) * S + lane] = uvc[c];
    }
  }
  if (tid < nf) Fr[tid] = fr0;
  if (tid + NT < nf) Fr[tid + NT] = fr1;
  for (int i = tid + 2 * NT; i < nf; i += NT) Fr[i] = fp[i];
  if (nbnd > 0) {
    for (int i = tid; i < nbnd * 4 * N; i += NT) {
      const int bl = i / (4 * N), k2 = i - bl * 4 * N;
      const int bf = a.bnd_pad[(size_t)shard * a.bnd_pitch + bl];
      if (k2 == 0) Bk[bl] = a.bface_kind[bf];
      Bv[i] = a.bval[(size_t)bf * 4 * N + k2];
    }
  }
  __syncthreads();
  flux_phase<N, FLUX, 0>(a, Us, Th, Fh, Fr, Bv, Bk, nullptr, HS, nf, tid);
  __syncthreads();

  double *red = Fh;
  double wrow[N][4];
#pragma unroll
  for (int m = 0; m < N; ++m)
#pragma unroll
    for (int c = 0; c < 4; ++c) wrow[m][c] = urow[c][m];
#define DFLO_ROWPK(Bq) row_update_pk<N, Bq, MODE>(a, Us, S, Fh, red, shard, lane, active, h, cref, wrow, ucur, uold, dt_step)
  if constexpr (N == 2) {
    if (row == 0) DFLO_ROWPK(0); else DFLO_ROWPK(1);
  } else if constexpr (N == 3) {
    if (row == 0) DFLO_ROWPK(0); else if (row == 1) DFLO_ROWPK(1); else DFLO_ROWPK(2);
  } else {
    if (row == 0) DFLO_ROWPK(0); else if (row == 1) DFLO_ROWPK(1); else if (row == 2) DFLO_ROWPK(2); else DFLO_ROWPK(3);
  }
#undef DFLO_ROWPK
  if constexpr (MODE == 2) return;
  __syncthreads();
  if (row == N - 1) {
    double avg[4], res = 0.0, dtmin = 1.0e20;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double v = 0;
#pragma unroll
      for (int b = 0; b < N; ++b) v += red[(b * 5 + c) * 64 + lane];
      avg[c] = v;
    }
#pragma unroll
    for (int b = 0; b < N; ++b) res += red[(b * 5 + 4) * 64 + lane];
    if (active) {
#pragma unroll
      for (int c = 0; c < 4; ++c) a.avg_new[((size_t)shard * 4 + c) * 64 + lane] = avg[c];
      if (a.want_dt) dtmin = cfl_dt(avg, h, a.cfl, a.degree);
    }
    res = wave_sum_lane63(res);
    if (a.want_dt) dtmin = wave_min_lane63(dtmin);
    if (lane == 63) {
      a.shard_res[shard] = res;
      if (a.want_dt) a.shard_dtmin[shard] = dtmin;
    }
  }
}

// apply_limiter_TVB_Pk (src/limiter.cc:377-516) then the Pk branch of apply_positivity_limiter
// (src/positivity.cc:100-109, 197-205); lane = cell, all modes in registers
template <int N>
__global__ __launch_bounds__(64) void limiter_pk_kernel(const LimArgs a) {
  constexpr int NM = N * (N + 1) / 2, NDOFM = 4 * NM;
  const int sidx = shard_of_block(blockIdx.x, a.n_list);
  if (sidx < 0) return;
  const int shard = a.shard_list ? a.shard_list[sidx] : sidx;
  const int lane = threadIdx.x;
  if (lane >= a.shard_count[shard]) return;
  double *up = a.U + (size_t)shard * NDOFM * 64 + lane;
  double U[4][NM], A[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int m = 0; m < NM; ++m) U[c][m] = up[(c * NM + m) * 64];
#pragma unroll
  for (int c = 0; c < 4; ++c) A[c] = a.avg[((size_t)shard * 4 + c) * 64 + lane];
  const double h = a.uniform_h ? a.h_uniform : a.cell_h[(size_t)shard * 64 + lane];
  bool changed = false;
  const double sqrt_3 = 1.7320508075688772935;
  if (a.tvb && (!a.shock || a.shock[(size_t)shard * 64 + lane] > 1.0)) {  // src/limiter.cc:263,406
    const double dx = h, Mdx2 = a.M * dx * dx, beta = 0.5 * a.beta;   // :396
    double Dx[4], Dy[4], dbx[4], dfx[4], dby[4], dfy[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      Dx[c] = U[c][1] * sqrt_3;       // mode (1,0)
      Dy[c] = U[c][N] * sqrt_3;       // mode (0,1) = index k+1
    }
    const int il = a.lrbt[((size_t)shard * 4 + 0) * 64 + lane], ir = a.lrbt[((size_t)shard * 4 + 1) * 64 + lane];
    const int ib = a.lrbt[((size_t)shard * 4 + 2) * 64 + lane], it = a.lrbt[((size_t)shard * 4 + 3) * 64 + lane];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      dbx[c] = il >= 0 ? A[c] - a.avg[((size_t)(il >> 6) * 4 + c) * 64 + (il & 63)] : Dx[c];
      dfx[c] = ir >= 0 ? a.avg[((size_t)(ir >> 6) * 4 + c) * 64 + (ir & 63)] - A[c] : Dx[c];
      dby[c] = ib >= 0 ? A[c] - a.avg[((size_t)(ib >> 6) * 4 + c) * 64 + (ib & 63)] : Dy[c];
      dfy[c] = it >= 0 ? a.avg[((size_t)(it >> 6) * 4 + c) * 64 + (it & 63)] - A[c] : Dy[c];
    }
    EigenXY e;
    if (a.char_lim) {
      e = eigen_at(A);
      to_char(e, 0, dbx); to_char(e, 0, dfx); to_char(e, 1, dby); to_char(e, 1, dfy);
      to_char(e, 0, Dx); to_char(e, 1, Dy);
    }
    double Dxn[4], Dyn[4], change_x = 0, change_y = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Dxn[i] = minmod(Dx[i], beta * dbx[i], beta * dfx[i], Mdx2);
      Dyn[i] = minmod(Dy[i], beta * dby[i], beta * dfy[i], Mdx2);
      change_x += fabs(Dxn[i] - Dx[i]);
      change_y += fabs(Dyn[i] - Dy[i]);
    }
    change_x /= 4;
    change_y /= 4;
    if (change_x + change_y > 1.0e-10) {
      if (a.char_lim) {
        to_con(e, 0, Dxn);
        to_con(e, 1, Dyn);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 1; m < NM; ++m) U[c][m] = m == 1 ? Dxn[c] * (1.0 / sqrt_3) : (m == N ? Dyn[c] * (1.0 / sqrt_3) : 0.0);
      changed = true;
    }
  }
  if (a.pos_lim) {
    const double eps = 1.0e-13;
    if (smin(A[RHO], pressure(A)) < eps) {
      atomicOr(&a.flags[0], 1);
    } else {
      // point value of component c at (Pt(xi), Pt(eta)) given the 1-D Legendre values
      auto point = [&](int c, const double *pxi, const double *peta) {
        double v = 0.0;
#pragma unroll
        for (int m = 0; m < NM; ++m) v += pxi[PB<N>::t.mi[m]] * peta[PB<N>::t.mj[m]] * U[c][m];
        return v;
      };
      double rho_min = 1.0e20;
      for (int l = 0; l < N; ++l)
        for (int g = 0; g < a.kb.Ng; ++g) {
          double pg[N], pl[N];
#pragma unroll
          for (int n = 0; n < N; ++n) { pg[n] = a.kb.PLg[g][n]; pl[n] = a.kb.PLx[l][n]; }
          rho_min = smin(smin(rho_min, point(RHO, pg, pl)), point(RHO, pl, pg));
        }
      const double rat = fabs(A[RHO] - eps) * frcp(fabs(A[RHO] - rho_min) + 1.0e-13);
      const double theta1 = smin(rat, 1.0);
      if (theta1 < 1.0) {
#pragma unroll
        for (int m = 1; m < NM; ++m) U[RHO][m] *= theta1;
        changed = true;
      }
      double theta2 = 1.0;
      bool fail = false;
      for (int dir = 0; dir < 2; ++dir)
        for (int l = 0; l < N; ++l)
          for (int g = 0; g < a.kb.Ng; ++g) {
            double pg[N], pl[N], W[4];
#pragma unroll
            for (int n = 0; n < N; ++n) { pg[n] = a.kb.PLg[g][n]; pl[n] = a.kb.PLx[l][n]; }
#pragma unroll
            for (int c = 0; c < 4; ++c) W[c] = dir == 0 ? point(c, pg, pl) : point(c, pl, pg);
            const double pre = kG1 * (W[EN] - 0.5 * (W[MX] * W[MX] + W[MY] * W[MY]) * frcp(W[RHO]));
            if (pre < eps) {
              const double drho = W[RHO] - A[RHO], dmx = W[MX] - A[MX], dmy = W[MY] - A[MY], dE = W[EN] - A[EN];
              const double a1 = 2.0 * drho * dE - (dmx * dmx + dmy * dmy);
              double b1 = 2.0 * drho * (A[EN] - eps / kG1) + 2.0 * A[RHO] * dE - 2.0 * (A[MX] * dmx + A[MY] * dmy);
              double c1 = 2.0 * A[RHO] * A[EN] - (A[MX] * A[MX] + A[MY] * A[MY]) - 2.0 * eps * A[RHO] / kG1;
              b1 /= a1;
              c1 /= a1;
              const double D = sqrt(fabs(b1 * b1 - 4.0 * c1));
              const double t1 = 0.5 * (-b1 - D), t2 = 0.5 * (-b1 + D);
              double t;
              if (t1 > -1.0e-12 && t1 < 1.0 + 1.0e-12) t = t1;
              else if (t2 > -1.0e-12 && t2 < 1.0 + 1.0e-12) t = t2;
              else { fail = true; t = 0.0; }
              t = smin(1.0, t);
              t = smax(0.0, t);
              if (fabs(1.0 - t) < 1.0e-14) t = 0.0;
              theta2 = smin(theta2, t);
            }
          }
      if (fail) atomicOr(&a.flags[1], 1);
      if (theta2 < 1.0) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int m = 1; m < NM; ++m) U[c][m] *= theta2;
        changed = true;
      }
    }
  }
  if (changed) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 1; m < NM; ++m) up[(c * NM + m) * 64] = U[c][m];
  }
}

// ------------------------------------------------------------------ KXRCF troubled-cell indicator
struct IndArgs {
  const double *U, *avg;
  double *shock;  // [n_slots]
  const int32_t *shard_count, *lrbt;
  const uint8_t *nbr_code;
  const double *cell_h;
  double h_uniform;
  int uniform_h, component, degree;
  const int32_t *shard_list;
  int n_list;
};

// value of one component at point q of local face f: Qk from the nodes on the line through the face point,
// Pk from all modes; u points at the component's first DoF of the cell (DoF stride 64)
template <int N, int PK>
__device__ __forceinline__ double face_point_value(const double *u, int f, int q) {
  double v = 0.0;
  if constexpr (PK == 0) {
    const int str0 = f < 2 ? 1 : N;
    const int base = (f < 2 ? N * q : q) + ((f & 1) ? (N - 1) * str0 : 0), str = (f & 1) ? -str0 : str0;
#pragma unroll
    for (int m = 0; m < N; ++m) v += CB<N>::t.L0[m] * u[(base + m * str) * 64];
  } else {
    constexpr int NM = N * (N + 1) / 2;
    double pxi[N], peta[N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
      double pq = PB<N>::t.Px[0][n];
#pragma unroll
      for (int qq = 1; qq < N; ++qq) pq = q == qq ? PB<N>::t.Px[qq][n] : pq;
      pxi[n] = f == 0 ? PB<N>::t.P0[n] : (f == 1 ? PB<N>::t.P1[n] : pq);
      peta[n] = f == 2 ? PB<N>::t.P0[n] : (f == 3 ? PB<N>::t.P1[n] : pq);
    }
#pragma unroll
    for (int m = 0; m < NM; ++m) v += pxi[PB<N>::t.mi[m]] * peta[PB<N>::t.mj[m]] * u[m * 64];
  }
  return v;
}

// compute_shock_indicator_kxrcf (src/indicator.cc:51-198), same-level faces, axis-aligned squares; lane = cell.
// Runs as its own pass between the stage update and the limiter: it reads the neighbours' unlimited DoFs.
template <int N, int PK>
__global__ __launch_bounds__(64) void indicator_kernel(const IndArgs a) {
  constexpr int NS = PK ? N * (N + 1) / 2 : N * N, NDOF = 4 * NS;
  const int sidx = shard_of_block(blockIdx.x, a.n_list);
  if (sidx < 0) return;
  const int shard = a.shard_list ? a.shard_list[sidx] : sidx;
  const int lane = threadIdx.x;
  if (lane >= a.shard_count[shard]) return;
  double A[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) A[c] = a.avg[((size_t)shard * 4 + c) * 64 + lane];
  const double vel[2] = {A[MX] / A[RHO], A[MY] / A[RHO]};  // :106-108
  const double h = a.uniform_h ? a.h_uniform : a.cell_h[(size_t)shard * 64 + lane];
  const double *uo = a.U + ((size_t)shard * NDOF + a.component * NS) * 64 + lane;
  double ind = 0.0, inflow = 0.0;
  for (int f = 0; f < 4; ++f) {
    const int code = a.nbr_code[((size_t)shard * 4 + f) * 64 + lane];
    if (!(code & 8)) continue;  // boundary (or periodic) face, :169-174
    const int ns = a.lrbt[((size_t)shard * 4 + f) * 64 + lane], nf = code & 3;
    const bool flip = (code & 4) != 0;
    const double *un = a.U + ((size_t)(ns >> 6) * NDOF + a.component * NS) * 64 + (ns & 63);
    const double vn = f == 0 ? -vel[0] : (f == 1 ? vel[0] : (f == 2 ? -vel[1] : vel[1]));
    const double inflow_status = vn < 0 ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < N; ++q) {
      const double jxw = CB<N>::t.w[q] * h;
      const double d = face_point_value<N, PK>(uo, f, q) - face_point_value<N, PK>(un, nf, flip ? N - 1 - q : q);
      ind += inflow_status * d * jxw;
      inflow += inflow_status * jxw;
    }
  }
  const double diameter = h * 1.4142135623730950488;
  const double denominator = pow(diameter, 0.5 * (a.degree + 1)) * inflow * A[a.component];  // :179-181
  a.shock[(size_t)shard * 64 + lane] = fabs(ind) / denominator;  // 0/0 -> NaN -> "not > 1": not limited, as in the reference
}

// ------------------------------------------------------------------ one translation unit per degree
// The stage kernels of one N (5 fluxes x 3 modes x 2 geometries x the limiter variants, Qk and Pk) take most of the compile
// time; build() compiles this file once per N with -DDFLO_STAGE_N=N (only this section is kept after the kernels above)
// and once without (everything else), in parallel.
typedef void (*stage_fn)(const StageArgs);
template <int N, int FLUX>
stage_fn pick_stage_m(int mode, int geo, int pos) {
  if (pos == 1 && mode != 2) {
    if (geo == 0) return mode == 0 ? stage_kernel<N, FLUX, 0, 0, 1> : stage_kernel<N, FLUX, 1, 0, 1>;
    return mode == 0 ? stage_kernel<N, FLUX, 0, 1, 1> : stage_kernel<N, FLUX, 1, 1, 1>;
  }
  if (pos == 2 && mode != 2 && geo == 0) return mode == 0 ? stage_kernel<N, FLUX, 0, 0, 2> : stage_kernel<N, FLUX, 1, 0, 2>;
  if (geo == 0) return mode == 0 ? stage_kernel<N, FLUX, 0, 0, 0> : (mode == 1 ? stage_kernel<N, FLUX, 1, 0, 0> : stage_kernel<N, FLUX, 2, 0, 0>);
  return mode == 0 ? stage_kernel<N, FLUX, 0, 1, 0> : (mode == 1 ? stage_kernel<N, FLUX, 1, 1, 0> : stage_kernel<N, FLUX, 2, 1, 0>);
}
template <int N>
stage_fn pick_stage_n(int flux, int mode, int geo, int pos) {
  switch (flux) {
    case DFLO_FLUX_LXF: return pick_stage_m<N, DFLO_FLUX_LXF>(mode, geo, pos);
    case DFLO_FLUX_SW: return pick_stage_m<N, DFLO_FLUX_SW>(mode, geo, pos);
    case DFLO_FLUX_KFVS: return pick_stage_m<N, DFLO_FLUX_KFVS>(mode, geo, pos);
    case DFLO_FLUX_ROE: return pick_stage_m<N, DFLO_FLUX_ROE>(mode, geo, pos);
    default: return pick_stage_m<N, DFLO_FLUX_HLLC>(mode, geo, pos);
  }
}
template <int N, int FLUX>
stage_fn pick_pk_m(int mode) {
  return mode == 0 ? stage_kernel_pk<N, FLUX, 0> : (mode == 1 ? stage_kernel_pk<N, FLUX, 1> : stage_kernel_pk<N, FLUX, 2>);
}
template <int N>
stage_fn pick_pk_n(int flux, int mode) {
  switch (flux) {
    case DFLO_FLUX_LXF: return pick_pk_m<N, DFLO_FLUX_LXF>(mode);
    case DFLO_FLUX_SW: return pick_pk_m<N, DFLO_FLUX_SW>(mode);
    case DFLO_FLUX_KFVS: return pick_pk_m<N, DFLO_FLUX_KFVS>(mode);
    case DFLO_FLUX_ROE: return pick_pk_m<N, DFLO_FLUX_ROE>(mode);
    default: return pick_pk_m<N, DFLO_FLUX_HLLC>(mode);
  }
}
stage_fn stage_of_2(int flux, int mode, int geo, int pos);
stage_fn stage_of_3(int flux, int mode, int geo, int pos);
stage_fn stage_of_4(int flux, int mode, int geo, int pos);
stage_fn stage_pk_of_2(int flux, int mode);
stage_fn stage_pk_of_3(int flux, int mode);
stage_fn stage_pk_of_4(int flux, int mode);
#ifdef DFLO_STAGE_N
#define DFLO_CAT_(a, b) a##b
#define DFLO_CAT(a, b) DFLO_CAT_(a, b)
stage_fn DFLO_CAT(stage_of_, DFLO_STAGE_N)(int flux, int mode, int geo, int pos) { return pick_stage_n<DFLO_STAGE_N>(flux, mode, geo, pos); }
stage_fn DFLO_CAT(stage_pk_of_, DFLO_STAGE_N)(int flux, int mode) { return pick_pk_n<DFLO_STAGE_N>(flux, mode); }
}  // namespace dflo
#else

// ------------------------------------------------------------------ boundary functions on the device
// The boundary values of integrate_boundary_term_explicit (FunctionParser::vector_value_list at the face
// quadrature points with set_time(bc_time), src/assemble_explicit.cc:161-165, src/claw.cc:736-745) evaluated by
// the device from postfix programs (dflo_hip_set_boundary_program): no host round trip per step for
// time-dependent boundary data (C4's moving shock on the top wall).
struct BcArgs {
  const int32_t *ops;       // [n][2] (dflo_expr_op, constant index)
  const double *consts;
  const int32_t *prog;      // [DFLO_MAX_BOUNDARIES][4][2] (first op, number of ops); 0 ops = values stay as uploaded
  const int32_t *bface_id;  // [n_bfaces]
  const double *bxy;        // [n_bfaces][N][2]
  double *bval0, *bval1;    // [n_bfaces][N][4] tables of RK stage 0 (time t) and of the later stages (t + dt)
  const double *dt_dev;     // [0] dt, [1] elapsed time
  double dt_host;
  const int32_t *faces;     // [n_faces] the boundary faces whose id has a program
  int n_faces, N, n_ops, n_consts;
};
constexpr int kBcLdsOps = 1024, kBcLdsConsts = 256;  // programs up to this size are interpreted out of LDS
constexpr int kExprStack = 16;
constexpr int kBcThreads = 512;   // 8 wavefronts: (component, table) pairs over 64 face points
// Postfix interpreter.  Every lane of the wave runs the SAME program (the caller loops over the programs and masks
// the store), so the opcode is wave-uniform: it is moved to a scalar register and the switch becomes one scalar
// jump -- with a per-lane opcode the compiler walks through all forty cases under an execution mask.  The top of
// the stack lives in a register, the rest in LDS (st[depth][thread]; a private array indexed by the stack pointer
// would be placed in scratch memory).
__device__ double run_program(const int32_t *ops, int n, const double *consts, double x, double y, double t, double *st) {
  double tos = 0.0;
  int sp = 0;   // entries below the top, kept in st[0 .. sp)
  for (int i = 0; i < n; ++i) {
    const int op = __builtin_amdgcn_readfirstlane(ops[2 * i]);
    if (op <= DFLO_OP_T) {  // pushes
      st[(sp++) * kBcThreads] = tos;
      tos = op == DFLO_OP_CONST ? consts[__builtin_amdgcn_readfirstlane(ops[2 * i + 1])] : (op == DFLO_OP_X ? x : (op == DFLO_OP_Y ? y : t));
      continue;
    }
    if (op == DFLO_OP_SEL) {
      const double b = tos, a = st[(--sp) * kBcThreads], c = st[(--sp) * kBcThreads];
      tos = c != 0.0 ? a : b;
      continue;
    }
    const bool binary = (op >= DFLO_OP_ADD && op <= DFLO_OP_OR) || op == DFLO_OP_MIN || op == DFLO_OP_MAX || op == DFLO_OP_ATAN2;
    double a = tos, b = 0.0;
    if (binary) {
      b = tos;
      a = st[(--sp) * kBcThreads];
    }
    double r;
    switch (op) {
      case DFLO_OP_NEG: r = -a; break;
      case DFLO_OP_ADD: r = a + b; break;
      case DFLO_OP_SUB: r = a - b; break;
      case DFLO_OP_MUL: r = a * b; break;
      case DFLO_OP_DIV: r = a / b; break;
      case DFLO_OP_POW: r = pow(a, b); break;
      case DFLO_OP_LT: r = a < b ? 1.0 : 0.0; break;
      case DFLO_OP_LE: r = a <= b ? 1.0 : 0.0; break;
      case DFLO_OP_GT: r = a > b ? 1.0 : 0.0; break;
      case DFLO_OP_GE: r = a >= b ? 1.0 : 0.0; break;
      case DFLO_OP_EQ: r = a == b ? 1.0 : 0.0; break;
      case DFLO_OP_NE: r = a != b ? 1.0 : 0.0; break;
      case DFLO_OP_AND: r = (a != 0.0 && b != 0.0) ? 1.0 : 0.0; break;
      case DFLO_OP_OR: r = (a != 0.0 || b != 0.0) ? 1.0 : 0.0; break;
      case DFLO_OP_SIN: r = sin(a); break;
      case DFLO_OP_COS: r = cos(a); break;
      case DFLO_OP_TAN: r = tan(a); break;
      case DFLO_OP_EXP: r = exp(a); break;
      case DFLO_OP_LOG: r = log(a); break;
      case DFLO_OP_SQRT: r = sqrt(a); break;
      case DFLO_OP_ABS: r = fabs(a); break;
      case DFLO_OP_MIN: r = fmin(a, b); break;
      case DFLO_OP_MAX: r = fmax(a, b); break;
      case DFLO_OP_ATAN2: r = atan2(a, b); break;
      case DFLO_OP_TANH: r = tanh(a); break;
      case DFLO_OP_SINH: r = sinh(a); break;
      case DFLO_OP_COSH: r = cosh(a); break;
      case DFLO_OP_ASIN: r = asin(a); break;
      case DFLO_OP_ACOS: r = acos(a); break;
      case DFLO_OP_ATAN: r = atan(a); break;
      case DFLO_OP_FLOOR: r = floor(a); break;
      case DFLO_OP_CEIL: r = ceil(a); break;
      case DFLO_OP_SIGN: r = a > 0.0 ? 1.0 : (a < 0.0 ? -1.0 : 0.0); break;
      case DFLO_OP_LOG10: r = log10(a); break;
      case DFLO_OP_ERF: r = erf(a); break;
      case DFLO_OP_ERFC: r = erfc(a); break;
      default: r = __builtin_nan(""); break;
    }
    tos = r;
  }
  return tos;
}
__global__ __launch_bounds__(kBcThreads) void bc_eval_kernel(const BcArgs a) {
  // the programs are a few hundred words: interpret them out of LDS, not with a dependent global load per opcode
  __shared__ int32_t s_ops[2 * kBcLdsOps];
  __shared__ double s_consts[kBcLdsConsts];
  __shared__ int32_t s_prog[DFLO_MAX_BOUNDARIES * 4 * 2];
  __shared__ double s_stack[(kExprStack + 1) * kBcThreads];
  const bool in_lds = a.n_ops <= kBcLdsOps && a.n_consts <= kBcLdsConsts;
  if (in_lds) {
    for (int k = threadIdx.x; k < 2 * a.n_ops; k += blockDim.x) s_ops[k] = a.ops[k];
    for (int k = threadIdx.x; k < a.n_consts; k += blockDim.x) s_consts[k] = a.consts[k];
  }
  for (int k = threadIdx.x; k < DFLO_MAX_BOUNDARIES * 4 * 2; k += blockDim.x) s_prog[k] = a.prog[k];
  __syncthreads();
  // a block takes 64 (listed face, point) pairs; wavefront w evaluates component w & 3 for the table w >> 2, so the
  // eight short programs of a point run side by side instead of one after the other in a single thread
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = wave & 3, which = wave >> 2;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const bool valid = j < a.n_faces * a.N;
  const int bf = a.faces[valid ? j / a.N : 0], i = bf * a.N + (valid ? j % a.N : 0);
  const int id = a.bface_id[bf];
  const double t = a.dt_dev[1] + (which ? (a.dt_host >= 0.0 ? a.dt_host : a.dt_dev[0]) : 0.0);
  const double x = a.bxy[2 * i], y = a.bxy[2 * i + 1];
  double *bval = which ? a.bval1 : a.bval0;
  for (int b = 0; b < DFLO_MAX_BOUNDARIES; ++b) {
    const int first = s_prog[(b * 4 + c) * 2], n = s_prog[(b * 4 + c) * 2 + 1];   // wave-uniform
    if (n == 0) continue;
    const double v = in_lds ? run_program(s_ops + 2 * first, n, s_consts, x, y, t, s_stack + threadIdx.x)
                            : run_program(a.ops + 2 * first, n, a.consts, x, y, t, s_stack + threadIdx.x);
    if (valid && id == b) bval[(size_t)i * 4 + c] = v;
  }
}

// accuracy probe of the reciprocal / square-root forms used by the flux functions
__global__ void debug_math_kernel(const double *x, double *rcp, double *sq, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    rcp[i] = frcp(x[i]);
    sq[i] = fsqrt(x[i]);
  }
}

// ------------------------------------------------------------------ small kernels
// quadrature weight of node j for the cell average: w_a w_b (squares) or w_a w_b det J / |K| (bilinear cells)
__device__ __forceinline__ double avg_weight(const KBasis &kb, int N, int j, const double *vert, int n_slots, int slot,
                                             double inv_area) {
  if (N < 0) return j == 0 ? 1.0 : 0.0;  // Pk (flagged by N < 0): the average is mode 0
  const double ww = kb.w[j % N] * kb.w[j / N];
  if (!vert) return ww;
  double v[8];
  for (int k = 0; k < 8; ++k) v[k] = vert[(size_t)k * n_slots + slot];
  const double xi = kb.x[j % N], eta = kb.x[j / N];
  const double xxi = (v[2] - v[0]) + eta * ((v[6] - v[4]) - (v[2] - v[0])), yxi = (v[3] - v[1]) + eta * ((v[7] - v[5]) - (v[3] - v[1]));
  const double xeta = (v[4] - v[0]) + xi * ((v[6] - v[2]) - (v[4] - v[0])), yeta = (v[5] - v[1]) + xi * ((v[7] - v[3]) - (v[5] - v[1]));
  return ww * (xxi * yeta - xeta * yxi) * inv_area;
}
__device__ __forceinline__ double cell_inv_area(const double *vert, int n_slots, int slot) {
  if (!vert) return 1.0;
  double v[8];
  for (int k = 0; k < 8; ++k) v[k] = vert[(size_t)k * n_slots + slot];
  return 1.0 / (0.5 * fabs((v[0] * v[3] - v[2] * v[1]) + (v[2] * v[7] - v[6] * v[3]) + (v[6] * v[5] - v[4] * v[7]) +
                           (v[4] * v[1] - v[0] * v[5])));
}
// user (dflo) layout <-> shard SoA layout
__global__ void scatter_kernel(const double *user, double *U, const int32_t *user_of, int n_slots, int ndof) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n_slots * ndof) return;
  const int slot = (int)(t / ndof), d = (int)(t - (long long)slot * ndof);
  const int uc = user_of[slot];
  const int ns = ndof / 4;
  // padding slots hold a harmless state (rho = 1, E = 1)
  const double v = uc >= 0 ? user[(size_t)uc * ndof + d] : ((d / ns) >= 2 ? 1.0 : 0.0);
  U[((size_t)(slot >> 6) * ndof + d) * 64 + (slot & 63)] = v;
}
__global__ void gather_kernel(double *user, const double *U, const int32_t *iid, int n_cells, int ndof) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n_cells * ndof) return;
  const int c = (int)(t / ndof), d = (int)(t - (long long)c * ndof);
  const int slot = iid[c];
  user[t] = U[((size_t)(slot >> 6) * ndof + d) * 64 + (slot & 63)];
}
// pack listed cells (internal slots) cell-major: buf[k][ndof]
__global__ void pack_kernel(double *buf, const double *U, const int32_t *slots, int n, int ndof) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * ndof) return;
  const int k = (int)(t / ndof), d = (int)(t - (long long)k * ndof);
  const int slot = slots[k];
  buf[t] = U[((size_t)(slot >> 6) * ndof + d) * 64 + (slot & 63)];
}
// ghost cells: staging buffer [g][ndof] -> ghost shards, and their cell averages
// one thread per (ghost cell, component): its DoFs travel buffer -> ghost shard (the buffer is read with unit stride
// along the thread's own run of n_s values, the shard rows are written 64 cells wide) and their average is formed
__global__ void unpack_ghost_kernel(const double *buf, double *U, double *avg, int first_slot, int n_ghost, int ndof,
                                    KBasis kb, int N, const double *vert, int n_slots) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_ghost * 4) return;
  const int g = t >> 2, c = t & 3;
  const int slot = first_slot + g, ns = ndof / 4;
  const double ia = cell_inv_area(vert, n_slots, slot);
  double m = 0;
  for (int j = 0; j < ns; ++j) {
    const double v = buf[(size_t)g * ndof + c * ns + j];
    U[((size_t)(slot >> 6) * ndof + c * ns + j) * 64 + (slot & 63)] = v;
    m += avg_weight(kb, N, j, vert, n_slots, slot, ia) * v;
  }
  avg[((size_t)(slot >> 6) * 4 + c) * 64 + (slot & 63)] = m;
}
__global__ void unpack_ghost_avg_kernel(const double *buf, double *avg, int first_slot, int n_ghost) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_ghost) return;
  const int slot = first_slot + g;
  for (int c = 0; c < 4; ++c) avg[((size_t)(slot >> 6) * 4 + c) * 64 + (slot & 63)] = buf[(size_t)g * 4 + c];
}
// compute_cell_average (src/claw.cc:562-597) for all slots (owned and ghost shards)
__global__ void average_kernel(const double *U, double *avg, int ndof, KBasis kb, int N, const double *vert, int n_slots) {
  const int shard = blockIdx.x;
  const int lane = threadIdx.x;
  const int ns = ndof / 4, slot = shard * 64 + lane;
  const double ia = cell_inv_area(vert, n_slots, slot);
  for (int c = 0; c < 4; ++c) {
    double m = 0;
    for (int j = 0; j < ns; ++j)
      m += avg_weight(kb, N, j, vert, n_slots, slot, ia) * U[((size_t)shard * ndof + c * ns + j) * 64 + lane];
    avg[((size_t)shard * 4 + c) * 64 + lane] = m;
  }
}
// compute_time_step_q (src/claw.cc:520-557): max of |v| + c over the 4 x 4 points of QIterated(QTrapez,3),
// dt = cfl h / lambda / (2k+1) with h = diameter / sqrt(2); per-shard minimum.  Lane = cell; the
// interpolation to the 16 points is sum-factorised (xi first, then eta).
template <int N>
__global__ __launch_bounds__(64) void dt_q_kernel(const double *U, const double *cell_h, const int32_t *shard_count,
                                                  double *shard_dtmin, KBasis kb, double cfl, int degree, double *dt_cell) {
  constexpr int NS = N * N, NDOF = 4 * NS;
  const int shard = blockIdx.x;
  const int lane = threadIdx.x;
  double dtmin = 1.0e20;
  if (lane < shard_count[shard]) {
    double u[NDOF];
#pragma unroll
    for (int j = 0; j < NDOF; ++j) u[j] = U[((size_t)shard * NDOF + j) * 64 + lane];
    dtmin = dt_q_cell<N>(u, kb, cell_h[(size_t)shard * 64 + lane], cfl, degree);
    if (dt_cell) dt_cell[(size_t)shard * 64 + lane] = dtmin;
  }
  dtmin = wave_min(dtmin);
  if (lane == 0) shard_dtmin[shard] = dtmin;
}
// compute_time_step_cartesian (src/claw.cc:486-511): per-shard minimum from the stored cell averages
__global__ void dt_kernel(const double *avg, const double *cell_h, double h_uniform, int uniform_h,
                          const int32_t *shard_count, double *shard_dtmin, double cfl, int degree, double *dt_cell) {
  const int shard = blockIdx.x;
  const int lane = threadIdx.x;
  double dtmin = 1.0e20;
  if (lane < shard_count[shard]) {
    double A[4];
    for (int c = 0; c < 4; ++c) A[c] = avg[((size_t)shard * 4 + c) * 64 + lane];
    const double h = uniform_h ? h_uniform : cell_h[(size_t)shard * 64 + lane];
    dtmin = cfl_dt(A, h, cfl, degree);
    if (dt_cell) dt_cell[(size_t)shard * 64 + lane] = dtmin;  // "time step type = local": dt(c), src/claw.cc:506
  }
  dtmin = wave_min(dtmin);
  if (lane == 0) shard_dtmin[shard] = dtmin;
}

// reductions over shards + the global-dt rules of compute_time_step (src/claw.cc:468-476)
struct FinalArgs {
  const double *shard_res, *shard_dtmin;
  double *res_sq;  // [3] per stage
  double *dt_dev;  // [0] dt, [1] elapsed time, [2] raw min before rules
  double *partial; // [kFinBlocks][4] workgroup partials
  int *counter;    // workgroups done
  int n_shards, n_stages, res_stride, do_dt, advance_time, global_rules;  // shard_res: [n_stages][res_stride]
  double time_step, final_time, dt_host;
};
constexpr int kFinBlocks = 32;   // workgroups of the two-level reduction
__global__ __launch_bounds__(256) void finalize_kernel(const FinalArgs a) {
  __shared__ double sred[4][4];
  __shared__ int is_last;
  // Two levels, both with a fixed assignment and a fixed combination order -> deterministic sums:
  // workgroup b reduces the shards [b chunk, (b+1) chunk) of every array (the residual partials of all stages
  // of the step and the CFL minima), the workgroup that finishes last combines the kFinBlocks partials in index order.
  const int n = a.n_shards, t = threadIdx.x, b = blockIdx.x;
  const int chunk = ((n + kFinBlocks - 1) / kFinBlocks + 255) & ~255;
  const int lo = b * chunk, hi = min(n, lo + chunk);
  double rs[3] = {0.0, 0.0, 0.0}, m = 1.0e20;
  for (int s = lo + t; s < hi; s += 256) {
    for (int st = 0; st < a.n_stages; ++st) rs[st] += a.shard_res[(size_t)st * a.res_stride + s];
    if (a.do_dt) m = fmin(m, a.shard_dtmin[s]);
  }
  for (int st = 0; st < 3; ++st) {
    const double r = wave_sum(rs[st]);
    if ((t & 63) == 0) sred[st][t >> 6] = r;
  }
  m = wave_min(m);
  if ((t & 63) == 0) sred[3][t >> 6] = m;
  __syncthreads();
  if (t == 0) {
    for (int st = 0; st < 3; ++st) a.partial[b * 4 + st] = (sred[st][0] + sred[st][1]) + (sred[st][2] + sred[st][3]);
    a.partial[b * 4 + 3] = fmin(fmin(sred[3][0], sred[3][1]), fmin(sred[3][2], sred[3][3]));
    __threadfence();
    is_last = atomicAdd(a.counter, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  __shared__ double spart[kFinBlocks * 4];
  if (t < (int)gridDim.x * 4) spart[t] = ((const volatile double *)a.partial)[t];  // one round trip for all partials
  __syncthreads();
  if (t != 0) return;
  *a.counter = 0;  // ready for the next launch (launches on one stream do not overlap)
  for (int st = 0; st < a.n_stages; ++st) {
    double tot = 0.0;
    for (int i = 0; i < (int)gridDim.x; ++i) tot += spart[i * 4 + st];
    a.res_sq[st] = tot;
  }
  if (a.do_dt) {
    double tt = a.dt_dev[1];
    if (a.advance_time) {  // elapsed_time += global_dt (src/claw.cc:1072) for the step just done
      tt += a.dt_host >= 0.0 ? a.dt_host : a.dt_dev[0];
      a.dt_dev[1] = tt;
    }
    double dt = spart[3];
    for (int i = 1; i < (int)gridDim.x; ++i) dt = fmin(dt, spart[i * 4 + 3]);
    a.dt_dev[2] = dt;
    if (a.global_rules) {  // src/claw.cc:469-476, only for "time step type = global"
      if (dt > 0 && a.time_step > 0) dt = fmin(dt, a.time_step);
      if (tt + dt > a.final_time) dt = a.final_time - tt;
    }
    a.dt_dev[0] = dt;
  }
}
// re-apply the rules after an external all-reduce(min) of dt_dev[2] (multi-device)
__global__ void dt_rules_kernel(double *dt_dev, double time_step, double final_time) {
  double dt = dt_dev[2], t = dt_dev[1];
  if (dt > 0 && time_step > 0) dt = fmin(dt, time_step);
  if (t + dt > final_time) dt = final_time - t;
  dt_dev[0] = dt;
}

}  // namespace dflo

// ====================================================================== host side
using namespace dflo;

struct dflo_hip_engine {
  Plan plan;
  BasisTables bt;
  KBasis kb;
  dflo_params_t prm;
  int degree = 1, N = 2, ns = 4, ndof = 16, mapping = DFLO_MAP_CARTESIAN, geo = 0, basis = DFLO_BASIS_QK;
  int n_rk = 2;
  double ark[3] = {0, 0, 0};
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  // device buffers
  double *U[3] = {nullptr, nullptr, nullptr};
  int cur = 0, old = 0;
  double *avg[2] = {nullptr, nullptr};
  int avg_cur = 0;
  double *rhs = nullptr, *user_buf = nullptr;
  double *bval[2] = {nullptr, nullptr};
  int32_t *bface_kind = nullptr;
  // device-evaluated boundary functions: host copies of the programs, rebuilt device tables when they change
  std::vector<int32_t> bc_ops[DFLO_MAX_BOUNDARIES][4];
  std::vector<double> bc_consts[DFLO_MAX_BOUNDARIES][4];
  int n_bc_programs = 0;
  bool bc_dirty = false;
  int32_t *d_bc_ops = nullptr, *d_bc_prog = nullptr, *d_bface_id = nullptr, *d_bc_faces = nullptr;
  int n_bc_faces = 0, n_bc_ops = 0, n_bc_consts = 0;
  double *d_bc_consts = nullptr, *d_bxy = nullptr;
  int32_t *d_shard_count = nullptr;
  uint32_t *d_faces_pad = nullptr;
  uint8_t *d_nbr_code = nullptr;
  double *d_shock = nullptr;  // [n_slots], only for shock indicator = density | energy
  int32_t *d_bnd_pad = nullptr;
  int bnd_pitch = 1;
  int4 *d_shard_hdr = nullptr;
  int32_t *d_halo_pad = nullptr;
  int face_pitch = 0, halo_pitch = 0, halo_stride = 0;
  uint16_t *d_cell_face = nullptr;
  int32_t *d_lrbt = nullptr, *d_user_of = nullptr, *d_iid = nullptr;
  double *d_cell_h = nullptr, *d_cell_vert = nullptr, *d_fgeom_pad = nullptr, *d_dt_cell = nullptr;
  double *shard_res = nullptr, *shard_dtmin = nullptr, *res_sq = nullptr, *dt_dev = nullptr, *fin_partial = nullptr;
  int *flags = nullptr;
  std::vector<double> bface_xy;  // [n_bfaces][N][2]
  int pending_rk = -1;
  int st_in = 0, st_old = 0, st_out = 0, st_avg_in = 0, st_rk = 0, st_which = 0;
  double st_dt = -1.0;
  int64_t t_stages = 0, t_seen = 0;   // stages timed / stages seen while timing is on
  bool t_sample = false;
  int32_t *d_rim_list = nullptr, *d_int_list = nullptr;
  hipEvent_t ev_rim = nullptr, ev_unpack = nullptr;
  bool unpack_pending = false;
  double pending_dt = -1.0;
  int32_t *d_send_slots = nullptr;
  int n_send = 0;
  double *ghost_stage = nullptr;
  size_t lds_bytes = 0;
  int stage_grid = 8, prefetch_ahead = 1 << 30;
  unsigned long long *phase_cycles = nullptr;
  int max_fp = 0;
  // timing
  bool timing = false;
  int dtq_parts = 0;   // last stage on bilinear cells: parts (1 rim, 2 interior) whose limiter pass also formed the time step
  bool fuse_pos = false;   // positivity limiter without TVB on Qk: applied inside the stage kernel (DFLO_FUSE_POS=0: separate pass)
  unsigned long long *lim_mask = nullptr;   // TVB on Qk squares: [n_shards], written by the stage kernel for the limiter pass (DFLO_LIM_MASK=0: off)
  bool aux_fresh = false;      // lim_mask belongs to the state the open stage has just produced
  // dflo_hip_advance replays a captured graph of `graph_steps` time steps (the buffer rotation repeats with
  // that period); built lazily for the state it was captured in
  hipGraphExec_t graph_exec = nullptr;
  int graph_steps = 0, graph_cur = -1, graph_avg = -1;
  hipStream_t graph_stream = nullptr;
  bool use_graph = false;  // opt-in (DFLO_GRAPH=1): on ROCm 7.2 / MI355X replay measured no faster than plain launches
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  double t_accum_ms = 0;
  int64_t t_count = 0;
  std::string err;
};

namespace {
std::string g_create_error;

#define HIPCHK(h, call)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
      return DFLO_ERR_HIP;                                                                   \
    }                                                                                        \
  } while (0)

template <typename T>
int upload(dflo_hip_engine *h, T **dst, const std::vector<T> &src) {
  size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
  HIPCHK(h, hipMalloc((void **)dst, bytes));
  if (!src.empty()) HIPCHK(h, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return DFLO_OK;
}

KBasis make_kbasis(const BasisTables &b) {
  KBasis k{};
  for (int i = 0; i < kMaxN; ++i) {
    k.w[i] = b.w[i];
    k.iw[i] = b.w[i] != 0.0 ? 1.0 / b.w[i] : 0.0;
    k.x[i] = b.x[i];
    k.L0[i] = b.L0[i];
    k.L1[i] = b.L1[i];
    for (int j = 0; j < kMaxN; ++j) {
      k.D[i][j] = b.D[i][j];
      k.DW[i][j] = b.D[i][j] * b.w[i];
    }
  }
  for (int g = 0; g < kMaxGLL; ++g)
    for (int j = 0; j < kMaxN; ++j) k.Pg[g][j] = b.Pg[g][j];
  for (int g = 0; g < kTrap; ++g)
    for (int j = 0; j < kMaxN; ++j) k.Pt[g][j] = b.Pt[g][j];
  k.Ng = b.Ng;
  k.pg_neg = 0.0;
  for (int g = 0; g < b.Ng; ++g) {
    double sneg = 0.0;
    for (int j = 0; j < b.N; ++j) sneg += std::max(-b.Pg[g][j], 0.0);
    k.pg_neg = std::max(k.pg_neg, sneg);
  }
  {
    long double g[kMaxGLL + 1];
    gauss_lobatto01(b.Ng, g);
    for (int gi = 0; gi < b.Ng; ++gi)
      for (int n = 0; n < b.N; ++n) k.PLg[gi][n] = legendre01(n, (double)g[gi]);
    for (int q = 0; q < b.N; ++q)
      for (int n = 0; n < b.N; ++n) k.PLx[q][n] = legendre01(n, b.x[q]);
  }
  return k;
}

stage_fn pick_pk(int N, int flux, int mode) {
  switch (N) {
    case 2: return dflo::stage_pk_of_2(flux, mode);
    case 3: return dflo::stage_pk_of_3(flux, mode);
    default: return dflo::stage_pk_of_4(flux, mode);
  }
}
stage_fn pick_stage(int N, int flux, int mode, int geo, int pos = 0) {
  switch (N) {
    case 2: return dflo::stage_of_2(flux, mode, geo, pos);
    case 3: return dflo::stage_of_3(flux, mode, geo, pos);
    default: return dflo::stage_of_4(flux, mode, geo, pos);
  }
}

int grid_for(int n_shards) { return ((n_shards + 7) / 8) * 8; }

void launch_dt_q(dflo_hip_engine *h);

void time_begin(dflo_hip_engine *h) {
  if (!h->t_sample) return;
  if (h->ev_used == h->ev_pool.size()) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    h->ev_pool.push_back({a, b});
  }
  hipEventRecord(h->ev_pool[h->ev_used].first, h->stream);
}
void time_end(dflo_hip_engine *h) {
  if (!h->t_sample) return;
  hipEventRecord(h->ev_pool[h->ev_used].second, h->stream);
  ++h->ev_used;
}
void time_collect(dflo_hip_engine *h) {
  for (size_t i = 0; i < h->ev_used; ++i) {
    float ms = 0;
    hipEventSynchronize(h->ev_pool[i].second);
    hipEventElapsedTime(&ms, h->ev_pool[i].first, h->ev_pool[i].second);
    h->t_accum_ms += ms;
    ++h->t_count;
  }
  h->ev_used = 0;
}

// ---- one RK stage on the host side.  A stage is opened once (buffer roles are fixed), its update and
// limiter kernels may then be launched for all shards or separately for the rim shards (those that read
// ghost cells) and the interior shards, and it is finished by the reductions.
int eval_boundary_programs(dflo_hip_engine *h, double dt_host);

int open_stage(dflo_hip_engine *h, int rk, double dt_host, bool residual_only, int which_override) {
  if (rk == 0 && !residual_only && h->n_bc_programs > 0) {  // boundary functions at t (stage 0) and t + dt (later stages)
    const int rc = eval_boundary_programs(h, dt_host);
    if (rc) return rc;
  }
  const bool last = rk == h->n_rk - 1;
  int out;
  if (residual_only) out = h->cur;
  else if (last && h->cur != h->old) out = h->old;
  else { out = 0; while (out == h->cur || out == h->old) ++out; }
  h->st_in = h->cur;
  h->st_old = h->old;
  h->st_out = out;
  h->st_avg_in = h->avg_cur;
  h->st_rk = rk;
  h->st_dt = dt_host;
  h->st_which = which_override >= 0 ? which_override : (rk == 0 ? 0 : 1);
  if (!residual_only) {
    h->cur = out;
    h->avg_cur = 1 - h->avg_cur;
    if (last) h->old = out;
    h->pending_rk = rk;
    h->pending_dt = dt_host;
    h->dtq_parts = 0;
    h->aux_fresh = false;
    // stage timing samples every fifth stage (5 is coprime to the 2 or 3 stages of a step, so every stage of the
    // step is sampled equally often): two event records per launch are not free
    h->t_sample = h->timing && (h->t_seen++ % 5 == 0);
    if (h->t_sample) ++h->t_stages;
  }
  return DFLO_OK;
}

int eval_boundary_programs(dflo_hip_engine *h, double dt_host) {
  const Plan &p = h->plan;
  const int nb = (int)p.bface_cell.size();
  if (nb == 0 || h->n_bc_programs == 0) return DFLO_OK;
  if (h->bc_dirty) {  // flatten the programs: one op / constant pool, a (first, count) entry per (id, component)
    std::vector<int32_t> ops, prog(DFLO_MAX_BOUNDARIES * 4 * 2, 0);
    std::vector<double> consts;
    for (int b = 0; b < DFLO_MAX_BOUNDARIES; ++b)
      for (int c = 0; c < 4; ++c) {
        const std::vector<int32_t> &o = h->bc_ops[b][c];
        prog[(b * 4 + c) * 2] = (int32_t)(ops.size() / 2);
        prog[(b * 4 + c) * 2 + 1] = (int32_t)(o.size() / 2);
        for (size_t k = 0; k < o.size(); k += 2) {
          ops.push_back(o[k]);
          ops.push_back(o[k] == DFLO_OP_CONST ? o[k + 1] + (int32_t)consts.size() : 0);
        }
        consts.insert(consts.end(), h->bc_consts[b][c].begin(), h->bc_consts[b][c].end());
      }
    h->n_bc_ops = (int)(ops.size() / 2);
    h->n_bc_consts = (int)consts.size();
    if (ops.empty()) ops.assign(2, 0);
    if (consts.empty()) consts.assign(1, 0.0);
    std::vector<int32_t> faces;   // only the faces of boundaries that carry a program are visited
    for (int b = 0; b < nb; ++b) {
      const int id = p.bface_id[b];
      bool any = false;
      for (int c = 0; c < 4; ++c) any |= !h->bc_ops[id][c].empty();
      if (any) faces.push_back(b);
    }
    h->n_bc_faces = (int)faces.size();
    if (faces.empty()) faces.push_back(0);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    hipFree(h->d_bc_ops); hipFree(h->d_bc_consts); hipFree(h->d_bc_prog); hipFree(h->d_bc_faces);
    h->d_bc_ops = nullptr; h->d_bc_consts = nullptr; h->d_bc_prog = nullptr; h->d_bc_faces = nullptr;
    int rc;
    if ((rc = upload(h, &h->d_bc_ops, ops)) || (rc = upload(h, &h->d_bc_consts, consts)) || (rc = upload(h, &h->d_bc_prog, prog)) ||
        (rc = upload(h, &h->d_bc_faces, faces)))
      return rc;
    if (!h->d_bface_id) {
      if ((rc = upload(h, &h->d_bface_id, p.bface_id)) || (rc = upload(h, &h->d_bxy, h->bface_xy))) return rc;
    }
    h->bc_dirty = false;
  }
  BcArgs a{};
  a.ops = h->d_bc_ops;
  a.consts = h->d_bc_consts;
  a.prog = h->d_bc_prog;
  a.bface_id = h->d_bface_id;
  a.bxy = h->d_bxy;
  a.bval0 = h->bval[0];
  a.bval1 = h->bval[1];
  a.dt_dev = h->dt_dev;
  a.dt_host = dt_host;
  a.faces = h->d_bc_faces;
  a.n_faces = h->n_bc_faces;
  a.N = h->N;
  a.n_ops = h->n_bc_ops;
  a.n_consts = h->n_bc_consts;
  if (a.n_faces == 0) return DFLO_OK;
  hipLaunchKernelGGL(bc_eval_kernel, dim3((a.n_faces * a.N + 63) / 64), dim3(kBcThreads), 0, h->stream, a);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

// residual + update kernel of the open stage; part 0: all shards, 1: rim shards, 2: interior shards
int launch_update(dflo_hip_engine *h, double *rhs_out, int part) {
  const Plan &p = h->plan;
  const int rk = h->st_rk;
  const bool last = rk == h->n_rk - 1;
  StageArgs a{};
  a.phase_cycles = h->phase_cycles;
  a.Ucur = h->U[h->st_in];
  a.Uold = h->U[h->st_old];
  a.Unew = h->U[h->st_out];
  a.avg_cur = h->avg[h->st_avg_in];
  a.avg_new = h->avg[1 - h->st_avg_in];
  a.rhs_out = rhs_out;
  a.shard_count = h->d_shard_count;
  a.shard_hdr = h->d_shard_hdr;
  a.halo_pad = h->d_halo_pad;
  a.halo_pitch = h->halo_pitch;
  a.halo_stride = h->halo_stride;
  a.faces_pad = h->d_faces_pad;
  a.face_pitch = h->face_pitch;
  a.bnd_pad = h->d_bnd_pad;
  a.bnd_pitch = h->bnd_pitch;
  a.cell_face = h->d_cell_face;
  a.cell_h = h->d_cell_h;
  a.cell_vert = h->d_cell_vert;
  a.fgeom_pad = h->d_fgeom_pad;
  a.n_slots = p.n_slots;
  a.bval = h->bval[h->st_which];
  a.bface_kind = h->bface_kind;
  a.dt_dev = h->dt_dev;
  a.dt_cell = h->d_dt_cell;  // null unless "time step type = local"
  a.shard_res = h->shard_res + (size_t)rk * std::max(h->plan.n_shards, 1);
  a.shard_dtmin = h->shard_dtmin;
  a.dt_host = h->st_dt;
  a.ark = h->ark[rk];
  a.gravity = h->prm.gravity;
  a.cfl = h->prm.cfl;
  a.h_uniform = p.h;
  a.n_shards = p.n_shards;
  a.max_fp = h->max_fp;
  a.max_bnd = h->plan.max_bnd;
  a.max_faces = (std::max(h->plan.max_faces, 1) + 1) & ~1;
  a.prefetch_ahead = h->prefetch_ahead;
  a.uniform_h = p.uniform_h ? 1 : 0;
  a.want_dt = last ? 1 : 0;
  a.degree = h->degree;
  a.kb = h->kb;
  a.shard_list = part == 1 ? h->d_rim_list : (part == 2 ? h->d_int_list : nullptr);
  a.n_list = part == 1 ? (int)p.rim_shards.size() : (part == 2 ? (int)p.interior_shards.size() : p.n_shards);
  if (a.n_list == 0) return DFLO_OK;
  const int mode_ = rhs_out ? 2 : (h->ark[rk] != 0.0 ? 1 : 0);
  a.flags = h->flags;
  a.lim_mask = h->lim_mask;
  a.tvb_M = h->prm.limiter_type == DFLO_LIMITER_TVB ? h->prm.M : -1.0;
  a.tvb_char = h->prm.char_lim;
  a.pos_check = h->prm.pos_lim;
  const int pos_ = h->fuse_pos ? 1 : (h->lim_mask ? 2 : 0);
  if (pos_ == 2 && mode_ != 2) h->aux_fresh = true;
  stage_fn fn = h->basis == DFLO_BASIS_PK ? pick_pk(h->N, h->prm.flux_type, mode_) : pick_stage(h->N, h->prm.flux_type, mode_, h->geo, pos_);
  time_begin(h);
  hipLaunchKernelGGL(fn, dim3(grid_for(a.n_list)), dim3(64 * h->N), h->lds_bytes, h->stream, a);
  time_end(h);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

// compute_shock_indicator (src/indicator.cc:17-30): nothing to do for type "limiter" (the limiter treats a
// missing indicator as 1e20 everywhere)
int launch_indicator(dflo_hip_engine *h, int part) {
  if (!h->d_shock) return DFLO_OK;
  const Plan &p = h->plan;
  IndArgs a{};
  a.U = h->U[h->cur];
  a.avg = h->avg[h->avg_cur];
  a.shock = h->d_shock;
  a.shard_count = h->d_shard_count;
  a.lrbt = h->d_lrbt;
  a.nbr_code = h->d_nbr_code;
  a.cell_h = h->d_cell_h;
  a.h_uniform = p.h;
  a.uniform_h = p.uniform_h ? 1 : 0;
  a.component = h->prm.shock_indicator == DFLO_IND_DENSITY ? RHO : EN;  // :70-82
  a.degree = h->degree;
  a.shard_list = part == 1 ? h->d_rim_list : (part == 2 ? h->d_int_list : nullptr);
  a.n_list = part == 1 ? (int)p.rim_shards.size() : (part == 2 ? (int)p.interior_shards.size() : p.n_shards);
  if (a.n_list == 0) return DFLO_OK;
  void (*fn)(const IndArgs);
  if (h->basis == DFLO_BASIS_PK) fn = h->N == 2 ? indicator_kernel<2, 1> : (h->N == 3 ? indicator_kernel<3, 1> : indicator_kernel<4, 1>);
  else fn = h->N == 2 ? indicator_kernel<2, 0> : (h->N == 3 ? indicator_kernel<3, 0> : indicator_kernel<4, 0>);
  hipLaunchKernelGGL(fn, dim3(grid_for(a.n_list)), dim3(64), 0, h->stream, a);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

int launch_limiter(dflo_hip_engine *h, int tvb, int pos, int part, bool stage_data = false) {
  const Plan &p = h->plan;
  LimArgs l{};
  l.U = h->U[h->cur];
  l.avg = h->avg[h->avg_cur];
  l.shard_count = h->d_shard_count;
  l.lrbt = h->d_lrbt;
  l.cell_h = h->d_cell_h;
  l.flags = h->flags;
  l.h_uniform = p.h;
  l.M = h->prm.M;
  l.beta = h->prm.beta;
  l.n_shards = p.n_shards;
  l.uniform_h = p.uniform_h ? 1 : 0;
  l.tvb = tvb;
  l.char_lim = h->prm.char_lim;
  l.pos_lim = pos;
  l.kb = h->kb;
  l.shock = tvb ? h->d_shock : nullptr;
  l.mask = (stage_data && tvb && h->aux_fresh) ? h->lim_mask : nullptr;
  l.dtq = (h->geo == 1 && h->basis == DFLO_BASIS_QK && h->pending_rk == h->n_rk - 1) ? 1 : 0;
  l.shard_dtmin = h->shard_dtmin;
  l.dt_cell = h->d_dt_cell;
  l.cfl = h->prm.cfl;
  l.degree = h->degree;
  l.shard_list = part == 1 ? h->d_rim_list : (part == 2 ? h->d_int_list : nullptr);
  l.n_list = part == 1 ? (int)p.rim_shards.size() : (part == 2 ? (int)p.interior_shards.size() : p.n_shards);
  if (l.dtq) h->dtq_parts |= part == 0 ? 3 : part;
  if (l.n_list == 0) return DFLO_OK;
  void (*lf)(const LimArgs) = h->N == 2 ? limiter_kernel<2> : (h->N == 3 ? limiter_kernel<3> : limiter_kernel<4>);
  if (h->basis == DFLO_BASIS_PK) lf = h->N == 2 ? limiter_pk_kernel<2> : (h->N == 3 ? limiter_pk_kernel<3> : limiter_pk_kernel<4>);
  hipLaunchKernelGGL(lf, dim3(grid_for(l.n_list)), dim3(64), 0, h->stream, l);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

// limiter of the open stage on one part of the shards
int launch_stage_limiter(dflo_hip_engine *h, int part) {
  if (h->pending_rk < 0) { h->err = "no stage pending"; return DFLO_ERR_BAD_PARAM; }
  const bool limited = h->prm.limiter_type != DFLO_LIMITER_NONE || h->prm.pos_lim;
  if (!limited) return DFLO_OK;
  if (h->fuse_pos) return DFLO_OK;   // positivity alone: the stage kernel has applied it on the way out
  if (h->prm.limiter_type == DFLO_LIMITER_TVB) {  // compute_shock_indicator(); apply_limiter();  src/claw.cc:763-764
    const int rc = launch_indicator(h, part);
    if (rc) return rc;
  }
  return launch_limiter(h, h->prm.limiter_type == DFLO_LIMITER_TVB, h->prm.pos_lim, part, true);
}

// reductions of the stage launched last
// workgroups of finalize_kernel: as many chunks of >= 256 shards as there are, at most kFinBlocks
int fin_grid(int n_shards) {
  const int chunk = ((n_shards + kFinBlocks - 1) / kFinBlocks + 255) & ~255;
  return std::max(1, (n_shards + chunk - 1) / chunk);
}

int launch_finish(dflo_hip_engine *h) {
  const Plan &p = h->plan;
  const int rk = h->pending_rk;
  if (rk < 0) { h->err = "no stage pending"; return DFLO_ERR_BAD_PARAM; }
  const bool last = rk == h->n_rk - 1;
  if (last && h->geo == 1) {  // bilinear cells: dt from the point values of the (limited) new solution,
    if (h->dtq_parts != 3) {   // unless the limiter pass of this stage has formed it on the way
      launch_dt_q(h);
      HIPCHK(h, hipGetLastError());
    }
  } else if (last && h->d_dt_cell) {  // local time stepping: the per-cell dt of the next step
    hipLaunchKernelGGL(dt_kernel, dim3(p.n_shards), dim3(64), 0, h->stream, h->avg[h->avg_cur], h->d_cell_h, p.h,
                       p.uniform_h ? 1 : 0, h->d_shard_count, h->shard_dtmin, h->prm.cfl, h->degree, h->d_dt_cell);
    HIPCHK(h, hipGetLastError());
  }
  h->pending_rk = -1;
  if (!last) return DFLO_OK;  // ||rhs|| of every stage is reduced once, after the last stage (it is only reported, src/claw.cc:768)
  FinalArgs f{};
  f.shard_res = h->shard_res;
  f.shard_dtmin = h->shard_dtmin;
  f.res_sq = h->res_sq;
  f.dt_dev = h->dt_dev;
  f.n_shards = p.n_shards;
  f.n_stages = h->n_rk;
  f.res_stride = std::max(p.n_shards, 1);
  f.do_dt = 1;
  f.advance_time = 1;
  f.dt_host = h->pending_dt;
  f.time_step = h->prm.time_step;
  f.final_time = h->prm.final_time;
  f.global_rules = h->prm.global_time_step;
  f.partial = h->fin_partial;
  f.counter = h->flags + 2;
  hipLaunchKernelGGL(finalize_kernel, dim3(fin_grid(f.n_shards)), dim3(256), 0, h->stream, f);
  HIPCHK(h, hipGetLastError());
  h->pending_rk = -1;
  return DFLO_OK;
}

int launch_limit_finalize(dflo_hip_engine *h) {
  int rc = launch_stage_limiter(h, 0);
  if (rc) return rc;
  return launch_finish(h);
}

int launch_stage(dflo_hip_engine *h, int rk, double dt_host, double *rhs_out, int which_override) {
  int rc = open_stage(h, rk, dt_host, rhs_out != nullptr, which_override);
  if (rc) return rc;
  rc = launch_update(h, rhs_out, 0);
  if (rc || rhs_out) return rc;
  return launch_limit_finalize(h);
}

void launch_dt_q(dflo_hip_engine *h) {
  const Plan &p = h->plan;
  auto fn = h->N == 2 ? dt_q_kernel<2> : (h->N == 3 ? dt_q_kernel<3> : dt_q_kernel<4>);
  hipLaunchKernelGGL(fn, dim3(p.n_shards), dim3(64), 0, h->stream, (const double *)h->U[h->cur], (const double *)h->d_cell_h,
                     (const int32_t *)h->d_shard_count, h->shard_dtmin, h->kb, h->prm.cfl, h->degree, h->d_dt_cell);
}

int launch_average(dflo_hip_engine *h) {
  const Plan &p = h->plan;
  const int all = p.n_shards + p.n_ghost_shards;
  hipLaunchKernelGGL(average_kernel, dim3(all), dim3(64), 0, h->stream, h->U[h->cur], h->avg[h->avg_cur], h->ndof, h->kb,
                     h->basis == DFLO_BASIS_PK ? -h->N : h->N, (const double *)h->d_cell_vert, p.n_slots);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

void drop_graph(dflo_hip_engine *h) {
  if (h->graph_exec) hipGraphExecDestroy(h->graph_exec);
  h->graph_exec = nullptr;
}

int check_handle(dflo_hip_handle h) { return h ? DFLO_OK : DFLO_ERR_BAD_PARAM; }

}  // namespace

extern "C" {

const char *dflo_hip_last_error(dflo_hip_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int dflo_hip_create(const dflo_mesh_t *mesh, const dflo_params_t *params, int device_id, dflo_hip_handle *out) {
  if (!mesh || !params || !out) { g_create_error = "null argument"; return DFLO_ERR_BAD_PARAM; }
  *out = nullptr;
  // consistency checks of the reference's parameter parsing (src/parameters.cc:536-550)
  if (mesh->degree < 1 || mesh->degree > DFLO_MAX_DEGREE) { g_create_error = "degree must be 1..3"; return DFLO_ERR_BAD_PARAM; }
  if (mesh->basis != DFLO_BASIS_QK && mesh->basis != DFLO_BASIS_PK) { g_create_error = "unknown basis"; return DFLO_ERR_BAD_PARAM; }
  if (mesh->basis == DFLO_BASIS_PK && mesh->mapping != DFLO_MAP_CARTESIAN) {
    g_create_error = "Pk basis is implemented for cartesian mapping only";
    return DFLO_ERR_UNSUPPORTED;
  }
  if (mesh->mapping != DFLO_MAP_CARTESIAN && mesh->mapping != DFLO_MAP_Q1) { g_create_error = "q2 mapping is not implemented"; return DFLO_ERR_UNSUPPORTED; }
  if (params->shock_indicator < DFLO_IND_LIMITER || params->shock_indicator >= DFLO_IND_U2) {
    g_create_error = "shock indicator must be limiter, density or energy (u2 belongs to the MOOD scheme)";
    return params->shock_indicator == DFLO_IND_U2 ? DFLO_ERR_UNSUPPORTED : DFLO_ERR_BAD_PARAM;
  }
  if (params->shock_indicator != DFLO_IND_LIMITER && mesh->mapping != DFLO_MAP_CARTESIAN) {
    g_create_error = "the KXRCF indicator is implemented for cartesian mapping";
    return DFLO_ERR_UNSUPPORTED;
  }
  if (params->limiter_type == DFLO_LIMITER_TVB && mesh->mapping != DFLO_MAP_CARTESIAN) {
    g_create_error = "TVB limiter is implemented only for cartesian mapping";  // src/parameters.cc:543-544
    return DFLO_ERR_BAD_PARAM;
  }
  if (params->flux_type < 0 || params->flux_type > DFLO_FLUX_HLLC) { g_create_error = "unknown flux"; return DFLO_ERR_BAD_PARAM; }

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    g_create_error = "no HIP device available: the dflo HIP engine has no CPU fallback";
    return DFLO_ERR_HIP;
  }
  if (device_id < 0 || device_id >= ndev) { g_create_error = "bad device id"; return DFLO_ERR_BAD_PARAM; }
  dflo_hip_engine *h = new dflo_hip_engine;
  h->device = device_id;
  h->prm = *params;
  h->degree = mesh->degree;
  h->N = mesh->degree + 1;
  h->basis = mesh->basis;
  if (const char *e = std::getenv("DFLO_GRAPH")) h->use_graph = std::atoi(e) != 0;
  h->ns = mesh->basis == DFLO_BASIS_PK ? h->N * (h->N + 1) / 2 : h->N * h->N;
  h->ndof = 4 * h->ns;
  h->mapping = mesh->mapping;
  h->geo = mesh->mapping == DFLO_MAP_CARTESIAN ? 0 : 1;
  int rc = build_plan(*mesh, 8, 8, h->plan, h->err);
  if (rc) { g_create_error = h->err; delete h; return rc; }
  h->bt = make_basis(h->degree);
  h->kb = make_kbasis(h->bt);
  // quadrature points of the boundary faces (fe_v.get_quadrature_points(), src/assemble_explicit.cc:164)
  h->bface_xy.resize(h->plan.bface_cell.size() * h->N * 2);
  for (size_t b = 0; b < h->plan.bface_cell.size(); ++b) {
    const double *v = &mesh->cell_vertices[(size_t)h->plan.bface_cell[b] * 8];
    const int f = h->plan.bface_face[b];
    for (int q = 0; q < h->N; ++q) {
      const double s = h->bt.x[q];
      const double xi = f == 0 ? 0.0 : (f == 1 ? 1.0 : s), eta = f == 2 ? 0.0 : (f == 3 ? 1.0 : s);
      for (int d = 0; d < 2; ++d)
        h->bface_xy[(b * h->N + q) * 2 + d] = (1 - xi) * (1 - eta) * v[d] + xi * (1 - eta) * v[2 + d] +
                                               (1 - xi) * eta * v[4 + d] + xi * eta * v[6 + d];
    }
  }
  // SSP-RK coefficients by degree (src/claw.cc:141-159)
  h->n_rk = h->degree == 1 ? 2 : 3;
  if (params->n_rk > 0) h->n_rk = std::min(params->n_rk, 3);
  if (h->n_rk == 2) { h->ark[0] = 0.0; h->ark[1] = 0.5; }
  if (h->n_rk == 3) { h->ark[0] = 0.0; h->ark[1] = 0.75; h->ark[2] = 1.0 / 3.0; }
  auto bail = [&](int code) { g_create_error = h->err; dflo_hip_destroy(h); return code; };
  if (hipSetDevice(device_id) != hipSuccess) { h->err = "hipSetDevice failed"; return bail(DFLO_ERR_HIP); }
  if (hipStreamCreate(&h->own_stream) != hipSuccess) { h->err = "hipStreamCreate failed"; return bail(DFLO_ERR_HIP); }
  h->stream = h->own_stream;
  const Plan &p = h->plan;
  const size_t nU = (size_t)p.n_slots * h->ndof;
  for (int i = 0; i < 3; ++i)
    if (hipMalloc((void **)&h->U[i], nU * sizeof(double)) != hipSuccess) { h->err = "hipMalloc(U) failed"; return bail(DFLO_ERR_NOMEM); }
  for (int i = 0; i < 2; ++i)
    if (hipMalloc((void **)&h->avg[i], (size_t)p.n_slots * 4 * sizeof(double)) != hipSuccess) { h->err = "hipMalloc(avg) failed"; return bail(DFLO_ERR_NOMEM); }
  if (hipMalloc((void **)&h->rhs, nU * sizeof(double)) != hipSuccess) { h->err = "hipMalloc(rhs) failed"; return bail(DFLO_ERR_NOMEM); }
  if (hipMalloc((void **)&h->user_buf, nU * sizeof(double)) != hipSuccess) { h->err = "hipMalloc(user_buf) failed"; return bail(DFLO_ERR_NOMEM); }
  for (int i = 0; i < 3; ++i) hipMemset(h->U[i], 0, nU * sizeof(double));
  std::vector<int32_t> kinds(p.bface_id.size());
  for (size_t b = 0; b < kinds.size(); ++b) kinds[b] = params->bc_kind[p.bface_id[b]];
  const size_t nb = std::max<size_t>(p.bface_cell.size(), 1) * h->N * 4;
  for (int w = 0; w < 2; ++w) {
    if (hipMalloc((void **)&h->bval[w], nb * sizeof(double)) != hipSuccess) { h->err = "hipMalloc(bval) failed"; return bail(DFLO_ERR_NOMEM); }
    hipMemset(h->bval[w], 0, nb * sizeof(double));
  }
  if ((rc = upload(h, &h->bface_kind, kinds))) return bail(rc);
  if ((rc = upload(h, &h->d_shard_count, p.shard_count))) return bail(rc);
  if ((rc = upload(h, &h->d_cell_face, p.cell_face))) return bail(rc);
  {  // fixed-pitch copies of the per-shard lists (+2 shards of slack: the kernel reads two shards ahead,
     // and 2*64*N face slots per shard so that unconditional loads stay in bounds)
    const int ns = p.n_shards + 2;
    h->face_pitch = 2 * 64 * h->N;
    if (p.max_faces > h->face_pitch) { h->err = "more than 2 faces per thread in a shard"; return bail(DFLO_ERR_UNSUPPORTED); }
    h->halo_pitch = std::max(p.max_halo, 1);
    std::vector<int4> hdr(ns, int4{0, 0, 0, 0});
    std::vector<int32_t> hp((size_t)ns * h->halo_pitch, 0);
    std::vector<uint32_t> fpad((size_t)ns * h->face_pitch, 0u);
    h->bnd_pitch = std::max(p.max_bnd, 1);
    std::vector<int32_t> bpad((size_t)ns * h->bnd_pitch, 0);
    for (int sidx = 0; sidx < p.n_shards; ++sidx) {
      const int nh = p.halo_begin[sidx + 1] - p.halo_begin[sidx], nf = p.face_begin[sidx + 1] - p.face_begin[sidx];
      hdr[sidx] = int4{p.shard_count[sidx], nf, nh, p.shard_bnd[sidx]};
      for (int k = 0; k < nh; ++k)
        hp[(size_t)sidx * h->halo_pitch + k] = p.halo_cells[p.halo_begin[sidx] + k] | (p.halo_faces[p.halo_begin[sidx] + k] << 28);
      for (int k = 0; k < nf; ++k) {
        const FaceRec &r = p.faces[p.face_begin[sidx] + k];
        fpad[(size_t)sidx * h->face_pitch + k] = pface_pack(r);
        if ((r.w0 >> 18) & 1) bpad[(size_t)sidx * h->bnd_pitch + ((r.w0 >> 20) & 0x3FF)] = r.w1;
      }
    }
    if ((rc = upload(h, &h->d_shard_hdr, hdr))) return bail(rc);
    if ((rc = upload(h, &h->d_halo_pad, hp))) return bail(rc);
    if ((rc = upload(h, &h->d_faces_pad, fpad))) return bail(rc);
    if ((rc = upload(h, &h->d_bnd_pad, bpad))) return bail(rc);
    if (h->prm.shock_indicator != DFLO_IND_LIMITER) {
      if ((rc = upload(h, &h->d_nbr_code, p.nbr_code))) return bail(rc);
      std::vector<double> z((size_t)p.n_slots, 0.0);
      if ((rc = upload(h, &h->d_shock, z))) return bail(rc);
    }
    if (h->geo == 1) {  // face geometry at the same pitch, [shard][3][face_pitch]
      std::vector<double> gpad((size_t)ns * 3 * h->face_pitch, 0.0);
      for (int sidx = 0; sidx < p.n_shards; ++sidx) {
        const int nf = p.face_begin[sidx + 1] - p.face_begin[sidx];
        for (int k = 0; k < nf; ++k)
          for (int j2 = 0; j2 < 3; ++j2)
            gpad[((size_t)sidx * 3 + j2) * h->face_pitch + k] = p.face_geom[((size_t)p.face_begin[sidx] + k) * 3 + j2];
      }
      if ((rc = upload(h, &h->d_fgeom_pad, gpad))) return bail(rc);
    }
  }
  if ((rc = upload(h, &h->d_lrbt, p.lrbt))) return bail(rc);
  if ((rc = upload(h, &h->d_rim_list, p.rim_shards))) return bail(rc);
  if ((rc = upload(h, &h->d_int_list, p.interior_shards))) return bail(rc);
  if ((rc = upload(h, &h->d_user_of, p.user_of))) return bail(rc);
  if ((rc = upload(h, &h->d_iid, p.iid))) return bail(rc);
  if ((rc = upload(h, &h->d_cell_h, p.cell_h))) return bail(rc);
  if (!params->global_time_step) {  // per-cell time steps, src/claw.cc:453,506
    std::vector<double> z((size_t)p.n_slots + 128, 0.0);
    if ((rc = upload(h, &h->d_dt_cell, z))) return bail(rc);
  }
  if (h->geo == 1) {
    if ((rc = upload(h, &h->d_cell_vert, p.cell_vert))) return bail(rc);
  }
  const size_t nsh = std::max(p.n_shards, 1);
  if (hipMalloc((void **)&h->shard_res, 3 * nsh * sizeof(double)) != hipSuccess ||
      hipMalloc((void **)&h->shard_dtmin, nsh * sizeof(double)) != hipSuccess ||
      hipMalloc((void **)&h->res_sq, 4 * sizeof(double)) != hipSuccess || hipMalloc((void **)&h->fin_partial, 4 * 32 * sizeof(double)) != hipSuccess ||
      hipMalloc((void **)&h->dt_dev, 4 * sizeof(double)) != hipSuccess || hipMalloc((void **)&h->flags, 4 * sizeof(int)) != hipSuccess) {
    h->err = "hipMalloc(scalars) failed";
    return bail(DFLO_ERR_NOMEM);
  }
  hipMemset(h->res_sq, 0, 4 * sizeof(double));
  hipMemset(h->dt_dev, 0, 4 * sizeof(double));
  hipMemset(h->flags, 0, 4 * sizeof(int));
  h->halo_stride = std::max(p.max_halo, 1) | 1;  // odd stride: the trace rows fall on different LDS banks
  h->max_fp = std::max(std::max(p.max_faces, 1) * h->N, 9 * 64 * h->N / 4 + 1);  // Fh also hosts the row partials (5 N rows of 64) and the positivity minima (3 N) or the slope partials (4 N)
  {
    const char *e = getenv("DFLO_FUSE_POS");
    h->fuse_pos = h->prm.pos_lim && h->prm.limiter_type == DFLO_LIMITER_NONE && h->basis == DFLO_BASIS_QK && !(e && e[0] == '0');
    const char *e2 = getenv("DFLO_LIM_MASK");
    // (measured: the marks cost the stage kernel ~11 %; the pass they shorten reads (k+1)^2 values per cell and component,
    //  which pays from k = 2 on -- C4 +7 % -- and not for k = 1 -- C3 -7 %; DFLO_LIM_MASK=1 forces them, 0 forbids them)
    const bool want_marks = e2 ? e2[0] != '0' : h->N >= 3;
    if (h->prm.limiter_type == DFLO_LIMITER_TVB && h->basis == DFLO_BASIS_QK && h->geo == 0 && want_marks) {
      const size_t nb = (size_t)std::max(p.n_shards, 1) * sizeof(unsigned long long);
      if (hipMalloc((void **)&h->lim_mask, nb) != hipSuccess) {
        h->err = "hipMalloc(limiter marks) failed";
        return bail(DFLO_ERR_NOMEM);
      }
      hipMemset(h->lim_mask, 0, nb);
    }
  }
  {
    const int rows = 4 * h->N * h->N + (h->prm.flux_type == DFLO_FLUX_LXF ? 3 : 0);  // nodal image (also for Pk)
    const int trows = 4 * h->N + (h->prm.flux_type == DFLO_FLUX_LXF ? 3 : 0);
    const size_t mf = (std::max(p.max_faces, 1) + 1) & ~1;   // = StageArgs::max_faces
    h->lds_bytes = ((size_t)rows * 65 + (size_t)trows * h->halo_stride + 4 * (size_t)h->max_fp + mf / 2 +
                    (size_t)p.max_bnd * 4 * h->N + (p.max_bnd + 2) / 2 + (h->geo == 1 ? 3 * mf : 0)) * sizeof(double);
  }

  if (h->lds_bytes > 160 * 1024) { h->err = "shard halo too large for LDS"; return bail(DFLO_ERR_UNSUPPORTED); }
  if (h->lds_bytes > 64 * 1024) {
    for (int mode = 0; mode < 3; ++mode) {
      stage_fn fn = h->basis == DFLO_BASIS_PK ? pick_pk(h->N, h->prm.flux_type, mode) : pick_stage(h->N, h->prm.flux_type, mode, h->geo, h->fuse_pos ? 1 : (h->lim_mask ? 2 : 0));
      if (hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes) != hipSuccess) {
        h->err = "cannot raise dynamic LDS limit";
        return bail(DFLO_ERR_HIP);
      }
    }
  }
  {  // persistent grid: as many workgroups as stay resident, a multiple of 8 (one run of shards per XCD)
    stage_fn fn = pick_stage(h->N, h->prm.flux_type, 1, h->geo);
    int per_cu = 0, n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device_id);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)fn, 64 * h->N, h->lds_bytes) != hipSuccess || per_cu < 1)
      per_cu = 1;
    if (n_cu < 1) n_cu = 256;
    h->stage_grid = grid_for(h->plan.n_shards);  // one workgroup per shard (per_cu of them resident per CU)
    // a workgroup touches the index data of the shard that the same XCD takes ~1.5 residency rounds later
    h->prefetch_ahead = std::max(8, (per_cu * n_cu / 8) * 3 / 2);
#ifdef DFLO_PHASE_TIMING
    hipMalloc((void **)&h->phase_cycles, (size_t)h->stage_grid * 32 * sizeof(unsigned long long));
    hipMemset(h->phase_cycles, 0, (size_t)h->stage_grid * 32 * sizeof(unsigned long long));
#endif
  }
  *out = h;
  return DFLO_OK;
}

int dflo_hip_destroy(dflo_hip_handle h) {
  if (!h) return DFLO_OK;
  hipSetDevice(h->device);
  if (h->own_stream) hipStreamSynchronize(h->own_stream);
  drop_graph(h);
  for (int i = 0; i < 3; ++i) hipFree(h->U[i]);
  for (int i = 0; i < 2; ++i) { hipFree(h->avg[i]); hipFree(h->bval[i]); }
  hipFree(h->rhs); hipFree(h->user_buf); hipFree(h->bface_kind);
  hipFree(h->d_bc_ops); hipFree(h->d_bc_consts); hipFree(h->d_bc_prog); hipFree(h->d_bc_faces); hipFree(h->d_bface_id); hipFree(h->d_bxy);
  hipFree(h->d_shard_count);
  hipFree(h->d_bnd_pad); hipFree(h->d_nbr_code); hipFree(h->d_shock); hipFree(h->lim_mask); hipFree(h->d_faces_pad); hipFree(h->d_shard_hdr); hipFree(h->d_halo_pad); hipFree(h->d_cell_face); hipFree(h->d_lrbt); hipFree(h->d_user_of); hipFree(h->d_iid);
  hipFree(h->d_rim_list); hipFree(h->d_int_list);
  hipFree(h->d_cell_h); hipFree(h->d_dt_cell); hipFree(h->d_cell_vert); hipFree(h->d_fgeom_pad); hipFree(h->shard_res); hipFree(h->shard_dtmin); hipFree(h->res_sq); hipFree(h->fin_partial); hipFree(h->dt_dev);
  hipFree(h->flags); hipFree(h->d_send_slots); hipFree(h->ghost_stage);
  for (auto &e : h->ev_pool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  if (h->ev_rim) { hipEventDestroy(h->ev_rim); hipEventDestroy(h->ev_unpack); }
  if (h->own_stream) hipStreamDestroy(h->own_stream);
  delete h;
  return DFLO_OK;
}

int dflo_hip_set_stream(dflo_hip_handle h, void *s) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  h->stream = s ? (hipStream_t)s : h->own_stream;
  return DFLO_OK;
}

int64_t dflo_hip_n_dofs(dflo_hip_handle h) { return h ? (int64_t)h->plan.n_cells * h->ndof : 0; }
int32_t dflo_hip_dofs_per_cell(dflo_hip_handle h) { return h ? h->ndof : 0; }
int32_t dflo_hip_n_rk(dflo_hip_handle h) { return h ? h->n_rk : 0; }
int32_t dflo_hip_n_boundary_faces(dflo_hip_handle h) { return h ? (int32_t)h->plan.bface_cell.size() : 0; }

int dflo_hip_set_solution(dflo_hip_handle h, const double *u) {
  if (check_handle(h) || !u) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  const Plan &p = h->plan;
  const size_t n = (size_t)p.n_cells * h->ndof;
  HIPCHK(h, hipMemcpyAsync(h->user_buf, u, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  const long long tot = (long long)p.n_slots * h->ndof;
  h->cur = h->old = 0;
  hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->user_buf, h->U[0],
                     h->d_user_of, p.n_slots, h->ndof);
  HIPCHK(h, hipGetLastError());
  int rc = launch_average(h);
  if (rc) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return DFLO_OK;
}

int dflo_hip_get_solution(dflo_hip_handle h, double *u) {
  if (check_handle(h) || !u) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  const Plan &p = h->plan;
  const long long tot = (long long)p.n_cells * h->ndof;
  hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->user_buf,
                     h->U[h->cur], h->d_iid, p.n_cells, h->ndof);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(u, h->user_buf, tot * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return DFLO_OK;
}

int dflo_hip_get_cell_average(dflo_hip_handle h, double *avg) {
  if (check_handle(h) || !avg) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  const Plan &p = h->plan;
  std::vector<double> tmp((size_t)p.n_slots * 4);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(tmp.data(), h->avg[h->avg_cur], tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int c = 0; c < p.n_cells; ++c) {
    const int s = p.iid[c];
    for (int k = 0; k < 4; ++k) avg[(size_t)c * 4 + k] = tmp[((size_t)(s >> 6) * 4 + k) * 64 + (s & 63)];
  }
  return DFLO_OK;
}

int dflo_hip_boundary_faces(dflo_hip_handle h, int32_t *cell, int32_t *face, int32_t *boundary_id, double *xy) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  const Plan &p = h->plan;
  for (size_t b = 0; b < p.bface_cell.size(); ++b) {
    if (cell) cell[b] = p.bface_cell[b];
    if (face) face[b] = p.bface_face[b];
    if (boundary_id) boundary_id[b] = p.bface_id[b];
  }
  if (xy) std::memcpy(xy, h->bface_xy.data(), h->bface_xy.size() * sizeof(double));
  return DFLO_OK;
}

int dflo_hip_set_boundary_values(dflo_hip_handle h, int which, const double *values) {
  if (check_handle(h) || which < 0 || which > 1 || !values) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  const size_t n = h->plan.bface_cell.size() * h->N * 4;
  if (n == 0) return DFLO_OK;
  HIPCHK(h, hipMemcpyAsync(h->bval[which], values, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return DFLO_OK;
}

int dflo_hip_get_boundary_values(dflo_hip_handle h, int which, double *values) {
  if (check_handle(h) || which < 0 || which > 1 || !values) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  const size_t n = h->plan.bface_cell.size() * h->N * 4;
  if (n == 0) return DFLO_OK;
  HIPCHK(h, hipMemcpyAsync(values, h->bval[which], n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return DFLO_OK;
}

int dflo_hip_set_boundary_program(dflo_hip_handle h, int32_t boundary_id, int32_t component, int32_t n_ops, const int32_t *ops,
                                  int32_t n_consts, const double *consts) {
  if (check_handle(h) || boundary_id < 0 || boundary_id >= DFLO_MAX_BOUNDARIES || component < 0 || component > 3 || n_ops < 0 ||
      n_consts < 0 || (n_ops > 0 && !ops) || (n_consts > 0 && !consts))
    return DFLO_ERR_BAD_PARAM;
  // validate: known opcodes, constant indices in range, stack discipline within kExprStack, one value left
  int sp = 0;
  for (int i = 0; i < n_ops; ++i) {
    const int op = ops[2 * i];
    if (op < 0 || op >= DFLO_OP_COUNT) { h->err = "boundary program: unknown opcode"; return DFLO_ERR_BAD_PARAM; }
    if (op == DFLO_OP_CONST && (ops[2 * i + 1] < 0 || ops[2 * i + 1] >= n_consts)) { h->err = "boundary program: constant index out of range"; return DFLO_ERR_BAD_PARAM; }
    const bool binary = (op >= DFLO_OP_ADD && op <= DFLO_OP_OR) || op == DFLO_OP_MIN || op == DFLO_OP_MAX || op == DFLO_OP_ATAN2;
    const int pops = op <= DFLO_OP_T ? 0 : (op == DFLO_OP_SEL ? 3 : (binary ? 2 : 1));
    if (sp < pops) { h->err = "boundary program: stack underflow"; return DFLO_ERR_BAD_PARAM; }
    sp += 1 - pops;
    if (sp > kExprStack) { h->err = "boundary program: stack too deep"; return DFLO_ERR_BAD_PARAM; }
  }
  if (n_ops > 0 && sp != 1) { h->err = "boundary program: must leave exactly one value"; return DFLO_ERR_BAD_PARAM; }
  std::vector<int32_t> &o = h->bc_ops[boundary_id][component];
  h->n_bc_programs += (n_ops > 0 ? 1 : 0) - (o.empty() ? 0 : 1);
  o.assign(ops, ops + 2 * (size_t)n_ops);
  h->bc_consts[boundary_id][component].assign(consts, consts + (n_ops > 0 ? n_consts : 0));
  h->bc_dirty = true;
  return DFLO_OK;
}

int dflo_hip_residual(dflo_hip_handle h, int which, double *rhs_out) {
  if (check_handle(h) || !rhs_out) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  int rc = launch_stage(h, 0, 0.0, h->rhs, which);
  if (rc) return rc;
  const Plan &p = h->plan;
  const long long tot = (long long)p.n_cells * h->ndof;
  hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->user_buf, h->rhs,
                     h->d_iid, p.n_cells, h->ndof);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(rhs_out, h->user_buf, tot * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return DFLO_OK;
}

int dflo_hip_compute_cell_average(dflo_hip_handle h) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  return launch_average(h);
}

int dflo_hip_compute_dt(dflo_hip_handle h, double elapsed_time, double *dt) {
  if (check_handle(h) || !dt) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  if (h->prm.global_time_step && h->prm.cfl <= 0.0) {  // src/claw.cc:456-460
    *dt = h->prm.time_step;
    return DFLO_OK;
  }
  // per-shard minima from the stored cell averages (src/claw.cc:486-511)
  const Plan &p = h->plan;
  if (h->geo == 0)
    hipLaunchKernelGGL(dt_kernel, dim3(p.n_shards), dim3(64), 0, h->stream, h->avg[h->avg_cur], h->d_cell_h, p.h,
                       p.uniform_h ? 1 : 0, h->d_shard_count, h->shard_dtmin, h->prm.cfl, h->degree, h->d_dt_cell);
  else
    launch_dt_q(h);
  HIPCHK(h, hipGetLastError());
  double tt[4] = {0, elapsed_time, 0, 0};
  HIPCHK(h, hipMemcpyAsync(h->dt_dev, tt, sizeof(tt), hipMemcpyHostToDevice, h->stream));
  FinalArgs f{};
  f.shard_res = h->shard_res;
  f.shard_dtmin = h->shard_dtmin;
  f.res_sq = h->res_sq;
  f.dt_dev = h->dt_dev;
  f.n_shards = p.n_shards;
  f.n_stages = 0;
  f.res_stride = 0;
  f.do_dt = 1;
  f.advance_time = 0;
  f.dt_host = -1.0;
  f.time_step = h->prm.time_step;
  f.final_time = h->prm.final_time;
  f.global_rules = h->prm.global_time_step;
  f.partial = h->fin_partial;
  f.counter = h->flags + 2;
  hipLaunchKernelGGL(finalize_kernel, dim3(fin_grid(f.n_shards)), dim3(256), 0, h->stream, f);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(tt, h->dt_dev, sizeof(tt), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  *dt = tt[0];
  return DFLO_OK;
}

int dflo_hip_stage(dflo_hip_handle h, int rk, double dt) {
  if (check_handle(h) || rk < 0 || rk >= h->n_rk) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  return launch_stage(h, rk, dt, nullptr, -1);
}

int dflo_hip_end_step(dflo_hip_handle h) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  h->old = h->cur;  // old_solution = current_solution (src/claw.cc:1110): a pointer swap here
  return DFLO_OK;
}

int dflo_hip_step(dflo_hip_handle h, double dt, double *res_norm0, double *res_norm) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  for (int rk = 0; rk < h->n_rk; ++rk) {
    int rc = launch_stage(h, rk, dt, nullptr, -1);
    if (rc) return rc;
  }
  h->old = h->cur;
  if (res_norm0 || res_norm) {
    double r[4];
    HIPCHK(h, hipMemcpyAsync(r, h->res_sq, sizeof(r), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (res_norm0) *res_norm0 = std::sqrt(r[0]);
    if (res_norm) *res_norm = std::sqrt(r[h->n_rk - 1]);
  }
  return dflo_hip_check(h);
}

int dflo_hip_advance(dflo_hip_handle h, int n_steps, double *elapsed_time_inout) {
  if (check_handle(h) || n_steps < 0 || !elapsed_time_inout) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  // first dt from the current cell averages, then dt and time stay on the device
  double dt0;
  int rc = dflo_hip_compute_dt(h, *elapsed_time_inout, &dt0);
  if (rc) return rc;
  if (h->prm.global_time_step && h->prm.cfl <= 0.0) {
    double tt[4] = {dt0, *elapsed_time_inout, dt0, 0};
    HIPCHK(h, hipMemcpyAsync(h->dt_dev, tt, sizeof(tt), hipMemcpyHostToDevice, h->stream));
  }
  int s = 0;
  if (h->use_graph && !h->timing && h->cur == h->old) {
    // the (solution, average) buffer indices come back to where they started after 2 steps (avg_cur flips
    // n_rk times per step): capture those once, replay them for the bulk of the steps
    const int period = 2;
    if (n_steps >= 2 * period) {
      if (h->graph_exec && (h->graph_cur != h->cur || h->graph_avg != h->avg_cur || h->graph_stream != h->stream)) drop_graph(h);
      if (!h->graph_exec) {
        const int cur0 = h->cur, avg0 = h->avg_cur;
        hipGraph_t g = nullptr;
        bool ok = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
          for (int k = 0; k < period && !rc; ++k) {
            for (int rk = 0; rk < h->n_rk && !rc; ++rk) rc = launch_stage(h, rk, -1.0, nullptr, -1);
            h->old = h->cur;
          }
          ok = hipStreamEndCapture(h->stream, &g) == hipSuccess && !rc && g;
          if (ok) ok = hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0) == hipSuccess;
          if (g) hipGraphDestroy(g);
          // capture records, it does not run: put the indices back
          h->cur = h->old = cur0;
          h->avg_cur = avg0;
          h->pending_rk = -1;
        }
        if (!ok || h->cur != cur0) {
          (void)hipGetLastError();
          h->graph_exec = nullptr;
          h->use_graph = false;  // fall back to plain launches for good
          rc = DFLO_OK;
        } else {
          h->graph_steps = period;
          h->graph_cur = cur0;
          h->graph_avg = avg0;
          h->graph_stream = h->stream;
        }
      }
      if (h->graph_exec) {
        for (; s + period <= n_steps; s += period) HIPCHK(h, hipGraphLaunch(h->graph_exec, h->stream));
      }
    }
  }
  for (; s < n_steps; ++s) {
    for (int rk = 0; rk < h->n_rk; ++rk) {
      rc = launch_stage(h, rk, -1.0, nullptr, -1);
      if (rc) return rc;
    }
    h->old = h->cur;
  }
  double tt[4];
  HIPCHK(h, hipMemcpyAsync(tt, h->dt_dev, sizeof(tt), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->timing) time_collect(h);
  *elapsed_time_inout = tt[1];
  return dflo_hip_check(h);
}

int dflo_hip_apply_limiter(dflo_hip_handle h) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  if (h->prm.limiter_type == DFLO_LIMITER_NONE) return DFLO_OK;
  const int rc = launch_indicator(h, 0);  // run(): compute_shock_indicator(); apply_limiter();  src/claw.cc:1000-1001
  if (rc) return rc;
  return launch_limiter(h, 1, 0, 0);
}

int dflo_hip_compute_shock_indicator(dflo_hip_handle h) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  return launch_indicator(h, 0);
}
int dflo_hip_get_shock_indicator(dflo_hip_handle h, double *out) {
  if (check_handle(h) || !out) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  const Plan &p = h->plan;
  if (!h->d_shock) {  // type "limiter": shock_indicator = 1e20 (src/indicator.cc:21)
    for (int c = 0; c < p.n_cells; ++c) out[c] = 1.0e20;
    return DFLO_OK;
  }
  std::vector<double> z((size_t)p.n_slots);
  HIPCHK(h, hipMemcpyAsync(z.data(), h->d_shock, z.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int c = 0; c < p.n_cells; ++c) out[c] = z[p.iid[c]];
  return DFLO_OK;
}
int dflo_hip_apply_positivity_limiter(dflo_hip_handle h) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  int rc = launch_limiter(h, 0, 1, 0);
  if (rc) return rc;
  return dflo_hip_check(h);
}

int dflo_hip_stage_update(dflo_hip_handle h, int rk, double dt) {
  if (check_handle(h) || rk < 0 || rk >= h->n_rk) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  int rc = open_stage(h, rk, dt, false, -1);
  if (rc) return rc;
  return launch_update(h, nullptr, 0);
}

int dflo_hip_stage_open(dflo_hip_handle h, int rk, double dt) {
  if (check_handle(h) || rk < 0 || rk >= h->n_rk) return DFLO_ERR_BAD_PARAM;
  return open_stage(h, rk, dt, false, -1);
}

int dflo_hip_stage_update_part(dflo_hip_handle h, int part) {
  if (check_handle(h) || part < 0 || part > 2 || h->pending_rk < 0) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  return launch_update(h, nullptr, part);
}

int dflo_hip_stage_limit_part(dflo_hip_handle h, int part) {
  if (check_handle(h) || part < 0 || part > 2) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  return launch_stage_limiter(h, part);
}

int dflo_hip_stage_finish(dflo_hip_handle h) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  return launch_finish(h);
}

int dflo_hip_n_rim_shards(dflo_hip_handle h) { return h ? (int)h->plan.rim_shards.size() : 0; }

// ---- the overlapped stage as four host calls (main stream = compute, comm stream = halo traffic):
//   stage_rim      main: wait for the previous unpack, open the stage, advance the rim shards
//   stage_rim_send comm: wait for the rim shards, (TVB: receive averages,) limit them, pack
//   stage_rim_recv comm: unpack what the transport delivered
//   stage_interior main: interior shards, their limiter, reductions
static int ensure_events(dflo_hip_engine *h) {
  if (h->ev_rim) return DFLO_OK;
  HIPCHK(h, hipEventCreateWithFlags(&h->ev_rim, hipEventDisableTiming));
  HIPCHK(h, hipEventCreateWithFlags(&h->ev_unpack, hipEventDisableTiming));
  return DFLO_OK;
}

int dflo_hip_stage_rim(dflo_hip_handle h, int rk, double dt, void *main_stream) {
  if (check_handle(h) || rk < 0 || rk >= h->n_rk) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  int rc = ensure_events(h);
  if (rc) return rc;
  h->stream = main_stream ? (hipStream_t)main_stream : h->own_stream;
  if (h->unpack_pending) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_unpack, 0));
  rc = open_stage(h, rk, dt, false, -1);
  if (rc) return rc;
  rc = launch_update(h, nullptr, 1);
  if (rc) return rc;
  HIPCHK(h, hipEventRecord(h->ev_rim, h->stream));
  return DFLO_OK;
}

// what = 0: limit the rim shards and pack their DoFs; 1: pack the cell averages (TVB, before the limiter);
// 2: unpack received averages (TVB), then limit and pack the DoFs
int dflo_hip_stage_rim_send(dflo_hip_handle h, void *comm_stream, int what, const void *avg_recv, void *send_buffer) {
  if (check_handle(h) || !comm_stream) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  hipStream_t main = h->stream;
  h->stream = (hipStream_t)comm_stream;
  int rc = DFLO_OK;
  if (what != 2) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_rim, 0));
  if (what == 1) {
    rc = dflo_hip_pack_send_avg(h, send_buffer);
  } else {
    if (what == 2) rc = dflo_hip_unpack_ghost_avg(h, avg_recv);
    if (!rc) rc = launch_stage_limiter(h, 1);
    if (!rc) rc = dflo_hip_pack_send(h, send_buffer);
  }
  h->stream = main;
  return rc;
}

int dflo_hip_stage_rim_recv(dflo_hip_handle h, void *comm_stream, const void *recv_buffer) {
  if (check_handle(h) || !comm_stream) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  hipStream_t main = h->stream;
  h->stream = (hipStream_t)comm_stream;
  int rc = dflo_hip_unpack_ghost(h, recv_buffer);
  if (!rc) {
    if (hipEventRecord(h->ev_unpack, h->stream) != hipSuccess) rc = DFLO_ERR_HIP;
    h->unpack_pending = true;
  }
  h->stream = main;
  return rc;
}

int dflo_hip_stage_interior(dflo_hip_handle h) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  int rc = launch_update(h, nullptr, 2);
  if (!rc) rc = launch_stage_limiter(h, 2);
  if (!rc) rc = launch_finish(h);
  return rc;
}

// main stream waits for the last unpack (before the state is read or a non-overlapped call follows)
int dflo_hip_stage_join(dflo_hip_handle h) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  if (h->unpack_pending) {
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_unpack, 0));
    h->unpack_pending = false;
  }
  return DFLO_OK;
}

int dflo_hip_stage_limit(dflo_hip_handle h) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  return launch_limit_finalize(h);
}

int dflo_hip_check(dflo_hip_handle h) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  int f[4];
  HIPCHK(h, hipMemcpyAsync(f, h->flags, sizeof(f), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (f[0]) { h->err = "Fatal: Negative states"; return DFLO_ERR_NEGATIVE_MEAN_STATE; }
  if (f[1]) { h->err = "Problem in positivity limiter"; return DFLO_ERR_POSITIVITY_NO_ROOT; }
  return DFLO_OK;
}

int dflo_hip_synchronize(dflo_hip_handle h) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return DFLO_OK;
}

int dflo_hip_stage_timing(dflo_hip_handle h, int enable, double *avg_ms, int64_t *n) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  time_collect(h);
  // a stage may have been launched in two parts (rim + interior): the average is per stage
  if (avg_ms) *avg_ms = h->t_stages ? h->t_accum_ms / (double)h->t_stages : 0.0;
  if (n) *n = h->t_stages;
  h->t_accum_ms = 0;
  h->t_count = 0;
  h->t_stages = 0;
  h->t_seen = 0;
  h->t_sample = false;
  h->timing = enable != 0;
  return DFLO_OK;
}

// ---------------------------------------------------------------- halo seam
int dflo_hip_set_send_cells(dflo_hip_handle h, int32_t n, const int32_t *cells) {
  if (check_handle(h) || n < 0 || (n > 0 && !cells)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  std::vector<int32_t> slots(n);
  for (int i = 0; i < n; ++i) {
    if (cells[i] < 0 || cells[i] >= h->plan.n_owned) { h->err = "send cell is not an owned cell"; return DFLO_ERR_COMM; }
    slots[i] = h->plan.iid[cells[i]];
  }
  hipFree(h->d_send_slots);
  h->d_send_slots = nullptr;
  h->n_send = n;
  return upload(h, &h->d_send_slots, slots);
}

int dflo_hip_pack_send(dflo_hip_handle h, void *device_buffer) {
  if (check_handle(h) || !device_buffer) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  if (h->n_send == 0) return DFLO_OK;
  const long long tot = (long long)h->n_send * h->ndof;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, (double *)device_buffer,
                     h->U[h->cur], h->d_send_slots, h->n_send, h->ndof);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

int dflo_hip_pack_send_avg(dflo_hip_handle h, void *device_buffer) {
  if (check_handle(h) || !device_buffer) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  if (h->n_send == 0) return DFLO_OK;
  const long long tot = (long long)h->n_send * 4;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, (double *)device_buffer,
                     h->avg[h->avg_cur], h->d_send_slots, h->n_send, 4);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

int dflo_hip_n_ghost_cells(dflo_hip_handle h) { return h ? h->plan.n_cells - h->plan.n_owned : 0; }

int dflo_hip_unpack_ghost(dflo_hip_handle h, const void *device_buffer) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  const Plan &p = h->plan;
  const int n_ghost = p.n_cells - p.n_owned;
  if (n_ghost == 0) return DFLO_OK;
  if (!device_buffer) return DFLO_ERR_BAD_PARAM;
  hipLaunchKernelGGL(unpack_ghost_kernel, dim3((4 * n_ghost + 127) / 128), dim3(128), 0, h->stream, (const double *)device_buffer,
                     h->U[h->cur], h->avg[h->avg_cur], p.n_shards * 64, n_ghost, h->ndof, h->kb,
                     h->basis == DFLO_BASIS_PK ? -h->N : h->N, (const double *)h->d_cell_vert, p.n_slots);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

int dflo_hip_unpack_ghost_avg(dflo_hip_handle h, const void *device_buffer) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  const Plan &p = h->plan;
  const int n_ghost = p.n_cells - p.n_owned;
  if (n_ghost == 0) return DFLO_OK;
  if (!device_buffer) return DFLO_ERR_BAD_PARAM;
  hipLaunchKernelGGL(unpack_ghost_avg_kernel, dim3((n_ghost + 63) / 64), dim3(64), 0, h->stream,
                     (const double *)device_buffer, h->avg[h->avg_cur], p.n_shards * 64, n_ghost);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

int dflo_hip_scalar_ptrs(dflo_hip_handle h, void **dt_ptr, void **res_ptr) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  if (dt_ptr) *dt_ptr = h->dt_dev;
  if (res_ptr) *res_ptr = h->res_sq;
  return DFLO_OK;
}

int dflo_hip_debug_math(int n, const double *x, double *rcp_out, double *sqrt_out) {
  double *dx = nullptr, *dr = nullptr, *ds = nullptr;
  if (hipMalloc((void **)&dx, n * sizeof(double)) != hipSuccess || hipMalloc((void **)&dr, n * sizeof(double)) != hipSuccess ||
      hipMalloc((void **)&ds, n * sizeof(double)) != hipSuccess)
    return DFLO_ERR_HIP;
  hipMemcpy(dx, x, n * sizeof(double), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(debug_math_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dr, ds, n);
  hipMemcpy(rcp_out, dr, n * sizeof(double), hipMemcpyDeviceToHost);
  hipMemcpy(sqrt_out, ds, n * sizeof(double), hipMemcpyDeviceToHost);
  hipFree(dx); hipFree(dr); hipFree(ds);
  return hipGetLastError() == hipSuccess ? DFLO_OK : DFLO_ERR_HIP;
}

/* developer probe: per-workgroup cycle counts of the stage kernel's phases (library built with
 * -DDFLO_PHASE_TIMING); returns the grid size, 0 when the probe is compiled out. */
int dflo_hip_debug_phase_cycles(dflo_hip_handle h, unsigned long long *out, int max_groups) {
  if (!h || !h->phase_cycles) return 0;
  const int n = std::min(max_groups, h->stage_grid);
  hipStreamSynchronize(h->stream);
  hipMemcpy(out, h->phase_cycles, (size_t)n * 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  return h->stage_grid;
}

int dflo_hip_apply_dt_rules(dflo_hip_handle h) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  hipLaunchKernelGGL(dt_rules_kernel, dim3(1), dim3(1), 0, h->stream, h->dt_dev, h->prm.time_step, h->prm.final_time);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

}  // extern "C"
#endif  // DFLO_STAGE_N
