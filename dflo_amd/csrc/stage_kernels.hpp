// stage_kernels.hpp -- the stage kernels (Qk: squares and bilinear cells; Pk) and their dispatch by (flux, mode, geometry, limiter variant)
// Part of the device side of engine.hip (see there for the layout of the data and of a stage).
#pragma once
#include "kernels_common.hpp"

namespace dflo {

// Wavefronts per SIMD the kernels are built for (register budgets: 3 -> 168, 2 -> 256).  Measured alternatives: Q2 at 4 (128
// registers: own row read back from LDS, later stages 190 -> 206 us), Q3 later stages at 3 (111 -> 118 us), P3 later stages at 2
// (225 registers: P3 KFVS 98 000 instead of 114 000 MDoF/s).
constexpr int kQ3FirstStageWaves = 3;   // first-stage Q3 kernel on squares: own row and own G row read back from the LDS image
constexpr int kQ2Waves = 3;
// (Q4 on squares built for 3 / 2 wavefronts per SIMD in the first / later stages: 121 700 against 122 100 MDoF/s; 3 / 3: spills, 87 000)
constexpr bool kPkLeanLater = true;     // P3: the later stages built like the first one (161 registers, no spills)
#ifdef DFLO_NO_PK_LEAN_HIGH
constexpr bool kPkLeanHigh = false;
#else
constexpr bool kPkLeanHigh = true;      // P4, P5: the own row and the own G row come back from the LDS image (as in the Qk kernels of those degrees)
#endif
// Q4 on squares, later stages: u(n) and the row's own u(s) are read where the update combines them (the first from memory, the second
// a second time, from the cache) instead of being held across the flux phase and phase C -- 226 -> <= 168 registers, so that TWO
// workgroups of five wavefronts fit a CU (75 KB of LDS each) like the first stage's
#ifdef DFLO_NO_LATE_Q4
constexpr bool kLateQ4 = false;
#else
constexpr bool kLateQ4 = true;
#endif
// ... and on bilinear cells at degree 5 (256 registers and 188-436 bytes of scratch per lane otherwise): u(n), and the row's own u(s)
// for the update, are read at the combine in every stage, and the row update takes its own row and its own eta part from the LDS image
// (LEANQ).  Measured on the forward-step mesh, HLLC: Q5 70 000 -> 80 700 MDoF/s; Q4 the same way (206-228 registers, no scratch, instead of
// 256 + 28-204 B) 81 500 -> 73 600, not applied; Q3 (C5) -1.7 %, not applied (profiles/r06/ab_bilinear_q4q5.txt, ab_hi7.txt, LAB R6.9)
template <int N, int GEO>
constexpr bool late_q1() { return kLateQ4 && N >= 6 && GEO == 1; }
template <int N, int MODE, int GEO>
constexpr bool late_loads() { return kLateQ4 && N == 5 && MODE == 1 && GEO == 0; }   // (Q3 and Q2 the same way, one workgroup more per CU: -12 .. -25 %, LAB R6.8)
// ------------------------------------------------------------------ the stage kernel
// One workgroup of N wavefronts per shard: lane = cell, wavefront = node row b of the (k+1)^2
// collocation nodes, so control flow is wave-uniform and every global access is a coalesced
// 512-byte line.  LDS image: Us[ndof (+3: u, v, c of the cell average for LxF)][65] the own 64 cells,
// Th[4N][halo_stride] one column per face: first the faces shared with halo cells (the column holds the halo cell's trace until
// the flux of the face replaces it), then the other faces (fluxes only); Av[3][halo_cols] (LxF: u, v, c of the halo cells'
// averages) and the shard's boundary data.  The packed face records stay in registers.

// MF (N = 4; north_star: "MFMA only for the dense per-element basis contractions at higher order"; src/assemble_explicit.cc:85-115):
// the eta-derivative of phase C on the matrix pipe.  Per cell  O[c][aa][B] = sum_q G[c][(aa, q)] Dm[q][B]  is four 4 x 4 x 4
// products, one per component: exactly one v_mfma_f64_4x4x4_4b_f64 (4 blocks).  Operand layout measured with
// tools/mfma_4x4_layout_probe.hip: A_b[i][k] in lane 16 k + 4 b + i, B_b[k][n] in lane 16 k + 4 b + n, D_b[i][n] in lane
// 16 i + 4 b + n.  With b = component, i = node aa, k = source row q, n = target row B, wave w takes the cells 16 w .. 16 w + 15:
// lane l fetches G[c][aa, q] of the cell from column `cell` of the LDS image (a gather over its 64 rows), the matrix pipe forms
// the sums for all four target rows at once, and the result goes back into the same column (a column belongs to one wave; all
// 16 columns are read before the first is written); after a barrier wave B reads its 16 sums, lane = cell, from the rows
// (B, c, aa).  Dm = DW (squares) or D (bilinear cells: the metric factors are folded into the rows beforehand).
template <int N, bool WEIGHTED>
__device__ __forceinline__ void eta_derivative_mfma(double *Us, const int S, const int lane, const int w) {
  static_assert(N == 4, "one 4x4x4 block per component: Q3 only");
  constexpr int NS = N * N;
  const int l = lane;
  const int ra = (((l >> 2) & 3) * NS + (l & 3) + N * (l >> 4)) * S;   // row of A: c = (l / 4) % 4, aa = l % 4, q = l / 16
  const int rd = ((l & 3) * NS + ((l >> 2) & 3) * N + (l >> 4)) * S;   // row of D: B = l % 4, c = (l / 4) % 4, aa = l / 16 -> O[B][c][aa]
  double bq = 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k)
#pragma unroll
    for (int n = 0; n < N; ++n) bq = ((l >> 4) == k && (l & 3) == n) ? (WEIGHTED ? CB<N>::t.DW[k][n] : CB<N>::t.D[k][n]) : bq;
  double av[16];
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) av[jj] = Us[ra + 16 * w + jj];
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) av[jj] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[jj], bq, 0.0, 0, 0, 0);
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) Us[rd + 16 * w + jj] = av[jj];
}

// phase C for node row B of every cell of the shard (lane = cell)
template <int N, int B, int MODE, int POS, int STREAM, int MF = 0>
__device__ __forceinline__ void row_update(const StageArgs &a, double *Us, const int S, const double *Fh,
                                           double *red, int shard, int lane, bool active, double h,
                                           const uint16_t (&cref)[4], const double (&uold)[4][N],
                                           const double (&Wrow)[N][4], double (&unew)[4][N], const double dt) {
  const int HS = a.halo_stride;   // row stride of the trace / flux table (Fh)
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
  // Q3, first stage (LEAN): that kernel is built for 3 wavefronts per SIMD (168 registers; measured 100 -> 85 us, while the
  // later stages, which also hold u(n), lose at 3) -- the state of the own row comes back from the LDS image (phase A put it
  // there; nobody else writes these rows) instead of being held across the flux phase, and the own G row is read back like the
  // others.  The same values, the same arithmetic.
  constexpr bool LEAN = (N == 4 && MODE == 0) || N >= 5;   // (N >= 5: every stage -- the later stages spill otherwise)
  double Gown[N][4], base[4][N];
#pragma unroll
  for (int aa = 0; aa < N; ++aa) {
    double Fx[4], Wa[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      Wa[c] = LEAN ? Us[(c * NS + aa + N * B) * S + lane] : Wrow[aa][c];
      if constexpr (!late_loads<N, MODE, 0>()) base[c][aa] = Wa[c];
    }
    flux_xy(Wa, Fx, Gown[aa]);
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
      R[MY][aa] += a.gravity * (-1.0 * Wa[RHO]) * jxw;
      R[EN][aa] += a.gravity * (-1.0 * Wa[MY]) * jxw;
    }
  }
  __syncthreads();
  if constexpr (MF && N == 4) {
    eta_derivative_mfma<N, true>(Us, S, lane, B);
    __syncthreads();
#pragma unroll
    for (int aa = 0; aa < N; ++aa) {
      const double wah = CB<N>::t.w[aa] * h;
#pragma unroll
      for (int c = 0; c < 4; ++c) R[c][aa] += wah * Us[(B * NS + c * N + aa) * S + lane];
    }
  } else {
  int ln = lane;
#pragma unroll
  for (int aa = 0; aa < N; ++aa) {
    const double wah = CB<N>::t.w[aa] * h;
    // LEAN: the reads of this node wait for the sums of the node before -- 4 N values of G in flight, not all 4 N^2 of them at
    // once (the compiler otherwise hoists every read to the top and spills to make room)
    if constexpr (LEAN) {
      if (aa > 0) asm volatile("" : "+v"(ln) : "v"(R[0][aa - 1]), "v"(R[1][aa - 1]), "v"(R[2][aa - 1]), "v"(R[3][aa - 1]));
    }
#pragma unroll
    for (int q = 0; q < N; ++q) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double gy = (q == B && !LEAN) ? Gown[aa][c] : Us[(c * NS + aa + N * q) * S + ln];
        R[c][aa] += gy * (wah * CB<N>::t.DW[q][B]);
      }
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
          const double fq = Fh[row_off<N>(c * N, qq, HS) + k] * jxw;
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
          for (int c = 0; c < 4; ++c) R[c][q] += Fh[row_off<N>(c * N, qq, HS) + k] * jxw;
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
      double ust[4][N];
      constexpr bool LATE = late_loads<N, MODE, 0>();
      const double *cp = a.Ucur + (size_t)shard * 4 * NS * 64 + (size_t)(N * B) * 64 + lane;
      const double *op = a.Uold + (size_t)shard * 4 * NS * 64 + (size_t)(N * B) * 64 + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < N; ++m) {
          const double ww = CB<N>::t.w[m] * CB<N>::t.w[B];
          const double invM = rh2 * (CB<N>::t.iw[m] * CB<N>::t.iw[B]);
          part[4] += R[c][m] * R[c][m];
          double u = LATE ? cp[(c * NS + m) * 64] : base[c][m];
          u += dt * R[c][m] * invM;
          if constexpr (MODE == 1) u = (1.0 - a.ark) * u + a.ark * (LATE ? __builtin_nontemporal_load(&op[(c * NS + m) * 64]) : uold[c][m]);
          ust[c][m] = u;
          if constexpr (POS) unew[c][m] = u;   // kept for the positivity step of the caller (which stores again if it scales)
          part[c] = __builtin_fma(ww, u, part[c]);   // (spelled out: cell_average_rows forms the same sums from the DoFs, bit for bit)
        }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < N; ++m) stream_store<STREAM>(&np[(c * NS + m + N * B) * 64], ust[c][m]);
    }
  } else if constexpr (POS) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < N; ++m) {
        if constexpr (late_loads<N, MODE, 0>()) unew[c][m] = a.Ucur[(size_t)shard * 4 * NS * 64 + (size_t)(c * NS + m + N * B) * 64 + lane];
        else unew[c][m] = base[c][m];
      }
  }
  // partial cell averages / residual of this row -> LDS (red aliases Fh, see the caller's barriers)
  if constexpr (MODE != 2) {
    __syncthreads();  // every wave is done reading Fh (and the G rows of Us)
#pragma unroll
    for (int c = 0; c < 5; ++c) red[(B * 5 + c) * 64 + lane] = part[c];
    if constexpr (POS == 1) positivity_row_bounds<N, B>(Us, lane, unew);   // the LDS image is free now
    if constexpr (POS == 2 && N >= 3) positivity_row_bounds<N, B>(Us, lane, unew);   // (the marks bound the TVB slopes with the box as well)
    if constexpr (POS == 2 && N >= 3) {   // x part of "dx * gradient of the cell average" (src/limiter.cc:283-289): l_m(1) - l_m(0) is
                                          // antisymmetric in m, pairing the nodes makes the slope of a constant state exactly zero
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double g = 0.0;
#pragma unroll
        for (int m = 0; m < N / 2; ++m) g += (CB<N>::t.L1[m] - CB<N>::t.L0[m]) * (unew[c][m] - unew[c][N - 1 - m]);
        red[(5 * N + c * N + B) * 64 + lane] = CB<N>::t.w[B] * g;
      }
    }
    if constexpr (POS == 2 && N == 2) {   // k = 1: the difference along the row is all the marks want (stage_kernel, last wave)
#pragma unroll
      for (int c = 0; c < 4; ++c) red[(5 * N + c * N + B) * 64 + lane] = unew[c][1] - unew[c][0];
    }
  }
}

// Edge of local face f of a bilinear cell with the vertices v = (x0, y0, .., x3, y3) in deal.II's lexicographic order: t runs
// along the increasing free coordinate; faces 1 (xi = 1) and 2 (eta = 0) have the cell on the left of t, faces 0 and 3 on the
// right (counter-clockwise cells) -- the outward normal is (ty, -tx) / |t| on faces 1 and 2 and (-ty, tx) / |t| on 0 and 3
// (plan.cc: the same edge, the same two vertices whichever of the two cells looks at it: |t|^2 = tx tx + ty ty comes out
// bit-identical on both sides, t itself with the opposite sign).
__device__ __forceinline__ void face_edge(const double (&v)[8], int f, double &tx, double &ty) {
  const int a = f == 1 ? 1 : (f == 3 ? 2 : 0), b = f == 0 ? 2 : (f == 2 ? 1 : 3);
  tx = v[2 * b] - v[2 * a];
  ty = v[2 * b + 1] - v[2 * a + 1];
}

// phase C on bilinear (Q1-mapped) cells (the reference gets all of this from
// FEValues with MappingQ1): J = [x_xi x_eta; y_xi y_eta] varies inside the cell,
//   int F.grad(phi) = sum_q w_q [ d(phi)/d(xi) (y_eta F - x_eta G) + d(phi)/d(eta) (-y_xi F + x_xi G) ],
// lumped mass M_j = w_j det J_j (src/claw.cc:223-227), face JxW = w_q |edge|.
template <int N, int B, int MODE, int POS, int STREAM, int MF = 0>
__device__ __forceinline__ void row_update_q1(const StageArgs &a, double *Us, const int S, const double *Fh,
                                              double *red, int shard, int lane, bool active,
                                              const double (&vx)[8], const uint16_t (&cref)[4],
                                              const double (&uold)[4][N], const double (&Wrow)[N][4], double (&unew)[4][N],
                                              const double dt) {
  const int HS = a.halo_stride;   // row stride of the trace / flux table (Fh)
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
  // degree 5 (LEANQ, round 6): as on squares the state of the own row comes back from the LDS image instead of being held across the
  // flux phase, and the own eta part is read back like the others (the kernels spilled 188-436 bytes per lane)
  constexpr bool LEANQ = late_q1<N, 1>();   // (degree 3, C5, the same way: 241 registers without scratch instead of 256 + 28 B, and 1.7 % slower)
  double Hown[N][4];
  {
    const double xxi = ax + CB<N>::t.x[B] * bx, yxi = ay + CB<N>::t.x[B] * by;
#pragma unroll
    for (int aa = 0; aa < N; ++aa) {
      const double xeta = cx + CB<N>::t.x[aa] * dx, yeta = cy + CB<N>::t.x[aa] * dy;
      double Fx[4], Gy[4], Wa[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) Wa[c] = LEANQ ? Us[(c * NS + aa + N * B) * S + lane] : Wrow[aa][c];
      flux_xy(Wa, Fx, Gy);
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
        R[MY][aa] += a.gravity * (-1.0 * Wa[RHO]) * jxw;
        R[EN][aa] += a.gravity * (-1.0 * Wa[MY]) * jxw;
      }
    }
  }
  __syncthreads();
  if constexpr (MF && N == 4) {   // (the metric factors sit in the rows already: the plain derivative matrix)
    eta_derivative_mfma<N, false>(Us, S, lane, B);
    __syncthreads();
#pragma unroll
    for (int aa = 0; aa < N; ++aa)
#pragma unroll
      for (int c = 0; c < 4; ++c) R[c][aa] += Us[(B * NS + c * N + aa) * S + lane];
  } else {
#pragma unroll
  for (int aa = 0; aa < N; ++aa)
#pragma unroll
    for (int q = 0; q < N; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double hy = (q == B && !LEANQ) ? Hown[aa][c] : Us[(c * NS + aa + N * q) * S + lane];
        R[c][aa] += hy * CB<N>::t.D[q][B];
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
      double etx, ety;   // face JxW = w_q |edge|, the edge from the cell's own vertices
      face_edge(vx, f, etx, ety);
      const double len = fsqrt(etx * etx + ety * ety);
      if (f < 2) {
        const int qq = flip ? N - 1 - B : B;
        const double jxw = sgn * CB<N>::t.w[B] * len;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double fq = Fh[row_off<N>(c * N, qq, HS) + k] * jxw;
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
          for (int c = 0; c < 4; ++c) R[c][q] += Fh[row_off<N>(c * N, qq, HS) + k] * jxw;
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
      constexpr bool LATE = late_q1<N, 1>();
      const double *cp = a.Ucur + (size_t)shard * 4 * NS * 64 + lane, *op = a.Uold + (size_t)shard * 4 * NS * 64 + lane;
#pragma unroll
      for (int m = 0; m < N; ++m) {
        const double det = xxi * (cy + CB<N>::t.x[m] * dy) - (cx + CB<N>::t.x[m] * dx) * yxi;
        const double wd = CB<N>::t.w[m] * CB<N>::t.w[B] * det;   // JxW of node (m, B) = its lumped mass
        const double invM = frcp(wd);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int d = c * NS + m + N * B;
          part[4] += R[c][m] * R[c][m];
          double u = LATE ? cp[d * 64] : Wrow[m][c];
          u += dt * R[c][m] * invM;
          if constexpr (MODE == 1) u = (1.0 - a.ark) * u + a.ark * (LATE ? __builtin_nontemporal_load(&op[d * 64]) : uold[c][m]);
          stream_store<STREAM>(&np[d * 64], u);
          unew[c][m] = u;   // kept: the positivity step and the time step of the new solution (dtq) work on it
          part[c] += wd * u;
        }
      }
    }
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < N; ++m) {
        if constexpr (late_q1<N, 1>()) unew[c][m] = a.Ucur[(size_t)shard * 4 * NS * 64 + (size_t)(c * NS + m + N * B) * 64 + lane];
        else unew[c][m] = Wrow[m][c];
      }
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
// neighbouring faces at the same q -> same LDS rows, different slots.  Reads LDS only.
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
constexpr int kGhostTrace = 1 << 30;   // halo entry: bits 0-27 hold a ghost-trace number instead of a cell slot
__device__ __forceinline__ int pface_slot(uint32_t w) { return w & 0x1FF; }
__device__ __forceinline__ int pface_face(uint32_t w) { return (w >> 9) & 3; }
__device__ __forceinline__ bool pface_bnd(uint32_t w) { return (w >> 11) & 1; }
__device__ __forceinline__ bool pface_flip(uint32_t w) { return (w >> 12) & 1; }
__device__ __forceinline__ int pface_other_face(uint32_t w) { return (w >> 13) & 3; }
__device__ __forceinline__ int pface_other_slot(uint32_t w) { return (w >> 15) & 0x1FF; }
__device__ __forceinline__ int pface_bnd_local(uint32_t w) { return (w >> 13) & 0x3FF; }

// The table Th[4N][HS] starts out with the traces of the halo entries in its first columns (entry e in column e); the flux of a
// face goes into the column of the face -- the face of halo entry e into column e, at the point index of the HALO cell, where
// it replaces the trace it was computed from (read and written by the same thread), the k-th other face into column
// halo_cols + k.  The face records come in registers (frr: the records of this thread's first three passes).
// face point p of a shard with nf faces -> face k (and point q): for N = 3 the N points of a face in consecutive lanes (measured
// 1 % faster there), otherwise p = q nf + k
template <int N>
__device__ __forceinline__ int face_of_point(int p, int nf, int &q) {
  if constexpr (N == 3) {
    const int k = p / N;
    q = p - k * N;
    return k;
  }
  q = p >= nf ? 1 : 0;
  if constexpr (N > 3) q += (p >= 2 * nf ? 1 : 0) + (p >= 3 * nf ? 1 : 0);
  if constexpr (N > 4) q += p >= 4 * nf ? 1 : 0;
  if constexpr (N > 5) q += p >= 5 * nf ? 1 : 0;
  return p - q * nf;
}
template <int N>
__device__ __forceinline__ int face_of_point(int p, int nf) {
  int q;
  return face_of_point<N>(p, nf, q);
}
template <int N, int FLUX, int GEO>
__device__ __forceinline__ void flux_phase(const StageArgs &a, const double *Us, double *Th, const double *Av,
                                           const uint32_t (&frr)[3], const uint32_t *fp, const double *Bv, const int *Bk,
                                           const double *Vx, const int HS, const int nf, const int nh, const int tid) {
  constexpr int NS = N * N, NDOF = 4 * NS, NT = 64 * N, S = 65;
  const int nfp = nf * N;
  int it = 0;
  for (int p = tid; p < nfp; p += NT, ++it) {
    // neighbouring lanes take neighbouring faces at the same point q (the plan sorts the faces by kind, local face, slot): the
    // lanes of a wavefront read the same rows of the LDS image at different cells (measured against "the N points of a face in
    // consecutive lanes": Q3 KFVS +3-4 %, Q1 LxF +1 %, Q2 HLLC -1 %)
    int q;
    const int k = face_of_point<N>(p, nf, q);
    const uint32_t r = it == 0 ? frr[0] : (it == 1 ? frr[1] : (it == 2 ? frr[2] : fp[k]));
    const int col = k < nh ? k : k - nh + a.halo_cols;
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
        for (int c = 0; c < 4; ++c) W[c] = Th[row_off<N>(c * N, qq, HS) + slot - 64];
        if constexpr (FLUX == DFLO_FLUX_LXF) {
#pragma unroll
          for (int c = 0; c < 3; ++c) A[c] = Av[c * a.halo_cols + slot - 64];
        }
      }
    };
    trace(slotL, fL, q, Wp, Ap);
    double nx, ny;  // outward unit normal of the integrating cell
    if constexpr (GEO == 0) {
      nx = fL == 0 ? -1.0 : (fL == 1 ? 1.0 : 0.0);
      ny = fL == 2 ? -1.0 : (fL == 3 ? 1.0 : 0.0);
    } else {
      // Vx[4 faces][2][64]: the outward normals of the own cells, formed in phase A from their vertices -- the integrating
      // cell's own normal, or the other cell's with the sign turned (the same edge, the same two vertices: the same bits).  A
      // per-face table of normals and lengths in memory, 4 % of the kernel's traffic and 3 HS doubles of LDS, is gone.
      const bool ownL = slotL < 64;
      const int sv = ownL ? slotL : pface_other_slot(r), fv = ownL ? fL : fR;
      const double sx = Vx[(2 * fv) * 64 + sv], sy = Vx[(2 * fv + 1) * 64 + sv];
      nx = ownL ? sx : -sx;
      ny = ownL ? sy : -sy;
    }
    int qs = q;   // the point index the flux is filed under
    if (!bnd) {
      const int slotR = pface_other_slot(r), qR = flip ? N - 1 - q : q;
      if (slotR >= 64) qs = qR;
      trace(slotR, fR, qR, Wm, Am);
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
    for (int c = 0; c < 4; ++c) Th[row_off<N>(c * N, qs, HS) + col] = F[c];
  }
}

// POS 2, what a box around the cell's nodal values settles for the limiter pass (see the marks at the end of stage_kernel):
// `settled` -- the positivity limiter has nothing to do (the arithmetic of positivity_box_settled, kernels_common.hpp); `open` --
// TVB cannot be ruled out from the box alone.  dsum bounds the sum over the components of (largest - smallest nodal value).
// "dx * gradient of the cell average" (src/limiter.cc:283-289) is a weighted sum of nodal values whose weights sum to zero: no
// component of it exceeds K (hi - lo), K = 1/2 sum_m |l_m(1) - l_m(0)|, in either direction, and the characteristic slopes
// together no more than ||L||_1 times the sum of that over the components, ||L||_1 bounded over the box (the cell average, where
// the pass forms L, is a convex combination of the nodal values).  Below `M dx^2` minmod hands back its first argument
// (src/limiter.cc:15-30), and a cell whose summed |slopes| stay below the rewrite threshold of src/limiter.cc:347 is left as it is
// whatever its neighbours hold (the pass's own early-out).
template <int N>
__device__ __forceinline__ void limiter_marks_from_box(const double (&lo)[4], const double (&hi)[4], bool fin, double dsum, double sn,
                                                       double tvb_M, int tvb_char, double h, bool &settled, bool &open) {
  const double rho_lo = lo[RHO] - (hi[RHO] - lo[RHO]) * sn, e_lo = lo[EN] - (hi[EN] - lo[EN]) * sn;
  const double dmx = (hi[MX] - lo[MX]) * sn, dmy = (hi[MY] - lo[MY]) * sn;
  const double mxa = fmax(fabs(lo[MX] - dmx), fabs(hi[MX] + dmx)), mya = fmax(fabs(lo[MY] - dmy), fabs(hi[MY] + dmy));
  const double ri_lo = frcp(rho_lo);
  const double p_lo = kG1 * (e_lo - 0.5 * (mxa * mxa + mya * mya) * ri_lo);
  settled = fin && rho_lo >= 1.0e-10 + 1.0e-8 * hi[RHO] && p_lo >= 1.0e-10 + 1.0e-8 * fabs(hi[EN]);
  open = false;
  if (tvb_M >= 0.0) {
    constexpr double K = []() constexpr {
      double t = 0.0;
      for (int m = 0; m < N; ++m) { const double g = CB<N>::t.L1[m] - CB<N>::t.L0[m]; t += g < 0 ? -g : g; }
      return 0.5 * t;
    }();
    double kappa = 1.0;
    if (tvb_char) {
      // columns of |Lx|, |Ly| (physics.hpp: to_char) with q >= |u| + |v|, phi2 <= 0.2 q^2, 1/c <= (1 + 1/c^2) / 2
      const double q = (mxa + mya) * ri_lo, ic2 = hi[RHO] * frcp(kGamma * p_lo);
      kappa = 1.0 + q + ic2 * (0.4 * q * q + 0.8 * q + 0.8) + 0.5 * (q + 1.0) * (1.0 + ic2);
    }
    const double Sb = 1.0001 * K * kappa * dsum;
    // (a box that is not settled has no bound on L: NaN or a negative pressure leave `open` true)
    open = !(settled && (0.5 * Sb <= 0.9998e-10 || Sb < tvb_M * h * h * (1.0 - 1.0e-9)));
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
//   STREAM 1: nothing reads the new state before the next stage kernel (no limiter pass over all cells follows): it is stored past
//         the caches (measured: C2 +2 %, C4 +3 %; with the Q1 limiter pass behind it C3 -2 %, hence a variant and not a rule).
//         A compile-time switch: behind a run-time branch the compiler merges the two store sequences and drops the hint.
//   AF 1 (LxF on squares, no limiter, no ghost cells): the (u, v, c) of the cell averages that the LxF flux wants
//         (src/equation.h:357-359) are formed from the DoFs the kernel loads anyway -- the own cells' and, in the halo gather, the
//         neighbours' (a thread takes one component of one halo cell: its N^2 nodes give the N traces and the average) -- in the
//         order in which the stage epilogue sums them (cell_average_rows): the same bits as the stored averages, which are then
//         neither read nor, in intermediate stages, written (15-20 % of the Q1 LxF kernel's memory traffic)
// The tail of a stage kernel that delivers (StageArgs::dl_*): the workgroup has stored its shard's new state; behind a barrier its
// threads read it back (same compute unit: the stores have completed and the L1 holds no older copy of rows nobody has read in this
// launch), form the traces on the shard's cut faces exactly as face_trace_kernel would (cell_face_trace: the same bits), and store
// them into the neighbours' tables.  Release at system scope, count, and the last workgroup that delivers publishes the number.
// the stage kernel's share of the exchange (kernels_common.hpp: await_words, deliver_face_traces, deliver_averages)
__device__ __forceinline__ bool await_traces(const StageArgs &a, const int shard) {   // false: a neighbour's traces never came
  if (a.wt_begin[shard + 1] == a.wt_begin[shard]) return true;   // workgroup-uniform: no cut face, no ghost trace
  return await_words(a.wt_flag, a.wt_n, a.wt_seq, a.wt_fail, a.wt_ticks);
}
template <int N>
__device__ __forceinline__ void deliver_traces(const StageArgs &a, const int shard) {
  deliver_face_traces<N>(a.dl_begin, a.dl_rec, a.dl_dst, a.dl_flag, a.dl_nflag, a.dl_total, a.dl_seq, a.dl_done, a.Unew, shard, a.dl_fence);
}
// does the limiter pass find this shard by the list of marked shards?  (not a shard on a cut where the pass takes the exchange
// along: those get a wavefront of their own whatever their marks)
__device__ __forceinline__ bool lists_itself(const StageArgs &a, const int shard, const int sidx) {
  return a.lim_cnt && sidx >= a.lim_list_from && !(a.dla_begin && a.dla_begin[shard + 1] > a.dla_begin[shard]);
}

template <int N, int FLUX, int MODE, int GEO, int POS, int STREAM, int AF = 0, int MF = 0>
__global__ __launch_bounds__(64 * N, (kLateQ4 && N == 5 && GEO == 0 && MODE != 2) ? 3 : N >= 5 ? 1 : ((N == 4 && GEO == 0 && MODE == 0) ? kQ3FirstStageWaves : (((GEO == 1 && N != 3) || N == 4) ? 2 : (N == 3 && GEO == 0 ? kQ2Waves : 3)))) void stage_kernel(const StageArgs a) {
  constexpr int NS = N * N, NDOF = 4 * NS, NT = 64 * N;
  constexpr int ROWS = NDOF + (FLUX == DFLO_FLUX_LXF ? 3 : 0);   // LxF: (u, v, c) of the cell average ride along
  constexpr int TROWS = 4 * N;                                   // trace / flux table: (component, point) rows
  constexpr int S = 65;                                          // own-cell row stride: 1 mod 32 doubles
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int sidx = shard_of_block(blockIdx.x, a.n_list, a.sweep_rev);
  if (sidx < 0) return;
  const int shard = a.shard_list ? a.shard_list[sidx] : sidx;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int row = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HS = a.halo_stride;
  double *Us = lds;                                   // [ROWS][S] DoFs (and averages) of the own cells
  double *Th = Us + ROWS * S;                         // [TROWS][HS] columns 0 .. halo_cols-1: traces of the halo cells on the shared
                                                      // faces, later the fluxes of those faces; from halo_cols: fluxes of the other faces
  double *Av = Th + TROWS * HS;                       // LxF: [3][halo_cols] (u, v, c) of the halo cells' averages
  double *Bv = Av + (FLUX == DFLO_FLUX_LXF ? 3 * a.halo_cols : 0);   // [max_bnd][N][4] boundary values of the shard
  int *Bk = (int *)(Bv + a.max_bnd * 4 * N);          // [max_bnd] boundary kinds
  double *Vx = (double *)(Bk + ((a.max_bnd + 1) & ~1)); // GEO 1: [4][2][64] outward unit normals of the own cells' faces

  // ---- all loads of the shard, issued back to back; the halo entries first (the halo values depend on them)
  // halo entries: thread t works on entry (t & 31) + 32 b of every block b of 32 entries (8x8 lattice shards have one
  // block, unstructured shards two or three): load them all now, the gathers below then depend on nothing else
  constexpr int HB = 3;
  int hentb[HB];
  if constexpr (AF) {   // a thread takes (entry, component) = (t >> 2, t & 3) of every block of 16 N entries
    hentb[0] = a.halo_pad[(size_t)shard * a.halo_pitch + min(tid >> 2, a.halo_pitch - 1)];
  } else {
#pragma unroll
  for (int b = 0; b < HB; ++b)   // (lattice shards have at most 32 entries: one load, the condition is uniform)
    hentb[b] = (b == 0 || a.halo_pitch > 32 * b) ? a.halo_pad[(size_t)shard * a.halo_pitch + min((tid & 31) + 32 * b, a.halo_pitch - 1)] : 0;
  }
  const int4 hdr = a.shard_hdr[shard];                // {cells, faces, halo entries, boundary faces}
  const int nf = hdr.y, nh = hdr.z, nbnd = hdr.w;
  const bool active = lane < (hdr.x & 0xFF);
  const int pat = hdr.x >> 8;   // index pattern of the shard: its face records and face references
  double urow[4][N];                                  // node row `row` of the own cells
  {
    const double *up = a.Ucur + (size_t)shard * NDOF * 64 + (size_t)(N * row) * 64 + lane;   // one base, constant offsets
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < N; ++m) {
        urow[c][m] = up[(c * NS + m) * 64];
      }
  }
  double uavg[4];
  if constexpr (FLUX == DFLO_FLUX_LXF && !AF) {
    if (row == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) uavg[c] = a.avg_cur[((size_t)shard * 4 + c) * 64 + lane];
    }
  }
  // face records of this thread's passes over the face points (see flux_phase for the numbering)
  const uint32_t *fp = a.faces_pad + (size_t)pat * a.face_pitch;
  uint32_t frr[3];
#pragma unroll
  for (int it = 0; it < 3; ++it) frr[it] = fp[min(face_of_point<N>(tid + it * NT, nf), a.face_pitch - 1)];
  uint16_t cref[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) cref[f] = a.cell_face[((size_t)pat * 4 + f) * 64 + lane];
  double h = 0.0, vx[8], hq = 0.0;
  if constexpr (GEO == 0) {
    h = a.uniform_h ? a.h_uniform : a.cell_h[(size_t)shard * 64 + lane];
  } else {
    if (a.dtq && row == N - 1) hq = a.cell_h[(size_t)shard * 64 + lane];   // diameter / sqrt(2), for the time step
#pragma unroll
    for (int k = 0; k < 8; ++k) vx[k] = a.cell_vert[(size_t)k * a.n_slots + (size_t)shard * 64 + lane];
  }
  // the time step of the update: fetched here with everything else (it used to be read in the middle of phase C, one more
  // trip to memory on every wave's critical path)
  double dt_step = 0.0;
  if constexpr (MODE != 2) dt_step = a.dt_cell ? a.dt_cell[(size_t)shard * 64 + lane] : (a.dt_host >= 0.0 ? a.dt_host : step_dt(a.dts, a.dt_dev));
  double uold[4][N];
  if constexpr (MODE == 1 && !late_loads<N, MODE, GEO>() && !late_q1<N, GEO>()) {
    const double *op = a.Uold + (size_t)shard * NDOF * 64 + (size_t)(N * row) * 64 + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < N; ++m) {
        uold[c][m] = __builtin_nontemporal_load(&op[(c * NS + m) * 64]);   // u(n) is read once per stage, by this thread only
      }
  }

  if (a.wt_n && !await_traces(a, shard)) return;   // (timed out: the failure word is up; nothing is computed from stale traces, nothing delivered)
  // ---- phase A: own rows -> LDS; halo: only the trace on the shared face is kept.
  //      halo item i -> (entry s = i % nh, q = (i / nh) % N, comp = i / (nh N)); an entry is
  //      (internal cell slot | local face << 28) of a face neighbour outside the shard
  //      A thread takes, of its entry in block b, the two (component, point) rows r = g and g + 2N (g = t >> 5):
  //      2N independent loads per block.
  if constexpr (AF) {
    for (int it = 0, item = tid; item < 4 * nh; item += NT, ++it) {
      const int sl = item >> 2, c = item & 3;
      const int e = it == 0 ? hentb[0] : a.halo_pad[(size_t)shard * a.halo_pitch + sl];
      const int ic = e & 0x0FFFFFFF, f = (e >> 28) & 3;
      const double *hp = a.Ucur + ((size_t)(ic >> 6) * NDOF + c * NS) * 64 + (ic & 63);
      double u[NS];
#pragma unroll
      for (int j = 0; j < NS; ++j) u[j] = hp[j * 64];
      // the lines through the face points, from the face inwards (selects, not indexed registers: the face differs from lane to lane)
      const bool yf = f >= 2, rev = f & 1;
#pragma unroll
      for (int q = 0; q < N; ++q) {
        double val[N];
#pragma unroll
        for (int m = 0; m < N; ++m) {
          const double fwd = yf ? u[q + N * m] : u[m + N * q], bwd = yf ? u[q + N * (N - 1 - m)] : u[(N - 1 - m) + N * q];
          val[m] = rev ? bwd : fwd;
        }
        Th[row_off<N>(0, c * N + q, HS) + sl] = trace_from_line<N>(val);
      }
      // the four components of an entry sit in the four lanes of a quad: everybody fetches the other three averages
      const double av = cell_average_rows<N>(u);
      double A[4], uvc[3];
      A[0] = dpp_f64<0x00, 0xf>(0.0, av);   // quad_perm: [0,0,0,0]
      A[1] = dpp_f64<0x55, 0xf>(0.0, av);   // [1,1,1,1]
      A[2] = dpp_f64<0xAA, 0xf>(0.0, av);   // [2,2,2,2]
      A[3] = dpp_f64<0xFF, 0xf>(0.0, av);   // [3,3,3,3]
      wave_speed_uvc(A, uvc);
      if (c < 3) Av[c * a.halo_cols + sl] = c == 0 ? uvc[0] : (c == 1 ? uvc[1] : uvc[2]);
    }
  } else {
    const int g = tid >> 5, l32 = tid & 31;
    for (int b = 0; b * 32 < nh; ++b) {
      const int sl = l32 + 32 * b;
      if (sl >= nh) continue;
      const int e = b == 0 ? hentb[0] : (b == 1 ? hentb[1] : (b == 2 ? hentb[2] : a.halo_pad[(size_t)shard * a.halo_pitch + sl]));
      const int ic = e & 0x0FFFFFFF, f = (e >> 28) & 3;
      if (e & kGhostTrace) {   // a ghost cell (multi-device): its owner has sent the trace, ic is the trace number
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int r = g + 2 * N * j;   // = c * N + q
          Th[row_off<N>(0, r, HS) + sl] = a.Tg[(size_t)ic * 4 * N + r];
        }
        continue;
      }
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
        Th[row_off<N>(c * N, q, HS) + sl] = trace_from_line<N>(val[j]);
      }
    }
  }
  if constexpr (FLUX == DFLO_FLUX_LXF && !AF) {  // lambda of the LxF flux comes from the cell averages (src/equation.h:357-359):
                                          // keep (u, v, c) of each average instead of the four components
    for (int sl = tid; sl < nh; sl += NT) {
      const int blk = sl >> 5;   // sl = tid + k NT: entry (tid & 31) + 32 blk is this thread's own preloaded one
      const int e = blk == 0 ? hentb[0] : (blk == 1 ? hentb[1] : (blk == 2 ? hentb[2] : a.halo_pad[(size_t)shard * a.halo_pitch + sl]));
      const int ic = (e & kGhostTrace) ? a.gt_slot[e & 0x0FFFFFFF] : (e & 0x0FFFFFFF);
      double A[4], uvc[3];
#pragma unroll
      for (int c = 0; c < 4; ++c) A[c] = a.avg_cur[((size_t)(ic >> 6) * 4 + c) * 64 + (ic & 63)];
      wave_speed_uvc(A, uvc);
#pragma unroll
      for (int c = 0; c < 3; ++c) Av[c * a.halo_cols + sl] = uvc[c];
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int m = 0; m < N; ++m) Us[(c * NS + m + N * row) * S + lane] = urow[c][m];
  if constexpr (FLUX == DFLO_FLUX_LXF && !AF) {
    if (row == 0) {
      double uvc[3];
      wave_speed_uvc(uavg, uvc);
#pragma unroll
      for (int c = 0; c < 3; ++c) Us[(NDOF + c) * S + lane] = uvc[c];
    }
  }
  if constexpr (AF) {   // this row's share of the own cells' averages: into free columns of the flux table (the flux phase writes there later)
    const double wr = CB<N>::t.w[row];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double pr = 0.0;
#pragma unroll
      for (int m = 0; m < N; ++m) pr = __builtin_fma(CB<N>::t.w[m] * wr, urow[c][m], pr);
      Th[(row * 4 + c) * HS + a.halo_cols + lane] = pr;
    }
  }
  if constexpr (GEO == 1) {
    // outward unit normals of the own cells' faces (every wave holds the vertices of its lanes' cells: wave r takes the faces
    // r, r + N, ..).  IEEE square root and quotients, as plan.cc / the reference's FEFaceValues form them: a flux whose branch
    // hangs on the sign of a vanishing normal velocity (KFVS: the jump of the reference's ERF polynomial at 0) must see the
    // normal the host would have computed.
    for (int f = row; f < 4; f += N) {
      double tx, ty;
      face_edge(vx, f, tx, ty);
      const double len = sqrt(tx * tx + ty * ty);
      const bool left = f == 1 || f == 2;
      Vx[(2 * f) * 64 + lane] = (left ? ty : -ty) / len;
      Vx[(2 * f + 1) * 64 + lane] = (left ? -tx : tx) / len;
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
  __syncthreads();
  if constexpr (AF) {   // the rows' shares meet: (u, v, c) of the own cells' averages, the rows added in the epilogue's order
    if (row == 0) {
      double A[4], uvc[3];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double v = 0;
#pragma unroll
        for (int b = 0; b < N; ++b) v += Th[(b * 4 + c) * HS + a.halo_cols + lane];
        A[c] = v;
      }
      wave_speed_uvc(A, uvc);
#pragma unroll
      for (int c = 0; c < 3; ++c) Us[(NDOF + c) * S + lane] = uvc[c];
    }
    __syncthreads();
  }

  // ---- phase B
  flux_phase<N, FLUX, GEO>(a, Us, Th, Av, frr, fp, Bv, Bk, Vx, HS, nf, nh, tid);
  __syncthreads();

  // ---- phase C: volume + lifting + RK update of node row `row`
  double *Fh = Th;   // the fluxes, as the row updates read them
  double *red = Th;  // reused after the barrier inside row_update
  double wrow[N][4];
  double unew[4][N];   // POS: the updated row, held back until the positivity step below
#pragma unroll
  for (int m = 0; m < N; ++m)
#pragma unroll
    for (int c = 0; c < 4; ++c) wrow[m][c] = urow[c][m];
#define DFLO_ROW(Bq)                                                                                     \
  do {                                                                                                   \
    if constexpr (GEO == 0) row_update<N, Bq, MODE, POS, STREAM, MF>(a, Us, S, Fh, red, shard, lane, active, h, cref, uold, wrow, unew, dt_step); \
    else row_update_q1<N, Bq, MODE, POS, STREAM, MF>(a, Us, S, Fh, red, shard, lane, active, vx, cref, uold, wrow, unew, dt_step); \
  } while (0)
  if constexpr (N == 2) {
    if (row == 0) DFLO_ROW(0); else DFLO_ROW(1);
  } else if constexpr (N == 3) {
    if (row == 0) DFLO_ROW(0); else if (row == 1) DFLO_ROW(1); else DFLO_ROW(2);
  } else if constexpr (N == 4) {
    if (row == 0) DFLO_ROW(0); else if (row == 1) DFLO_ROW(1); else if (row == 2) DFLO_ROW(2); else DFLO_ROW(3);
  } else if constexpr (N == 5) {
    if (row == 0) DFLO_ROW(0); else if (row == 1) DFLO_ROW(1); else if (row == 2) DFLO_ROW(2); else if (row == 3) DFLO_ROW(3); else DFLO_ROW(4);
  } else {
    if (row == 0) DFLO_ROW(0); else if (row == 1) DFLO_ROW(1); else if (row == 2) DFLO_ROW(2); else if (row == 3) DFLO_ROW(3);
    else if (row == 4) DFLO_ROW(4); else DFLO_ROW(5);
  }
#undef DFLO_ROW
  if constexpr (MODE == 2) return;
  __syncthreads();
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
      if (!settled && row == 0) {       // statistics (rare path): cells that go through the limiter proper
        const unsigned long long mk = __ballot(!ok && active);
        if (lane == 0) atomicAdd(a.pos_stats, (unsigned long long)__popcll(mk));
      }
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
    if (bad && active && row == 0) raise_flag(a.flags, 0, a.step_ctr);
    double *pm = red + 5 * N * 64;   // [3][N][64] minima of the rows: density, theta2 (speculative), theta2 (after theta1)
    // theta2 of this wave's points (:138-178) for the current unew / Us
    auto pressure_theta = [&](bool &fail) {
      double th = 1.0;
      for_gll_points(a.kb.Ng, [&](auto kind, int g) {
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
          double W[4];
#pragma unroll
          for (int c = 0; c < 4; ++c)
            W[c] = gll_point<N, decltype(kind)::value>(a.kb, g, [&](int m) { return dir == 0 ? unew[c][m] : Us[(c * NS2 + row + N * m) * S + lane]; });
          th = smin(th, positivity_theta2(W, A, eps, fail));
        }
      });
      return th;
    };
    // first round: the density minimum and, on the guess theta1 = 1 (true almost everywhere), theta2 as well
    bool fail = false;
    {
      double rmin = 1.0e20;
      for_gll_points(a.kb.Ng, [&](auto kind, int g) {
        const double px = gll_point<N, decltype(kind)::value>(a.kb, g, [&](int m) { return unew[RHO][m]; });
        const double py = gll_point<N, decltype(kind)::value>(a.kb, g, [&](int m) { return Us[(RHO * NS2 + row + N * m) * S + lane]; });
        rmin = smin(smin(rmin, px), py);
      });
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
    else if (fail && active) raise_flag(a.flags, 1, a.step_ctr);
    if (row == 0) {   // statistics: cells the limiter changes
      const unsigned long long mk = __ballot(active && (t1 || theta2 < 1.0));
      if (lane == 0 && mk) atomicAdd(a.pos_stats + 1, (unsigned long long)__popcll(mk));
    }
    if (active && (t1 || theta2 < 1.0)) {   // rare: the rows stored by the update are replaced
      double *np = a.Unew + (size_t)shard * 4 * NS2 * 64 + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < N; ++m) {
          unew[c][m] = theta2 < 1.0 ? positivity_blend(theta2, unew[c][m], A[c]) : unew[c][m];
          np[(c * NS2 + m + N * row) * 64] = unew[c][m];
        }
    }
    }   // !settled
  }
  if constexpr (GEO == 1 && MODE != 2) {
    // compute_time_step_q (src/claw.cc:520-557) of the new solution, when no limiter pass follows that could form it: max of
    // |v| + c over the 4 x 4 points of QIterated(QTrapez, 3).  Sum-factorised across the waves: wave b interpolates its row in
    // xi (registers), the rows meet in the LDS image (free by now), wave p finishes the point columns p, p + N, .. in eta --
    // the arithmetic of dt_q_cell, the separate pass over the whole state (dt_q_kernel) is not launched.
    if (a.dtq) {
      // the LDS image has 4 N^2 rows, the xi-interpolated rows are 16 N: one pass for N = 4, two (two point columns each) for
      // N = 2, 3, four for N = 1
      constexpr int PP = N >= 4 ? 4 : (N >= 2 ? 2 : 1), PASSES = kTrap / PP;
      double maxeig = 0.0;
#pragma unroll
      for (int pass = 0; pass < PASSES; ++pass) {
        __syncthreads();   // the row bounds / the limiter's copy of the rows / the previous pass have been read by every wave
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int pl = 0; pl < PP; ++pl) {
            double t = 0;
#pragma unroll
            for (int aa = 0; aa < N; ++aa) t += a.kb.Pt[pass * PP + pl][aa] * unew[c][aa];
            Us[((c * PP + pl) * N + row) * S + lane] = t;
          }
        __syncthreads();
        for (int pl = row; pl < PP; pl += N) {
          double v[4][N];
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int b = 0; b < N; ++b) v[c][b] = Us[((c * PP + pl) * N + b) * S + lane];
#pragma unroll
          for (int pb = 0; pb < kTrap; ++pb) {
            double w[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              double t = 0;
#pragma unroll
              for (int b = 0; b < N; ++b) t += a.kb.Pt[pb][b] * v[c][b];
              w[c] = t;
            }
            maxeig = fmax(maxeig, max_eigenvalue(w));
          }
        }
      }
      red[(8 * N + row) * 64 + lane] = maxeig;
      __syncthreads();
    }
  }
  if constexpr (POS == 2 && GEO == 0 && MODE != 2 && N >= 3) {
    // Which cells can the limiter pass change?  One bit per cell, formed by wave 0 (the last wave has the averages and the
    // reductions to write).  The pass itself is unchanged for the marked cells, so the results are those of the plain pass.
    //  * positivity: the nodal box test.
    //  * TVB, first from the nodal box alone (limiter_marks_from_box): states constant up to rounding, most of a shock tube and
    //    of the double Mach reflection, are settled there in a few dozen instructions.
    //  * Only wavefronts with a cell the box cannot settle form the slopes themselves (row partials in LDS, characteristic
    //    variables when `char_lim` is on), x then y, with a margin on the thresholds so that the pass, which forms the same
    //    slopes once more from the DoFs, can never disagree in the other direction.
    //  (k = 1: wave 0's block below)
    if (row == 0) {
      double lo[4], hi[4];
      bool fin = true;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        lo[c] = Us[((2 * c) * N) * 64 + lane];
        hi[c] = Us[((2 * c + 1) * N) * 64 + lane];
        if (c == RHO) fin = fin && lo[c] == lo[c];
#pragma unroll
        for (int b = 1; b < N; ++b) {
          const double l = Us[((2 * c) * N + b) * 64 + lane];
          if (c == RHO) fin = fin && l == l;
          lo[c] = fmin(lo[c], l);
          hi[c] = fmax(hi[c], Us[((2 * c + 1) * N + b) * 64 + lane]);
        }
      }
      bool settled, open;
      limiter_marks_from_box<N>(lo, hi, fin, (hi[0] - lo[0]) + (hi[1] - lo[1]) + (hi[2] - lo[2]) + (hi[3] - lo[3]), a.kb.pg_neg,
                                a.tvb_M, a.tvb_char, h, settled, open);
      bool need = a.pos_check && !settled;
      if (__any(open && active)) {
        double A[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double v = 0.0;
#pragma unroll
          for (int b = 0; b < N; ++b) v += red[(b * 5 + c) * 64 + lane];
          A[c] = v;
        }
        // margin: relative on the threshold, and absolute against the rounding of the slopes (formed here from row
        // partials, in the pass from the DoFs; both errors are a few ulp of the state)
        const double thr = a.tvb_M * h * h * (1.0 - 1.0e-9) - 1.0e-11 * (fabs(A[0]) + fabs(A[1]) + fabs(A[2]) + fabs(A[3]));
        EigenXY e;
        if (a.tvb_char) e = eigen_at(A);
        bool over = false;     // a slope at or above the threshold (and not zero)
        double sum = 0.0;      // |slopes| of both directions
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
          double D[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            double g = 0.0;
            if (dir == 0) {
#pragma unroll
              for (int b = 0; b < N; ++b) g += red[(5 * N + c * N + b) * 64 + lane];
            } else {
#pragma unroll
              for (int m = 0; m < N / 2; ++m)   // the row sums of the average are w_m * (sum of the row): w_m = w_(N-1-m)
                g += (CB<N>::t.L1[m] - CB<N>::t.L0[m]) * CB<N>::t.iw[m] * (red[(m * 5 + c) * 64 + lane] - red[((N - 1 - m) * 5 + c) * 64 + lane]);
            }
            D[c] = g;
          }
          if (a.tvb_char) to_char(e, dir, D);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            over = over || !(fabs(D[i]) < thr || D[i] == 0.0);
            sum += fabs(D[i]);
          }
        }
        // The second way out (the pass's own early-out, see limiter_kernel): minmod changes a slope by at most its size, so the
        // "change" that decides whether the cell is rewritten (src/limiter.cc:347: > 1e-10, a quarter of the summed |changes| of
        // both directions) is at most a quarter of the sum of the |slopes|.  The slopes here and in the pass are the same sums
        // in another order (a few ulp of the slopes themselves): the margin on the threshold covers that.
        need = need || (over && !(0.25 * sum <= 0.9998e-10));
      }
      const unsigned long long m = __ballot(need && active);
      if (lane == 0 && m) {
        atomicOr(&a.lim_mask[shard], m);
        if (lists_itself(a, shard, sidx)) a.lim_list[atomicAdd(a.lim_cnt, 1)] = make_ulonglong2((unsigned long long)shard, m);
      }
    }
  }
  if constexpr (POS == 2 && GEO == 0 && MODE != 2 && N == 2) {
    if (row == 0) {   // wave 0, beside the last wave's averages and reductions (the averages: the same partials in the same order)
    double avg[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double v = 0;
#pragma unroll
      for (int b = 0; b < N; ++b) v += red[(b * 5 + c) * 64 + lane];
      avg[c] = v;
    }
    // The marks at k = 1, where every instruction of a 500-instruction kernel shows: no row extremes, no slopes -- the four
    // nodal values of a component are (row mean) -+ (half the difference along the row), so they lie within
    // d = |mean_1 - mean_0| + max |difference| of each other and of their average, and the box `average -+ sum of the d` holds
    // every component.  The row means are the partials of the average, the differences were left in LDS
    // by the row updates.  What the box cannot settle is marked (no second look at the slopes: with M = 0 there is nothing
    // between "constant up to rounding" and "limited").
    double sd = 1.0e-15 * (fabs(avg[0]) + fabs(avg[1]) + fabs(avg[2]) + fabs(avg[3]));   // (rounding of the partials)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double dm = red[(1 * 5 + c) * 64 + lane] - red[(0 * 5 + c) * 64 + lane];   // w_b (row sum): half the row mean
      sd += 2.0 * fabs(dm) + fmax(fabs(red[(5 * N + c * N) * 64 + lane]), fabs(red[(5 * N + c * N + 1) * 64 + lane]));
    }
    double lo[4], hi[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { lo[c] = avg[c] - sd; hi[c] = avg[c] + sd; }
    bool settled, open;
    limiter_marks_from_box<N>(lo, hi, sd == sd, sd, a.kb.pg_neg, a.tvb_M, a.tvb_char, h, settled, open);
    const unsigned long long m = __ballot(((a.pos_check && !settled) || open) && active);
    if (lane == 0 && m) {
      atomicOr(&a.lim_mask[shard], m);
      if (lists_itself(a, shard, sidx)) a.lim_list[atomicAdd(a.lim_cnt, 1)] = make_ulonglong2((unsigned long long)shard, m);
    }
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
      for (int c = 0; c < 4; ++c)
        if (a.store_avg) a.avg_new[((size_t)shard * 4 + c) * 64 + lane] = avg[c];
      if constexpr (GEO == 0) {
        if (a.want_dt) dtmin = cfl_dt(avg, h, a.cfl, a.degree);
      }
    }
    bool have_dt = GEO == 0 && a.want_dt;
    if constexpr (GEO == 1 && MODE != 2) {
      if (a.dtq) {
        have_dt = true;
        if (active) {
          double maxeig = red[(8 * N) * 64 + lane];
#pragma unroll
          for (int b = 1; b < N; ++b) maxeig = fmax(maxeig, red[(8 * N + b) * 64 + lane]);
          dtmin = a.cfl * hq / maxeig / (2.0 * a.degree + 1.0);
          if (a.dt_cell_out) a.dt_cell_out[(size_t)shard * 64 + lane] = dtmin;
        }
      }
    }
    res = wave_sum_lane63(res);
    if (have_dt) dtmin = wave_min_lane63(dtmin);
    if (lane == 63) {
      a.shard_res[shard] = res;
      if (have_dt) a.shard_dtmin[shard] = dtmin;
    }
  }
  if (a.dl_begin) deliver_traces<N>(a, shard);
  if (a.dla_begin) deliver_averages(a.dla_begin, a.dla_slot, a.dla_dst, a.dla_flag, a.dla_nflag, a.dla_total, a.dla_seq, a.dla_done, a.avg_new, shard, a.dl_fence);
  if (a.tail_word && sidx == 0 && tid == 0) {   // (StageArgs::tail_word: the launch does not end before that word is up)
    const long long t0 = wall_clock64();
    // (relaxed: nothing is read behind the word here -- the kernel only must not END before it is up; an acquire per poll would
    //  invalidate this XCD's caches under the workgroups that are still computing)
    while (__hip_atomic_load(a.tail_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < a.tail_seq) {
      __builtin_amdgcn_s_sleep(32);
      if (a.wt_ticks > 0 && wall_clock64() - t0 > a.wt_ticks) {
        if (a.wt_fail) __hip_atomic_store(a.wt_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
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
// node row B of the own cells from their modes in the LDS image Um[4 NM][S] (every wave has put the modes it owns there)
template <int N, int B>
__device__ __forceinline__ void modal_to_row(const double *Um, const int S, const int lane, double (&urow)[4][N]) {
  constexpr int NM = N * (N + 1) / 2;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int aa = 0; aa < N; ++aa) urow[c][aa] = 0.0;
#pragma unroll
    for (int m = 0; m < NM; ++m) {   // (mode by mode: one value in flight per node sum, the sums in the order m = 0, 1, ..)
      const double um = Um[(c * NM + m) * S + lane];
#pragma unroll
      for (int aa = 0; aa < N; ++aa) urow[c][aa] += PB<N>::t.T[aa + N * B][m] * um;
    }
  }
}

// phase C for node row B, then projection of the nodal residual on the modes and the modal update of the
// modes this wave owns (m = B, B+N, ...)
// GEO 1 (bilinear cells; FE_DGP lives on the reference cell, src/claw.cc:91-119 with mapping q1): the nodal residual of the row
// is formed with the metric terms of row_update_q1; the mass matrix is the DIAGONAL the reference keeps,
// M_mm = sum_q psi_m(x_q)^2 JxW_q (src/claw.cc:228-258: "not exact for general cells"), and the cell average is the quadrature
// of the modal expansion over the cell, sum_m U_m (sum_q psi_m JxW_q) / |K| (src/claw.cc:589-593), no longer mode 0.
// MF (N = 4, squares; north_star's per-element MFMA contraction, src/main.cc:44-48 FE_DGP + src/assemble_explicit.cc:85-115): the
// two dense tables of the modal element on the matrix pipe, v_mfma_f64_16x16x4_f64 with 16 cells along the columns.  Operand layout
// measured with tools/mfma_f64_16x16_probe.hip: A[i][k] in lane i + 16 k, B[k][n] in lane n + 16 k, D[i][n] in lane n + 16 (i % 4),
// register i / 4.  Wave w takes the cells 16 w .. 16 w + 15 and lane (g, n) = (lane / 16, lane % 16) holds the modes g, g + 4, g + 8 of cell
// 16 w + n in its registers t = 0, 1, 2 -- which IS the B operand of k-step t (mode 4 t + g) of
//     nodal = T modes:   D[node i][cell] = sum_m T[i][m] U_m        (A = T, 16 x 12, the columns of the modes 10, 11 zero)
// and the layout D leaves the result of
//     modal = T^T nodal: D[mode i][cell] = sum_j T[j][i] R_j        (A = T^T, rows 10 .. 15 zero; K = the 16 nodes, read from the LDS image)
// in (mode 4 r + g in register r): the modes are loaded, updated and stored where the matrix instruction wants them, four 128-byte
// segments per wavefront access instead of one 512-byte line.
template <int N>
__device__ __forceinline__ double pk_table_entry(int node, int mode) {   // T[node][mode], 0 beyond the last mode (per-lane index: a load from the constant table)
  constexpr int NM = N * (N + 1) / 2;
  const double v = PB<N>::t.T[node][mode < NM ? mode : NM - 1];
  return mode < NM ? v : 0.0;
}

template <int N, int B, int MODE, int STREAM, int GEO = 0, int MF = 0>
__device__ __forceinline__ void row_update_pk(const StageArgs &a, double *Us, const int S, const double *Fh, double *red,
                                              int shard, int lane, bool active, double h, const double (&vx)[8], const uint16_t (&cref)[4],
                                              const double (&Wrow)[N][4], const double (&ucur)[4][(N * (N + 1) / 2 + N - 1) / N],
                                              const double (&uold)[4][(N * (N + 1) / 2 + N - 1) / N], const double dt,
                                              const bool active_x = false, const double h_x = 0.0, const double dt_x = 0.0) {
  const int HS = a.halo_stride;   // row stride of the trace / flux table (Fh)
  constexpr int NS = N * N, NM = N * (N + 1) / 2;
  double R[4][N];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int m = 0; m < N; ++m) R[c][m] = 0.0;
  // metric terms of the bilinear map (GEO 1): x_xi depends on eta only, x_eta on xi only
  const double ax = vx[2] - vx[0], bx = (vx[6] - vx[4]) - ax, ay = vx[3] - vx[1], by = (vx[7] - vx[5]) - ay;
  const double cx = vx[4] - vx[0], dx = (vx[6] - vx[2]) - cx, cy = vx[5] - vx[1], dy = (vx[7] - vx[3]) - cy;
  if constexpr (GEO == 1) {
    // as row_update_q1: xi part lifted here, eta part (x_xi G - y_xi F) w w through the LDS image
    double Hown[N][4];
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
        double etx, ety;
        face_edge(vx, f, etx, ety);
        const double len = fsqrt(etx * etx + ety * ety);
        if (f < 2) {
          const int qq = flip ? N - 1 - B : B;
          const double jxw = sgn * CB<N>::t.w[B] * len;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const double fq = Fh[row_off<N>(c * N, qq, HS) + k] * jxw;
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
            for (int c = 0; c < 4; ++c) R[c][q] += Fh[row_off<N>(c * N, qq, HS) + k] * jxw;
          }
        }
      }
    }
  } else {
  // P3, first stage (LEAN, as in row_update): built for 3 wavefronts per SIMD -- the nodal values of the own row and the own G
  // row come back from the LDS image instead of being held in registers, the G values are taken node by node
  constexpr bool LEAN = (N == 4 && (MF || MODE == 0 || (MODE == 1 && kPkLeanLater))) || (kPkLeanHigh && N >= 5);   // (N >= 5, round 6: the later stages spilled 60-512 bytes per lane otherwise)
  double Gown[N][4];
#pragma unroll
  for (int aa = 0; aa < N; ++aa) {
    double Fx[4], Wa[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) Wa[c] = LEAN ? Us[(c * NS + aa + N * B) * S + lane] : Wrow[aa][c];
    flux_xy(Wa, Fx, Gown[aa]);
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
      R[MY][aa] += a.gravity * (-1.0 * Wa[RHO]) * jxw;
      R[EN][aa] += a.gravity * (-1.0 * Wa[MY]) * jxw;
    }
  }
  __syncthreads();
  int ln = lane;
#pragma unroll
  for (int aa = 0; aa < N; ++aa) {
    const double wah = CB<N>::t.w[aa] * h;
    if constexpr (LEAN) {
      if (aa > 0) asm volatile("" : "+v"(ln) : "v"(R[0][aa - 1]), "v"(R[1][aa - 1]), "v"(R[2][aa - 1]), "v"(R[3][aa - 1]));
    }
#pragma unroll
    for (int q = 0; q < N; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double gy = (q == B && !LEAN) ? Gown[aa][c] : Us[(c * NS + aa + N * q) * S + ln];
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
          const double fq = Fh[row_off<N>(c * N, qq, HS) + k] * jxw;
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
          for (int c = 0; c < 4; ++c) R[c][q] += Fh[row_off<N>(c * N, qq, HS) + k] * jxw;
        }
      }
    }
  }
  }   // GEO
  // ---- nodal residual -> modal residual: rhs_m = sum over rows of sum_a psi_m(x_(a,b)) R_b[.][a], the rows added in a
  //      fixed order (row 0 first).  Every wave leaves its row of the nodal residual in the now unused LDS image and, after ONE
  //      barrier, forms the sums of the modes it owns (m = B, B + N, ..) from all rows -- the waves used to take turns at
  //      adding their row to every mode through LDS, N barriers one after the other.  The same sums in the same order.
  __syncthreads();  // every wave is done with the G exchange
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int aa = 0; aa < N; ++aa) Us[(c * NS + aa + N * B) * S + lane] = R[c][aa];
  __syncthreads();
  constexpr int MSB = (NM - B + N - 1) / N;   // modes of this wave
  double part[5] = {0, 0, 0, 0, 0};
  if constexpr (MF && N == 4 && GEO == 0) {
    typedef double mf_d4 __attribute__((ext_vector_type(4)));
    constexpr int MS = (NM + N - 1) / N;
    const int g = lane >> 4, cx = 16 * B + (lane & 15);
    double At[4];   // A = T^T: lane (i, k) = (lane % 16, lane / 16) holds T[node 4 ks + k][mode i] of k-step ks
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) At[ks] = pk_table_entry<N>(4 * ks + (lane >> 4), lane & 15);
    const double rh2 = frcp(h_x * h_x);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      mf_d4 d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)   // node 4 ks + g = (aa, b) = (g, ks) of cell cx
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(At[ks], Us[(c * NS + g + N * ks) * S + cx], d, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < MS; ++t) {
        const int m = 4 * t + g;
        if (m < NM && active_x) {
          const double rm = d[t];
          if constexpr (MODE == 2) {
            a.rhs_out[((size_t)shard * 4 * NM + c * NM + m) * 64 + cx] = rm;
          } else {
            part[4] += rm * rm;
            double u = ucur[c][t];
            u += dt_x * rm * rh2;
            if constexpr (MODE == 1) u = (1.0 - a.ark) * u + a.ark * uold[c][t];
            stream_store<STREAM>(&a.Unew[((size_t)shard * 4 * NM + c * NM + m) * 64 + cx], u);
            if (m == 0) part[c] = u;   // the cell average is mode 0
          }
        }
      }
    }
    if constexpr (MODE != 2) {
      __syncthreads();  // every wave is done reading Fh
#pragma unroll
      for (int c = 0; c < 5; ++c) red[(g * 5 + c) * 64 + cx] = part[c];   // lane group g in the place of wave g: the epilogue's sums over the rows
    }
    return;
  }
  {
    const double rh2 = frcp(h * h);  // inverse mass of the orthonormal modes: 1/|K|
    double invM[MSB > 0 ? MSB : 1], cavg[MSB > 0 ? MSB : 1];   // GEO 1: per owned mode, from the cell's own metric
    if constexpr (GEO == 1) {
      const double area = 0.5 * fabs((vx[0] * vx[3] - vx[2] * vx[1]) + (vx[2] * vx[7] - vx[6] * vx[3]) +
                                     (vx[6] * vx[5] - vx[4] * vx[7]) + (vx[4] * vx[1] - vx[0] * vx[5]));
      const double ia = 1.0 / area;
#pragma unroll
      for (int t = 0; t < MSB; ++t) {
        double mm = 0.0, cm = 0.0;
#pragma unroll
        for (int b = 0; b < N; ++b)
#pragma unroll
          for (int aa = 0; aa < N; ++aa) {
            const double det = (ax + CB<N>::t.x[b] * bx) * (cy + CB<N>::t.x[aa] * dy) - (cx + CB<N>::t.x[aa] * dx) * (ay + CB<N>::t.x[b] * by);
            const double jxw = CB<N>::t.w[aa] * CB<N>::t.w[b] * det, psi = PB<N>::t.T[aa + N * b][B + N * t];
            mm += psi * psi * jxw;
            cm += psi * jxw;
          }
        invM[t] = frcp(mm);
        cavg[t] = cm * ia;
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double rmv[MSB > 0 ? MSB : 1];
#pragma unroll
      for (int b = 0; b < N; ++b) {
        double pr[MSB > 0 ? MSB : 1];
#pragma unroll
        for (int t = 0; t < MSB; ++t) pr[t] = 0.0;
#pragma unroll
        for (int aa = 0; aa < N; ++aa) {
          const double rn = Us[(c * NS + aa + N * b) * S + lane];
#pragma unroll
          for (int t = 0; t < MSB; ++t) pr[t] += PB<N>::t.T[aa + N * b][B + N * t] * rn;
        }
#pragma unroll
        for (int t = 0; t < MSB; ++t) rmv[t] = b == 0 ? pr[t] : rmv[t] + pr[t];
      }
      if (active) {
      int t = 0;
#pragma unroll
      for (int m = B; m < NM; m += N, ++t) {
        const double rm = rmv[t];
        if constexpr (MODE == 2) {
          a.rhs_out[((size_t)shard * 4 * NM + c * NM + m) * 64 + lane] = rm;
        } else {
          part[4] += rm * rm;
          double u = ucur[c][t];
          u += dt * rm * (GEO == 1 ? invM[t] : rh2);
          if constexpr (MODE == 1) u = (1.0 - a.ark) * u + a.ark * uold[c][t];
          stream_store<STREAM>(&a.Unew[((size_t)shard * 4 * NM + c * NM + m) * 64 + lane], u);
          if constexpr (GEO == 1) part[c] += u * cavg[t];   // this wave's modes' share of the cell average
          else if (m == 0) part[c] = u;  // squares: the cell average is mode 0
        }
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

template <int N, int FLUX, int MODE, int STREAM, int GEO = 0, int MF = 0>
__global__ __launch_bounds__(64 * N, N >= 5 ? 1 : (N == 4 ? (((MF || MODE == 0 || (MODE == 1 && kPkLeanLater)) && FLUX != DFLO_FLUX_LXF && GEO == 0) ? kQ3FirstStageWaves : 2) : (GEO == 1 ? 2 : 3))) void stage_kernel_pk(const StageArgs a) {
  static_assert(!MF || (N == 4 && GEO == 0), "the matrix-pipe tables: P3 on squares");
  constexpr int NS = N * N, NM = N * (N + 1) / 2, NDOFM = 4 * NM, NT = 64 * N, MS = (NM + N - 1) / N;
  constexpr int ROWS = 4 * NS + (FLUX == DFLO_FLUX_LXF ? 3 : 0);
  constexpr int TROWS = 4 * N;
  constexpr int S = 65;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int sidx = shard_of_block(blockIdx.x, a.n_list, a.sweep_rev);
  if (sidx < 0) return;
  const int shard = a.shard_list ? a.shard_list[sidx] : sidx;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int row = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HS = a.halo_stride;
  double *Us = lds;
  double *Th = Us + ROWS * S;
  double *Av = Th + TROWS * HS;
  double *Bv = Av + (FLUX == DFLO_FLUX_LXF ? 3 * a.halo_cols : 0);
  int *Bk = (int *)(Bv + a.max_bnd * 4 * N);
  double *Vx = (double *)(Bk + ((a.max_bnd + 1) & ~1));   // GEO 1: [4][2][64] outward unit normals of the own cells' faces

  // ---- loads: a wave reads the modes it will update (m = row, row + N, ..) of u(s) and u(n); the node rows need all modes of
  //      the cell, which the waves hand each other through LDS (each of them used to load all of them: 4 NM loads and as many
  //      registers per wave instead of 4 NM / N)
  // LxF: lambda comes from the cell averages (src/equation.h:357-359), and the average of a modal function IS its mode 0 -- the
  // kernel takes it from the modes it loads anyway (own cells: wave 0 holds mode 0; halo cells: the gather below), the array of
  // averages is not read.  The gather then runs with (entry, component) = (t >> 2, t & 3), the four components of an entry in the
  // four lanes of a quad, instead of the components spread over half-waves.
  constexpr bool LXF = FLUX == DFLO_FLUX_LXF && GEO == 0;   // (bilinear cells: the average is not mode 0 -- the arrays of averages are read)
  int hent[1];
  hent[0] = a.halo_pad[(size_t)shard * a.halo_pitch + (LXF ? min(tid >> 2, a.halo_pitch - 1) : (tid & 31))];
  const int4 hdr = a.shard_hdr[shard];
  const int nf = hdr.y, nh = hdr.z, nbnd = hdr.w;
  const bool active = lane < (hdr.x & 0xFF);
  const int pat = hdr.x >> 8;   // index pattern of the shard: its face records and face references
  const uint32_t *fp = a.faces_pad + (size_t)pat * a.face_pitch;
  uint32_t frr[3];
#pragma unroll
  for (int it = 0; it < 3; ++it) frr[it] = fp[min(face_of_point<N>(tid + it * NT, nf), a.face_pitch - 1)];
  uint16_t cref[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) cref[f] = a.cell_face[((size_t)pat * 4 + f) * 64 + lane];
  const double h = GEO == 1 ? 0.0 : (a.uniform_h ? a.h_uniform : a.cell_h[(size_t)shard * 64 + lane]);
  double vx[8] = {0, 0, 0, 0, 0, 0, 0, 0}, uavg[4];
  if constexpr (GEO == 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) vx[k] = a.cell_vert[(size_t)k * a.n_slots + (size_t)shard * 64 + lane];
    if constexpr (FLUX == DFLO_FLUX_LXF) {
      if (row == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) uavg[c] = a.avg_cur[((size_t)shard * 4 + c) * 64 + lane];
      }
    }
  }
  double dt_step = 0.0;   // fetched with the other loads, not in the middle of phase C
  if constexpr (MODE != 2) dt_step = a.dt_cell ? a.dt_cell[(size_t)shard * 64 + lane] : (a.dt_host >= 0.0 ? a.dt_host : step_dt(a.dts, a.dt_dev));
  // MF: lane (g, n) = (lane / 16, lane % 16) of wave w holds the modes g + 4 t of cell cx = 16 w + n (see row_update_pk); otherwise wave w
  // holds the modes w + N t of cell `lane`
  const int mg = MF ? lane >> 4 : row, cx = MF ? 16 * row + (lane & 15) : lane;
  double ucur[4][MS], uold[4][MS];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < MS; ++t) {
      const int m = min(mg + N * t, NM - 1);
      ucur[c][t] = a.Ucur[((size_t)shard * NDOFM + c * NM + m) * 64 + cx];
      if constexpr (MODE == 1) uold[c][t] = __builtin_nontemporal_load(&a.Uold[((size_t)shard * NDOFM + c * NM + m) * 64 + cx]);   // read once per stage
    }
  bool active_x = active;
  double h_x = h, dt_x = dt_step;
  if constexpr (MF) {   // the same three of the cell the lane updates in the matrix layout
    active_x = cx < (hdr.x & 0xFF);
    h_x = a.uniform_h ? a.h_uniform : a.cell_h[(size_t)shard * 64 + cx];
    if constexpr (MODE != 2) dt_x = a.dt_cell ? a.dt_cell[(size_t)shard * 64 + cx] : dt_step;
  }

  double urow[4][N];
  if constexpr (MF) {
    // ---- phase A on the matrix pipe: nodal = T modes, straight from the registers the modes were loaded into; the result lands in the
    //      rows of the nodal image (node 4 r + g = (aa, b) = (g, r) in register r), every wave reads its row back in phase C (LEAN)
    typedef double mf_d4 __attribute__((ext_vector_type(4)));
    double Am[MS];   // A = T: lane (i, k) = (lane % 16, lane / 16) holds T[node i][mode 4 ks + k] of k-step ks
#pragma unroll
    for (int ks = 0; ks < MS; ++ks) Am[ks] = pk_table_entry<N>(lane & 15, 4 * ks + (lane >> 4));
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      mf_d4 d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < MS; ++ks) d = __builtin_amdgcn_mfma_f64_16x16x4f64(Am[ks], ucur[c][ks], d, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) Us[(c * NS + mg + N * r) * S + cx] = d[r];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < N; ++m) urow[c][m] = 0.0;   // (phase C reads its row from the image)
  } else {
  // ---- phase A: the modes meet in LDS (in the rows the nodal image will take), every wave forms its node row from them
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < MS; ++t) {
      const int m = row + N * t;
      if (m < NM) Us[(c * NM + m) * S + lane] = ucur[c][t];
    }
  __syncthreads();
  if constexpr (N == 2) {
    if (row == 0) modal_to_row<N, 0>(Us, S, lane, urow); else modal_to_row<N, 1>(Us, S, lane, urow);
  } else if constexpr (N == 3) {
    if (row == 0) modal_to_row<N, 0>(Us, S, lane, urow); else if (row == 1) modal_to_row<N, 1>(Us, S, lane, urow); else modal_to_row<N, 2>(Us, S, lane, urow);
  } else if constexpr (N == 4) {
    if (row == 0) modal_to_row<N, 0>(Us, S, lane, urow); else if (row == 1) modal_to_row<N, 1>(Us, S, lane, urow);
    else if (row == 2) modal_to_row<N, 2>(Us, S, lane, urow); else modal_to_row<N, 3>(Us, S, lane, urow);
  } else if constexpr (N == 5) {
    if (row == 0) modal_to_row<N, 0>(Us, S, lane, urow); else if (row == 1) modal_to_row<N, 1>(Us, S, lane, urow);
    else if (row == 2) modal_to_row<N, 2>(Us, S, lane, urow); else if (row == 3) modal_to_row<N, 3>(Us, S, lane, urow); else modal_to_row<N, 4>(Us, S, lane, urow);
  } else {
    if (row == 0) modal_to_row<N, 0>(Us, S, lane, urow); else if (row == 1) modal_to_row<N, 1>(Us, S, lane, urow);
    else if (row == 2) modal_to_row<N, 2>(Us, S, lane, urow); else if (row == 3) modal_to_row<N, 3>(Us, S, lane, urow);
    else if (row == 4) modal_to_row<N, 4>(Us, S, lane, urow); else modal_to_row<N, 5>(Us, S, lane, urow);
  }
  __syncthreads();   // every wave has read the modes: the nodal image may take their place
  }   // !MF
  // halo: traces of the neighbour's modal expansion on the shared face, psi_m = Pt_i(xi) Pt_j(eta) with one coordinate fixed at
  // 0 or 1 and the other at the N face points.  A thread takes one (entry, component): it loads the NM modes once, sums over
  // the fixed direction (both ways, the face decides which one counts) and evaluates the N points from the N sums -- the modes
  // used to be loaded once per face point, N NM loads for N values.
  for (int i = tid; i < ((nh + 31) & ~31) * 4; i += NT) {
    const int sl = LXF ? i >> 2 : (i & 31) + ((i >> 5) >> 2) * 32, c = LXF ? i & 3 : (i >> 5) & 3;
    if (sl >= nh) continue;
    const int e = (LXF ? i < NT : sl < 32) ? hent[0] : a.halo_pad[(size_t)shard * a.halo_pitch + sl];
    const int ic = e & 0x0FFFFFFF, f = (e >> 28) & 3;
    const double *hp = a.Ucur + ((size_t)(ic >> 6) * NDOFM + c * NM) * 64 + (ic & 63);
    double um[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) um[m] = hp[m * 64];
    double ax[N], ay[N];   // ax[j] = sum_i Pt_i(0|1) U_(i,j): the face xi = 0|1;  ay[i] = sum_j Pt_j(0|1) U_(i,j): the face eta = 0|1
#pragma unroll
    for (int n = 0; n < N; ++n) ax[n] = ay[n] = 0.0;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int mi = PB<N>::t.mi[m], mj = PB<N>::t.mj[m];
      ax[mj] += ((f & 1) ? PB<N>::t.P1[mi] : PB<N>::t.P0[mi]) * um[m];
      ay[mi] += ((f & 1) ? PB<N>::t.P1[mj] : PB<N>::t.P0[mj]) * um[m];
    }
#pragma unroll
    for (int q = 0; q < N; ++q) {
      double v = 0.0;
#pragma unroll
      for (int n = 0; n < N; ++n) v += PB<N>::t.Px[q][n] * (f < 2 ? ax[n] : ay[n]);
      Th[row_off<N>(0, c * N + q, HS) + sl] = v;
    }
    if constexpr (LXF) {   // (u, v, c) of the entry's average = its four modes 0, met through the quad
      double A[4], uvc[3];
      A[0] = dpp_f64<0x00, 0xf>(0.0, um[0]);
      A[1] = dpp_f64<0x55, 0xf>(0.0, um[0]);
      A[2] = dpp_f64<0xAA, 0xf>(0.0, um[0]);
      A[3] = dpp_f64<0xFF, 0xf>(0.0, um[0]);
      wave_speed_uvc(A, uvc);
      if (c < 3) Av[c * a.halo_cols + sl] = c == 0 ? uvc[0] : (c == 1 ? uvc[1] : uvc[2]);
    }
  }
  if constexpr (!MF) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int m = 0; m < N; ++m) Us[(c * NS + m + N * row) * S + lane] = urow[c][m];
  }
  if constexpr (LXF) {
    if (mg == 0) {   // wave 0 (MF: lane group 0 of every wave) owns mode 0 (ucur[.][0])
      const double A[4] = {ucur[0][0], ucur[1][0], ucur[2][0], ucur[3][0]};
      double uvc[3];
      wave_speed_uvc(A, uvc);
#pragma unroll
      for (int c = 0; c < 3; ++c) Us[(4 * NS + c) * S + cx] = uvc[c];
    }
  }
  if constexpr (FLUX == DFLO_FLUX_LXF && GEO == 1) {   // (u, v, c) of the stored averages: own cells and halo entries
    if (row == 0) {
      double uvc[3];
      wave_speed_uvc(uavg, uvc);
#pragma unroll
      for (int c = 0; c < 3; ++c) Us[(4 * NS + c) * S + lane] = uvc[c];
    }
    for (int sl = tid; sl < nh; sl += NT) {
      const int ic = (sl < 32 ? hent[0] : a.halo_pad[(size_t)shard * a.halo_pitch + sl]) & 0x0FFFFFFF;
      double A[4], uvc[3];
#pragma unroll
      for (int c = 0; c < 4; ++c) A[c] = a.avg_cur[((size_t)(ic >> 6) * 4 + c) * 64 + (ic & 63)];
      wave_speed_uvc(A, uvc);
#pragma unroll
      for (int c = 0; c < 3; ++c) Av[c * a.halo_cols + sl] = uvc[c];
    }
  }
  if constexpr (GEO == 1) {   // outward unit normals of the own cells' faces (as in stage_kernel)
    for (int f = row; f < 4; f += N) {
      double tx, ty;
      face_edge(vx, f, tx, ty);
      const double len = sqrt(tx * tx + ty * ty);
      const bool left = f == 1 || f == 2;
      Vx[(2 * f) * 64 + lane] = (left ? ty : -ty) / len;
      Vx[(2 * f + 1) * 64 + lane] = (left ? -tx : tx) / len;
    }
  }
  if (nbnd > 0) {
    for (int i = tid; i < nbnd * 4 * N; i += NT) {
      const int bl = i / (4 * N), k2 = i - bl * 4 * N;
      const int bf = a.bnd_pad[(size_t)shard * a.bnd_pitch + bl];
      if (k2 == 0) Bk[bl] = a.bface_kind[bf];
      Bv[i] = a.bval[(size_t)bf * 4 * N + k2];
    }
  }
  __syncthreads();
  flux_phase<N, FLUX, GEO>(a, Us, Th, Av, frr, fp, Bv, Bk, Vx, HS, nf, nh, tid);
  __syncthreads();

  double *Fh = Th, *red = Th;
  double wrow[N][4];
#pragma unroll
  for (int m = 0; m < N; ++m)
#pragma unroll
    for (int c = 0; c < 4; ++c) wrow[m][c] = urow[c][m];
#define DFLO_ROWPK(Bq) row_update_pk<N, Bq, MODE, STREAM, GEO, MF>(a, Us, S, Fh, red, shard, lane, active, h, vx, cref, wrow, ucur, uold, dt_step, active_x, h_x, dt_x)
  if constexpr (N == 2) {
    if (row == 0) DFLO_ROWPK(0); else DFLO_ROWPK(1);
  } else if constexpr (N == 3) {
    if (row == 0) DFLO_ROWPK(0); else if (row == 1) DFLO_ROWPK(1); else DFLO_ROWPK(2);
  } else if constexpr (N == 4) {
    if (row == 0) DFLO_ROWPK(0); else if (row == 1) DFLO_ROWPK(1); else if (row == 2) DFLO_ROWPK(2); else DFLO_ROWPK(3);
  } else if constexpr (N == 5) {
    if (row == 0) DFLO_ROWPK(0); else if (row == 1) DFLO_ROWPK(1); else if (row == 2) DFLO_ROWPK(2); else if (row == 3) DFLO_ROWPK(3); else DFLO_ROWPK(4);
  } else {
    if (row == 0) DFLO_ROWPK(0); else if (row == 1) DFLO_ROWPK(1); else if (row == 2) DFLO_ROWPK(2); else if (row == 3) DFLO_ROWPK(3);
    else if (row == 4) DFLO_ROWPK(4); else DFLO_ROWPK(5);
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
    const bool have_dt = a.want_dt && GEO == 0;   // (bilinear cells: compute_time_step_q, a pass of its own over the new state)
    if (active) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (a.store_avg) a.avg_new[((size_t)shard * 4 + c) * 64 + lane] = avg[c];
      if (have_dt) dtmin = cfl_dt(avg, h, a.cfl, a.degree);
    }
    res = wave_sum_lane63(res);
    if (have_dt) dtmin = wave_min_lane63(dtmin);
    if (lane == 63) {
      a.shard_res[shard] = res;
      if (have_dt) a.shard_dtmin[shard] = dtmin;
    }
  }
}


// ------------------------------------------------------------------ dispatch
// The kernels of one N (5 fluxes x 3 modes x 2 geometries x the limiter variants, Qk and Pk) are instantiated in their own
// translation unit (stage_inst.hip, -DDFLO_STAGE_N=N) and reached through stage_of_N / stage_pk_of_N.
typedef void (*stage_fn)(const StageArgs);
// MF 1: the matrix-pipe variants (N = 4 only: Q3's eta-derivative, P3's modal <-> nodal tables on squares), their own translation
// units (stage_inst.hip with -DDFLO_STAGE_MF=1), reached through stage_mf_of_4 / stage_pk_mf_of_4 when the engine runs with DFLO_MFMA=1
template <int N, int FLUX, int MF = 0>
stage_fn pick_stage_m(int mode, int geo, int pos, int nt) {
#define DFLO_SK(MODE, GEO, POS, STREAM) stage_kernel<N, FLUX, MODE, GEO, POS, STREAM, 0, MF>
#define DFLO_SK_AF(MODE, STREAM) stage_kernel<N, FLUX, MODE, 0, 0, STREAM, 1, MF>
  if constexpr (FLUX == DFLO_FLUX_LXF) {
    if (pos == 3) {   // LxF without the arrays of cell averages (AF; squares, no limiter): pos is free to carry the request
      if (mode == 2) return DFLO_SK_AF(2, 0);
      if (nt) return mode == 0 ? DFLO_SK_AF(0, 1) : DFLO_SK_AF(1, 1);
      return mode == 0 ? DFLO_SK_AF(0, 0) : DFLO_SK_AF(1, 0);
    }
  }
  if (pos == 1 && mode != 2) {   // the limiter has been applied on the way out: nothing re-reads the state
    if (geo == 0) return mode == 0 ? DFLO_SK(0, 0, 1, 1) : DFLO_SK(1, 0, 1, 1);
    return mode == 0 ? DFLO_SK(0, 1, 1, 1) : DFLO_SK(1, 1, 1, 1);
  }
  if (pos == 2 && mode != 2 && geo == 0)   // the limiter pass reads the marked cells only
    return mode == 0 ? DFLO_SK(0, 0, 2, 1) : DFLO_SK(1, 0, 2, 1);
  if (mode == 2) return geo == 0 ? DFLO_SK(2, 0, 0, 0) : DFLO_SK(2, 1, 0, 0);
  if (nt) {
    if (geo == 0) return mode == 0 ? DFLO_SK(0, 0, 0, 1) : DFLO_SK(1, 0, 0, 1);
    return mode == 0 ? DFLO_SK(0, 1, 0, 1) : DFLO_SK(1, 1, 0, 1);
  }
  if (geo == 0) return mode == 0 ? DFLO_SK(0, 0, 0, 0) : DFLO_SK(1, 0, 0, 0);
  return mode == 0 ? DFLO_SK(0, 1, 0, 0) : DFLO_SK(1, 1, 0, 0);
#undef DFLO_SK
#undef DFLO_SK_AF
}
template <int N, int MF = 0>
stage_fn pick_stage_n(int flux, int mode, int geo, int pos, int nt) {
  switch (flux) {
    case DFLO_FLUX_LXF: return pick_stage_m<N, DFLO_FLUX_LXF, MF>(mode, geo, pos, nt);
    case DFLO_FLUX_SW: return pick_stage_m<N, DFLO_FLUX_SW, MF>(mode, geo, pos, nt);
    case DFLO_FLUX_KFVS: return pick_stage_m<N, DFLO_FLUX_KFVS, MF>(mode, geo, pos, nt);
    case DFLO_FLUX_ROE: return pick_stage_m<N, DFLO_FLUX_ROE, MF>(mode, geo, pos, nt);
    default: return pick_stage_m<N, DFLO_FLUX_HLLC, MF>(mode, geo, pos, nt);
  }
}
template <int N, int FLUX, int MF = 0>
stage_fn pick_pk_m(int mode, int nt) {   // nt bit 0: nothing re-reads the new state before the next stage kernel (see STREAM); bit 1: bilinear cells
  if (nt & 2) {   // (bilinear cells: the tables stay on the vector units, the mass matrix is per cell and mode there)
    if (mode == 2) return stage_kernel_pk<N, FLUX, 2, 0, 1>;
    if (nt & 1) return mode == 0 ? stage_kernel_pk<N, FLUX, 0, 1, 1> : stage_kernel_pk<N, FLUX, 1, 1, 1>;
    return mode == 0 ? stage_kernel_pk<N, FLUX, 0, 0, 1> : stage_kernel_pk<N, FLUX, 1, 0, 1>;
  }
  if (mode == 2) return stage_kernel_pk<N, FLUX, 2, 0, 0, MF>;
  if (nt) return mode == 0 ? stage_kernel_pk<N, FLUX, 0, 1, 0, MF> : stage_kernel_pk<N, FLUX, 1, 1, 0, MF>;
  return mode == 0 ? stage_kernel_pk<N, FLUX, 0, 0, 0, MF> : stage_kernel_pk<N, FLUX, 1, 0, 0, MF>;
}
template <int N, int MF = 0>
stage_fn pick_pk_n(int flux, int mode, int nt) {
  switch (flux) {
    case DFLO_FLUX_LXF: return pick_pk_m<N, DFLO_FLUX_LXF, MF>(mode, nt);
    case DFLO_FLUX_SW: return pick_pk_m<N, DFLO_FLUX_SW, MF>(mode, nt);
    case DFLO_FLUX_KFVS: return pick_pk_m<N, DFLO_FLUX_KFVS, MF>(mode, nt);
    case DFLO_FLUX_ROE: return pick_pk_m<N, DFLO_FLUX_ROE, MF>(mode, nt);
    default: return pick_pk_m<N, DFLO_FLUX_HLLC, MF>(mode, nt);
  }
}
stage_fn stage_mf_of_4(int flux, int mode, int geo, int pos, int nt);
stage_fn stage_pk_mf_of_4(int flux, int mode, int nt);
stage_fn stage_of_1(int flux, int mode, int geo, int pos, int nt);
stage_fn stage_of_2(int flux, int mode, int geo, int pos, int nt);
stage_fn stage_of_3(int flux, int mode, int geo, int pos, int nt);
stage_fn stage_of_4(int flux, int mode, int geo, int pos, int nt);
// N = 5, 6: one translation unit per flux (stage_inst.hip with -DDFLO_STAGE_FLUX)
#define DFLO_DECL_BY_FLUX(N)                                                                                                   \
  stage_fn stage_of_##N##_f0(int, int, int, int); stage_fn stage_of_##N##_f1(int, int, int, int); stage_fn stage_of_##N##_f2(int, int, int, int); \
  stage_fn stage_of_##N##_f3(int, int, int, int); stage_fn stage_of_##N##_f4(int, int, int, int);                                \
  stage_fn stage_pk_of_##N##_f0(int, int); stage_fn stage_pk_of_##N##_f1(int, int); stage_fn stage_pk_of_##N##_f2(int, int);    \
  stage_fn stage_pk_of_##N##_f3(int, int); stage_fn stage_pk_of_##N##_f4(int, int);                                              \
  inline stage_fn stage_of_##N(int flux, int mode, int geo, int pos, int nt) {                                                   \
    switch (flux) {                                                                                                              \
      case 0: return stage_of_##N##_f0(mode, geo, pos, nt); case 1: return stage_of_##N##_f1(mode, geo, pos, nt);                 \
      case 2: return stage_of_##N##_f2(mode, geo, pos, nt); case 3: return stage_of_##N##_f3(mode, geo, pos, nt);                 \
      default: return stage_of_##N##_f4(mode, geo, pos, nt);                                                                     \
    }                                                                                                                            \
  }                                                                                                                              \
  inline stage_fn stage_pk_of_##N(int flux, int mode, int nt) {                                                                  \
    switch (flux) {                                                                                                              \
      case 0: return stage_pk_of_##N##_f0(mode, nt); case 1: return stage_pk_of_##N##_f1(mode, nt);                               \
      case 2: return stage_pk_of_##N##_f2(mode, nt); case 3: return stage_pk_of_##N##_f3(mode, nt);                               \
      default: return stage_pk_of_##N##_f4(mode, nt);                                                                            \
    }                                                                                                                            \
  }
DFLO_DECL_BY_FLUX(5)
DFLO_DECL_BY_FLUX(6)
#undef DFLO_DECL_BY_FLUX
stage_fn stage_pk_of_1(int flux, int mode, int nt);
stage_fn stage_pk_of_2(int flux, int mode, int nt);
stage_fn stage_pk_of_3(int flux, int mode, int nt);
stage_fn stage_pk_of_4(int flux, int mode, int nt);


}  // namespace dflo
