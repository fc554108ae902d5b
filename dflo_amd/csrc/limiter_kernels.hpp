// limiter_kernels.hpp -- time step of bilinear cells, TVB / positivity limiter passes (Qk, Pk), KXRCF indicator
// Part of the device side of engine.hip (see there for the layout of the data and of a stage).
#pragma once
#include "kernels_common.hpp"
#include "bc_program.hpp"

namespace dflo {

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
  int *flags;  // [0] negative mean state, [1] positivity root failure (raise_flag)
  const int *step_ctr;   // as StageArgs::step_ctr
  double h_uniform, M, beta;
  int n_shards, uniform_h, tvb, char_lim, pos_lim;
  int conserve_ang_mom;   // Pk: src/limiter.cc:496-500
  const int32_t *shard_list;
  int n_list;
  int sweep_rev;
  const double *shock;  // KXRCF indicator per cell, or null: "shock indicator = limiter" marks every cell (1e20)
  unsigned long long *mask;   // [n_shards] from the stage kernel: the cells this pass can change (cleared here), or null: all cells
  // with marks, on launches over all shards: the marked shards as a list (StageArgs::lim_list).  The grid is a few hundred
  // wavefronts that walk it; mark_cnt is the stage kernel's count, mark_cnt_next the counter the next stage kernel will use
  // (zeroed here: the two alternate).  null: one wavefront per shard looks at its word.
  const ulonglong2 *mark_list;   // (shard, the word the stage kernel OR-ed into mask[shard])
  const int *mark_cnt;
  int *mark_cnt_next;
  // multi-device, TVB: the averages of the ghost cells as their owners sent them, [n_ghost][4] in ghost order (the receive area
  // itself: no unpack kernel between the arrival and this pass), or null: they are in `avg` like everybody's
  const double *ghost_avg;
  int first_ghost_slot;
  // bilinear cells, last stage: the time step of the limited solution is formed here, while the cell is in registers
  double *shard_dtmin, *dt_cell;
  double cfl;
  int degree, dtq;
  // the pass behind the last stage of a step also forms the step's reductions (the residual norms, the CFL minimum, the next
  // time step): its first fin_blocks wavefronts each take one workgroup's share of finalize_kernel before their shard -- one
  // launch and its latency less per step (C3: 6 of 139 us).  0: finalize_kernel is launched as usual.
  int fin_blocks;
  FinalArgs fin;
  // the pass behind stage 0 can take the boundary programs along: bc_blocks extra wavefronts at the end of the grid fill the table
  // of the later stages (values at t + dt) while the others limit -- stage 0 itself has read the table the previous step's later
  // stages used, whose program-evaluated entries are the values at this step's t, bit for bit (the clock adds the same dt to the
  // same t).  bc_eval_kernel and its 11 us then leave the path between two steps (C4).  0: nothing to take along.
  int bc_blocks;
  BcArgs bc;
  // The pass can take the exchange of a multi-device TVB stage along (one process per GPU over mapped tables): rim_blocks extra
  // wavefronts, one per shard on a cut (rim_list), whatever its marks -- each waits for the neighbours' averages of this stage (the
  // words wt_*; the stage kernel that ran before delivered this rank's), limits its shard with them (ghost_avg) and then delivers
  // the traces of the limited state on the shard's cut faces (dl_*, as the stage kernel does where no pass sits in between).
  // The stage kernel keeps those shards off the list of marked shards.  0: nothing to take along.
  int rim_blocks;
  const int32_t *rim_list;
  const int32_t *dl_begin;
  const int2 *dl_rec;
  double *const *dl_dst;
  unsigned long long *const *dl_flag;
  int dl_nflag, dl_total;
  unsigned long long dl_seq;
  unsigned int *dl_done;
  int dl_fence;
  const unsigned long long *const *wt_flag;
  int wt_n;
  unsigned long long wt_seq;
  int *wt_fail;
  long long wt_ticks;
  // limiter_rim_ghost_kernel (multi-device TVB stage with one exchange): the launch's list also holds the ghost shards (numbers from
  // n_shards on); a ghost shard's wavefront limits its cells whatever the marks say and then forms the traces gt_begin[k] ..
  // gt_begin[k + 1] of ghost shard k -- (ghost cell, face) pairs gt_slot / gt_face -- into gt_out [trace][4][N]
  const int32_t *gt_begin, *gt_slot, *gt_face;
  double *gt_out;
  KBasis kb;
};

// apply_limiter_TVB_Qk (src/limiter.cc:225-370) then apply_positivity_limiter
// (src/positivity.cc:17-208), lane = cell, all DoFs of the cell in registers.
template <int N>
__device__ __forceinline__ void limiter_shard(const LimArgs &a, const int shard, const bool listed = false, const unsigned long long word = 0,
                                              const bool masked = true) {
  constexpr int NS = N * N, NDOF = 4 * NS;
  const int lane = threadIdx.x;
  const bool active = lane < a.shard_count[shard];   // padding lanes hold a harmless state and run along
  const KBasis &kb = a.kb;
  double *up = a.U + (size_t)shard * NDOF * 64 + lane;
  double U[NDOF], A[4];
  // With the stage kernel's marks only the cells the limiters can change go through the pass (the others are provably left
  // as they are, see the stage kernel): most wavefronts return after one load.
  bool marked = true;
  if (a.mask && masked) {
    const unsigned long long m = listed ? word : a.mask[shard];   // listed: the word came with the list entry (one round trip less)
    if (m == 0) return;
    if (lane == 0) a.mask[shard] = 0;   // consumed (the load above has returned: m was compared)
    marked = (m >> lane) & 1;
  }
  int nb[4] = {-1, -1, -1, -1};   // left, right, bottom, top: asked for with the DoFs (one memory round trip less on the path of a
                                  // wavefront that has to look at its neighbours)
  if (marked) {
#pragma unroll
    for (int d = 0; d < NDOF; ++d) U[d] = up[d * 64];
    if (a.tvb) {
#pragma unroll
      for (int f = 0; f < 4; ++f) nb[f] = a.lrbt[((size_t)shard * 4 + f) * 64 + lane];
    }
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
    // dx * cell-average gradient in direction dir (D0) and its characteristic projection (D); l_m(1) - l_m(0) is antisymmetric
    // in m, and pairing the nodes makes the slope of a constant state exactly zero
    auto slopes = [&](int dir, double *D0, double *D) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double g = 0;
#pragma unroll
        for (int b = 0; b < N; ++b)
#pragma unroll
          for (int m = 0; m < N / 2; ++m) {
            const int j0 = dir == 0 ? m + N * b : b + N * m, j1 = dir == 0 ? (N - 1 - m) + N * b : b + N * (N - 1 - m);
            g += CB<N>::t.w[b] * (CB<N>::t.L1[m] - CB<N>::t.L0[m]) * (U[c * NS + j0] - U[c * NS + j1]);
          }
        D0[c] = D[c] = g;
      }
      if (a.char_lim) to_char(e, dir, D);
    };
    // minmod hands back its first argument, zero, or something of the same sign and no larger: |limited - slope| <= |slope|.
    // So the "change" that decides whether the cell is rewritten (src/limiter.cc:347: > 1e-10) is at most a quarter of the sum
    // of the |slopes| -- where that is below the threshold in every cell of the wavefront (states that are constant up to
    // rounding: most of a shock tube), the neighbours need not be looked at at all.
    bool tiny;
    {
      double D0[4], Dx[4], Dy[4];
      slopes(0, D0, Dx);
      slopes(1, D0, Dy);
      double bx = 0, by = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) { bx += fabs(Dx[i]); by += fabs(Dy[i]); }
      tiny = 0.25 * bx + 0.25 * by <= 1.0e-10;
    }
    if (!__all(tiny || !active)) {
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
      double D[4], D0[4], db[4], df[4];
      // (the boundary case "no neighbour: difference = own slope" (:296-316) is resolved before the projection: D0)
      slopes(dir, D0, D);
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
      const int ib = nb[2 * dir], ifw = nb[2 * dir + 1];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double ab = (a.ghost_avg && ib >= a.first_ghost_slot) ? a.ghost_avg[(size_t)(ib - a.first_ghost_slot) * 4 + c]
                                                                    : a.avg[((size_t)(max(ib, 0) >> 6) * 4 + c) * 64 + (max(ib, 0) & 63)];
        const double af = (a.ghost_avg && ifw >= a.first_ghost_slot) ? a.ghost_avg[(size_t)(ifw - a.first_ghost_slot) * 4 + c]
                                                                     : a.avg[((size_t)(max(ifw, 0) >> 6) * 4 + c) * 64 + (max(ifw, 0) & 63)];
        db[c] = ib >= 0 ? A[c] - ab : D0[c];
        df[c] = ifw >= 0 ? af - A[c] : D0[c];
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
    const bool bad = smin(A[RHO], pressure(A)) < eps;
    // The bound of the stage kernels (positivity_box_settled): a point value on a line through Gauss nodes lies within
    // [lo - d s, hi + d s] of the cell's nodal extremes; if the lowest density and pressure of that box are safely positive,
    // theta1 = theta2 = 1 and the limiter leaves the cell alone.  A wavefront whose cells all pass skips the limiter proper
    // (the Sod tube of C3: most of them; measured 28.7 -> see DESIGN 3.2).
    bool settled;
    {
      double lo[4], hi[4], chk = 0.0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        lo[c] = hi[c] = U[c * NS];
        chk += U[c * NS];
#pragma unroll
        for (int j = 1; j < NS; ++j) {
          lo[c] = fmin(lo[c], U[c * NS + j]);
          hi[c] = fmax(hi[c], U[c * NS + j]);
          chk += U[c * NS + j];
        }
      }
      const double sn = kb.pg_neg;
      const double rho_lo = lo[RHO] - (hi[RHO] - lo[RHO]) * sn, e_lo = lo[EN] - (hi[EN] - lo[EN]) * sn;
      const double dmx = (hi[MX] - lo[MX]) * sn, dmy = (hi[MY] - lo[MY]) * sn;
      const double mxa = fmax(fabs(lo[MX] - dmx), fabs(hi[MX] + dmx)), mya = fmax(fabs(lo[MY] - dmy), fabs(hi[MY] + dmy));
      const double p_lo = kG1 * (e_lo - 0.5 * (mxa * mxa + mya * mya) * frcp(rho_lo));
      settled = (chk - chk == 0.0) && rho_lo >= 1.0e-10 + 1.0e-8 * hi[RHO] && p_lo >= 1.0e-10 + 1.0e-8 * fabs(hi[EN]);
    }
    if (bad) {  // "Fatal: Negative states" :26-38
      if (active) raise_flag(a.flags, 0, a.step_ctr);
    } else if (!__all(settled || !active)) {
      // density at GLL(Ng) x Gauss(N) and Gauss(N) x GLL(Ng)  (:43-47, :72-78)
      double rho_min = 1.0e20;
#pragma unroll
      for (int l = 0; l < N; ++l)
        for_gll_points(a.kb.Ng, [&](auto kind, int g) {
          const double vx = gll_point<N, decltype(kind)::value>(kb, g, [&](int m) { return U[RHO * NS + m + N * l]; });
          const double vy = gll_point<N, decltype(kind)::value>(kb, g, [&](int m) { return U[RHO * NS + l + N * m]; });
          rho_min = smin(smin(rho_min, vx), vy);
        });
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
          for_gll_points(a.kb.Ng, [&](auto kind, int g) {
            double W[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
              W[c] = gll_point<N, decltype(kind)::value>(kb, g, [&](int m) { return dir == 0 ? U[c * NS + m + N * l] : U[c * NS + l + N * m]; });
            theta2 = smin(theta2, positivity_theta2(W, A, eps, fail));
          });
      if (fail && active) raise_flag(a.flags, 1, a.step_ctr);
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

template <int N>
__global__ __launch_bounds__(64) void limiter_kernel(const LimArgs a) {
  extern __shared__ unsigned char bc_lds[];   // kBcWaveLds bytes when bc_blocks > 0, else none
  const int grid = (int)gridDim.x - a.bc_blocks - a.rim_blocks;   // the wavefronts that limit the shards the launch is about
  if ((int)blockIdx.x >= grid + a.rim_blocks) {
    bc_eval_wave(a.bc, (int)blockIdx.x - grid - a.rim_blocks, 1, bc_lds);
    return;
  }
  const bool rim = (int)blockIdx.x >= grid;   // a shard on a cut: averages in, limiter, traces out
  if (a.mark_list && !rim) {
    // the marked shards from the stage kernel's list, in the order they were appended (any order gives the same bits: a shard's
    // pass rewrites its own cells and reads averages, which limiting does not change).  The step's reductions ride on the LAST
    // wavefronts of the grid, which have a list entry only when the list is longer than the grid.
    ulonglong2 e = a.mark_list[blockIdx.x];   // asked for with the count, not behind it (an entry beyond the count is not used)
    const int cnt = __builtin_amdgcn_readfirstlane(*(const volatile int *)a.mark_cnt);
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.mark_cnt_next = 0;
    const int fb = (int)blockIdx.x - (grid - a.fin_blocks);
    if (a.fin_blocks > 0 && fb >= 0) finalize_by_wave(a.fin, fb, a.fin_blocks);
    for (int k = blockIdx.x; k < cnt; k += grid) {
      if (k != (int)blockIdx.x) e = a.mark_list[k];
      limiter_shard<N>(a, (int)e.x, true, e.y);
    }
    return;
  }
  // (one call of limiter_shard for the wavefront of a shard on a cut and for the wavefront-per-shard launch: a third copy of it in
  //  this kernel cost limiter_kernel<3> 18 more registers and 2.4 KB of scratch per lane, and the pass 25 us)
  int shard;
  if (rim) {
    shard = a.rim_list[(int)blockIdx.x - grid];
    if (a.wt_n && !await_words(a.wt_flag, a.wt_n, a.wt_seq, a.wt_fail, a.wt_ticks)) return;   // (timed out: neither limited with stale averages nor delivered)
  } else {
    if (a.fin_blocks > 0 && (int)blockIdx.x < a.fin_blocks) finalize_by_wave(a.fin, blockIdx.x, a.fin_blocks);
    const int sidx = shard_of_block(blockIdx.x, a.n_list, a.sweep_rev);
    if (sidx < 0) return;
    shard = a.shard_list ? a.shard_list[sidx] : sidx;
  }
  limiter_shard<N>(a, shard);
  if (rim) deliver_face_traces<N>(a.dl_begin, a.dl_rec, a.dl_dst, a.dl_flag, a.dl_nflag, a.dl_total, a.dl_seq, a.dl_done, a.U, shard, a.dl_fence);
}

// The rim shards and the ghost shards in one launch (LimArgs::gt_begin): what a part does with its neighbours' unlimited cut cells.
// A ghost cell is limited as its owner limits the original -- same inputs (the record: DoFs, average, the owner's side of the
// neighbourhood; this part's own averages across the cut), same arithmetic --, then its traces on the cut faces are formed as
// face_trace_kernel forms them (cell_face_trace: the bits the owner would have sent after its own limiter pass).
template <int N>
__global__ __launch_bounds__(64) void limiter_rim_ghost_kernel(const LimArgs a) {
  const int sidx = shard_of_block(blockIdx.x, a.n_list, a.sweep_rev);
  if (sidx < 0) return;
  const int shard = a.shard_list[sidx];
  const bool ghost = shard >= a.n_shards;   // wave-uniform
  limiter_shard<N>(a, shard, false, 0, !ghost);
  if (!ghost) return;
  const int b0 = a.gt_begin[shard - a.n_shards], n = a.gt_begin[shard - a.n_shards + 1] - b0;
  if (n == 0) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the limited cells are in U (same compute unit: deliver_face_traces)
  __syncthreads();
  for (int t = threadIdx.x; t < n * 4 * N; t += 64) {
    const int j = b0 + t / (4 * N), r = t - (t / (4 * N)) * (4 * N);
    a.gt_out[(size_t)j * 4 * N + r] = cell_face_trace<N>(a.U, a.gt_slot[j], a.gt_face[j], r / N, r % N);
  }
}

// apply_limiter_TVB_Pk (src/limiter.cc:377-516) then the Pk branch of apply_positivity_limiter
// (src/positivity.cc:100-109, 197-205); lane = cell, all modes in registers
template <int N>
__global__ __launch_bounds__(64) void limiter_pk_kernel(const LimArgs a) {
  constexpr int NM = N * (N + 1) / 2, NDOFM = 4 * NM;
  const int sidx = shard_of_block(blockIdx.x, a.n_list, a.sweep_rev);
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
    const double ang_mom = Dx[MY] - Dy[MX];   // angular momentum of a square cell = v_x - u_y, :453
    const double Dx0[4] = {Dx[0], Dx[1], Dx[2], Dx[3]}, Dy0[4] = {Dy[0], Dy[1], Dy[2], Dy[3]};
    EigenXY e;
    if (a.char_lim) {
      e = eigen_at(A);
      to_char(e, 0, Dx); to_char(e, 1, Dy);
    }
    double Dxn[4], Dyn[4], change_x = 0, change_y = 0;
    // |minmod(a, ..) - a| <= |a|: where a quarter of the sum of the |slopes| is below the rewrite threshold in every cell of the
    // wavefront, the neighbours are not looked at (see limiter_kernel)
    double bx = 0, by = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { bx += fabs(Dx[i]); by += fabs(Dy[i]); }
    if (!__all(bx / 4 + by / 4 <= 1.0e-10)) {   // (padding lanes have left the kernel)
    const int il = a.lrbt[((size_t)shard * 4 + 0) * 64 + lane], ir = a.lrbt[((size_t)shard * 4 + 1) * 64 + lane];
    const int ib = a.lrbt[((size_t)shard * 4 + 2) * 64 + lane], it = a.lrbt[((size_t)shard * 4 + 3) * 64 + lane];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      dbx[c] = il >= 0 ? A[c] - a.avg[((size_t)(il >> 6) * 4 + c) * 64 + (il & 63)] : Dx0[c];
      dfx[c] = ir >= 0 ? a.avg[((size_t)(ir >> 6) * 4 + c) * 64 + (ir & 63)] - A[c] : Dx0[c];
      dby[c] = ib >= 0 ? A[c] - a.avg[((size_t)(ib >> 6) * 4 + c) * 64 + (ib & 63)] : Dy0[c];
      dfy[c] = it >= 0 ? a.avg[((size_t)(it >> 6) * 4 + c) * 64 + (it & 63)] - A[c] : Dy0[c];
    }
    if (a.char_lim) {
      to_char(e, 0, dbx); to_char(e, 0, dfx); to_char(e, 1, dby); to_char(e, 1, dfy);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Dxn[i] = minmod(Dx[i], beta * dbx[i], beta * dfx[i], Mdx2);
      Dyn[i] = minmod(Dy[i], beta * dby[i], beta * dfy[i], Mdx2);
      change_x += fabs(Dxn[i] - Dx[i]);
      change_y += fabs(Dyn[i] - Dy[i]);
    }
    }
    change_x /= 4;
    change_y /= 4;
    if (change_x + change_y > 1.0e-10) {
      if (a.char_lim) {
        to_con(e, 0, Dxn);
        to_con(e, 1, Dyn);
      }
      if (a.conserve_ang_mom) {   // :496-500
        Dyn[MX] = 0.5 * (Dyn[MX] - (ang_mom - Dxn[MY]));
        Dxn[MY] = ang_mom + Dyn[MX];
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
      raise_flag(a.flags, 0, a.step_ctr);
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
      if (fail) raise_flag(a.flags, 1, a.step_ctr);
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
  int sweep_rev;
  const double *cell_vert;   // bilinear cells: [8][n_slots] vertices (normals and edge lengths of the faces), or null: squares
  int n_slots;
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

// compute_shock_indicator_kxrcf (src/indicator.cc:51-198), same-level faces; lane = cell.  Squares, or bilinear cells (cell_vert:
// the outward normal and the length of a straight edge from its two vertices, as the stage kernels form them -- on such cells
// the reference's limiters do not run, src/parameters.cc:543-544, and the indicator is the diagnostic of src/claw.cc:763, 1000).
// Runs as its own pass between the stage update and the limiter: it reads the neighbours' unlimited DoFs.
template <int N, int PK>
__global__ __launch_bounds__(64) void indicator_kernel(const IndArgs a) {
  constexpr int NS = PK ? N * (N + 1) / 2 : N * N, NDOF = 4 * NS;
  const int sidx = shard_of_block(blockIdx.x, a.n_list, a.sweep_rev);
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
    double vn = f == 0 ? -vel[0] : (f == 1 ? vel[0] : (f == 2 ? -vel[1] : vel[1])), flen = h;
    if (a.cell_vert) {
      double vx[8], tx, ty;
#pragma unroll
      for (int k = 0; k < 8; ++k) vx[k] = a.cell_vert[(size_t)k * a.n_slots + (size_t)shard * 64 + lane];
      const int va = f == 1 ? 1 : (f == 3 ? 2 : 0), vb = f == 0 ? 2 : (f == 2 ? 1 : 3);   // (face_edge of the stage kernels)
      tx = vx[2 * vb] - vx[2 * va];
      ty = vx[2 * vb + 1] - vx[2 * va + 1];
      flen = sqrt(tx * tx + ty * ty);
      const bool left = f == 1 || f == 2;
      vn = vel[0] * ((left ? ty : -ty) / flen) + vel[1] * ((left ? -tx : tx) / flen);
    }
    const double inflow_status = vn < 0 ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < N; ++q) {
      const double jxw = CB<N>::t.w[q] * flen;
      const double d = face_point_value<N, PK>(uo, f, q) - face_point_value<N, PK>(un, nf, flip ? N - 1 - q : q);
      ind += inflow_status * d * jxw;
      inflow += inflow_status * jxw;
    }
  }
  const double diameter = h * 1.4142135623730950488;
  const double denominator = pow(diameter, 0.5 * (a.degree + 1)) * inflow * A[a.component];  // :179-181
  a.shock[(size_t)shard * 64 + lane] = fabs(ind) / denominator;  // 0/0 -> NaN -> "not > 1": not limited, as in the reference
}


}  // namespace dflo
