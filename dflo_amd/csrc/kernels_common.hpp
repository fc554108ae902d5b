// kernels_common.hpp -- kernel argument blocks and the helpers every kernel uses
// Part of the device side of engine.hip (see there for the layout of the data and of a stage).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "abi.h"
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


constexpr int kDtSlots = 16;   // parts of a multi-device run (dflo_hip_multi_create: n_devices <= 16)
// reductions over shards + the global-dt rules of compute_time_step (src/claw.cc:468-476)
struct FinalArgs {
  const double *shard_res, *shard_dtmin;
  double *res_sq;  // [3] per stage
  double *dt_dev;  // [0] dt, [1] elapsed time, [2] raw min before rules
  double *partial; // [kFinBlocks][4] workgroup partials
  int *counter;    // [0] workgroups done, [2 + p] index of the time step in flight when its parity is p (StageArgs::step_ctr)
  int step_par;    // parity of the step these reductions end
  int n_shards, n_stages, res_stride, do_dt, advance_time, global_rules;  // shard_res: [n_stages][res_stride]
  int fixed_dt;    // "time step type = global" with cfl <= 0: dt = time_step (src/claw.cc:455-460)
  // several engines (multi-device): the raw CFL minima of all parts meet in a table mins[2][kDtSlots] on every engine -- row
  // p holds the minima the step of parity p reads -- which the engine that forms a minimum writes into, on its own device and on
  // the peers' (stores over xGMI peer access; rank mode: an all-reduce in place on the single slot).  The consumers of the time
  // step take the minimum over their row and apply the rules themselves (step_dt): no kernel between the reductions of a step
  // and the first kernel of the next.  null: one engine, dt_dev[0] is the time step.
  double *mins;
  double *peer_mins[kDtSlots];
  int my_slot, n_slots;
  double time_step, final_time, dt_host;
};
// what a consumer of the time step needs to form it from the table (StageArgs, BcArgs)
struct DtSrc {
  const double *row;   // the row of this step's parity, or null: dt_dev[0]
  int n;
  int global_rules, fixed_dt;
  double time_step, final_time;
};
constexpr int kFinBlocks = 64;   // workgroups of the two-level reduction (one lane of the last workgroup's first wavefront each)

struct StageArgs {
  const double *Ucur, *Uold;
  double *Unew;
  const double *avg_cur;
  double *avg_new;
  double *rhs_out;  // parity hook: write the assembled rhs instead of updating
  const int32_t *shard_count;
  const int4 *shard_hdr;      // {cells, faces, halo cells, 0}
  const int32_t *halo_pad;    // [n_shards][halo_pitch]: internal cell slot | local face << 28
  int halo_pitch;             // entries per shard in halo_pad
  int halo_stride;            // row stride of the stage kernel's LDS table of traces and fluxes: halo_cols + the most other faces of a shard
  const uint32_t *faces_pad;  // [n_shards][face_pitch] packed face records (pface_*)
  const int32_t *bnd_pad;     // [n_shards][bnd_pitch] boundary-face index of the shard's l-th boundary face
  int bnd_pitch;
  int face_pitch;
  const uint16_t *cell_face;
  const double *cell_h;
  const double *cell_vert;    // GEO 1: [8][n_slots]
  int n_slots;
  const double *bval;
  const int32_t *bface_kind;
  const double *dt_dev;   // device-resident global dt (used when dt_host < 0): [0] dt, [1] elapsed time
  DtSrc dts;              // several engines: the table of the parts' minima the time step is formed from (step_dt)
  const double *dt_cell;  // local time stepping: per internal slot, else null
  double *shard_res, *shard_dtmin;
  double dt_host, ark, gravity, cfl, h_uniform;
  int n_shards, max_bnd, uniform_h, want_dt, degree;
  int halo_cols;              // columns of that table that belong to halo entries (>= the most halo entries of a shard)
  double *dt_cell_out;        // dtq with "time step type = local": the per-cell time step of the next step
  int store_avg;              // 0: nobody reads the cell averages of this stage (no LxF flux, limiter or indicator; not the last stage)
  int dtq;                    // bilinear cells, last stage, no limiter pass behind it: the kernel forms compute_time_step_q on the way out
  const int32_t *shard_list;  // null: all shards; else the n_list shards of this launch (rim / interior)
  int n_list;
  int sweep_rev;              // see shard_of_block
  int *flags;   // POS 1: [0] negative mean state, [1] positivity root failure (as LimArgs::flags)
  const int *step_ctr;  // device count of the time steps completed since set_solution: read only when a flag is raised, so that a
                        // replayed graph reports the step it is in, not the one it was captured in.  Two slots that alternate with
                        // the parity of the step: the reductions of step s write the slot of step s + 1, never the one the kernels of
                        // step s read -- a flag raised by a pass that also carries the step's reductions names step s on every run
  const double *Tg;        // multi-device: traces of the ghost cells on the cut faces, [n_ghost_traces][4][N], of the state being read
  const int32_t *gt_slot;  // internal slot of the ghost cell of a trace (its cell average: LxF)
  unsigned long long *pos_stats;  // POS 1: [0] cells that failed the nodal-box bound (limiter proper), [1] cells it changed
  unsigned long long *lim_mask;   // POS 2: [n_shards] bit = the limiter pass may have something to do in that cell
  int *lim_cnt;                   // POS 2, launches over all shards: the shards with a mark also go on a list (one append per
  int lim_list_from;              //   (a launch over rim + ring of a multi-device part: only the shards from this index of the launch's list on --
                                  //    the ring -- go on the list; the rim shards are limited by a pass of their own, ahead of the others)
  ulonglong2 *lim_list;           //   marked shard and launch: a single wavefront writes a shard's word) as (shard, word), so that the pass is a few
                                  //   hundred wavefronts walking that list instead of one per shard that reads a word and leaves; or null
  double tvb_M;                   // POS 2: TVB constant M, < 0: the limiter pass has no TVB part
  int tvb_char, pos_check;        // POS 2: characteristic limiting; the positivity limiter runs in the pass
  // Delivery by the stage kernel itself (one process per GPU over IPC-mapped tables, launches over all shards): a workgroup whose
  // shard has cut faces forms the traces of its NEW state on them and stores them straight into the neighbours' trace tables, and
  // the last such workgroup publishes the exchange's number in the neighbours' sequence words -- no rim launch of its own, no
  // pack kernel, no second stream (dflo_hip_set_deliver / dflo_hip_stage_deliver).  null: nothing to deliver.
  const int32_t *dl_begin;              // [n_shards + 1] the shard's records
  const int2 *dl_rec;                   // (cell slot, local face) of a record
  double *const *dl_dst;                // where its 4 N doubles go (the receive area of this exchange)
  unsigned long long *const *dl_flag;   // [dl_nflag] the receivers' sequence words
  int dl_nflag, dl_total;               // dl_total: workgroups that deliver (shards with records)
  unsigned long long dl_seq;
  unsigned int *dl_done;                // their counter (zero between launches)
  int dl_fence;                         // the destinations are plain device memory: a release per delivering workgroup
  // ... and the arrival of the neighbours' traces of the stage before is awaited by the workgroups that read them (the shards with
  // records above), behind their own loads, instead of by a kernel of its own in front of this launch: wt_n words to reach wt_seq
  const unsigned long long *const *wt_flag;
  const int32_t *wt_begin;              // the records by which a shard knows that it reads ghost traces (dl_begin, or dla_begin)
  int wt_n;
  unsigned long long wt_seq;
  int *wt_fail;                         // host-mapped: a word that did not arrive within wt_ticks
  long long wt_ticks;                   // of the 100 MHz clock (DFLO_IPC_TIMEOUT_S, default 120 s; 0: no limit)
  // This launch may not END before a word (fine-grained memory) has reached tail_seq: its first workgroup polls for it when its own
  // work is done (dflo_hip_stage_tail_wait).  The multi-device schedule orders the compute stream's NEXT kernel behind a kernel of
  // the comm stream that way -- stream order does the rest -- instead of by a wait packet in front of that next kernel, which costs
  // the compute stream 8.4 us even when the event has long been set (tools/stop_event_probe.hip).
  const unsigned long long *tail_word;
  unsigned long long tail_seq;
  // TVB: what leaves from the stage kernel are the AVERAGES of the cells on a cut (the neighbours' limiter reads them; the traces
  // leave from the limiter pass, LimArgs) -- the same arrangement with cell records
  const int32_t *dla_begin;
  const int32_t *dla_slot;
  double *const *dla_dst;
  unsigned long long *const *dla_flag;
  int dla_nflag, dla_total;
  unsigned long long dla_seq;
  unsigned int *dla_done;
  KBasis kb;
};

// Failure flags (the reference's AssertThrow / exit(0), src/positivity.cc:28-38,160-169).  They live in host memory
// mapped into the device -- written only when a kernel fails, never on the hot path -- so the host's step loop reads
// them without a copy: [0] negative mean state, [1] positivity root failure, [2] 1 + index of the time step in which
// the first flag went up.
__device__ __forceinline__ void raise_flag(int *flags, int which, const int *step_ctr) {
  volatile int *f = flags;
  if (f[2] == 0) f[2] = *(const volatile int *)step_ctr + 1;
  f[which] = 1;
}

template <int NT>
__device__ __forceinline__ void stream_store(double *p, double v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// Trace of a Qk function on a face from the N nodal values on the line through the face point, ordered from the face inwards
// (l_m(1) = l_(N-1-m)(0): the faces at 1 walk their line backwards and use the weights l_m(0) as well).  One function for the
// stage kernel's halo gather and for the kernels that pack / initialise ghost traces: the same bits everywhere.
template <int N>
__device__ __forceinline__ double trace_from_line(const double (&val)[N]) {
  double v = 0.0;
#pragma unroll
  for (int m = 0; m < N; ++m) v += CB<N>::t.L0[m] * val[m];
  return v;
}
// Value at Gauss-Lobatto point g of a line of N nodal values (ascending along the line) -- what the positivity limiter samples
// (src/positivity.cc:43-47, 72-78).  The two end points ARE face quadrature points, and the limiter pins the pressure there to
// 1e-13: they are summed exactly as the stage kernels sum a trace (trace_from_line: from the face inwards, weights l_m(0)), so
// that the value the limiter has made admissible is, bit for bit, the value the next flux evaluation takes the square root of.
// (The reference is consistent with itself in the same way -- one set of shape values serves both; with two orders of summation
// the flux saw p = 1e-13 - O(1e-13) at degree 5 and went NaN where the reference does not: fuzz case 2312 of seed 4243.)
// KIND 0: the first point (the face at 0), 2: the last (the face at 1: the line walked backwards, weights l_m(0), like a trace),
// 1: an interior point g (interpolation weights Pg).  get(m): nodal value m of the line, ascending.  The caller peels the first and
// the last point off its loop over g, so that every variant is straight-line code with compile-time indices.
template <int N, int KIND, class F>
__device__ __forceinline__ double gll_point(const KBasis &kb, int g, F get) {
  double v = 0.0;
#pragma unroll
  for (int m = 0; m < N; ++m) {
    if constexpr (KIND == 0) v += CB<N>::t.L0[m] * get(m);
    else if constexpr (KIND == 2) v += CB<N>::t.L0[m] * get(N - 1 - m);
    else v += kb.Pg[g][m] * get(m);
  }
  return v;
}
// run body(kind, g) for every Gauss-Lobatto point g = 0 .. Ng - 1 with its kind as a compile-time constant
template <class B>
__device__ __forceinline__ void for_gll_points(int Ng, B body) {
  body(std::integral_constant<int, 0>(), 0);
  for (int g = 1; g < Ng - 1; ++g) body(std::integral_constant<int, 1>(), g);
  body(std::integral_constant<int, 2>(), Ng - 1);
}
// component c of the trace of cell `slot` on its local face f at face point q (Qk, shard layout U)
template <int N>
__device__ __forceinline__ double cell_face_trace(const double *U, int slot, int f, int c, int q) {
  constexpr int NS = N * N, NDOF = 4 * NS;
  const int str0 = f < 2 ? 1 : N, str = (f & 1) ? -str0 : str0;
  const int base = (f < 2 ? N * q : q) + ((f & 1) ? (N - 1) * str0 : 0);
  const double *hp = U + ((size_t)(slot >> 6) * NDOF + c * NS) * 64 + (slot & 63);
  double val[N];
#pragma unroll
  for (int m = 0; m < N; ++m) val[m] = hp[(base + m * str) * 64];
  return trace_from_line<N>(val);
}

// ---- the exchange from inside the kernels (one process per GPU over mapped tables; engine.hip: dflo_hip_set_deliver ..).
// await_words: the first n threads of the workgroup poll one sequence word each (fine-grained memory, acquire at system scope)
// until it has reached seq; the workgroup meets at a barrier.  `ticks` of the 100 MHz clock (DFLO_IPC_TIMEOUT_S; 0: wait for ever): a
// neighbour that died or fell out of step -- the failure word goes up and the WHOLE workgroup learns of it (false): it must neither
// compute with the stale records nor deliver or count itself as having delivered (the exchange then never completes anywhere, every
// rank ends in DFLO_ERR_COMM instead of going on with wrong numbers).
__device__ __forceinline__ bool await_words(const unsigned long long *const *flag, const int n, const unsigned long long seq, int *fail, const long long ticks) {
  int late = 0;
  if ((int)threadIdx.x < n) {
    const unsigned long long *w = flag[threadIdx.x];
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      __builtin_amdgcn_s_sleep(8);
      if (ticks > 0 && wall_clock64() - t0 > ticks) {
        __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        late = 1;
        break;
      }
    }
  }
  return __syncthreads_or(late) == 0;
}
// what every delivering workgroup does last: its stores have completed (the caller waited for them), count, and the last one
// publishes the exchange's number in the receivers' words
__device__ __forceinline__ void deliver_publish(unsigned long long *const *flag, const int nflag, const int total, const unsigned long long seq, unsigned int *done) {
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (atomicAdd(done, 1u) != (unsigned)total - 1u) return;
  __threadfence_system();   // once per launch: the workgroup that publishes
  __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int i = 0; i < nflag; ++i) __hip_atomic_store(flag[i], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// deliver_face_traces: the workgroup has stored its shard's new state in U; behind a barrier its threads read it back (same compute
// unit: the stores have completed and the L1 holds no older copy of rows nobody has read in this launch), form the traces on the
// shard's cut faces exactly as face_trace_kernel would (cell_face_trace: the same bits) and store them into the neighbours' tables.
// The values leave as system-scope stores, which fine-grained memory takes written through, and the workgroup WAITS for them: no
// fence per workgroup (a release at agent or system scope writes the XCD's whole L2 back: with one per delivering workgroup the
// launch was 8 us longer).  That a plain store would do as well is true of a neighbour on another device, whose memory is never
// cached here -- not of a neighbour PROCESS on this device (the tests), whose table sits in this device's memory: a plain store
// may stay dirty in this XCD's L2 while a reader on another XCD, told by the word, reads the old line from memory (seen as a
// 2e-9 difference in one run of three ranks on one GPU out of many).  `fence`: the destination is plain device memory
// (DFLO_PEER_FINEGRAINED=0), where only a release writes back.
template <int N>
__device__ __forceinline__ void deliver_face_traces(const int32_t *begin, const int2 *rec, double *const *dst, unsigned long long *const *flag, const int nflag,
                                                    const int total, const unsigned long long seq, unsigned int *done, const double *U, const int shard,
                                                    const int fence) {
  const int b0 = begin[shard], n = begin[shard + 1] - b0;   // wave-uniform
  if (n == 0) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = threadIdx.x; t < n * 4 * N; t += blockDim.x) {
    const int j = b0 + t / (4 * N), r = t - (t / (4 * N)) * (4 * N);
    const int2 rc = rec[j];
    __hip_atomic_store(&dst[j][r], cell_face_trace<N>(U, rc.x, rc.y, r / N, r % N), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (fence) __threadfence_system();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  deliver_publish(flag, nflag, total, seq, done);
}
// deliver_averages: the same for the cell averages of the shard's cells on a cut (what the neighbours' TVB limiter reads), 4 doubles
// per record, from the array the epilogue has just written
__device__ __forceinline__ void deliver_averages(const int32_t *begin, const int32_t *slot, double *const *dst, unsigned long long *const *flag, const int nflag,
                                                 const int total, const unsigned long long seq, unsigned int *done, const double *avg, const int shard,
                                                 const int fence) {
  const int b0 = begin[shard], n = begin[shard + 1] - b0;
  if (n == 0) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = threadIdx.x; t < n * 4; t += blockDim.x) {
    const int j = b0 + (t >> 2), c = t & 3, sl = slot[j];
    __hip_atomic_store(&dst[j][c], avg[((size_t)(sl >> 6) * 4 + c) * 64 + (sl & 63)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (fence) __threadfence_system();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  deliver_publish(flag, nflag, total, seq, done);
}

// Cell average of one component of a Qk function on a square from its N^2 nodal values (node (m, b) at u[m + N b]), summed
// exactly as the stage kernels' epilogue sums it -- the row b: p_b = fma(w_m w_b, u_mb, p_b) for m = 0, 1, ..; then 0 + p_0 + p_1 + ..
// -- so that an average formed from the DoFs carries the bits of the stored one (src/claw.cc:589-593).
template <int N>
__device__ __forceinline__ double cell_average_rows(const double (&u)[N * N]) {
  double v = 0.0;
#pragma unroll
  for (int b = 0; b < N; ++b) {
    double pr = 0.0;
#pragma unroll
    for (int m = 0; m < N; ++m) pr = __builtin_fma(CB<N>::t.w[m] * CB<N>::t.w[b], u[m + N * b], pr);
    v += pr;
  }
  return v;
}

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

// Row offset in the flux table (row * stride).  v_mul_lo_u32 issues at a quarter of the rate of the 24-bit multiply; rows and
// stride are far inside 24 bits.  Measured (one box, two runs each): Q1 LxF +2.1 %, C3 (Q1 Roe) +1.6 %, Q2 HLLC +-0, P2 +0.4 %,
// Q3 KFVS -0.6 %, C5 -1.3 % (the scheduler's order changes with it) -- so the 24-bit form at k <= 1 only.
// row (c N + q) of the table, c a compile-time index, q per lane (k >= 2: the expression the kernels were tuned with, untouched)
template <int N>
__device__ __forceinline__ int row_off(int cN, int q, int stride) {
  if constexpr (N <= 2) return cN * stride + __mul24(q, stride);
  else return (cN + q) * stride;
}

// blockIdx -> shard so that every XCD (block b runs on XCD b % 8) sweeps one contiguous run of
// the Morton-ordered shards: halo re-reads then hit that XCD's own L2.
// rev: the XCD walks its run backwards -- a launch that sweeps against the previous one starts on the shards that one
// touched last, which are still in the XCD's L2.
__device__ __forceinline__ int shard_of_block(int b, int n_shards, int rev = 0) {
  const int chunk = (n_shards + 7) >> 3;
  const int i = b >> 3;
  const int s = (b & 7) * chunk + (rev ? chunk - 1 - i : i);
  return i < chunk && s < n_shards ? s : -1;
}

// the global-time-step rules of compute_time_step (src/claw.cc:455-476) applied to the raw CFL minimum
__device__ __forceinline__ double dt_rules(double dt, double t, double time_step, double final_time, int global_rules, int fixed_dt) {
  if (fixed_dt) return time_step;
  if (global_rules) {
    if (dt > 0 && time_step > 0) dt = fmin(dt, time_step);
    if (t + dt > final_time) dt = fmax(final_time - t, 0.0);   // (never a step backwards once t has rounded past final_time)
  }
  return dt;
}
// what the one thread that holds the totals of a step writes: the squared residual norms, the clock, the raw CFL minimum and
// the time step of the next step
// the time step of the step in flight, as every consumer forms it (stage kernels, boundary programs, the clock): the minimum
// over the parts' raw CFL minima + the rules, or dt_dev[0] where one engine has applied them already
__device__ __forceinline__ double step_dt(const DtSrc &d, const double *dt_dev) {
  if (!d.row) return dt_dev[0];
  double raw = d.row[0];
  for (int i = 1; i < d.n; ++i) raw = fmin(raw, d.row[i]);
  return dt_rules(raw, dt_dev[1], d.time_step, d.final_time, d.global_rules, d.fixed_dt);
}
__device__ __forceinline__ void finalize_publish(const FinalArgs &a, const double (&tot)[3], double dt) {
  *a.counter = 0;  // ready for the next launch (launches on one stream do not overlap)
  for (int st = 0; st < a.n_stages; ++st) a.res_sq[st] = tot[st];
  if (a.do_dt) {
    double tt = a.dt_dev[1];
    if (a.advance_time) {  // elapsed_time += global_dt (src/claw.cc:1072) for the step just done
      const DtSrc d{a.mins ? a.mins + a.step_par * kDtSlots : nullptr, a.n_slots, a.global_rules, a.fixed_dt, a.time_step, a.final_time};
      tt += a.dt_host >= 0.0 ? a.dt_host : step_dt(d, a.dt_dev);
      a.dt_dev[1] = tt;
      a.counter[2 + (a.step_par ^ 1)] = a.counter[2 + a.step_par] + 1;   // the next step's index, in the next step's slot
    }
    a.dt_dev[2] = dt;
    if (a.mins) {   // into the row the step that will use this minimum reads: the next one, or (compute_dt) the one about to run
      const int at = (a.advance_time ? a.step_par ^ 1 : a.step_par) * kDtSlots + a.my_slot;
      a.mins[at] = dt;
      for (int q = 0; q < a.n_slots; ++q)
        if (a.peer_mins[q]) a.peer_mins[q][at] = dt;
    }
    a.dt_dev[0] = dt_rules(dt, tt, a.time_step, a.final_time, a.global_rules, a.fixed_dt);
  }
}
// finalize_kernel's workgroup `b` of `nb`, run by ONE wavefront (the first workgroups of a limiter pass that ends a step take it
// on, see limiter_kernel): lane l plays the threads l, l + 64, l + 128, l + 192 of the 256, so every sum is formed in the order
// finalize_kernel forms it -- the same bits.
__device__ __forceinline__ void finalize_by_wave(const FinalArgs &a, int b, int nb) {
  const int n = a.n_shards, l = threadIdx.x & 63;
  const int chunk = ((n + kFinBlocks - 1) / kFinBlocks + 255) & ~255;
  const int lo = b * chunk, hi = min(n, lo + chunk);
  double rs[4][3], m[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    rs[w][0] = rs[w][1] = rs[w][2] = 0.0;
    m[w] = 1.0e20;
  }
  for (int base = lo; base < hi; base += 4 * 256) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      double v[4][3], d[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int s = base + l + 64 * w + j * 256;
        const bool in = s < hi;
#pragma unroll
        for (int st = 0; st < 3; ++st) v[j][st] = (in && st < a.n_stages) ? a.shard_res[(size_t)st * a.res_stride + s] : 0.0;
        d[j] = (in && a.do_dt) ? a.shard_dtmin[s] : 1.0e20;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int st = 0; st < 3; ++st) rs[w][st] += v[j][st];
        m[w] = fmin(m[w], d[j]);
      }
    }
  }
  double sred[4][4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
#pragma unroll
    for (int st = 0; st < 3; ++st) sred[st][w] = wave_sum(rs[w][st]);
    sred[3][w] = wave_min(m[w]);
  }
  int last = 0;
  if (l == 0) {
    for (int st = 0; st < 3; ++st) a.partial[b * 4 + st] = (sred[st][0] + sred[st][1]) + (sred[st][2] + sred[st][3]);
    a.partial[b * 4 + 3] = fmin(fmin(sred[3][0], sred[3][1]), fmin(sred[3][2], sred[3][3]));
    __threadfence();
    last = atomicAdd(a.counter, 1) == nb - 1;
  }
  last = __shfl(last, 0);
  if (!last) return;
  __threadfence();
  const bool have = l < nb;
  double tot[3], dt = have ? ((const volatile double *)a.partial)[l * 4 + 3] : 1.0e20;
  for (int st = 0; st < 3; ++st) tot[st] = wave_sum(have ? ((const volatile double *)a.partial)[l * 4 + st] : 0.0);
  dt = wave_min(dt);
  if (l == 0) finalize_publish(a, tot, dt);
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


}  // namespace dflo
