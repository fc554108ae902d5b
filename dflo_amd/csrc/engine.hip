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
//
// Device code: kernels_common.hpp (argument blocks, helpers), stage_kernels.hpp (instantiated per degree in stage_inst.hip),
// limiter_kernels.hpp, small_kernels.hpp.  This file: the engine object and the C ABI.
#include <hip/hip_ext.h>
#include "stage_kernels.hpp"
#include "small_kernels.hpp"
#include "tunables.h"


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
  // Which of the two tables stage 0 reads (the other one: the later stages).  The roles swap from step to step where the limiter
  // pass behind stage 0 takes the boundary programs along (LimArgs::bc_blocks): the values the later stages of step n used, at
  // t_n + dt_n, ARE the values of stage 0 of step n + 1.  Valid while the uploaded (not program-evaluated) entries of the two tables
  // are the same (bval_equal: the host copies of the last uploads) and the clock has not been set from outside since.
  int bv_first = 0;
  std::vector<double> bval_host[2];
  bool bval_equal = true, bc_fuse = true, bc_take_along = false;
  int64_t bc_later_step = -1;   // the step whose later-stage table was evaluated with the time step its clock advance will use
  int32_t *bface_kind = nullptr;
  // device-evaluated boundary functions: host copies of the programs, rebuilt device tables when they change
  std::vector<int32_t> bc_ops[DFLO_MAX_BOUNDARIES][4];
  std::vector<double> bc_consts[DFLO_MAX_BOUNDARIES][4];
  int n_bc_programs = 0;
  bool bc_rich = false;   // the boundary programs use a transcendental function (bc_eval_kernel<true>)
  bool bc_dirty = false;
  int32_t *d_bc_ops = nullptr, *d_bc_prog = nullptr, *d_bface_id = nullptr, *d_bc_faces = nullptr;
  int n_bc_faces = 0, n_bc_ops = 0, n_bc_consts = 0;
  double *d_bc_consts = nullptr, *d_bxy = nullptr, *d_bc_pts = nullptr;
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
  double *d_cell_h = nullptr, *d_cell_vert = nullptr, *d_dt_cell = nullptr;
  double *shard_res = nullptr, *shard_dtmin = nullptr, *res_sq = nullptr, *dt_dev = nullptr, *fin_partial = nullptr;
  int *flags = nullptr;         // device view of flags_host (kernels_common.hpp: raise_flag)
  volatile int *flags_host = nullptr;   // [0] negative mean state, [1] positivity root failure, [2] 1 + step of the first
  int *fin_counter = nullptr;   // [0] finalize_kernel: workgroups done; [2 + p] index of the time step in flight while its parity is p
  bool af = false;              // LxF on squares without limiter / ghost cells: no array of cell averages on the path (stage_kernel AF)
  unsigned long long *pos_stats = nullptr;   // [2] positivity limiter inside the stage kernel: cells through the limiter proper, cells changed
  // several engines: the table of the parts' raw CFL minima, [2][kDtSlots] (row = parity of the step that reads it), this
  // engine's slot in it, and the peers' tables it writes its own minimum into (FinalArgs::mins, step_dt)
  double *dt_mins = nullptr;
  double *dt_peer[kDtSlots] = {};
  int dt_my = 0, dt_n = 0;
  int64_t steps_done = 0;       // time steps since set_solution (the index recorded with a failure flag)
  hipEvent_t ev_chunk[2] = {nullptr, nullptr};   // dflo_hip_advance: the host stays at most two chunks of steps ahead
  std::vector<double> bface_xy;  // [n_bfaces][N][2]
  int pending_rk = -1;
  int st_in = 0, st_old = 0, st_out = 0, st_avg_in = 0, st_rk = 0, st_which = 0;
  double st_dt = -1.0;
  int64_t t_stages = 0, t_seen = 0;   // stages timed / stages seen while timing is on
  int t_every = 5;                    // every t_every-th stage is timed (dflo_hip_stage_timing)
  bool t_sample = false;
  int32_t *d_rim_list = nullptr, *d_int_list = nullptr, *d_rim2_list = nullptr, *d_rest2_list = nullptr;
  double pending_dt = -1.0;
  int32_t *d_send_slots = nullptr;
  int n_send = 0;
  // delivery by the stage kernel (dflo_hip_set_deliver / dflo_hip_stage_deliver)
  std::vector<int32_t> h_sendf_slot, h_sendf_face;
  int32_t *d_dl_begin = nullptr;
  int2 *d_dl_rec = nullptr;
  double **d_dl_dst[2] = {nullptr, nullptr};
  unsigned long long **d_dl_flag = nullptr;
  int dl_nflag = 0, dl_total = 0, dl_armed = -1;
  unsigned long long dl_seq = 0;
  unsigned long long **d_wt_flag = nullptr;   // dflo_hip_set_arrival_words: this engine's own words for the neighbours' traces
  int wt_n = 0;
  int *wt_fail = nullptr;
  long long wt_ticks = 0;   // how long a workgroup polls a neighbour's word before it raises wt_fail (DFLO_IPC_TIMEOUT_S x 100 MHz; 0: for ever)
  unsigned long long wt_seq = 0;
  bool wt_armed = false;
  unsigned long long *pub_word = nullptr;          // dflo_hip_pack_publish: the next pack kernel's first thread stores pub_seq there
  unsigned long long pub_seq = 0;
  const unsigned long long *tail_word = nullptr;   // dflo_hip_stage_tail_wait: the next stage launch does not end before this word is at tail_seq
  unsigned long long tail_seq = 0;
  int dl_fence = 0;   // dflo_hip_deliver_to_plain_memory: the destinations are plain (not fine-grained) device memory
  // TVB: the averages of the cells on a cut leave from the stage kernel, the traces from the limiter pass (dflo_hip_set_deliver_averages ..)
  std::vector<int32_t> h_send_slots;
  int32_t *d_dla_begin = nullptr, *d_dla_slot = nullptr;
  double **d_dla_dst[2] = {nullptr, nullptr};
  unsigned long long **d_dla_flag = nullptr, **d_wta_flag = nullptr;
  int dla_nflag = 0, dla_total = 0, dla_armed = -1, wta_n = 0;
  unsigned long long dla_seq = 0;
  int lim_x_area = -1;                       // dflo_hip_limit_exchange: the next limiter pass over all shards takes the exchange along
  unsigned long long lim_x_seq = 0, lim_x_await = 0;
  bool lim_x_poll = false;
  hipEvent_t next_stop = nullptr;      // dflo_hip_attach_event: the next stage / limiter kernel launched carries this event as its completion signal
  unsigned int *send_done = nullptr;   // [3] workgroup counters of the signalling pack kernels, by kind (dflo_hip_pack_send_to_signal)
  bool peer_fine = false;              // DFLO_PEER_FINEGRAINED=1: what a peer's kernel writes lives in fine-grained memory
  const double *ghost_avg_src = nullptr;   // dflo_hip_ghost_avg_source: where the Qk limiter pass finds the ghost cells' averages
  int32_t *d_rim_ghost_list = nullptr, *d_gt_begin = nullptr;   // the rim shards and behind them the ghost shards; first trace of every ghost shard
  const double *ghost_ride_rec = nullptr;  // armed by dflo_hip_limit_ghost_cells: the records the next rim limiter pass takes along
  int ghost_ride_table = -1;
  double *d_gnb = nullptr;                 // [n_ghost][4 faces][4]: averages of the ghost cells' neighbours on their owners' side (dflo_hip_limit_ghost_cells)
  // ghost cells known by their face traces (Qk without the KXRCF indicator): two buffers, the stage kernels read Tg[tg_cur]
  // while the neighbours' next traces arrive in the other one
  bool trace_halo = false;
  bool tg_external = false, dt_external = false;   // Tg / dt_mins belong to the caller (dflo_hip_set_ghost_trace_buffers, _set_dt_table_buffer)
  double *Tg[2] = {nullptr, nullptr};
  int tg_cur = 0, n_gt = 0;
  int32_t *d_gt_slot = nullptr, *d_gt_face = nullptr, *d_sendf_slot = nullptr, *d_sendf_face = nullptr;
  int n_send_faces = 0;
  double *ghost_stage = nullptr;
  size_t lds_bytes = 0;
  int stage_grid = 8;
  bool lazy_avg = false, avg_valid = true;   // lazy_avg: intermediate stages do not store the cell averages (nobody reads them)
  bool resident = false;                     // inside dflo_hip_advance's loop: nobody can ask for anything between the steps
  int n_patterns = 0;   // distinct (face records, face references) among the shards
  int sweep_mode = 1, sweep_dir = 0;   // every launch over all shards walks them against the previous one (DFLO_SWEEP=0: always forward)
  bool fuse_dtq = true;                // DFLO_FUSE_DTQ=0: bilinear cells always take the separate time-step pass (dt_q_kernel)
  bool fuse_fin = true;                // DFLO_FUSE_FIN=0: finalize_kernel always as its own launch
  int stream_override = -1;            // DFLO_STREAM=0/1 forces the streaming-store variant off / on
  // timing
  bool timing = false;
  int dtq_parts = 0;   // last stage on bilinear cells: parts (1 rim, 2 interior) whose limiter pass also formed the time step
  bool finish_enqueued = false;   // the last dflo_hip_stage_finish put a kernel on the stream (dflo_hip_finish_enqueued)
  bool mfma = false;       // DFLO_MFMA=1, degree 3: the per-element contractions on the matrix pipe (Q3: eta-derivative; P3 on squares: modal <-> nodal)
  bool fuse_pos = false;   // positivity limiter without TVB on Qk: applied inside the stage kernel (DFLO_FUSE_POS=0: separate pass)
  unsigned long long *lim_mask = nullptr;   // TVB on Qk squares: [n_shards], written by the stage kernel for the limiter pass (DFLO_LIM_MASK=0: off)
  bool aux_fresh = false;      // lim_mask belongs to the state the open stage has just produced
  // the marked shards as a list (launches over all shards; DFLO_LIM_LIST=0: off): two counters that alternate from one marked
  // stage launch to the next -- the pass that walks the list of one zeroes the counter of the other -- and what the host knows
  // about them (a counter whose pass never ran is cleared by a memset before it is used again)
  int *lim_cnt = nullptr;
  ulonglong2 *lim_list = nullptr;
  int lim_epoch = 0, lim_open = -1, lim_grid = 1024;
  int lim_parts = -1;   // the launch that appended last to the open list: 0 all shards, 3 rim + ring, 4 the rest
  bool fin_done = false;   // a limiter pass of the open stage has carried the step's reductions
  bool lim_clean[2] = {true, true};
  // dflo_hip_advance replays a captured graph of `graph_steps` time steps (the buffer rotation repeats with
  // that period); built lazily for the state it was captured in
  hipGraphExec_t graph_exec = nullptr;
  int graph_steps = 0, graph_cur = -1, graph_avg = -1;
  int graph_lim = 0;   // parity of lim_epoch the graph was captured at (its launches name the two list counters in that order)
  int graph_par = 0;   // parity of the step count it was captured at (its launches name the step-index slots in that order)
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

// memory another device's kernel writes while this device works (ghost-trace tables, the table of the parts' time-step minima):
// plain device memory -- coherent at kernel boundaries, which is what the schedule relies on -- or, DFLO_PEER_FINEGRAINED=1,
// fine-grained device memory, coherent at every access
// DFLO_POISON_ALLOC=1 (developer switch, tests/test_gpu_round6.py): every device buffer of an engine starts out as 0xFF bytes (NaN as a
// double, -1 as an integer) instead of whatever the allocator hands over -- fresh memory reads as zero, a block the process has used
// before does not (LAB R5.15 / R6.3: the first engine created after a multi-device driver took the driver's freed window of sequence
// words).  Nothing may depend on the difference.
bool poison_alloc() {
  static const bool on = [] { const char *e = std::getenv("DFLO_POISON_ALLOC"); return e && std::atoi(e) != 0; }();
  return on;
}
hipError_t dmalloc(void **p, size_t bytes) {
  const hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess && poison_alloc()) (void)hipMemset(*p, 0xFF, bytes);
  return e;
}
hipError_t peer_malloc(void **p, size_t bytes, bool fine) {
  if (!fine) return dmalloc(p, bytes);
  const hipError_t e = hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained);
  if (e == hipSuccess && poison_alloc()) (void)hipMemset(*p, 0xFF, bytes);
  return e;
}

template <typename T>
int upload(dflo_hip_engine *h, T **dst, const std::vector<T> &src) {
  size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
  HIPCHK(h, dmalloc((void **)dst, bytes));
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

// kernel template K<N> for the engine's N = degree + 1 (the limiter-type kernels are never launched for N = 1)
#define DFLO_BY_N(n, K) ((n) == 1 ? K<1> : ((n) == 2 ? K<2> : ((n) == 3 ? K<3> : ((n) == 4 ? K<4> : ((n) == 5 ? K<5> : K<6>)))))
#define DFLO_BY_N2(n, K, P) ((n) <= 2 ? K<2, P> : ((n) == 3 ? K<3, P> : ((n) == 4 ? K<4, P> : ((n) == 5 ? K<5, P> : K<6, P>))))
#define DFLO_BY_N_LIM(n, K) ((n) <= 2 ? K<2> : ((n) == 3 ? K<3> : ((n) == 4 ? K<4> : ((n) == 5 ? K<5> : K<6>))))

// mf: the matrix-pipe variants of degree 3 (DFLO_MFMA=1; stage_kernels.hpp: eta_derivative_mfma, row_update_pk)
stage_fn pick_pk(int N, int flux, int mode, int nt = 0, bool mf = false) {   // nt: bit 0 streaming stores, bit 1 bilinear cells
  if (mf && N == 4) return dflo::stage_pk_mf_of_4(flux, mode, nt);
  switch (N) {
    case 1: return dflo::stage_pk_of_1(flux, mode, nt);
    case 2: return dflo::stage_pk_of_2(flux, mode, nt);
    case 3: return dflo::stage_pk_of_3(flux, mode, nt);
    case 4: return dflo::stage_pk_of_4(flux, mode, nt);
    case 5: return dflo::stage_pk_of_5(flux, mode, nt);
    default: return dflo::stage_pk_of_6(flux, mode, nt);
  }
}
stage_fn pick_stage(int N, int flux, int mode, int geo, int pos = 0, int nt = 0, bool mf = false) {
  if (mf && N == 4) return dflo::stage_mf_of_4(flux, mode, geo, pos, nt);
  switch (N) {
    case 1: return dflo::stage_of_1(flux, mode, geo, pos, nt);
    case 2: return dflo::stage_of_2(flux, mode, geo, pos, nt);
    case 3: return dflo::stage_of_3(flux, mode, geo, pos, nt);
    case 4: return dflo::stage_of_4(flux, mode, geo, pos, nt);
    case 5: return dflo::stage_of_5(flux, mode, geo, pos, nt);
    default: return dflo::stage_of_6(flux, mode, geo, pos, nt);
  }
}

// does nothing read the new state between this stage kernel and the next (no limiter / indicator pass over all cells)?
int streams_out(const dflo_hip_engine *h);
int next_sweep(dflo_hip_engine *h, int part);

int grid_for(int n_shards) { return ((n_shards + 7) / 8) * 8; }

// shard set of a partial launch: 0 all, 1 rim (shards that read ghost cells), 2 interior (the others), 3 rim + the ring of
// shards next to it, 4 the others
void part_list(const dflo_hip_engine *h, int part, const int32_t **list, int *n);

void launch_dt_q(dflo_hip_engine *h);
int launch_average(dflo_hip_engine *h);
// A pass launched outside a stage reads the stored cell averages: form them if the last stage kept its own to itself (lazy_avg)
int ensure_avg(dflo_hip_engine *h) {
  if (h->avg_valid) return DFLO_OK;
  const int rc = launch_average(h);
  if (!rc) h->avg_valid = true;
  return rc;
}
int launch_face_traces(dflo_hip_engine *h, double *out, const int32_t *slots, const int32_t *faces, int n, bool publish = false);

// the table of the parts' CFL minima as the reductions (FinalArgs) and the consumers of the time step (DtSrc) see it
void dt_table_args(const dflo_hip_engine *h, FinalArgs &f) {
  f.mins = h->dt_n > 0 ? h->dt_mins : nullptr;
  f.my_slot = h->dt_my;
  f.n_slots = h->dt_n;
  for (int q = 0; q < kDtSlots; ++q) f.peer_mins[q] = q < h->dt_n ? h->dt_peer[q] : nullptr;
}
DtSrc dt_source(const dflo_hip_engine *h) {
  DtSrc d{};
  d.row = h->dt_n > 0 ? h->dt_mins + (size_t)(h->steps_done & 1) * kDtSlots : nullptr;
  d.n = h->dt_n;
  d.global_rules = h->prm.global_time_step;
  d.fixed_dt = h->prm.global_time_step && h->prm.cfl <= 0.0;
  d.time_step = h->prm.time_step;
  d.final_time = h->prm.final_time;
  return d;
}

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
BcArgs bc_args(const dflo_hip_engine *h, double dt_host);
static bool bc_op_is_rich(int op) {
  switch (op) {
    case DFLO_OP_POW: case DFLO_OP_SIN: case DFLO_OP_COS: case DFLO_OP_TAN: case DFLO_OP_EXP: case DFLO_OP_LOG: case DFLO_OP_ATAN2:
    case DFLO_OP_TANH: case DFLO_OP_SINH: case DFLO_OP_COSH: case DFLO_OP_ASIN: case DFLO_OP_ACOS: case DFLO_OP_ATAN:
    case DFLO_OP_LOG10: case DFLO_OP_ERF: case DFLO_OP_ERFC: return true;
    default: return false;
  }
}

int open_stage(dflo_hip_engine *h, int rk, double dt_host, bool residual_only, int which_override) {
  if (rk == 0 && !residual_only && h->n_bc_programs > 0) {  // boundary functions at t (stage 0) and t + dt (later stages)
    // The stage's limiter pass over all shards -- or, in the multi-device schedule, over everything but the rim -- can evaluate the
    // later stages' table beside its own work, and stage 0 reads the table the previous step's later stages used.  A stage that is
    // finished without such a pass gets the table from bc_eval_kernel then (launch_finish).
    const bool limited = h->prm.limiter_type != DFLO_LIMITER_NONE || h->prm.pos_lim;
    const bool take = h->bc_fuse && !h->bc_dirty && !h->bc_rich && h->n_bc_ops <= kBcWaveOps && h->n_bc_consts <= kBcWaveConsts &&
                      h->bval_equal && limited && !h->fuse_pos && h->basis == DFLO_BASIS_QK && h->plan.n_shards > 0 && !h->use_graph &&
                      h->steps_done > 0 && h->bc_later_step == h->steps_done - 1;
    h->bc_take_along = take;
    if (take) {
      h->bv_first ^= 1;
    } else {
      const int rc = eval_boundary_programs(h, dt_host);
      if (rc) return rc;
      h->bc_later_step = h->steps_done;
    }
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
    h->lim_open = -1;
    h->lim_parts = -1;
    h->fin_done = false;
    if (h->lim_cnt) {
      // the list counter this stage's first marked launch will open is cleared HERE, on the stream that opens the stage: in the
      // multi-device schedule that launch (rim + ring) runs on the comm stream and the launch that joins its list (the rest) on
      // the compute stream; a memset issued with the former would be unordered against the appends of the latter
      const int i = (h->lim_epoch + 1) & 1;
      if (!h->lim_clean[i]) {
        HIPCHK(h, hipMemsetAsync(h->lim_cnt + i, 0, sizeof(int), h->stream));
        h->lim_clean[i] = true;
      }
    }
    // stage timing samples every fifth stage (5 is coprime to the 2 or 3 stages of a step, so every stage of the
    // step is sampled equally often): two event records per launch are not free
    h->t_sample = h->timing && (h->t_seen++ % h->t_every == 0);
    if (h->t_sample) ++h->t_stages;
  }
  return DFLO_OK;
}

BcArgs bc_args(const dflo_hip_engine *h, double dt_host) {
  BcArgs a{};
  a.ops = h->d_bc_ops;
  a.consts = h->d_bc_consts;
  a.prog = h->d_bc_prog;
  a.bface_id = h->d_bface_id;
  a.bxy = h->d_bxy;
  a.bval0 = h->bval[h->bv_first];
  a.bval1 = h->bval[h->bv_first ^ 1];
  a.dt_dev = h->dt_dev;
  a.dts = dt_source(h);
  a.dt_host = dt_host;
  a.faces = h->d_bc_faces;
  a.pts = h->d_bc_pts;
  a.n_faces = h->n_bc_faces;
  a.N = h->N;
  a.n_ops = h->n_bc_ops;
  a.n_consts = h->n_bc_consts;
  return a;
}

int eval_boundary_programs(dflo_hip_engine *h, double dt_host) {
  const Plan &p = h->plan;
  const int nb = (int)p.bface_cell.size();
  if (nb == 0 || h->n_bc_programs == 0) return DFLO_OK;
  if (h->bc_dirty) {  // flatten the programs: one op / constant pool, a (first, count) entry per (id, component)
    std::vector<int32_t> ops, prog(DFLO_MAX_BOUNDARIES * 4 * 2, 0);
    std::vector<double> consts;
    bool rich = false;   // any transcendental function among the programs?
    for (int b = 0; b < DFLO_MAX_BOUNDARIES; ++b)
      for (int c = 0; c < 4; ++c) {
        const std::vector<int32_t> &o = h->bc_ops[b][c];
        prog[(b * 4 + c) * 2] = (int32_t)(ops.size() / 2);
        prog[(b * 4 + c) * 2 + 1] = (int32_t)(o.size() / 2);
        for (size_t k = 0; k < o.size(); k += 2) {
          rich |= bc_op_is_rich(o[k]);
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
    std::vector<double> pts;   // per listed point: x, y, index in the value tables, boundary id
    for (int b : faces)
      for (int q = 0; q < h->N; ++q) {
        const size_t i = (size_t)b * h->N + q;
        pts.insert(pts.end(), {h->bface_xy[2 * i], h->bface_xy[2 * i + 1], (double)i, (double)p.bface_id[b]});
      }
    if (pts.empty()) pts.assign(4, 0.0);
    if (faces.empty()) faces.push_back(0);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    hipFree(h->d_bc_ops); hipFree(h->d_bc_consts); hipFree(h->d_bc_prog); hipFree(h->d_bc_faces); hipFree(h->d_bc_pts);
    h->d_bc_ops = nullptr; h->d_bc_consts = nullptr; h->d_bc_prog = nullptr; h->d_bc_faces = nullptr; h->d_bc_pts = nullptr;
    int rc;
    if ((rc = upload(h, &h->d_bc_ops, ops)) || (rc = upload(h, &h->d_bc_consts, consts)) || (rc = upload(h, &h->d_bc_prog, prog)) ||
        (rc = upload(h, &h->d_bc_faces, faces)) || (rc = upload(h, &h->d_bc_pts, pts)))
      return rc;
    if (!h->d_bface_id) {
      if ((rc = upload(h, &h->d_bface_id, p.bface_id)) || (rc = upload(h, &h->d_bxy, h->bface_xy))) return rc;
    }
    h->bc_rich = rich;
    h->bc_dirty = false;
  }
  const BcArgs a = bc_args(h, dt_host);
  if (a.n_faces == 0) return DFLO_OK;
  if (h->bc_rich) hipLaunchKernelGGL(bc_eval_kernel<true>, dim3((a.n_faces * a.N + 63) / 64), dim3(kBcThreads), 0, h->stream, a);
  else hipLaunchKernelGGL(bc_eval_kernel<false>, dim3((a.n_faces * a.N + 63) / 64), dim3(kBcThreads), 0, h->stream, a);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

// The event a multi-device driver wants recorded behind the next kernel rides on that kernel's own completion signal
// (hipExtLaunchKernel's stopEvent) instead of a record packet behind it: 3.5 us less between two kernels of a stream
// (tools/stop_event_probe.hip).  A launch that turns out to be empty records the event the plain way.
template <class K, class A>
void launch_with_event(dflo_hip_engine *h, K fn, dim3 grid, dim3 block, size_t lds, const A &args) {
  if (h->next_stop) {
    hipExtLaunchKernelGGL(fn, grid, block, (std::uint32_t)lds, h->stream, nullptr, h->next_stop, 0, args);
    h->next_stop = nullptr;
  } else {
    hipLaunchKernelGGL(fn, grid, block, lds, h->stream, args);
  }
}
void drop_attached_event(dflo_hip_engine *h) {
  if (!h->next_stop) return;
  hipEventRecord(h->next_stop, h->stream);
  h->next_stop = nullptr;
}

// residual + update kernel of the open stage; part 0: all shards, 1: rim shards, 2: interior shards
int launch_update(dflo_hip_engine *h, double *rhs_out, int part) {
  const Plan &p = h->plan;
  const int rk = h->st_rk;
  const bool last = rk == h->n_rk - 1;
  StageArgs a{};
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
  a.n_slots = p.n_slots;
  a.bval = h->bval[h->st_which == 0 ? h->bv_first : h->bv_first ^ 1];
  a.bface_kind = h->bface_kind;
  a.dt_dev = h->dt_dev;
  a.dts = dt_source(h);
  a.dt_cell = h->d_dt_cell;  // null unless "time step type = local"
  a.shard_res = h->shard_res + (size_t)rk * std::max(h->plan.n_shards, 1);
  a.shard_dtmin = h->shard_dtmin;
  a.dt_host = h->st_dt;
  a.ark = h->ark[rk];
  a.gravity = h->prm.gravity;
  a.cfl = h->prm.cfl;
  a.h_uniform = p.h;
  a.n_shards = p.n_shards;
  a.halo_cols = h->plan.halo_cols;
  a.max_bnd = h->plan.max_bnd;
  a.uniform_h = p.uniform_h ? 1 : 0;
  a.want_dt = last ? 1 : 0;
  // the averages of a stage go to memory when somebody reads them: the LxF flux and the limiter / indicator passes of the next
  // stage, the time step and the caller after the last one (a caller that asks in between gets them formed afresh)
  // (inside the device-resident loop of dflo_hip_advance not even the last stage stores them: the time step comes from the
  //  kernel's own minima, and a caller who asks afterwards gets them formed on demand, in the same order of summation)
  a.store_avg = ((last && !h->resident) || !h->lazy_avg) ? 1 : 0;
  if (!rhs_out) h->avg_valid = a.store_avg != 0;
  // bilinear cells: compute_time_step_q is formed by the last stage kernel itself when no limiter pass follows it (the
  // positivity limiter, if any, has been applied inside the kernel) -- otherwise by that pass, or by dt_q_kernel
  {
    const bool pass_follows = h->prm.limiter_type != DFLO_LIMITER_NONE || (h->prm.pos_lim && !h->fuse_pos);
    a.dtq = (h->fuse_dtq && last && !rhs_out && h->geo == 1 && h->basis == DFLO_BASIS_QK && !pass_follows && part <= 2) ? 1 : 0;
    if (a.dtq) h->dtq_parts |= part == 0 ? 3 : part;
    a.dt_cell_out = h->d_dt_cell;
  }
  a.degree = h->degree;
  a.kb = h->kb;
  part_list(h, part, &a.shard_list, &a.n_list);
  if (a.n_list == 0) {
    drop_attached_event(h);
    if (part == 0) { h->dl_armed = h->dla_armed = -1; h->wt_armed = false; }   // (nothing to launch: nothing delivers, nothing waits)
    return DFLO_OK;
  }
  a.sweep_rev = next_sweep(h, part);
  const int mode_ = rhs_out ? 2 : (h->ark[rk] != 0.0 ? 1 : 0);
  a.flags = h->flags;
  a.step_ctr = h->fin_counter + 2 + (int)(h->steps_done & 1);
  a.Tg = h->Tg[h->tg_cur];
  a.gt_slot = h->d_gt_slot;
  a.pos_stats = h->pos_stats;
  a.lim_mask = h->lim_mask;
  a.lim_cnt = nullptr;
  a.lim_list = h->lim_list;
  if (h->dl_armed >= 0 && part == 0 && !rhs_out) {   // this launch delivers its cut faces' traces itself (dflo_hip_stage_deliver)
    a.dl_begin = h->d_dl_begin;
    a.dl_rec = h->d_dl_rec;
    a.dl_dst = h->d_dl_dst[h->dl_armed];
    a.dl_flag = h->d_dl_flag;
    a.dl_nflag = h->dl_nflag;
    a.dl_total = h->dl_total;
    a.dl_seq = h->dl_seq;
    a.dl_done = h->send_done + 3;
    a.dl_fence = h->dl_fence;
    h->dl_armed = -1;
  }
  if (h->dla_armed >= 0 && part == 0 && !rhs_out) {   // TVB: it delivers the averages of its cells on a cut
    a.dla_begin = h->d_dla_begin;
    a.dla_slot = h->d_dla_slot;
    a.dla_dst = h->d_dla_dst[h->dla_armed];
    a.dla_flag = h->d_dla_flag;
    a.dla_nflag = h->dla_nflag;
    a.dla_total = h->dla_total;
    a.dla_seq = h->dla_seq;
    a.dla_done = h->send_done + 1;
    a.dl_fence = h->dl_fence;
    a.store_avg = 1;
    h->avg_valid = true;
    h->dla_armed = -1;
  }
  if (h->wt_armed && part == 0 && !rhs_out && (a.dl_begin || a.dla_begin)) {
    // ... and waits for the neighbours' traces of the stage before in its workgroups that read them (the shards with records)
    a.wt_begin = a.dl_begin ? a.dl_begin : a.dla_begin;
    a.wt_flag = h->d_wt_flag;
    a.wt_n = h->wt_n;
    a.wt_seq = h->wt_seq;
    a.wt_fail = h->wt_fail;
    a.wt_ticks = h->wt_ticks;
    h->wt_armed = false;
  }
  if (h->tail_word && !rhs_out) {
    a.tail_word = h->tail_word;
    a.tail_seq = h->tail_seq;
    a.wt_fail = h->wt_fail;
    a.wt_ticks = h->wt_ticks;
    h->tail_word = nullptr;
  }
  a.tvb_M = h->prm.limiter_type == DFLO_LIMITER_TVB ? h->prm.M : -1.0;
  a.tvb_char = h->prm.char_lim;
  a.pos_check = h->prm.pos_lim;
  const int pos_ = h->fuse_pos ? 1 : (h->lim_mask ? 2 : (h->af ? 3 : 0));
  if (pos_ == 2 && mode_ != 2) {
    h->aux_fresh = true;
    if (!(part == 4 && h->lim_parts == 3)) h->lim_open = -1;
    // The list of marked shards: launches over all shards, and the two launches of a multi-device part whose limiter pass
    // over "everything but the rim" comes later -- rim + ring (3: only the ring goes on the list, the rim has a pass of its own)
    // and the rest (4), which share one list; 4 joins the list 3 opened.
    if (h->lim_cnt && (part == 0 || part == 3 || part == 4)) {
      if (part == 4 && h->lim_parts == 3) {
        a.lim_cnt = h->lim_cnt + h->lim_open;
      } else {
        const int i = ++h->lim_epoch & 1;
        if (!h->lim_clean[i]) HIPCHK(h, hipMemsetAsync(h->lim_cnt + i, 0, sizeof(int), h->stream));
        h->lim_clean[i] = false;
        h->lim_open = i;
        a.lim_cnt = h->lim_cnt + i;
      }
      h->lim_parts = part;
      a.lim_list_from = part == 3 ? (int)p.rim_shards.size() : 0;
    }
  }
  stage_fn fn = h->basis == DFLO_BASIS_PK ? pick_pk(h->N, h->prm.flux_type, mode_, streams_out(h) | (h->geo << 1), h->mfma) : pick_stage(h->N, h->prm.flux_type, mode_, h->geo, pos_, streams_out(h), h->mfma);
  time_begin(h);
  launch_with_event(h, fn, dim3(grid_for(a.n_list)), dim3(64 * h->N), h->lds_bytes, a);
  time_end(h);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

// compute_shock_indicator (src/indicator.cc:17-30): nothing to do for type "limiter" (the limiter treats a
// missing indicator as 1e20 everywhere)
int launch_indicator(dflo_hip_engine *h, int part) {
  if (!h->d_shock) return DFLO_OK;
  if (h->pending_rk < 0) { const int rc = ensure_avg(h); if (rc) return rc; }
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
  a.cell_vert = h->geo == 1 ? h->d_cell_vert : nullptr;
  a.n_slots = p.n_slots;
  part_list(h, part, &a.shard_list, &a.n_list);
  if (a.n_list == 0) return DFLO_OK;
  a.sweep_rev = next_sweep(h, part);
  void (*fn)(const IndArgs);
  if (h->basis == DFLO_BASIS_PK) fn = DFLO_BY_N2(h->N, indicator_kernel, 1);
  else fn = DFLO_BY_N2(h->N, indicator_kernel, 0);
  hipLaunchKernelGGL(fn, dim3(grid_for(a.n_list)), dim3(64), 0, h->stream, a);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

// workgroups of finalize_kernel: as many chunks of >= 256 shards as there are, at most kFinBlocks
int fin_grid(int n_shards) {
  const int chunk = ((n_shards + kFinBlocks - 1) / kFinBlocks + 255) & ~255;
  return std::max(1, (n_shards + chunk - 1) / chunk);
}

int launch_limiter(dflo_hip_engine *h, int tvb, int pos, int part, bool stage_data = false, const FinalArgs *fin = nullptr) {
  const Plan &p = h->plan;
  if (h->pending_rk < 0) { const int rc = ensure_avg(h); if (rc) return rc; }   // standalone (apply_limiter / apply_positivity_limiter)
  LimArgs l{};
  l.U = h->U[h->cur];
  l.avg = h->avg[h->avg_cur];
  l.shard_count = h->d_shard_count;
  l.lrbt = h->d_lrbt;
  l.cell_h = h->d_cell_h;
  l.flags = h->flags;
  l.step_ctr = h->fin_counter + 2 + (int)(h->steps_done & 1);
  l.h_uniform = p.h;
  l.M = h->prm.M;
  l.beta = h->prm.beta;
  l.n_shards = p.n_shards;
  l.uniform_h = p.uniform_h ? 1 : 0;
  l.tvb = tvb;
  l.char_lim = h->prm.char_lim;
  l.pos_lim = pos;
  l.conserve_ang_mom = h->prm.conserve_angular_momentum;
  l.kb = h->kb;
  l.shock = tvb ? h->d_shock : nullptr;
  l.mask = (stage_data && tvb && h->aux_fresh) ? h->lim_mask : nullptr;
  l.ghost_avg = h->basis == DFLO_BASIS_QK ? h->ghost_avg_src : nullptr;
  l.first_ghost_slot = p.n_shards * 64;
  l.dtq = (h->geo == 1 && h->basis == DFLO_BASIS_QK && h->pending_rk == h->n_rk - 1) ? 1 : 0;
  l.shard_dtmin = h->shard_dtmin;
  l.dt_cell = h->d_dt_cell;
  l.cfl = h->prm.cfl;
  l.degree = h->degree;
  part_list(h, part, &l.shard_list, &l.n_list);
  if (l.dtq) h->dtq_parts |= part == 0 ? 3 : part;
  if (l.n_list == 0) { drop_attached_event(h); return DFLO_OK; }
  l.sweep_rev = next_sweep(h, part);
  if (fin) {
    l.fin = *fin;
    l.fin_blocks = fin_grid(fin->n_shards);
  }
  void (*lf)(const LimArgs) = DFLO_BY_N_LIM(h->N, limiter_kernel);
  if (h->basis == DFLO_BASIS_PK) lf = DFLO_BY_N_LIM(h->N, limiter_pk_kernel);
  if (h->ghost_ride_rec && part == 1 && stage_data) {   // the neighbours' unlimited cut cells ride along (dflo_hip_limit_ghost_cells)
    const int n_ghost = p.n_cells - p.n_owned;
    const long long tot = (long long)n_ghost * (h->ndof + kFatExtra);
    hipLaunchKernelGGL(unpack_fat_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->ghost_ride_rec, h->U[h->cur],
                       h->avg[h->avg_cur], h->d_gnb, p.n_shards * 64, n_ghost, h->ndof);
    HIPCHK(h, hipGetLastError());
    l.shard_list = h->d_rim_ghost_list;
    l.n_list += p.n_ghost_shards;
    l.ghost_avg = h->d_gnb;          // "slots" from n_slots on (Plan::ghost_lrbt): entry 4 g + f of the records' neighbour averages;
    l.first_ghost_slot = p.n_slots;  // the rim cells find their ghost neighbours' averages in the array, where the unpack has put them
    l.gt_begin = h->d_gt_begin;
    l.gt_slot = h->d_gt_slot;
    l.gt_face = h->d_gt_face;
    l.gt_out = h->Tg[h->ghost_ride_table];
    lf = DFLO_BY_N_LIM(h->N, limiter_rim_ghost_kernel);
    h->ghost_ride_rec = nullptr;
    h->ghost_ride_table = -1;
  }
  int grid = grid_for(l.n_list);
  if (l.mask && (part == 0 || part == 2) && h->lim_open >= 0 && h->basis == DFLO_BASIS_QK && (part == 0) == (h->lim_parts == 0)) {
    l.mark_list = h->lim_list;
    l.mark_cnt = h->lim_cnt + h->lim_open;
    l.mark_cnt_next = h->lim_cnt + (h->lim_open ^ 1);
    h->lim_clean[h->lim_open ^ 1] = true;
    h->lim_open = -1;
    grid = std::min(grid, std::max(h->lim_grid, l.fin_blocks));
  }
  grid = std::max(grid, l.fin_blocks);   // (a launch over a few shards that carries the reductions of all of them)
  if (h->lim_x_area >= 0 && part == 0 && stage_data) {   // the exchange of a multi-device TVB stage rides along (dflo_hip_limit_exchange)
    if (!l.mark_list || h->basis != DFLO_BASIS_QK) { h->lim_x_area = -1; h->err = "limit_exchange: this pass does not walk a list of marked shards"; return DFLO_ERR_UNSUPPORTED; }
    l.rim_blocks = (int)p.rim_shards.size();
    l.rim_list = h->d_rim_list;
    l.dl_begin = h->d_dl_begin;
    l.dl_rec = h->d_dl_rec;
    l.dl_dst = h->d_dl_dst[h->lim_x_area];
    l.dl_flag = h->d_dl_flag;
    l.dl_nflag = h->dl_nflag;
    l.dl_total = h->dl_total;
    l.dl_seq = h->lim_x_seq;
    l.dl_done = h->send_done + 3;
    l.dl_fence = h->dl_fence;
    if (h->lim_x_poll) {
      l.wt_flag = h->d_wta_flag;
      l.wt_n = h->wta_n;
      l.wt_seq = h->lim_x_await;
      l.wt_fail = h->wt_fail;
      l.wt_ticks = h->wt_ticks;
    }
    h->lim_x_area = -1;
  }
  size_t lds = 0;
  if (stage_data && h->bc_take_along && h->pending_rk == 0 && (part == 0 || part == 2) && h->basis == DFLO_BASIS_QK) {
    // the later stages' boundary values, by extra wavefronts behind the ones that limit
    l.bc = bc_args(h, h->st_dt);
    l.bc_blocks = 4 * ((l.bc.n_faces * l.bc.N + 63) / 64);
    lds = l.bc_blocks > 0 ? kBcWaveLds : 0;
    h->bc_take_along = false;
    h->bc_later_step = h->steps_done;
  }
  launch_with_event(h, lf, dim3(grid + l.rim_blocks + l.bc_blocks), dim3(64), lds, l);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

// limiter of the open stage on one part of the shards
int launch_stage_limiter(dflo_hip_engine *h, int part, const FinalArgs *fin = nullptr) {
  if (h->pending_rk < 0) { h->err = "no stage pending"; return DFLO_ERR_BAD_PARAM; }
  const bool limited = h->prm.limiter_type != DFLO_LIMITER_NONE || h->prm.pos_lim;
  if (!limited || h->fuse_pos) {   // (fuse_pos: positivity alone, the stage kernel has applied it on the way out)
    drop_attached_event(h);
    return DFLO_OK;
  }
  if (h->prm.limiter_type == DFLO_LIMITER_TVB) {  // compute_shock_indicator(); apply_limiter();  src/claw.cc:763-764
    const int rc = launch_indicator(h, part);
    if (rc) return rc;
  }
  return launch_limiter(h, h->prm.limiter_type == DFLO_LIMITER_TVB, h->prm.pos_lim, part, true, fin);
}

// reductions of the stage launched last
void final_args(dflo_hip_engine *h, FinalArgs &f) {
  const Plan &p = h->plan;
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
  f.fixed_dt = h->prm.global_time_step && h->prm.cfl <= 0.0;
  f.partial = h->fin_partial;
  f.counter = h->fin_counter;
  f.step_par = (int)(h->steps_done & 1);
  dt_table_args(h, f);
}

int launch_finish(dflo_hip_engine *h, bool reductions_done = false) {
  const Plan &p = h->plan;
  const int rk = h->pending_rk;
  if (rk < 0) { h->err = "no stage pending"; return DFLO_ERR_BAD_PARAM; }
  const bool last = rk == h->n_rk - 1;
  h->finish_enqueued = false;
  if (h->bc_take_along) {   // no limiter pass of this stage took the boundary programs along: the kernel of their own, before the next stage
    h->bc_take_along = false;
    h->finish_enqueued = true;
    const int rc = eval_boundary_programs(h, h->st_dt);
    if (rc) return rc;
    h->bc_later_step = h->steps_done;
  }
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
  if (last) h->finish_enqueued = true;   // (time step / reductions: a caller that orders streams behind the last stage records plainly anyway)
  if (!last || reductions_done || h->fin_done) return DFLO_OK;  // ||rhs|| of every stage is reduced once, after the last stage (it is only reported, src/claw.cc:768)
  FinalArgs f{};
  final_args(h, f);
  hipLaunchKernelGGL(finalize_kernel, dim3(fin_grid(f.n_shards)), dim3(256), 0, h->stream, f);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

// limiter and reductions of the open stage.  The TVB pass over all shards that ends a step on squares takes the reductions
// along (LimArgs::fin): nothing that finalize_kernel reads is written by the pass there (the averages, and with them the CFL
// minima, do not change under limiting; on bilinear cells the pass itself forms the time step, so not there).
int launch_limit_finalize(dflo_hip_engine *h) {
  const bool last = h->pending_rk == h->n_rk - 1;
  const bool fuse = h->fuse_fin && last && h->geo == 0 && !h->d_dt_cell && h->basis == DFLO_BASIS_QK &&
                    h->prm.limiter_type == DFLO_LIMITER_TVB && !h->fuse_pos && h->plan.n_shards > 0;
  FinalArgs f{};
  if (fuse) final_args(h, f);
  int rc = launch_stage_limiter(h, 0, fuse ? &f : nullptr);
  if (rc) return rc;
  return launch_finish(h, fuse);
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
  if (h->basis == DFLO_BASIS_PK) {   // from the modes
    auto fm = DFLO_BY_N(h->N, modal_geo_kernel);
    hipLaunchKernelGGL(fm, dim3(p.n_shards), dim3(64), 0, h->stream, (const double *)h->U[h->cur], (double *)nullptr, h->shard_dtmin,
                       (const double *)h->d_cell_h, (const double *)h->d_cell_vert, p.n_slots, (const int32_t *)h->d_shard_count, h->kb,
                       h->prm.cfl, h->degree, h->d_dt_cell);
    return;
  }
  auto fn = DFLO_BY_N(h->N, dt_q_kernel);
  hipLaunchKernelGGL(fn, dim3(p.n_shards), dim3(64), 0, h->stream, (const double *)h->U[h->cur], (const double *)h->d_cell_h,
                     (const int32_t *)h->d_shard_count, h->shard_dtmin, h->kb, h->prm.cfl, h->degree, h->d_dt_cell);
}

int launch_average(dflo_hip_engine *h) {
  const Plan &p = h->plan;
  const int all = p.n_shards + p.n_ghost_shards;
  if (h->basis == DFLO_BASIS_QK && h->geo == 0) {   // squares: the epilogue's order of summation
    auto fn = DFLO_BY_N(h->N, average_rows_kernel);
    hipLaunchKernelGGL(fn, dim3(all), dim3(64), 0, h->stream, (const double *)h->U[h->cur], h->avg[h->avg_cur]);
    HIPCHK(h, hipGetLastError());
    return DFLO_OK;
  }
  if (h->basis == DFLO_BASIS_PK && h->geo == 1) {   // bilinear cells: the average of a modal function is not its mode 0
    auto fm = DFLO_BY_N(h->N, modal_geo_kernel);
    hipLaunchKernelGGL(fm, dim3(all), dim3(64), 0, h->stream, (const double *)h->U[h->cur], h->avg[h->avg_cur], (double *)nullptr,
                       (const double *)h->d_cell_h, (const double *)h->d_cell_vert, p.n_slots, (const int32_t *)h->d_shard_count, h->kb,
                       h->prm.cfl, h->degree, (double *)nullptr);
    HIPCHK(h, hipGetLastError());
    return DFLO_OK;
  }
  hipLaunchKernelGGL(average_kernel, dim3(all), dim3(64), 0, h->stream, h->U[h->cur], h->avg[h->avg_cur], h->ndof, h->kb,
                     h->basis == DFLO_BASIS_PK ? -h->N : h->N, (const double *)h->d_cell_vert, p.n_slots);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

// (what dflo_hip_pack_publish has armed goes to the pack kernel launched next, once)
static unsigned long long *take_pub(dflo_hip_engine *h, unsigned long long *seq) {
  unsigned long long *w = h->pub_word;
  *seq = h->pub_seq;
  h->pub_word = nullptr;
  return w;
}
int launch_face_traces(dflo_hip_engine *h, double *out, const int32_t *slots, const int32_t *faces, int n, bool publish) {
  if (n == 0) return DFLO_OK;
  const long long tot = (long long)n * 4 * h->N;
  auto fn = DFLO_BY_N(h->N, face_trace_kernel);
  unsigned long long ps = 0, *pw = publish ? take_pub(h, &ps) : nullptr;
  hipLaunchKernelGGL(fn, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, out, (const double *)h->U[h->cur], slots, faces, n, pw, ps);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

void drop_graph(dflo_hip_engine *h) {
  if (h->graph_exec) hipGraphExecDestroy(h->graph_exec);
  h->graph_exec = nullptr;
}

int check_handle(dflo_hip_handle h) { return h ? DFLO_OK : DFLO_ERR_BAD_PARAM; }

int streams_out(const dflo_hip_engine *h) {
  const bool limited = h->prm.limiter_type != DFLO_LIMITER_NONE || h->prm.pos_lim;
  const bool full_pass = limited && !h->fuse_pos && !h->lim_mask;
  if (h->stream_override >= 0) return h->stream_override;
  return (full_pass || h->d_shock) ? 0 : 1;
}

// direction of the next launch over all shards (parts of a rim / interior split keep the forward order)
int next_sweep(dflo_hip_engine *h, int part) {
  if (!h->sweep_mode || part != 0) return 0;
  const int d = h->sweep_dir;
  h->sweep_dir ^= 1;
  return d;
}

void part_list(const dflo_hip_engine *h, int part, const int32_t **list, int *n) {
  const Plan &p = h->plan;
  switch (part) {
    case 1: *list = h->d_rim_list; *n = (int)p.rim_shards.size(); break;
    case 2: *list = h->d_int_list; *n = (int)p.interior_shards.size(); break;
    case 3: *list = h->d_rim2_list; *n = (int)p.rim2_shards.size(); break;
    case 4: *list = h->d_rest2_list; *n = (int)p.rest2_shards.size(); break;
    default: *list = nullptr; *n = p.n_shards; break;
  }
}

// the failure flags as they stand (no synchronisation: the caller decides how fresh they have to be)
int flags_status(dflo_hip_engine *h) {
  if (h->flags_host[0]) { h->err = "Fatal: Negative states"; return DFLO_ERR_NEGATIVE_MEAN_STATE; }
  if (h->flags_host[1]) { h->err = "Problem in positivity limiter"; return DFLO_ERR_POSITIVITY_NO_ROOT; }
  return DFLO_OK;
}

}  // namespace

extern "C" {

const char *dflo_hip_last_error(dflo_hip_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int dflo_hip_create(const dflo_mesh_t *mesh, const dflo_params_t *params, int device_id, dflo_hip_handle *out) {
  return dflo_hip_create_with_cell_size(mesh, params, device_id, out, 0.0);
}

}  // extern "C"

int dflo_hip_create_with_cell_size(const dflo_mesh_t *mesh, const dflo_params_t *params, int device_id, dflo_hip_handle *out, double h_hint,
                                    bool pass_takes_exchange) {
  if (!mesh || !params || !out) { g_create_error = "null argument"; return DFLO_ERR_BAD_PARAM; }
  *out = nullptr;
  // consistency checks of the reference's parameter parsing (src/parameters.cc:536-550)
  if (mesh->degree < 0 || mesh->degree > DFLO_MAX_DEGREE) { g_create_error = "degree must be 0..5"; return DFLO_ERR_BAD_PARAM; }
  if (mesh->basis != DFLO_BASIS_QK && mesh->basis != DFLO_BASIS_PK) { g_create_error = "unknown basis"; return DFLO_ERR_BAD_PARAM; }
  // `mapping = q2` (MappingQ<dim>(2), src/claw.cc:173-176): on cells with straight edges -- all that the flat mesh can describe
  // (four vertices per cell), and all the reference has: its curved boundary description is commented out, src/claw.cc:976-979 --
  // the biquadratic map IS the bilinear one (its extra support points are the edge midpoints and the centre's image under the
  // harmonic extension of bilinear boundary data, i.e. under the bilinear map itself).  Taken as q1.
  if (mesh->mapping != DFLO_MAP_CARTESIAN && mesh->mapping != DFLO_MAP_Q1 && mesh->mapping != DFLO_MAP_Q2) { g_create_error = "unknown mapping"; return DFLO_ERR_BAD_PARAM; }
  if (params->shock_indicator < DFLO_IND_LIMITER || params->shock_indicator >= DFLO_IND_U2) {
    g_create_error = "shock indicator must be limiter, density or energy (u2 belongs to the MOOD scheme)";
    return params->shock_indicator == DFLO_IND_U2 ? DFLO_ERR_UNSUPPORTED : DFLO_ERR_BAD_PARAM;
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
  if (mesh->degree == 0) {
    // piecewise constants: apply_limiter_TVB_* and apply_positivity_limiter return at once (src/limiter.cc:379,
    // src/positivity.cc:19 -- before the "Negative states" check), and nothing reads the shock indicator
    h->prm.limiter_type = DFLO_LIMITER_NONE;
    h->prm.pos_lim = 0;
    h->prm.shock_indicator = DFLO_IND_LIMITER;
  }
  h->degree = mesh->degree;
  h->N = mesh->degree + 1;
  h->basis = mesh->basis;
  const Tunables tun = read_tunables();
  h->use_graph = tun.graph;
  h->fuse_fin = tun.fuse_fin;
  h->sweep_mode = tun.sweep ? 1 : 0;
  h->stream_override = tun.stream;
  h->fuse_dtq = tun.fuse_dtq;
  h->peer_fine = tun.peer_finegrained;
  h->bc_fuse = tun.bc_fuse;
  h->mfma = tun.mfma && h->N == 4;
  h->wt_ticks = (long long)tun.ipc_timeout_s * 100000000LL;
  h->ns = mesh->basis == DFLO_BASIS_PK ? h->N * (h->N + 1) / 2 : h->N * h->N;
  h->ndof = 4 * h->ns;
  h->mapping = mesh->mapping == DFLO_MAP_Q2 ? DFLO_MAP_Q1 : mesh->mapping;
  h->geo = mesh->mapping == DFLO_MAP_CARTESIAN ? 0 : 1;
  int rc = build_plan(*mesh, 8, 8, h->plan, h->err, h_hint);
  if (rc) { g_create_error = h->err; delete h; return rc; }
  h->bt = make_basis(h->degree);
  h->kb = make_kbasis(h->bt);
  {  // multi-device: ghost cells as face traces (N*4 doubles per cut face) when nothing needs more of them -- the stage
     // kernels read only their traces and averages; the KXRCF indicator and the Pk element read their DoFs
    h->n_gt = (int)h->plan.gt_cell.size();
    h->trace_halo = h->n_gt > 0 && h->basis == DFLO_BASIS_QK && h->prm.shock_indicator == DFLO_IND_LIMITER && !tun.halo_cells;
  }
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
  h->n_rk = h->degree == 0 ? 1 : (h->degree == 1 ? 2 : 3);
  if (params->n_rk > 0) h->n_rk = std::min(params->n_rk, 3);
  if (h->n_rk == 1) h->ark[0] = 0.0;
  if (h->n_rk == 2) { h->ark[0] = 0.0; h->ark[1] = 0.5; }
  if (h->n_rk == 3) { h->ark[0] = 0.0; h->ark[1] = 0.75; h->ark[2] = 1.0 / 3.0; }
  auto bail = [&](int code) { g_create_error = h->err; dflo_hip_destroy(h); return code; };
  if (hipSetDevice(device_id) != hipSuccess) { h->err = "hipSetDevice failed"; return bail(DFLO_ERR_HIP); }
  if (hipStreamCreate(&h->own_stream) != hipSuccess) { h->err = "hipStreamCreate failed"; return bail(DFLO_ERR_HIP); }
  h->stream = h->own_stream;
  const Plan &p = h->plan;
  const size_t nU = (size_t)p.n_slots * h->ndof;
  for (int i = 0; i < 3; ++i)
    if (dmalloc((void **)&h->U[i], nU * sizeof(double)) != hipSuccess) { h->err = "hipMalloc(U) failed"; return bail(DFLO_ERR_NOMEM); }
  for (int i = 0; i < 2; ++i)
    if (dmalloc((void **)&h->avg[i], (size_t)p.n_slots * 4 * sizeof(double)) != hipSuccess) { h->err = "hipMalloc(avg) failed"; return bail(DFLO_ERR_NOMEM); }
  if (dmalloc((void **)&h->rhs, nU * sizeof(double)) != hipSuccess) { h->err = "hipMalloc(rhs) failed"; return bail(DFLO_ERR_NOMEM); }
  if (dmalloc((void **)&h->user_buf, nU * sizeof(double)) != hipSuccess) { h->err = "hipMalloc(user_buf) failed"; return bail(DFLO_ERR_NOMEM); }
  for (int i = 0; i < 3; ++i) hipMemset(h->U[i], 0, nU * sizeof(double));
  std::vector<int32_t> kinds(p.bface_id.size());
  for (size_t b = 0; b < kinds.size(); ++b) kinds[b] = params->bc_kind[p.bface_id[b]];
  const size_t nb = std::max<size_t>(p.bface_cell.size(), 1) * h->N * 4;
  for (int w = 0; w < 2; ++w) {
    if (dmalloc((void **)&h->bval[w], nb * sizeof(double)) != hipSuccess) { h->err = "hipMalloc(bval) failed"; return bail(DFLO_ERR_NOMEM); }
    hipMemset(h->bval[w], 0, nb * sizeof(double));
  }
  if ((rc = upload(h, &h->bface_kind, kinds))) return bail(rc);
  {   // (behind the owned shards' counts those of the ghost shards: the limiter pass over the ghost cells, dflo_hip_limit_ghost_cells)
    std::vector<int32_t> counts(p.shard_count);
    counts.insert(counts.end(), p.ghost_count.begin(), p.ghost_count.end());
    if ((rc = upload(h, &h->d_shard_count, counts))) return bail(rc);
  }
  {  // fixed-pitch copies of the per-shard lists (+2 shards of slack: the kernel reads two shards ahead,
     // and 2*64*N face slots per shard so that unconditional loads stay in bounds)
    const int ns = p.n_shards + 2;
    h->face_pitch = std::max(2 * 64 * h->N, (p.max_faces + 63) & ~63);   // (a thread preloads two records; the rest is read in the loop)
    h->halo_pitch = std::max(p.max_halo, 1);
    std::vector<int4> hdr(ns, int4{0, 0, 0, 0});
    std::vector<int32_t> hp((size_t)ns * h->halo_pitch, 0);
    std::vector<uint32_t> fpad((size_t)ns * h->face_pitch, 0u);
    h->bnd_pitch = std::max(p.max_bnd, 1);
    std::vector<int32_t> bpad((size_t)ns * h->bnd_pitch, 0);
    // The face records and the cells' face references are local to a shard (slots, columns), so shards of the same shape
    // share them: on a lattice a handful of patterns (interior, edges, corners) serve every shard, and the tables stay in the
    // caches instead of being streamed from memory with the state (1.1 KB per shard and launch on C2).  The pattern of a shard
    // rides in its header, above the cell count.
    std::unordered_map<std::string, int> pattern_of;
    std::vector<uint16_t> cf_pat;
    fpad.clear();
    for (int sidx = 0; sidx < p.n_shards; ++sidx) {
      const int nh = p.halo_begin[sidx + 1] - p.halo_begin[sidx], nf = p.face_begin[sidx + 1] - p.face_begin[sidx];
      for (int k = 0; k < nh; ++k) {
        const int he = p.halo_begin[sidx] + k;
        const int who = (h->trace_halo && p.halo_gt[he] >= 0) ? (p.halo_gt[he] | kGhostTrace) : p.halo_cells[he];
        hp[(size_t)sidx * h->halo_pitch + k] = who | (p.halo_faces[he] << 28);
      }
      std::vector<uint32_t> rec(nf);
      for (int k = 0; k < nf; ++k) {
        const FaceRec &r = p.faces[p.face_begin[sidx] + k];
        rec[k] = pface_pack(r);
        if ((r.w0 >> 18) & 1) bpad[(size_t)sidx * h->bnd_pitch + ((r.w0 >> 20) & 0x3FF)] = r.w1;
      }
      const uint16_t *cf = &p.cell_face[(size_t)sidx * 4 * 64];
      std::string key((const char *)rec.data(), rec.size() * sizeof(uint32_t));
      key.append((const char *)cf, 4 * 64 * sizeof(uint16_t));
      auto it = pattern_of.find(key);
      int pat;
      if (it != pattern_of.end()) pat = it->second;
      else {
        pat = (int)pattern_of.size();
        pattern_of.emplace(std::move(key), pat);
        fpad.resize((size_t)(pat + 1) * h->face_pitch, 0u);
        std::copy(rec.begin(), rec.end(), fpad.begin() + (size_t)pat * h->face_pitch);
        cf_pat.insert(cf_pat.end(), cf, cf + 4 * 64);
      }
      hdr[sidx] = int4{p.shard_count[sidx] | (pat << 8), nf, nh, p.shard_bnd[sidx]};
    }
    if (fpad.empty()) { fpad.assign(h->face_pitch, 0u); cf_pat.assign(4 * 64, kNoFace); }
    h->n_patterns = (int)pattern_of.size();
    if ((rc = upload(h, &h->d_shard_hdr, hdr))) return bail(rc);
    if ((rc = upload(h, &h->d_halo_pad, hp))) return bail(rc);
    if ((rc = upload(h, &h->d_faces_pad, fpad))) return bail(rc);
    if ((rc = upload(h, &h->d_cell_face, cf_pat))) return bail(rc);
    if ((rc = upload(h, &h->d_bnd_pad, bpad))) return bail(rc);
    if (h->prm.shock_indicator != DFLO_IND_LIMITER) {
      if ((rc = upload(h, &h->d_nbr_code, p.nbr_code))) return bail(rc);
      std::vector<double> z((size_t)p.n_slots, 0.0);
      if ((rc = upload(h, &h->d_shock, z))) return bail(rc);
    }
  }
  if (h->trace_halo) {
    if ((rc = upload(h, &h->d_gt_slot, p.gt_cell)) || (rc = upload(h, &h->d_gt_face, p.gt_face))) return bail(rc);
    for (int i = 0; i < 2; ++i) {
      if (peer_malloc((void **)&h->Tg[i], (size_t)h->n_gt * 4 * h->N * sizeof(double), h->peer_fine) != hipSuccess) { h->err = "hipMalloc(ghost traces) failed"; return bail(DFLO_ERR_NOMEM); }
      hipMemset(h->Tg[i], 0, (size_t)h->n_gt * 4 * h->N * sizeof(double));
    }
  }
  {   // (... and the ghost cells' neighbours behind the owned cells')
    std::vector<int32_t> nb(p.lrbt);
    nb.insert(nb.end(), p.ghost_lrbt.begin(), p.ghost_lrbt.end());
    if ((rc = upload(h, &h->d_lrbt, nb))) return bail(rc);
  }
  if ((rc = upload(h, &h->d_rim_list, p.rim_shards))) return bail(rc);
  {   // (the limiter pass over the rim that takes the ghost cells along: dflo_hip_limit_ghost_cells)
    std::vector<int32_t> rg(p.rim_shards), gb(p.n_ghost_shards + 1, 0);
    for (int k = 0; k < p.n_ghost_shards; ++k) rg.push_back(p.n_shards + k);
    for (int32_t sl : p.gt_cell) ++gb[sl / 64 - p.n_shards + 1];
    for (int k = 0; k < p.n_ghost_shards; ++k) gb[k + 1] += gb[k];
    if ((rc = upload(h, &h->d_rim_ghost_list, rg)) || (rc = upload(h, &h->d_gt_begin, gb))) return bail(rc);
  }
  if ((rc = upload(h, &h->d_int_list, p.interior_shards))) return bail(rc);
  if ((rc = upload(h, &h->d_rim2_list, p.rim2_shards))) return bail(rc);
  if ((rc = upload(h, &h->d_rest2_list, p.rest2_shards))) return bail(rc);
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
  if (dmalloc((void **)&h->shard_res, 3 * nsh * sizeof(double)) != hipSuccess ||
      dmalloc((void **)&h->shard_dtmin, nsh * sizeof(double)) != hipSuccess ||
      dmalloc((void **)&h->res_sq, 4 * sizeof(double)) != hipSuccess || dmalloc((void **)&h->fin_partial, 4 * kFinBlocks * sizeof(double)) != hipSuccess ||
      dmalloc((void **)&h->dt_dev, 4 * sizeof(double)) != hipSuccess || dmalloc((void **)&h->fin_counter, 4 * sizeof(int)) != hipSuccess ||
      peer_malloc((void **)&h->dt_mins, 2 * kDtSlots * sizeof(double), h->peer_fine) != hipSuccess || dmalloc((void **)&h->pos_stats, 2 * sizeof(unsigned long long)) != hipSuccess ||
      dmalloc((void **)&h->send_done, 4 * sizeof(unsigned int)) != hipSuccess) {
    h->err = "hipMalloc(scalars) failed";
    return bail(DFLO_ERR_NOMEM);
  }
  {  // failure flags: host memory the kernels write through the mapping (only when something fails)
    void *fh = nullptr, *fd = nullptr;
    if (hipHostMalloc(&fh, 4 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(&fd, fh, 0) != hipSuccess) {
      h->err = "hipHostMalloc(flags) failed";
      return bail(DFLO_ERR_NOMEM);
    }
    h->flags_host = (volatile int *)fh;
    h->flags = (int *)fd;
    for (int i = 0; i < 4; ++i) h->flags_host[i] = 0;
  }
  hipMemset(h->res_sq, 0, 4 * sizeof(double));
  hipMemset(h->dt_dev, 0, 4 * sizeof(double));
  {
    std::vector<double> big(2 * kDtSlots, 1.0e300);   // an unused slot never wins a minimum
    hipMemcpy(h->dt_mins, big.data(), big.size() * sizeof(double), hipMemcpyHostToDevice);
  }
  hipMemset(h->fin_counter, 0, 4 * sizeof(int));
  hipMemset(h->send_done, 0, 4 * sizeof(unsigned int));
  hipMemset(h->pos_stats, 0, 2 * sizeof(unsigned long long));
  // row stride of the stage kernel's trace / flux table: a column per halo entry (its trace, then the flux of its face) and one
  // per other face; the 4 N rows also host the row partials (5 N rows of 64), the positivity minima (3 N) or the slope
  // partials (4 N), and the point maxima of the time step (N): 9 N rows of 64.  Odd: the rows fall on different LDS banks.
  h->halo_stride = std::max(p.halo_cols + std::max(p.max_inner, 1), 9 * 64 / 4) | 1;
  {
    h->fuse_pos = h->prm.pos_lim && h->prm.limiter_type == DFLO_LIMITER_NONE && h->basis == DFLO_BASIS_QK && tun.fuse_pos;
    {  // who reads the averages of an intermediate stage?  The LxF flux (lambda from cell means), a limiter pass, the indicator,
       // local time stepping -- otherwise they stay in the kernel (DFLO_LAZY_AVG=0: always stored)
      const bool pass = h->prm.limiter_type != DFLO_LIMITER_NONE || (h->prm.pos_lim && !h->fuse_pos) || h->prm.shock_indicator != DFLO_IND_LIMITER;
      // The LxF flux takes (u, v, c) of the averages from the DoFs themselves where those give the stored average bit for bit:
      // Pk always (mode 0); Qk on squares when no limiter touches the state between the epilogue that forms the average and the
      // next stage (the reference's cell_average is the one of before the limiters, src/claw.cc:762-766) and no ghost cell
      // brings an average of its owner's (stage_kernel AF; the partial sums of the own cells borrow 64 columns of the flux table)
      const bool limited = h->prm.limiter_type != DFLO_LIMITER_NONE || h->prm.pos_lim;
      h->af = h->prm.flux_type == DFLO_FLUX_LXF && h->basis == DFLO_BASIS_QK && h->geo == 0 && !limited &&
              h->prm.shock_indicator == DFLO_IND_LIMITER && p.n_cells == p.n_owned && p.halo_cols + 64 <= h->halo_stride && tun.lxf_from_dofs;
      const bool lxf_reads_avg = h->prm.flux_type == DFLO_FLUX_LXF && !h->af && !(h->basis == DFLO_BASIS_PK && h->geo == 0 && p.n_cells == p.n_owned);
      h->lazy_avg = tun.lazy_avg && !lxf_reads_avg && !pass && h->prm.global_time_step;
    }
    // (measured: the marks pay from k = 2 on -- C4 +10 %, the slab pair +3.5 % with the box test of round 3 --; at k = 1, where the
    //  pass reads only 16 values per cell, the box-only marks give the single engine +1.5 % (C3) and cost a part of a
    //  multi-device run 7 % (its rim / ring launches are short already): on there without ghost cells only;
    //  DFLO_LIM_MASK=1 forces them, 0 forbids them)
    //  ... and on where the limiter pass takes the exchange along (one launch over all shards again: the single engine's case, and
    //  the shards on a cut can only leave a list; C3 through the IPC transport against itself: 0.80 -> 0.91 of the plain engine)
    //  Round 6: with the list of marked shards shared by the rim + ring and the rest launches (the pass over "everything but the rim"
    //  walks it instead of sending one wavefront to each of 8 000 shards) the marks pay at k = 1 on a part as well -- C3 against itself,
    //  RCCL 0.787 -> 0.892 of the plain engine, delivering pack kernels 0.795 -> 0.89 (LAB R6.12): on wherever the list is.
    const bool want_marks = h->N >= 2 && (tun.lim_mask >= 0 ? tun.lim_mask != 0 : (h->N >= 3 || p.n_cells == p.n_owned || pass_takes_exchange || tun.lim_list));
    if (h->prm.limiter_type == DFLO_LIMITER_TVB && h->basis == DFLO_BASIS_QK && h->geo == 0 && want_marks) {
      const size_t nb = (size_t)std::max(p.n_shards, 1) * sizeof(unsigned long long);
      if (dmalloc((void **)&h->lim_mask, nb) != hipSuccess) {
        h->err = "hipMalloc(limiter marks) failed";
        return bail(DFLO_ERR_NOMEM);
      }
      hipMemset(h->lim_mask, 0, nb);
      if (tun.lim_list) {
        if (dmalloc((void **)&h->lim_cnt, 2 * sizeof(int)) != hipSuccess ||
            dmalloc((void **)&h->lim_list, (size_t)(p.n_shards + 8) * sizeof(ulonglong2)) != hipSuccess) {
          h->err = "hipMalloc(limiter list) failed";
          return bail(DFLO_ERR_NOMEM);
        }
        hipMemset(h->lim_cnt, 0, 2 * sizeof(int));
        hipMemset(h->lim_list, 0, (size_t)(p.n_shards + 8) * sizeof(ulonglong2));
        h->lim_grid = std::max(64, tun.lim_grid);
      }
    }
  }
  {
    const int rows = 4 * h->N * h->N + (h->prm.flux_type == DFLO_FLUX_LXF ? 3 : 0);  // nodal image (also for Pk)
    h->lds_bytes = ((size_t)rows * 65 + (size_t)4 * h->N * h->halo_stride + (h->prm.flux_type == DFLO_FLUX_LXF ? 3 * (size_t)p.halo_cols : 0) +
                    (size_t)p.max_bnd * 4 * h->N + (p.max_bnd + 2) / 2 + (h->geo == 1 ? 8 * 64 : 0)) * sizeof(double);
  }

  if (h->lds_bytes > 160 * 1024) { h->err = "shard halo too large for LDS"; return bail(DFLO_ERR_UNSUPPORTED); }
  if (h->lds_bytes > 64 * 1024) {
    for (int mode = 0; mode < 3; ++mode) {
      stage_fn fn = h->basis == DFLO_BASIS_PK ? pick_pk(h->N, h->prm.flux_type, mode, streams_out(h) | (h->geo << 1), h->mfma) : pick_stage(h->N, h->prm.flux_type, mode, h->geo, h->fuse_pos ? 1 : (h->lim_mask ? 2 : (h->af ? 3 : 0)), streams_out(h), h->mfma);
      if (hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes) != hipSuccess) {
        h->err = "cannot raise dynamic LDS limit";
        return bail(DFLO_ERR_HIP);
      }
    }
  }
  {  // persistent grid: as many workgroups as stay resident, a multiple of 8 (one run of shards per XCD)
    stage_fn fn = h->basis == DFLO_BASIS_PK ? pick_pk(h->N, h->prm.flux_type, 1, h->geo << 1, h->mfma) : pick_stage(h->N, h->prm.flux_type, 1, h->geo, 0, 0, h->mfma);
    int per_cu = 0, n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device_id);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)fn, 64 * h->N, h->lds_bytes) != hipSuccess || per_cu < 1)
      per_cu = 1;
    if (n_cu < 1) n_cu = 256;
    if (tun.verbose)
      std::fprintf(stderr, "dflo_hip: stage kernel N=%d: %zu bytes of LDS per workgroup, %d workgroups (%d wavefronts) resident per CU; %d shards, %d index patterns\n",
                   h->N, h->lds_bytes, per_cu, per_cu * h->N, h->plan.n_shards, h->n_patterns);
    h->stage_grid = grid_for(h->plan.n_shards);  // one workgroup per shard (per_cu of them resident per CU)
  }
  *out = h;
  return DFLO_OK;
}

extern "C" {

int dflo_hip_destroy(dflo_hip_handle h) {
  if (!h) return DFLO_OK;
  hipSetDevice(h->device);
  if (h->own_stream) hipStreamSynchronize(h->own_stream);
  drop_graph(h);
  for (int i = 0; i < 3; ++i) hipFree(h->U[i]);
  for (int i = 0; i < 2; ++i) { hipFree(h->avg[i]); hipFree(h->bval[i]); }
  hipFree(h->rhs); hipFree(h->user_buf); hipFree(h->bface_kind);
  hipFree(h->d_bc_ops); hipFree(h->d_bc_consts); hipFree(h->d_bc_prog); hipFree(h->d_bc_faces); hipFree(h->d_bc_pts); hipFree(h->d_bface_id); hipFree(h->d_bxy);
  hipFree(h->d_shard_count);
  hipFree(h->d_bnd_pad); hipFree(h->d_nbr_code); hipFree(h->d_shock); hipFree(h->lim_mask); hipFree(h->lim_cnt); hipFree(h->lim_list); hipFree(h->d_faces_pad); hipFree(h->d_shard_hdr); hipFree(h->d_halo_pad); hipFree(h->d_cell_face); hipFree(h->d_lrbt); hipFree(h->d_user_of); hipFree(h->d_iid);
  hipFree(h->d_rim_list); hipFree(h->d_int_list); hipFree(h->d_rim2_list); hipFree(h->d_rest2_list);
  hipFree(h->d_cell_h); hipFree(h->d_dt_cell); hipFree(h->d_cell_vert); hipFree(h->shard_res); hipFree(h->shard_dtmin); hipFree(h->res_sq); hipFree(h->fin_partial); hipFree(h->dt_dev);
  if (h->flags_host) hipHostFree((void *)h->flags_host);
  hipFree(h->fin_counter); hipFree(h->pos_stats); hipFree(h->send_done);
  if (!h->dt_external) hipFree(h->dt_mins);
  if (!h->tg_external) { hipFree(h->Tg[0]); hipFree(h->Tg[1]); }
  hipFree(h->d_wt_flag); hipFree(h->d_dla_begin); hipFree(h->d_dla_slot); hipFree(h->d_dla_dst[0]); hipFree(h->d_dla_dst[1]); hipFree(h->d_dla_flag); hipFree(h->d_wta_flag);
  hipFree(h->d_dl_begin); hipFree(h->d_dl_rec); hipFree(h->d_dl_dst[0]); hipFree(h->d_dl_dst[1]); hipFree(h->d_dl_flag);
  hipFree(h->d_gnb); hipFree(h->d_rim_ghost_list); hipFree(h->d_gt_begin);
  hipFree(h->d_gt_slot); hipFree(h->d_gt_face); hipFree(h->d_sendf_slot); hipFree(h->d_sendf_face); hipFree(h->d_send_slots); hipFree(h->ghost_stage);
  for (int i = 0; i < 2; ++i) if (h->ev_chunk[i]) hipEventDestroy(h->ev_chunk[i]);
  for (auto &e : h->ev_pool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
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
  h->steps_done = 0;
  h->bc_later_step = -1;
  h->ghost_avg_src = nullptr;
  HIPCHK(h, hipMemsetAsync(h->fin_counter + 1, 0, 3 * sizeof(int), h->stream));
  for (int i = 0; i < 4; ++i) h->flags_host[i] = 0;   // a new state: the flags of an earlier run are history
  hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->user_buf, h->U[0],
                     h->d_user_of, p.n_slots, h->ndof);
  HIPCHK(h, hipGetLastError());
  int rc = launch_average(h);
  if (rc) return rc;
  h->avg_valid = true;
  if (h->trace_halo) {   // the ghost cells' traces from the ghost cells' DoFs of the initial state, into both buffers
    for (int i = 0; i < 2; ++i)
      if ((rc = launch_face_traces(h, h->Tg[i], h->d_gt_slot, h->d_gt_face, h->n_gt))) return rc;
    h->tg_cur = 0;
  }
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
  {   // an intermediate stage that kept its averages to itself
    const int rc = ensure_avg(h);
    if (rc) return rc;
  }
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
  HIPCHK(h, hipMemcpyAsync(h->bval[which == 0 ? h->bv_first : h->bv_first ^ 1], values, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  // (an upload overwrites program-evaluated entries too: the next step evaluates both tables afresh)
  h->bval_host[which].assign(values, values + n);
  h->bval_equal = h->bval_host[0].size() == h->bval_host[1].size() &&
                  (h->bval_host[0].empty() || std::memcmp(h->bval_host[0].data(), h->bval_host[1].data(), n * sizeof(double)) == 0);
  h->bc_later_step = -1;
  return DFLO_OK;
}

int dflo_hip_get_boundary_values(dflo_hip_handle h, int which, double *values) {
  if (check_handle(h) || which < 0 || which > 1 || !values) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  const size_t n = h->plan.bface_cell.size() * h->N * 4;
  if (n == 0) return DFLO_OK;
  HIPCHK(h, hipMemcpyAsync(values, h->bval[which == 0 ? h->bv_first : h->bv_first ^ 1], n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
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
  h->bc_later_step = -1;
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
  const int rc = launch_average(h);
  if (!rc) h->avg_valid = true;
  return rc;
}

// compute_time_step() of the current state into the device-resident (dt, t): launches only, nothing comes back to the host
static int launch_compute_dt(dflo_hip_engine *h, double elapsed_time) {
  // "time step type = global" with cfl <= 0: the time step of the input file (src/claw.cc:455-460; the reference leaves
  // global_dt unset on this path, the engine advances the clock by the time step it uses)
  const int fixed = h->prm.global_time_step && h->prm.cfl <= 0.0;
  // per-shard minima from the stored cell averages (src/claw.cc:486-511)
  const Plan &p = h->plan;
  if (!fixed && h->geo == 0) {   // the last stage kept its averages to itself (lazy_avg)
    const int rc = ensure_avg(h);
    if (rc) return rc;
  }
  if (fixed) {
  } else if (h->geo == 0)
    hipLaunchKernelGGL(dt_kernel, dim3(p.n_shards), dim3(64), 0, h->stream, h->avg[h->avg_cur], h->d_cell_h, p.h,
                       p.uniform_h ? 1 : 0, h->d_shard_count, h->shard_dtmin, h->prm.cfl, h->degree, h->d_dt_cell);
  else
    launch_dt_q(h);
  HIPCHK(h, hipGetLastError());
  double tt[4] = {0, elapsed_time, 0, 0};
  h->bc_later_step = -1;   // the clock is set from outside: the later stages' table of the last step is not "this step's t" any more
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
  f.fixed_dt = fixed;
  f.partial = h->fin_partial;
  f.counter = h->fin_counter;
  f.step_par = (int)(h->steps_done & 1);
  dt_table_args(h, f);
  hipLaunchKernelGGL(finalize_kernel, dim3(fin_grid(f.n_shards)), dim3(256), 0, h->stream, f);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

int dflo_hip_compute_dt(dflo_hip_handle h, double elapsed_time, double *dt) {
  if (check_handle(h) || !dt) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  const int rc = launch_compute_dt(h, elapsed_time);
  if (rc) return rc;
  double tt[4];
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
  ++h->steps_done;
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
  ++h->steps_done;
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
  // first dt from the current cell averages -- formed and left on the device like every later one: the host does not wait
  // for it (the stage kernels and the boundary programs read the device's (dt, t)), it only waits at the end
  // stage-timing events nobody has asked for yet are read here, before anything of this call is launched (outside the work the
  // caller times), so that a caller who enables the timing and never queries it does not grow the pool without bound
  if (h->ev_used > 2048) time_collect(h);
  int rc = launch_compute_dt(h, *elapsed_time_inout);
  if (rc) return rc;
  struct Resident {   // (every way out of this function leaves the loop)
    dflo_hip_engine *h;
    ~Resident() { h->resident = false; }
  } resident_guard{h};
  h->resident = true;
  int s = 0;
  if (h->use_graph && !h->timing && h->cur == h->old) {
    // the (solution, average) buffer indices come back to where they started after 2 steps (avg_cur flips
    // n_rk times per step): capture those once, replay them for the bulk of the steps
    const int period = 2;
    if (n_steps >= 2 * period) {
      const int lim_par = h->lim_cnt ? (h->lim_epoch & 1) : 0;
      if (h->graph_exec && (h->graph_cur != h->cur || h->graph_avg != h->avg_cur || h->graph_stream != h->stream || h->graph_lim != lim_par ||
                            h->graph_par != (int)(h->steps_done & 1))) drop_graph(h);
      if (!h->graph_exec) {
        const int cur0 = h->cur, avg0 = h->avg_cur, par0 = (int)(h->steps_done & 1);
        hipGraph_t g = nullptr;
        bool ok = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
          for (int k = 0; k < period && !rc; ++k) {
            for (int rk = 0; rk < h->n_rk && !rc; ++rk) rc = launch_stage(h, rk, -1.0, nullptr, -1);
            h->old = h->cur;
            ++h->steps_done;
          }
          ok = hipStreamEndCapture(h->stream, &g) == hipSuccess && !rc && g;
          if (ok) ok = hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0) == hipSuccess;
          if (g) hipGraphDestroy(g);
          // capture records, it does not run: put the indices back
          h->cur = h->old = cur0;
          h->avg_cur = avg0;
          h->pending_rk = -1;
          h->steps_done -= period;
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
          h->graph_lim = lim_par;
          h->graph_par = par0;
        }
      }
      if (h->graph_exec) {
        if (h->lim_cnt && s + period <= n_steps) {   // the first marked launch of a replay finds its list counter at zero
          const int i = (h->lim_epoch + 1) & 1;
          if (!h->lim_clean[i]) {
            HIPCHK(h, hipMemsetAsync(h->lim_cnt + i, 0, sizeof(int), h->stream));
            h->lim_clean[i] = true;
          }
        }
        for (; s + period <= n_steps; s += period) {
          HIPCHK(h, hipGraphLaunch(h->graph_exec, h->stream));
          h->steps_done += period;
        }
      }
    }
  }
  // The reference stops inside the stage that fails (src/positivity.cc:28-38,160-169).  Here the launches run ahead of the
  // device, so the host looks at the failure flags (mapped host memory, no copy) once per chunk of kCheckEvery steps,
  // after waiting for the chunk before the previous one: the device never runs dry and a failed run stops within
  // three chunks instead of marching on NaNs to the end.
  constexpr int kCheckEvery = 32;
  for (int i = 0; i < 2; ++i)
    if (!h->ev_chunk[i]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_chunk[i], hipEventDisableTiming));
  int chunk = 0;
  bool failed = false;
  for (int s0 = s; s < n_steps; ++s) {
    if ((s - s0) % kCheckEvery == 0 && s > s0) {
      HIPCHK(h, hipEventRecord(h->ev_chunk[chunk & 1], h->stream));
      ++chunk;
      if (chunk >= 2) {
        HIPCHK(h, hipEventSynchronize(h->ev_chunk[chunk & 1]));
        if (h->flags_host[0] | h->flags_host[1]) { failed = true; break; }
      }
    }
    for (int rk = 0; rk < h->n_rk; ++rk) {
      rc = launch_stage(h, rk, -1.0, nullptr, -1);
      if (rc) return rc;
    }
    h->old = h->cur;
    ++h->steps_done;
  }
  (void)failed;
  double tt[4];
  HIPCHK(h, hipMemcpyAsync(tt, h->dt_dev, sizeof(tt), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  // (the stage-timing events of this call are read when dflo_hip_stage_timing asks for them, not here inside the caller's clock)
  *elapsed_time_inout = tt[1];
  return flags_status(h);
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
  if (check_handle(h) || part < 0 || part > 4 || h->pending_rk < 0) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  return launch_update(h, nullptr, part);
}

int dflo_hip_set_deliver(dflo_hip_handle h, int area, int n_segments, const int32_t *first, void *const *dst, void *const *flags) {
  if (check_handle(h) || area < 0 || area > 1 || n_segments < 1 || n_segments > kMaxSegs || !first || !dst || !flags) return DFLO_ERR_BAD_PARAM;
  if (!h->trace_halo || h->basis != DFLO_BASIS_QK) { h->err = "delivery by the stage kernel needs ghost cells known by their traces (Qk)"; return DFLO_ERR_BAD_PARAM; }
  const int n = h->n_send_faces;
  if (n == 0 || first[0] != 0 || first[n_segments] != n || (int)h->h_sendf_slot.size() != n) { h->err = "set_deliver: the segments must cover the send list of set_send_faces"; return DFLO_ERR_COMM; }
  hipSetDevice(h->device);
  // the records sorted by the shard of their cell (stable: send-list order inside a shard)
  const int ns = h->plan.n_shards;
  std::vector<int32_t> begin(ns + 1, 0), order(n);
  for (int k = 0; k < n; ++k) ++begin[(h->h_sendf_slot[k] >> 6) + 1];
  int total = 0;
  for (int sh = 0; sh < ns; ++sh) { total += begin[sh + 1] > 0; begin[sh + 1] += begin[sh]; }
  {
    std::vector<int32_t> at(begin.begin(), begin.end() - 1);
    for (int k = 0; k < n; ++k) order[at[h->h_sendf_slot[k] >> 6]++] = k;
  }
  std::vector<int2> rec(n);
  std::vector<double *> to(n);
  const size_t w = (size_t)4 * h->N;
  for (int j = 0; j < n; ++j) {
    const int k = order[j];
    rec[j] = int2{h->h_sendf_slot[k], h->h_sendf_face[k]};
    int i = 0;
    while (i + 1 < n_segments && k >= first[i + 1]) ++i;
    to[j] = (double *)dst[i] + (size_t)(k - first[i]) * w;
  }
  std::vector<unsigned long long *> fl(n_segments);
  for (int i = 0; i < n_segments; ++i) fl[i] = (unsigned long long *)flags[i];
  HIPCHK(h, hipStreamSynchronize(h->stream));
  // the per-shard record tables follow the send list as it is NOW (a caller may have changed it since the other area was set: ADVICE r5)
  hipFree(h->d_dl_begin); h->d_dl_begin = nullptr;
  hipFree(h->d_dl_rec); h->d_dl_rec = nullptr;
  {
    int rc;
    if ((rc = upload(h, &h->d_dl_begin, begin)) || (rc = upload(h, &h->d_dl_rec, rec))) return rc;
  }
  hipFree(h->d_dl_dst[area]);
  h->d_dl_dst[area] = nullptr;
  hipFree(h->d_dl_flag);
  h->d_dl_flag = nullptr;
  int rc;
  if ((rc = upload(h, &h->d_dl_dst[area], to)) || (rc = upload(h, &h->d_dl_flag, fl))) return rc;
  h->dl_nflag = n_segments;
  h->dl_total = total;
  return DFLO_OK;
}

int dflo_hip_set_arrival_words(dflo_hip_handle h, int n, void *const *words, void *fail) {
  if (check_handle(h) || n < 0 || n > kMaxSegs || (n > 0 && (!words || !fail))) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  hipFree(h->d_wt_flag);
  h->d_wt_flag = nullptr;
  h->wt_n = n;
  h->wt_fail = (int *)fail;
  if (n == 0) return DFLO_OK;
  std::vector<unsigned long long *> w(n);
  for (int i = 0; i < n; ++i) w[i] = (unsigned long long *)words[i];
  return upload(h, &h->d_wt_flag, w);
}

int dflo_hip_stage_await(dflo_hip_handle h, uint64_t seq) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  if (!h->d_wt_flag || !(h->d_dl_begin || h->d_dla_begin)) { h->err = "stage_await: dflo_hip_set_deliver / dflo_hip_set_arrival_words have not been called"; return DFLO_ERR_BAD_PARAM; }
  h->wt_armed = true;
  h->wt_seq = seq;
  return DFLO_OK;
}

int dflo_hip_stage_tail_wait(dflo_hip_handle h, const void *word, uint64_t seq) {
  if (check_handle(h) || !word) return DFLO_ERR_BAD_PARAM;
  if (h->basis != DFLO_BASIS_QK) { h->err = "stage_tail_wait: Qk stage kernels only"; return DFLO_ERR_UNSUPPORTED; }
  h->tail_word = (const unsigned long long *)word;
  h->tail_seq = seq;
  return DFLO_OK;
}

int dflo_hip_set_deliver_averages(dflo_hip_handle h, int area, int n_segments, const int32_t *first, void *const *dst, void *const *flags,
                                  int n_words, void *const *words, void *fail) {
  if (check_handle(h) || area < 0 || area > 1 || n_segments < 1 || n_segments > kMaxSegs || !first || !dst || !flags || n_words < 0 || n_words > kMaxSegs ||
      (n_words > 0 && (!words || !fail)))
    return DFLO_ERR_BAD_PARAM;
  const int n = h->n_send;
  if (n == 0 || first[0] != 0 || first[n_segments] != n) { h->err = "set_deliver_averages: the segments must cover the send list of set_send_cells"; return DFLO_ERR_COMM; }
  hipSetDevice(h->device);
  const int ns = h->plan.n_shards;
  std::vector<int32_t> begin(ns + 1, 0), order(n), slot(n);
  for (int k = 0; k < n; ++k) ++begin[(h->h_send_slots[k] >> 6) + 1];
  int total = 0;
  for (int sh = 0; sh < ns; ++sh) { total += begin[sh + 1] > 0; begin[sh + 1] += begin[sh]; }
  {
    std::vector<int32_t> at(begin.begin(), begin.end() - 1);
    for (int k = 0; k < n; ++k) order[at[h->h_send_slots[k] >> 6]++] = k;
  }
  std::vector<double *> to(n);
  for (int j = 0; j < n; ++j) {
    const int k = order[j];
    slot[j] = h->h_send_slots[k];
    int i = 0;
    while (i + 1 < n_segments && k >= first[i + 1]) ++i;
    to[j] = (double *)dst[i] + (size_t)(k - first[i]) * 4;
  }
  std::vector<unsigned long long *> fl(n_segments), wd(std::max(n_words, 1), nullptr);
  for (int i = 0; i < n_segments; ++i) fl[i] = (unsigned long long *)flags[i];
  for (int i = 0; i < n_words; ++i) wd[i] = (unsigned long long *)words[i];
  HIPCHK(h, hipStreamSynchronize(h->stream));
  int rc;
  hipFree(h->d_dla_begin); h->d_dla_begin = nullptr;   // (as in set_deliver: the tables of the send list as it is now)
  hipFree(h->d_dla_slot); h->d_dla_slot = nullptr;
  if ((rc = upload(h, &h->d_dla_begin, begin)) || (rc = upload(h, &h->d_dla_slot, slot))) return rc;
  hipFree(h->d_dla_dst[area]); h->d_dla_dst[area] = nullptr;
  hipFree(h->d_dla_flag); h->d_dla_flag = nullptr;
  hipFree(h->d_wta_flag); h->d_wta_flag = nullptr;
  if ((rc = upload(h, &h->d_dla_dst[area], to)) || (rc = upload(h, &h->d_dla_flag, fl)) || (rc = upload(h, &h->d_wta_flag, wd))) return rc;
  h->dla_nflag = n_segments;
  h->dla_total = total;
  h->wta_n = n_words;
  if (fail) h->wt_fail = (int *)fail;
  return DFLO_OK;
}

int dflo_hip_stage_deliver_averages(dflo_hip_handle h, int area, uint64_t seq) {
  if (check_handle(h) || area < 0 || area > 1) return DFLO_ERR_BAD_PARAM;
  if (!h->d_dla_dst[area]) { h->err = "stage_deliver_averages: dflo_hip_set_deliver_averages has not been called for this receive area"; return DFLO_ERR_BAD_PARAM; }
  h->dla_armed = area;
  h->dla_seq = seq;
  return DFLO_OK;
}

int dflo_hip_limit_exchange(dflo_hip_handle h, int trace_area, uint64_t trace_seq, uint64_t average_seq, int poll_in_kernel) {
  if (check_handle(h) || trace_area < 0 || trace_area > 1) return DFLO_ERR_BAD_PARAM;
  if (!h->d_dl_dst[trace_area] || !h->d_wta_flag || !h->lim_list) {
    h->err = "limit_exchange: needs dflo_hip_set_deliver, dflo_hip_set_deliver_averages and a limiter pass that walks the list of marked shards";
    return DFLO_ERR_BAD_PARAM;
  }
  h->lim_x_area = trace_area;
  h->lim_x_seq = trace_seq;
  h->lim_x_await = average_seq;
  h->lim_x_poll = poll_in_kernel != 0;
  return DFLO_OK;
}

int dflo_hip_deliver_to_plain_memory(dflo_hip_handle h, int plain) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  h->dl_fence = plain ? 1 : 0;
  return DFLO_OK;
}

int dflo_hip_limiter_walks_list(dflo_hip_handle h) { return (h && h->lim_list) ? 1 : 0; }

int dflo_hip_stage_deliver(dflo_hip_handle h, int area, uint64_t seq) {
  if (check_handle(h) || area < 0 || area > 1) return DFLO_ERR_BAD_PARAM;
  if (!h->d_dl_dst[area]) { h->err = "stage_deliver: dflo_hip_set_deliver has not been called for this receive area"; return DFLO_ERR_BAD_PARAM; }
  h->dl_armed = area;
  h->dl_seq = seq;
  return DFLO_OK;
}

int dflo_hip_attach_event(dflo_hip_handle h, void *event) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  h->next_stop = (hipEvent_t)event;
  return DFLO_OK;
}

int dflo_hip_stage_limit_part(dflo_hip_handle h, int part) {
  if (check_handle(h) || part < 0 || part > 4) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  // The pass over everything but the rim (2) is the last limiter launch of a stage in the multi-device schedule; behind the last
  // stage of a TVB run on squares it carries the step's reductions like the single engine's pass (launch_limit_finalize).  The
  // caller has ordered it behind the rim's UPDATE (it reads the rim cells' new averages), which is all the reductions need of
  // the rim: the limiter changes neither averages nor residuals.
  const bool fuse = part == 2 && h->fuse_fin && h->pending_rk == h->n_rk - 1 && h->geo == 0 && !h->d_dt_cell && h->basis == DFLO_BASIS_QK &&
                    h->prm.limiter_type == DFLO_LIMITER_TVB && !h->fuse_pos && !h->plan.interior_shards.empty() && !h->d_shock;
  if (!fuse) return launch_stage_limiter(h, part);
  FinalArgs f{};
  final_args(h, f);
  const int rc = launch_stage_limiter(h, part, &f);
  if (!rc) h->fin_done = true;
  return rc;
}

int dflo_hip_stage_finish(dflo_hip_handle h) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  return launch_finish(h);
}

int dflo_hip_finish_enqueued(dflo_hip_handle h) { return (h && h->finish_enqueued) ? 1 : 0; }
int dflo_hip_n_rim_shards(dflo_hip_handle h) { return h ? (int)h->plan.rim_shards.size() : 0; }
int dflo_hip_n_part_shards(dflo_hip_handle h, int part) {
  if (!h || part < 0 || part > 4) return 0;
  const int32_t *list = nullptr;
  int n = 0;
  part_list(h, part, &list, &n);
  return n;
}

int dflo_hip_stage_limit(dflo_hip_handle h) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  return launch_limit_finalize(h);
}

int dflo_hip_check(dflo_hip_handle h) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return flags_status(h);
}

int dflo_hip_positivity_stats(dflo_hip_handle h, int64_t *counts, int reset) {
  if (check_handle(h) || !counts) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  unsigned long long v[2] = {0, 0};
  HIPCHK(h, hipMemcpyAsync(v, h->pos_stats, sizeof(v), hipMemcpyDeviceToHost, h->stream));
  if (reset) HIPCHK(h, hipMemsetAsync(h->pos_stats, 0, sizeof(v), h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  counts[0] = (int64_t)v[0];
  counts[1] = (int64_t)v[1];
  return DFLO_OK;
}

int dflo_hip_failure_step(dflo_hip_handle h, int64_t *step) {
  if (check_handle(h) || !step) return DFLO_ERR_BAD_PARAM;
  *step = h->flags_host[2] ? (int64_t)h->flags_host[2] - 1 : -1;
  return DFLO_OK;
}

int dflo_hip_synchronize(dflo_hip_handle h) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return DFLO_OK;
}

int dflo_hip_uses_mfma(dflo_hip_handle h) { return h && h->mfma ? 1 : 0; }

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
  // enable > 1: sample every enable-th stage (coprime to 2 and 3 keeps the stages of a step equally represented); 1: every fifth.
  // A sampled launch sits between two timed event records and costs its stream ~7 us of bubbles: a long run samples sparsely.
  h->t_every = enable > 1 ? enable : 5;
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
  HIPCHK(h, hipStreamSynchronize(h->stream));   // (as in set_send_faces: set_deliver_averages again for both areas)
  for (int a = 0; a < 2; ++a) { hipFree(h->d_dla_dst[a]); h->d_dla_dst[a] = nullptr; }
  hipFree(h->d_dla_begin); h->d_dla_begin = nullptr;
  hipFree(h->d_dla_slot); h->d_dla_slot = nullptr;
  h->dla_armed = -1;
  h->n_send = n;
  h->h_send_slots = slots;
  return upload(h, &h->d_send_slots, slots);
}

int dflo_hip_pack_send(dflo_hip_handle h, void *device_buffer) {
  if (check_handle(h) || !device_buffer) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  if (h->n_send == 0) return DFLO_OK;
  const long long tot = (long long)h->n_send * h->ndof;
  unsigned long long ps = 0, *pw = take_pub(h, &ps);
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, (double *)device_buffer,
                     h->U[h->cur], h->d_send_slots, h->n_send, h->ndof, pw, ps);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

int dflo_hip_pack_send_avg(dflo_hip_handle h, void *device_buffer) {
  if (check_handle(h) || !device_buffer) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  if (h->n_send == 0) return DFLO_OK;
  const long long tot = (long long)h->n_send * 4;
  unsigned long long ps = 0, *pw = take_pub(h, &ps);
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, (double *)device_buffer,
                     h->avg[h->avg_cur], h->d_send_slots, h->n_send, 4, pw, ps);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

int dflo_hip_pack_send_cells(dflo_hip_handle h, void *device_buffer) {
  if (check_handle(h) || !device_buffer) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  if (h->n_send == 0) return DFLO_OK;
  const long long tot = (long long)h->n_send * (h->ndof + 4);
  unsigned long long ps = 0, *pw = take_pub(h, &ps);
  hipLaunchKernelGGL(pack_cells_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, (double *)device_buffer,
                     h->U[h->cur], h->avg[h->avg_cur], h->d_send_slots, h->n_send, h->ndof, pw, ps);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

int dflo_hip_pack_send_cells_unlimited(dflo_hip_handle h, void *device_buffer) {
  if (check_handle(h) || !device_buffer) return DFLO_ERR_BAD_PARAM;
  if (h->basis != DFLO_BASIS_QK) { h->err = "pack_send_cells_unlimited: Qk only"; return DFLO_ERR_UNSUPPORTED; }
  hipSetDevice(h->device);
  if (h->n_send == 0) return DFLO_OK;
  const long long tot = (long long)h->n_send * (h->ndof + kFatExtra);
  unsigned long long ps = 0, *pw = take_pub(h, &ps);
  hipLaunchKernelGGL(pack_fat_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, (double *)device_buffer,
                     (const double *)h->U[h->cur], (const double *)h->avg[h->avg_cur], (const int32_t *)h->d_lrbt, (const int32_t *)h->d_send_slots,
                     h->n_send, h->ndof, pw, ps);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

int dflo_hip_pack_publish(dflo_hip_handle h, void *word, uint64_t seq) {
  if (check_handle(h) || !word) return DFLO_ERR_BAD_PARAM;
  h->pub_word = (unsigned long long *)word;
  h->pub_seq = seq;
  return DFLO_OK;
}

// The receiving side of the one-exchange TVB stage: the neighbours' cut cells as pack_send_cells_unlimited left them.  The NEXT
// dflo_hip_stage_limit_part(h, 1) -- the limiter pass over the rim shards -- takes them along: an unpack kernel puts the records
// into the ghost shards, the averages' array and d_gnb; the pass then runs over the rim shards AND the ghost shards (the routine
// that limits the owned cells, with a neighbour's average from this part's array where it is an owned cell and from the record
// where it lives with the ghost's owner: the owner's inputs, arithmetic and bits) and the wavefronts of the ghost shards form the
// traces of their limited cells into trace table `table` (launch_limiter).
int dflo_hip_limit_ghost_cells(dflo_hip_handle h, const void *records, int table) {
  if (check_handle(h) || table < 0 || table > 1) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  const Plan &p = h->plan;
  const int n_ghost = p.n_cells - p.n_owned;
  if (n_ghost == 0) return DFLO_OK;
  if (!records) return DFLO_ERR_BAD_PARAM;
  if (!h->trace_halo || h->basis != DFLO_BASIS_QK || h->prm.limiter_type != DFLO_LIMITER_TVB || h->d_shock) {
    h->err = "limit_ghost_cells: needs ghost cells known by their traces (Qk) and a TVB limiter without the KXRCF indicator";
    return DFLO_ERR_UNSUPPORTED;
  }
  if (!h->d_gnb && dmalloc((void **)&h->d_gnb, (size_t)n_ghost * 16 * sizeof(double)) != hipSuccess) { h->err = "hipMalloc(ghost neighbours) failed"; return DFLO_ERR_NOMEM; }
  h->ghost_ride_rec = (const double *)records;
  h->ghost_ride_table = table;
  return DFLO_OK;
}

int dflo_hip_unpack_ghost_cells(dflo_hip_handle h, const void *device_buffer) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  const Plan &p = h->plan;
  const int n_ghost = p.n_cells - p.n_owned;
  if (n_ghost == 0) return DFLO_OK;
  if (!device_buffer) return DFLO_ERR_BAD_PARAM;
  const long long tot = (long long)n_ghost * (h->ndof + 4);
  hipLaunchKernelGGL(unpack_cells_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, (const double *)device_buffer,
                     h->U[h->cur], h->avg[h->avg_cur], p.n_shards * 64, n_ghost, h->ndof);
  HIPCHK(h, hipGetLastError());
  if (h->trace_halo) return launch_face_traces(h, h->Tg[h->tg_cur], h->d_gt_slot, h->d_gt_face, h->n_gt);
  return DFLO_OK;
}

int dflo_hip_halo_traces(dflo_hip_handle h) { return (h && h->trace_halo) ? 1 : 0; }
int dflo_hip_n_ghost_traces(dflo_hip_handle h) { return h ? h->n_gt : 0; }

int dflo_hip_set_send_faces(dflo_hip_handle h, int32_t n, const int32_t *cells, const int32_t *faces) {
  if (check_handle(h) || n < 0 || (n > 0 && (!cells || !faces))) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  std::vector<int32_t> slots(n), ff(n);
  for (int i = 0; i < n; ++i) {
    if (cells[i] < 0 || cells[i] >= h->plan.n_owned || faces[i] < 0 || faces[i] > 3) { h->err = "send face is not a face of an owned cell"; return DFLO_ERR_COMM; }
    slots[i] = h->plan.iid[cells[i]];
    ff[i] = faces[i];
  }
  hipFree(h->d_sendf_slot); hipFree(h->d_sendf_face);
  h->d_sendf_slot = h->d_sendf_face = nullptr;
  // a delivery arrangement made for the list before is void: set_deliver has to be called again for both receive areas
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int a = 0; a < 2; ++a) { hipFree(h->d_dl_dst[a]); h->d_dl_dst[a] = nullptr; }
  hipFree(h->d_dl_begin); h->d_dl_begin = nullptr;
  hipFree(h->d_dl_rec); h->d_dl_rec = nullptr;
  h->dl_armed = -1;
  h->n_send_faces = n;
  h->h_sendf_slot = slots;
  h->h_sendf_face = ff;
  int rc = upload(h, &h->d_sendf_slot, slots);
  return rc ? rc : upload(h, &h->d_sendf_face, ff);
}

int dflo_hip_pack_send_traces(dflo_hip_handle h, void *device_buffer) {
  if (check_handle(h) || !device_buffer) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  return launch_face_traces(h, (double *)device_buffer, h->d_sendf_slot, h->d_sendf_face, h->n_send_faces, true);
}

int dflo_hip_pack_send_to(dflo_hip_handle h, int kind, int n_segments, const int32_t *first, void *const *dst) {
  return dflo_hip_pack_send_to_signal(h, kind, n_segments, first, dst, nullptr, 0);
}

int dflo_hip_pack_send_to_signal(dflo_hip_handle h, int kind, int n_segments, const int32_t *first, void *const *dst, void *const *flags,
                                 uint64_t seq) {
  if (check_handle(h) || kind < 0 || kind > 3 || n_segments < 0 || n_segments > kMaxSegs || (n_segments > 0 && (!first || !dst))) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  const int n = kind == 2 ? h->n_send_faces : h->n_send;
  if (n == 0 || n_segments == 0) return DFLO_OK;
  SendSegs seg{};
  seg.n = n_segments;
  for (int i = 0; i < n_segments; ++i) {
    seg.first[i] = first[i];
    seg.dst[i] = (double *)dst[i];
    seg.flag[i] = flags ? (unsigned long long *)flags[i] : nullptr;
  }
  seg.seq = seq;
  seg.pub = take_pub(h, &seg.pub_seq);
  seg.done = flags ? h->send_done + (kind == 3 ? 0 : kind) : nullptr;   // (kind 3 takes the place of kind 0: never both in one run)
  seg.first[n_segments] = first[n_segments];
  if (first[0] != 0 || first[n_segments] != n) { h->err = "pack_send_to: the segments must cover the send list"; return DFLO_ERR_COMM; }
  if (kind == 2) {
    const long long tot = (long long)n * 4 * h->N;
    auto fn = DFLO_BY_N(h->N, face_trace_to_kernel);
    hipLaunchKernelGGL(fn, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, seg, (const double *)h->U[h->cur],
                       (const int32_t *)h->d_sendf_slot, (const int32_t *)h->d_sendf_face, n);
  } else if (kind == 3) {   // unlimited cells with their neighbours' averages (dflo_hip_pack_send_cells_unlimited)
    const long long tot = (long long)n * (h->ndof + kFatExtra);
    hipLaunchKernelGGL(pack_fat_to_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, seg, (const double *)h->U[h->cur],
                       (const double *)h->avg[h->avg_cur], (const int32_t *)h->d_lrbt, (const int32_t *)h->d_send_slots, n, h->ndof);
  } else {
    const int w = kind == 0 ? h->ndof + 4 : 4;
    const long long tot = (long long)n * w;
    hipLaunchKernelGGL(pack_to_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, seg, (const double *)h->U[h->cur],
                       (const double *)h->avg[h->avg_cur], (const int32_t *)h->d_send_slots, n, h->ndof, kind == 0 ? 1 : 0);
  }
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

int dflo_hip_ghost_trace_buffer(dflo_hip_handle h, int which, void **ptr) {
  if (check_handle(h) || which < 0 || which > 1 || !ptr) return DFLO_ERR_BAD_PARAM;
  *ptr = h->Tg[which];
  return DFLO_OK;
}

int dflo_hip_set_ghost_trace_buffers(dflo_hip_handle h, void *table0, void *table1) {
  if (check_handle(h) || !table0 || !table1 || table0 == table1) return DFLO_ERR_BAD_PARAM;
  if (!h->trace_halo) { h->err = "this engine does not read ghost cells by their traces"; return DFLO_ERR_BAD_PARAM; }
  hipSetDevice(h->device);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const size_t bytes = (size_t)h->n_gt * 4 * h->N * sizeof(double);
  void *nt[2] = {table0, table1};
  for (int i = 0; i < 2; ++i) {   // what the tables hold now moves along (after set_solution: the initial state's traces)
    HIPCHK(h, hipMemcpy(nt[i], h->Tg[i], bytes, hipMemcpyDeviceToDevice));
    if (!h->tg_external) hipFree(h->Tg[i]);
    h->Tg[i] = (double *)nt[i];
  }
  h->tg_external = true;
  return DFLO_OK;
}

int dflo_hip_set_dt_table_buffer(dflo_hip_handle h, void *table) {
  if (check_handle(h) || !table) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(table, h->dt_mins, 2 * kDtSlots * sizeof(double), hipMemcpyDeviceToDevice));
  if (!h->dt_external) hipFree(h->dt_mins);
  h->dt_mins = (double *)table;
  h->dt_external = true;
  drop_graph(h);
  return DFLO_OK;
}

int dflo_hip_use_ghost_traces(dflo_hip_handle h, int which) {
  if (check_handle(h) || which < 0 || which > 1) return DFLO_ERR_BAD_PARAM;
  h->tg_cur = which;
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
  // whole cells handed to an engine that reads ghost cells by their traces: bring the current trace table up to date
  if (h->trace_halo) return launch_face_traces(h, h->Tg[h->tg_cur], h->d_gt_slot, h->d_gt_face, h->n_gt);
  return DFLO_OK;
}

int dflo_hip_unpack_ghost_avg(dflo_hip_handle h, const void *device_buffer) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  hipSetDevice(h->device);
  const Plan &p = h->plan;
  const int n_ghost = p.n_cells - p.n_owned;
  if (n_ghost == 0) return DFLO_OK;
  if (!device_buffer) return DFLO_ERR_BAD_PARAM;
  h->ghost_avg_src = nullptr;   // the averages are in the array again
  hipLaunchKernelGGL(unpack_ghost_avg_kernel, dim3((n_ghost + 63) / 64), dim3(64), 0, h->stream,
                     (const double *)device_buffer, h->avg[h->avg_cur], p.n_shards * 64, n_ghost);
  HIPCHK(h, hipGetLastError());
  return DFLO_OK;
}

int dflo_hip_ghost_avg_source(dflo_hip_handle h, const void *device_buffer) {
  if (check_handle(h)) return DFLO_ERR_BAD_PARAM;
  h->ghost_avg_src = (const double *)device_buffer;
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

int dflo_hip_debug_exp(int n, const double *x, double *exp_library, double *exp_flux) {
  double *dx = nullptr, *dl = nullptr, *df = nullptr;
  if (hipMalloc((void **)&dx, n * sizeof(double)) != hipSuccess || hipMalloc((void **)&dl, n * sizeof(double)) != hipSuccess ||
      hipMalloc((void **)&df, n * sizeof(double)) != hipSuccess)
    return DFLO_ERR_HIP;
  hipMemcpy(dx, x, n * sizeof(double), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(debug_exp_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dl, df, n);
  hipMemcpy(exp_library, dl, n * sizeof(double), hipMemcpyDeviceToHost);
  hipMemcpy(exp_flux, df, n * sizeof(double), hipMemcpyDeviceToHost);
  hipFree(dx); hipFree(dl); hipFree(df);
  return hipGetLastError() == hipSuccess ? DFLO_OK : DFLO_ERR_HIP;
}

int dflo_hip_dt_table(dflo_hip_handle h, void **base) {
  if (check_handle(h) || !base) return DFLO_ERR_BAD_PARAM;
  *base = h->dt_mins;
  return DFLO_OK;
}

int dflo_hip_dt_exchange(dflo_hip_handle h, int my_slot, int n_slots, void *const *peer_tables) {
  if (check_handle(h) || n_slots < 0 || n_slots > kDtSlots || (n_slots > 0 && (my_slot < 0 || my_slot >= n_slots))) return DFLO_ERR_BAD_PARAM;
  h->dt_n = n_slots;
  h->dt_my = n_slots > 0 ? my_slot : 0;
  for (int q = 0; q < kDtSlots; ++q) h->dt_peer[q] = (peer_tables && q < n_slots && q != my_slot) ? (double *)peer_tables[q] : nullptr;
  return DFLO_OK;
}

int dflo_hip_dt_slot(dflo_hip_handle h, void **slot) {
  if (check_handle(h) || !slot) return DFLO_ERR_BAD_PARAM;
  *slot = h->dt_mins + (size_t)(h->steps_done & 1) * kDtSlots + h->dt_my;
  return DFLO_OK;
}

}  // extern "C"
