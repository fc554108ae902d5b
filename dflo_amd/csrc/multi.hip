// multi.hip -- several MI355X behind one handle: the native multi-device driver of include/dflo_hip.h.
//
// Replaces what the MPI variant of dflo gets from deal.II + MPI (src_mpi/):
//   parallel::distributed::Triangulation, locally owned + ghost cells     src_mpi/claw.h:220      -> dflo_mesh_partition_ex
//   current_solution.update_ghost_values() after update and after limiter  src_mpi/claw.cc:793,
//                                                                          src_mpi/limiter.cc:232  -> pack / transport / unpack
//   Utilities::MPI::min(global_dt)                                         src_mpi/claw.cc:579     -> peer reads | ncclAllReduce(min)
//   right_hand_side.l2_norm()                                              src_mpi/claw.cc:777     -> sum of the per-part squares
//   right_hand_side.compress(add)                                          src_mpi/assemble_explicit.cc:580 -> not needed (faces on
//        a cut are integrated by both owners with the same integrating side, bit-identical flux)
//
// Two ways to own the devices, one stage schedule:
//   dflo_hip_multi_create       one process, n_devices engines.  A stream pair (compute M, comm C) and a host thread per GROUP
//                               of parts -- a part with a device of its own is a group, parts that share a device form at most
//                               two (struct Group below): a single thread cannot feed more than two or three devices (~20 API
//                               calls per part and stage against a stage of ~0.2 ms; DFLO_MULTI_THREADS=0 restores the
//                               single-threaded driver).  The pack kernels deliver: they write the halo records straight into
//                               the peers' receive areas (xGMI peer access); the time-step minimum is read from the peers'
//                               device slots (no host hop).  Where one group's stream waits for a peer's event, the host
//                               threads are sequenced by counters (an event has to be recorded before it can be waited for).
//   dflo_hip_multi_create_rank  one process per GPU (what torchrun / mpirun start); halos move with grouped
//                               ncclSend/ncclRecv, the time step with an 8-byte ncclAllReduce(min), all on the comm stream.
//                               RCCL is loaded with dlopen the first time it is needed: single-GPU users do not need it.
// One RK stage of a group (stage_phase below), the rim shards on C next to the interior shards on M:
//   C: [wait: M's previous stage, the step's time step]  rim shards (those that read ghost cells; TVB: + the ring beside them,
//      the averages leave, the neighbours' averages arrive, the rim is limited)  -> ev_rim;  the traces of the cut faces leave
//   M: [wait: ev_rim of the previous stage]  interior shards, their limiter, (last stage: wait ev_rim) the stage's reductions
//   C: the neighbours' traces arrive in the table the next stage's rim kernel reads
// This file is host code over the public halo seam of the engine (dflo_hip_stage_open / _update_part / _limit_part /
// _finish / _pack_send* / _unpack_ghost* ...): a dflo maintainer with another transport can write the same against the header.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and enums only: the functions are resolved with dlsym

#include <algorithm>
#include "plan.h"
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "abi.h"
#include "basis.h"
#include "tunables.h"

namespace {

std::string g_multi_error;

// ---------------------------------------------------------------- RCCL through dlopen
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;   // optional
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;      // optional (reporting only)
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;   // optional
};
Rccl g_rccl;

bool load_rccl(std::string &err) {
  if (g_rccl.lib) return true;
  // The RCCL that belongs to the HIP runtime this process runs on: first the librccl next to the loaded libamdhip64
  // (a process may hold a second ROCm, e.g. the one bundled with PyTorch, whose RCCL talks to its own, uninitialised
  // copy of the HSA runtime), then the loader's search path.
  std::vector<std::string> names;
  Dl_info info;
  if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    if (slash != std::string::npos) {
      dir.resize(slash + 1);
      names.push_back(dir + "librccl.so.1");
      names.push_back(dir + "librccl.so");
    }
  }
  names.push_back("librccl.so.1");
  names.push_back("librccl.so");
  names.push_back("/opt/rocm/lib/librccl.so.1");
  void *lib = nullptr;
  for (const std::string &n : names)
    if ((lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL))) break;
  if (!lib) { err = std::string("cannot load RCCL (librccl.so): ") + dlerror(); return false; }
#define DFLO_SYM(field, name)                                                            \
  do {                                                                                   \
    *(void **)(&g_rccl.field) = dlsym(lib, name);                                        \
    if (!g_rccl.field) { err = std::string("RCCL has no symbol ") + name; dlclose(lib); return false; } \
  } while (0)
  DFLO_SYM(GetUniqueId, "ncclGetUniqueId");
  DFLO_SYM(CommInitRank, "ncclCommInitRank");
  DFLO_SYM(CommDestroy, "ncclCommDestroy");
  DFLO_SYM(GroupStart, "ncclGroupStart");
  DFLO_SYM(GroupEnd, "ncclGroupEnd");
  DFLO_SYM(Send, "ncclSend");
  DFLO_SYM(Recv, "ncclRecv");
  DFLO_SYM(AllReduce, "ncclAllReduce");
  DFLO_SYM(GetErrorString, "ncclGetErrorString");
#undef DFLO_SYM
  *(void **)(&g_rccl.CommAbort) = dlsym(lib, "ncclCommAbort");
  *(void **)(&g_rccl.CommCount) = dlsym(lib, "ncclCommCount");
  *(void **)(&g_rccl.CommUserRank) = dlsym(lib, "ncclCommUserRank");
  g_rccl.lib = lib;
  return true;
}

enum Chan { CH_CELLS = 0, CH_AVG = 1, CH_TRACES = 2, CH_FIN = 3 };

// ---------------------------------------------------------------- one process per GPU without a transport library on the path
// DFLO_RANK_TRANSPORT=ipc: every rank maps its neighbours' receive areas (ghost-trace tables, average / cell areas), every rank's
// table of time-step minima and every rank's block of sequence words through hipIpcGetMemHandle / hipIpcOpenMemHandle once, at
// create (the handles travel through the communicator or the host program's all-reduce: a bootstrap, nothing more).  Per stage
// the pack kernel stores the records into the neighbour's area and then the exchange's number into the neighbour's sequence
// word for (kind, this rank); the receiver's comm stream runs a one-wavefront kernel that polls its words.  No RCCL kernel (which
// beside a full-size interior launch takes as long as that launch, profiles/r05), no rendezvous, no host hop.
constexpr int kFlagWords = 4 * 16;   // [kind][source rank]
__host__ __device__ inline int flag_index(int kind, int src) { return kind * 16 + src; }
inline int tail_word_index(int part) { return 64 + part; }   // (behind the words of the four kinds: setup_tail_wait)

struct WaitArgs {
  int n;
  const unsigned long long *flag[16];
  unsigned long long seq;
  int *fail;          // host-mapped: set when a word has not arrived after `ticks`
  long long ticks;    // of the 100 MHz wall clock (DFLO_IPC_TIMEOUT_S, default 120 s); 0: wait for ever
};
__global__ void wait_flags_kernel(const WaitArgs w) {
  const int i = threadIdx.x;
  if (i >= w.n) return;
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(w.flag[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < w.seq) {
    __builtin_amdgcn_s_sleep(16);
    if (w.ticks > 0 && wall_clock64() - t0 > w.ticks) {   // a peer that died or fell out of step: give up, tell the host
      __hip_atomic_store(w.fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
}
struct SignalArgs {
  int n;
  unsigned long long *flag[16];
  unsigned long long seq;
};
// behind a kernel whose stores the receivers are waiting for (the step's reductions, which write this rank's CFL minimum into
// every rank's table): the kernel boundary has completed those stores, this one publishes the sequence number
__global__ void signal_kernel(const SignalArgs a) {
  const int i = threadIdx.x;
  if (i >= a.n) return;
  __threadfence_system();
  __hip_atomic_store(a.flag[i], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// both in one launch (each tiny kernel costs its stream ~4.7 us): publish, then wait -- for the words of `w` and, where the stage
// kernel delivers, for the neighbours' traces of the step's last stage (`w2`) as well
__global__ void signal_wait_kernel(const SignalArgs a, const WaitArgs w, const WaitArgs w2) {
  const int i = threadIdx.x;
  if (i < a.n) {
    __threadfence_system();
    __hip_atomic_store(a.flag[i], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const WaitArgs &ww = i < 16 ? w : w2;
  const int k = i < 16 ? i : i - 16;
  if (i >= 32 || k >= ww.n) return;
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(ww.flag[k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < ww.seq) {
    __builtin_amdgcn_s_sleep(16);
    if (ww.ticks > 0 && wall_clock64() - t0 > ww.ticks) {
      __hip_atomic_store(ww.fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
}

// what rank q lets this rank write (IPC mappings; self-halo: this rank's own pointers) and where in q's areas this rank's records go
struct PeerMap {
  double *tg[2] = {nullptr, nullptr}, *recv_a[2] = {nullptr, nullptr}, *recv_u[2] = {nullptr, nullptr}, *dt_table = nullptr;
  unsigned long long *flags = nullptr;
  int32_t ro = 0, rfo = 0;     // q's receive offsets for this rank's cells / face traces, in records
  void *opened[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

// Host-side sequencing between the host threads of the parts (one-process mode).  A stream can only wait for an event that has
// been recorded: before a thread calls hipStreamWaitEvent on a peer's event it waits until the peer's counter says the record
// has been issued.  With the single-threaded driver the counters are always ahead (all parts post before any part waits).
struct Sync {
  std::atomic<int64_t> posted[3];   // exchanges of each kind whose copies and `sent` record this part has issued
  std::atomic<int64_t> used[3];     // strict mode: exchanges of each kind whose consumers (unpack / rim kernel) this part has issued
  std::atomic<int64_t> fin;         // steps whose `fin` record (CFL minimum published) this part has issued
  Sync() { reset(); }
  void reset() {
    for (int k = 0; k < 3; ++k) { posted[k].store(0); used[k].store(0); }
    fin.store(0);
  }
};

struct Part {
  int index = 0;               // part number in the partition (= rank in rank mode)
  int device = 0;
  dflo_hip_handle eng = nullptr;
  dflo_mesh_t *sub = nullptr;  // owned + ghost sub-mesh (cell_global_id: ids in the undivided mesh)
  const int32_t *send_cells = nullptr, *send_off = nullptr, *recv_off = nullptr;   // [P+1] offsets, owned by `sub`
  int n_owned = 0, n_cells = 0, n_send = 0, n_ghost = 0;
  std::vector<int> peers;      // parts this one exchanges cells with
  int group = 0;               // the device group (streams, host thread) this part belongs to
  hipStream_t M = nullptr, C = nullptr;   // the group's compute and comm stream
  hipEvent_t ev_dt = nullptr;
  hipEvent_t ev_fin[2] = {nullptr, nullptr};                 // by the parity of the step
  hipEvent_t ev_sent[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};   // by kind of record and receive area
  hipEvent_t ev_used[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};   // strict mode: area consumed
  Sync *sy = nullptr;
  int64_t n_post[3] = {0, 0, 0}, n_arr[3] = {0, 0, 0}, n_fin = 0;   // this part's own counts (what it has issued / expects of the peers)
  double *send_u = nullptr, *send_a = nullptr;
  double *recv_u[2] = {nullptr, nullptr}, *recv_a[2] = {nullptr, nullptr};   // alternate from exchange to exchange
  // face-trace records (dflo_hip_halo_traces): what goes to peer q are the traces of the faces sendf_off[q] .. of the send
  // list, what comes from q lands in the engine's trace table at recvf_off[q] (no staging buffer, no unpack)
  bool trace = false;
  std::vector<int32_t> sendf_off, recvf_off;
  double *send_t = nullptr;
  void *tg[2] = {nullptr, nullptr};
  void *dt_table = nullptr;    // the engine's table of the parts' CFL minima (dflo_hip_dt_table)
  void *dt_ptr = nullptr, *res_ptr = nullptr;
  int steps_run = 0;             // threaded advance: steps this part's thread issued
  std::vector<int32_t> bface_global;   // global boundary-face number of the engine's boundary faces
  // exchange timing (dflo_hip_multi_exchange_timing): event pairs on the comm stream around every fifth exchange -- rank mode: the
  // grouped send / receive (rendezvous with the peers included); one process: the wait for the peers' records
  std::vector<std::pair<hipEvent_t, hipEvent_t>> x_pool;
  size_t x_used = 0;
  int64_t x_seen = 0;
  bool x_on = false, x_open = false;
};

// A group = one compute stream, one comm stream and one host thread.  A part with a device of its own is a group of its own;
// parts that share a device (a test arrangement, and what a caller with more parts than devices gets) form at most TWO groups
// on it: the runtime maps streams onto four hardware queues per device, and with more stream pairs than that the small kernels
// of one part's comm stream queue behind another part's interior launch (measured, four parts on one device, C2: a stream pair
// each 67 000 MDoF/s, GPU_MAX_HW_QUEUES=8 worse still; one pair for all 147 000 -- the interior launches then run strictly one
// after the other, every one with its own ramp and tail; two pairs: see DESIGN).  DFLO_MULTI_GROUP=part | device forces one group
// per part / per device.
struct Group {
  int device = 0;
  hipStream_t M = nullptr, C = nullptr;
  std::vector<int> parts;      // indices into dflo_hip_multi::parts
  // the order between the two streams, once per group and stage whatever the number of parts (what one stream has to wait for
  // is the LAST launch of the other stream's phase; the launches before it are covered by stream order)
  hipEvent_t ev_open = nullptr, ev_rim = nullptr, ev_rim_prev = nullptr, ev_ring = nullptr, ev_unpack = nullptr, ev_chunk[2] = {nullptr, nullptr};
  bool unpack_pending = false;   // the comm stream has work the compute stream has not waited for
  bool rim_pending = false;      // ... among it the rim of the previous stage
  bool open_attached = false;    // ev_open rides on the compute stream's last kernel of the previous stage (dflo_hip_attach_event)
};
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = true, quit = false;
  int rc = 0;
};

}  // namespace

struct dflo_hip_multi {
  std::vector<Part> parts;     // the parts this process owns
  std::unique_ptr<Sync[]> sync;
  std::vector<Group> groups;
  std::vector<std::unique_ptr<Worker>> workers;   // one per group; empty: the calling thread drives every part
  bool direct = true;          // one process: the pack kernels write into the peers' receive areas (DFLO_MULTI_COPY=1: staging buffer + hipMemcpyPeerAsync)
  bool need_avg = true;        // somebody reads the ghost cells' averages (LxF flux, TVB limiter): they travel with the traces
  bool avg_in_place = false;   // TVB without the LxF flux, face-trace halos: only the rim limiter reads ghost averages, from the receive area
  // TVB stages with ONE exchange instead of the reference's two (src_mpi/limiter.cc:232 + src_mpi/claw.cc:793): the cut cells travel
  // unlimited, as CH_CELLS records widened by the averages of their face neighbours, and every part limits its ghost cells itself
  // (dflo_hip_limit_ghost_cells).  Needs every cut cell to border on one other part only (one_neighbour), ghost cells known by traces
  bool tvb_one = false, one_neighbour = true;
  bool dt_on_comm = false;     // DFLO_DT_ON_COMM=1 (reduce_dt_rank)
  // The compute stream's order behind the comm stream's rim launch WITHOUT a wait packet (setup_tail_wait): a one-thread kernel behind
  // the rim (+ ring) update publishes the stage's number in a word; the interior launch of the same stage does not end before it has
  // seen it (dflo_hip_stage_tail_wait) -- so whatever follows the interior launch on the compute stream follows the rim launch as well
  bool tail_wait = false;
  unsigned long long tail_count = 0;   // stages whose rim launch has been followed by its word (never reset: the word only grows)
  std::atomic<bool> abort{false};                 // a part's thread has failed: the others stop waiting for it
  std::atomic<int64_t> stop_at{INT64_MAX};        // threaded advance: the step at which every thread leaves the loop
  bool strict = false;         // DFLO_MULTI_STRICT=1: a sender waits for an explicit "consumed" event of the receive area
  int n_parts = 1;             // parts of the partition (= n_ranks in rank mode)
  int rank = 0;                // rank mode: this process's part
  bool rank_mode = false, loopback = false;
  // self-halo (dflo_hip_multi_create_self): ONE part that is its own neighbour across a virtual cut -- the whole schedule of a
  // rank of a multi-GPU run (rim / interior split, pack, transport, trace tables, time-step reduction) on a single device
  bool self_halo = false;
  int self_virtual = 1;
  // DFLO_RANK_TRANSPORT=ipc (see PeerMap): sequence words of this rank (fine-grained device memory), the peers' mappings, the
  // exchanges of each kind this rank has sent / expects (never reset: the words only grow), the wait kernels' failure word
  bool created = false, fatal = false;              // create ran to its end; a device or transport failure has been seen (agree)
  bool ipc = false;
  std::vector<int> pend_from;  // fused: the wait for the traces of a step's last stage rides with the time step's wait kernel
  unsigned long long pend_seq = 0;
  bool pend = false;
  bool kwait = false;          // fused, trace tables fine-grained (or this device's own writes): the NEXT stage kernel's workgroups wait
  bool want_ipc = false;       // the IPC transport has been asked for (known before the engines are made)
  bool fused_tvb = false;      // ... with a TVB limiter: averages leave from the stage kernel, traces from the limiter pass
  bool fused = false;          // ... and the stage kernel delivers its cut faces' traces itself (one launch per stage, one stream)
  void *win_data = nullptr, *win_sync = nullptr;   // what the peers map (see IpcExport)
  bool ipc_fine = false;                            // win_data is fine-grained memory
  bool recv_in_window = false;                      // the receive areas lie in win_data (not allocations of their own)
  unsigned long long *flags = nullptr;
  std::vector<PeerMap> pmap;
  unsigned long long ipc_post[4] = {0, 0, 0, 0}, ipc_arr[4] = {0, 0, 0, 0};
  unsigned long long ipc_post_met[4] = {0, 0, 0, 0};   // ipc_post at the ranks' last barrier (destroy: has anything been stored into a neighbour since?)
  volatile int *ipc_fail_host = nullptr;
  int *ipc_fail = nullptr;
  long long ipc_ticks = 0;     // DFLO_IPC_TIMEOUT_S in ticks of the 100 MHz clock (0: the wait kernels wait for ever)
  ncclComm_t comm = nullptr;
  // one process per GPU with the host program's own transport (MPI, ...) instead of RCCL
  dflo_exchange_fn x_exchange = nullptr;
  dflo_allreduce_fn x_allreduce = nullptr;
  void *x_user = nullptr;
  dflo_params_t prm{};
  int degree = 1, basis = 0, ndof = 16, N = 2, n_rk = 2;
  int64_t n_cells_global = 0;
  bool tvb = false, kxrcf = false, limited = false, sep_limiter = false;
  // Exchanges of the state since the last set_solution.  Exchange n (0, 1, ..) fills receive area (1 + n) & 1 of the DoF / trace
  // records and n & 1 of the averages: never the trace table the engines read at that moment (they start with table 0).
  int64_t nx = 0;
  int64_t n_steps_fin = 0;     // steps whose time step has been reduced over the parts (parity: which published slot)
  double *scal = nullptr;      // rank mode: device scratch for small all-reduces (on parts[0].device)
  std::vector<int32_t> gb_cell, gb_face, gb_id;   // boundary faces of the undivided mesh, MeshWorker order
  std::vector<double> gb_xy;
  hipEvent_t ev_chunk[2] = {nullptr, nullptr};
  std::mutex err_mu;
  std::string err;
  double t_phase[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // DFLO_MULTI_VERBOSE: host time spent issuing each phase (single-threaded driver)
  bool verbose = false;
};

namespace {

void set_err(dflo_hip_multi *m, const std::string &s) {
  std::lock_guard<std::mutex> lk(m->err_mu);
  m->err = s;
}

#define MHIP(m, call)                                                           \
  do {                                                                          \
    hipError_t e_ = (call);                                                     \
    if (e_ != hipSuccess) {                                                     \
      set_err((m), std::string(#call) + ": " + hipGetErrorString(e_));          \
      return DFLO_ERR_HIP;                                                      \
    }                                                                           \
  } while (0)
#define MNCCL(m, call)                                                          \
  do {                                                                          \
    ncclResult_t r_ = (call);                                                   \
    if (r_ != ncclSuccess) {                                                    \
      set_err((m), std::string(#call) + ": " + g_rccl.GetErrorString(r_));      \
      return DFLO_ERR_COMM;                                                     \
    }                                                                           \
  } while (0)
#define MENG(m, part, call)                                                     \
  do {                                                                          \
    int rc_ = (call);                                                           \
    if (rc_) {                                                                  \
      set_err((m), std::string("part ") + std::to_string((part).index) + ": " + dflo_hip_last_error((part).eng)); \
      return rc_;                                                               \
    }                                                                           \
  } while (0)

Part *local_part(dflo_hip_multi *m, int index) {
  for (Part &p : m->parts)
    if (p.index == index) return &p;
  return nullptr;
}

// wait (host) until a peer's counter has reached `want`
int wait_count(dflo_hip_multi *m, const std::atomic<int64_t> &c, int64_t want) {
  if (c.load(std::memory_order_acquire) >= want) return DFLO_OK;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 1;; ++spin) {
    if (c.load(std::memory_order_acquire) >= want) return DFLO_OK;
    if (m->abort.load(std::memory_order_acquire)) return DFLO_ERR_COMM;   // the failing thread has left its message
    if (m->workers.empty()) { set_err(m, "multi-device schedule out of step: a part waits for records no part has posted"); return DFLO_ERR_COMM; }
    if (spin > 4096) std::this_thread::yield();
    if ((spin & 0xFFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
      set_err(m, "multi-device schedule: a peer part has not posted its records for two minutes");
      return DFLO_ERR_COMM;
    }
  }
}

// Run fn(group) for every group: on the groups' own threads when there are any, else one after the other on the calling
// thread.  Returns the first failure (the groups still run to their end: a thread that fails raises `abort`, which ends the
// others' waits).
template <class F>
int for_groups(dflo_hip_multi *m, F fn) {
  if (m->workers.empty()) {
    for (Group &g : m->groups) {
      const int rc = fn(g);
      if (rc) return rc;
    }
    return DFLO_OK;
  }
  m->abort.store(false);
  for (size_t i = 0; i < m->workers.size(); ++i) {
    Worker &w = *m->workers[i];
    Group *gp = &m->groups[i];
    std::lock_guard<std::mutex> lk(w.mu);
    w.job = [m, gp, &fn]() {
      const int rc = fn(*gp);
      if (rc) m->abort.store(true, std::memory_order_release);
      return rc;
    };
    w.has_job = true;
    w.done = false;
    w.cv.notify_all();
  }
  int first = DFLO_OK;
  for (auto &wp : m->workers) {
    Worker &w = *wp;
    std::unique_lock<std::mutex> lk(w.mu);
    w.cv.wait(lk, [&] { return w.done; });
    // a thread that gave up because another one failed reports DFLO_ERR_COMM: the first real failure wins
    if (w.rc && (!first || first == DFLO_ERR_COMM)) first = w.rc;
  }
  return first;
}

void worker_main(Worker *w) {
  for (;;) {
    std::function<int()> job;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv.wait(lk, [&] { return w->has_job || w->quit; });
      if (w->quit) return;
      job = std::move(w->job);
      w->has_job = false;
    }
    const int rc = job();
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->rc = rc;
      w->done = true;
    }
    w->cv.notify_all();
  }
}

void stop_workers(dflo_hip_multi *m) {
  for (auto &wp : m->workers) {
    {
      std::lock_guard<std::mutex> lk(wp->mu);
      wp->quit = true;
    }
    wp->cv.notify_all();
    if (wp->th.joinable()) wp->th.join();
  }
  m->workers.clear();
}

// ---- transport.  Three kinds of record travel: whole cells (DoFs + average), cell averages, face traces.  post: the
// records packed in `send` leave for the peers; arrive: the comm stream waits until the peers' records are in this part's
// receive area `par`.
struct ChanView {
  const int32_t *so, *ro;   // per-peer offsets of what this part sends / receives, in records
  int width;                // doubles per record
  double *recv;             // where this part receives (area `par`)
};
ChanView chan(dflo_hip_multi *m, Part &p, int kind, int par) {
  if (kind == CH_AVG) return {p.send_off, p.recv_off, 4, p.recv_a[par]};
  if (kind == CH_TRACES) return {p.sendf_off.data(), p.recvf_off.data(), 4 * m->N, (double *)p.tg[par]};
  return {p.send_off, p.recv_off, m->ndof + (m->tvb_one ? 20 : 4), p.recv_u[par]};
}
// the engine's number for the records of a channel (dflo_hip_pack_send_to)
int eng_kind(const dflo_hip_multi *m, int kind) { return kind == CH_CELLS ? (m->tvb_one ? 3 : 0) : (kind == CH_AVG ? 1 : 2); }

void xt_begin(Part &p, hipStream_t st = nullptr) {
  p.x_open = p.x_on && (p.x_seen++ % 5 == 0) && p.x_used < 4096;
  if (!p.x_open) return;
  if (p.x_used == p.x_pool.size()) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    p.x_pool.push_back({a, b});
  }
  hipEventRecord(p.x_pool[p.x_used].first, st ? st : p.C);
}
void xt_end(Part &p, hipStream_t st = nullptr) {
  if (!p.x_open) return;
  hipEventRecord(p.x_pool[p.x_used].second, st ? st : p.C);
  ++p.x_used;
  p.x_open = false;
}

// pack the records of this kind and send them off (the engine's launches go to the comm stream, set by the caller)
int post(dflo_hip_multi *m, Part &p, int kind, int par) {
  if (p.peers.empty()) return DFLO_OK;
  const ChanView v = chan(m, p, kind, par);
  const size_t w = (size_t)v.width;
  const bool direct = m->direct && !m->rank_mode && !m->loopback;
  double *send = kind == CH_AVG ? p.send_a : (kind == CH_TRACES ? p.send_t : p.send_u);
  if (m->rank_mode && m->ipc) {
    // the pack kernel stores into the neighbours' areas and then publishes this exchange's number in their sequence words.  Why
    // area `par` of rank q is free: as in the one-process driver below (q's readers of the exchange two back precede, on q's comm
    // stream, q's own send of the last exchange, whose number this rank's comm stream has seen before it got here)
    const unsigned long long seq = ++m->ipc_post[kind];
    int32_t first[17];
    void *dst[16], *fl[16];
    int nseg = 0;
    for (int q : p.peers) {
      const size_t n = (size_t)(v.so[q + 1] - v.so[q]);
      if (!n) continue;
      if (nseg == 16) { set_err(m, "more than 16 neighbouring ranks"); return DFLO_ERR_UNSUPPORTED; }
      const PeerMap &pm = m->pmap[q];
      double *base = kind == CH_AVG ? pm.recv_a[par] : (kind == CH_TRACES ? pm.tg[par] : pm.recv_u[par]);
      first[nseg] = v.so[q];
      dst[nseg] = base + (size_t)(kind == CH_TRACES ? pm.rfo : pm.ro) * w;
      fl[nseg++] = pm.flags + flag_index(kind, p.index);
    }
    if (nseg) {
      first[nseg] = v.so[m->n_parts];
      MENG(m, p, dflo_hip_pack_send_to_signal(p.eng, eng_kind(m, kind), nseg, first, dst, fl, seq));
    }
    return DFLO_OK;
  }
  if (!direct) {   // into the staging buffer first
    if (kind == CH_AVG) MENG(m, p, dflo_hip_pack_send_avg(p.eng, send));
    else if (kind == CH_TRACES) MENG(m, p, dflo_hip_pack_send_traces(p.eng, send));
    else if (m->tvb_one) MENG(m, p, dflo_hip_pack_send_cells_unlimited(p.eng, send));
    else MENG(m, p, dflo_hip_pack_send_cells(p.eng, send));
  }
  if (m->rank_mode && m->x_exchange) {
    std::vector<int> peers;
    std::vector<const void *> sp;
    std::vector<void *> rp;
    std::vector<size_t> sb, rb;
    for (int q : p.peers) {
      peers.push_back(q);
      sp.push_back(send + (size_t)v.so[q] * w);
      sb.push_back((size_t)(v.so[q + 1] - v.so[q]) * w * sizeof(double));
      rp.push_back(v.recv + (size_t)v.ro[q] * w);
      rb.push_back((size_t)(v.ro[q + 1] - v.ro[q]) * w * sizeof(double));
    }
    xt_begin(p);
    const int xrc = m->x_exchange(m->x_user, (int)peers.size(), peers.data(), sp.data(), sb.data(), rp.data(), rb.data(), (void *)p.C);
    xt_end(p);
    if (xrc) {
      set_err(m, "the host program's exchange callback failed");
      return DFLO_ERR_COMM;
    }
    return DFLO_OK;
  }
  if (m->rank_mode) {
    // (the receives are posted by the receiver itself, on its comm stream behind everything of its own that read the
    //  receive area: no peer can overwrite what is still in use)
    xt_begin(p);
    MNCCL(m, g_rccl.GroupStart());
    for (int q : p.peers) {
      const size_t ns = (size_t)(v.so[q + 1] - v.so[q]) * w, nr = (size_t)(v.ro[q + 1] - v.ro[q]) * w;
      if (ns) MNCCL(m, g_rccl.Send(send + (size_t)v.so[q] * w, ns, ncclDouble, q, m->comm, p.C));
      if (nr) MNCCL(m, g_rccl.Recv(v.recv + (size_t)v.ro[q] * w, nr, ncclDouble, q, m->comm, p.C));
    }
    MNCCL(m, g_rccl.GroupEnd());
    xt_end(p);
    return DFLO_OK;
  }
  // One process: this part writes into its peers' receive areas -- the pack kernel itself does (stores over xGMI peer access
  // where the peer sits on another device), or a copy per peer out of the staging buffer.  Why area `par` of peer q is free:
  // q's consumers of the exchange two back (unpack kernel / rim kernel reading the trace table) precede q's own send of the last
  // exchange on q's comm stream, this part's comm stream has waited for that send (arrive), and q lists this part as a peer
  // exactly when this part lists q (checked at create).  DFLO_MULTI_STRICT=1 does not rely on that chain: the writes wait for
  // q's "consumed" event.
  const int64_t seq = ++p.n_post[kind];
  int32_t first[17];
  void *dst[16];
  int nseg = 0;
  for (int q : p.peers) {
    Part *dp = local_part(m, q);
    const ChanView dv = chan(m, *dp, kind, par);
    const size_t n = (size_t)(v.so[q + 1] - v.so[q]) * w;
    if (!n) continue;
    double *to = dv.recv + (size_t)dv.ro[p.index] * w;
    const double *from = send + (size_t)v.so[q] * w;
    if (m->strict && seq > 2) {
      const int rc = wait_count(m, dp->sy->used[kind], seq - 2);
      if (rc) return rc;
      if (dp->C != p.C) MHIP(m, hipStreamWaitEvent(p.C, dp->ev_used[kind][par], 0));
    }
    if (direct) {
      if (nseg == 16) { set_err(m, "more than 16 neighbouring parts"); return DFLO_ERR_UNSUPPORTED; }
      first[nseg] = v.so[q];
      dst[nseg++] = to;
    } else if (m->loopback) {   // test transport: the same copy as a self send/recv pair through RCCL
      MNCCL(m, g_rccl.GroupStart());
      MNCCL(m, g_rccl.Send(from, n, ncclDouble, 0, m->comm, p.C));
      MNCCL(m, g_rccl.Recv(to, n, ncclDouble, 0, m->comm, p.C));
      MNCCL(m, g_rccl.GroupEnd());
    } else {
      MHIP(m, hipMemcpyPeerAsync(to, dp->device, from, p.device, n * sizeof(double), p.C));
    }
  }
  if (direct && nseg) {
    first[nseg] = v.so[m->n_parts];
    MENG(m, p, dflo_hip_pack_send_to(p.eng, eng_kind(m, kind), nseg, first, dst));
  }
  bool foreign = false;   // a receiver on another stream (its own device's): it waits for this record
  for (int q : p.peers) foreign |= local_part(m, q)->C != p.C;
  if (foreign) MHIP(m, hipEventRecord(p.ev_sent[kind][par], p.C));
  p.sy->posted[kind].store(seq, std::memory_order_release);
  return DFLO_OK;
}

// the comm (or compute) stream waits until the listed ranks' words of this kind have reached `seq`
int fill_wait(dflo_hip_multi *m, WaitArgs &w, int kind, const std::vector<int> &from, unsigned long long seq) {
  w = WaitArgs{};
  for (int q : from) {
    if (w.n == 16) { set_err(m, "more than 16 ranks to wait for"); return DFLO_ERR_UNSUPPORTED; }
    w.flag[w.n++] = m->flags + flag_index(kind, q);
  }
  w.seq = seq;
  w.fail = m->ipc_fail;
  w.ticks = m->ipc_ticks;
  return DFLO_OK;
}
int wait_words(dflo_hip_multi *m, hipStream_t st, int kind, const std::vector<int> &from, unsigned long long seq) {
  WaitArgs w{};
  const int rc = fill_wait(m, w, kind, from, seq);
  if (rc) return rc;
  if (!w.n) return DFLO_OK;
  hipLaunchKernelGGL(wait_flags_kernel, dim3(1), dim3(64), 0, st, w);
  MHIP(m, hipGetLastError());
  return DFLO_OK;
}

int arrive(dflo_hip_multi *m, Part &p, int kind, int par) {
  if (m->rank_mode && m->ipc) {
    if (p.peers.empty()) return DFLO_OK;
    const unsigned long long seq = ++m->ipc_arr[kind];
    const ChanView v = chan(m, p, kind, par);
    std::vector<int> from;
    for (int q : p.peers)
      if (v.ro[q + 1] > v.ro[q]) from.push_back(q);
    xt_begin(p);
    const int rc = wait_words(m, p.C, kind, from, seq);
    xt_end(p);
    return rc;
  }
  if (m->rank_mode) return DFLO_OK;   // the receives were part of the group posted on this stream
  if (p.peers.empty()) return DFLO_OK;
  const int64_t seq = ++p.n_arr[kind];
  xt_begin(p);
  for (int q : p.peers) {
    Part *src = local_part(m, q);
    const ChanView sv = chan(m, *src, kind, 0);
    if (sv.so[p.index + 1] == sv.so[p.index]) continue;
    const int rc = wait_count(m, src->sy->posted[kind], seq);   // the record below exists
    if (rc) return rc;
    if (src->C == p.C) continue;   // the sender's launches are ahead of this point on the very same stream
    MHIP(m, hipStreamWaitEvent(p.C, src->ev_sent[kind][par], 0));
  }
  xt_end(p);
  return DFLO_OK;
}

// strict mode: everything that reads receive area `par` of this kind has been issued on the comm stream
int mark_used(dflo_hip_multi *m, Part &p, int kind, int par) {
  if (!m->strict || m->rank_mode || p.peers.empty() || p.n_arr[kind] == 0) return DFLO_OK;
  bool foreign = false;
  for (int q : p.peers) foreign |= local_part(m, q)->C != p.C;
  if (foreign) MHIP(m, hipEventRecord(p.ev_used[kind][par], p.C));
  p.sy->used[kind].store(p.n_arr[kind], std::memory_order_release);
  return DFLO_OK;
}

// The new state of the cut cells leaves (engine launches go to the comm stream, set by the caller): whole cells, or face
// traces (+ the averages, if anybody reads them, unless they have travelled already, as they do when a TVB limiter sits between
// update and send)
int send_state(dflo_hip_multi *m, Part &p, int upar, int apar, bool with_avg) {
  if (!p.trace) return post(m, p, CH_CELLS, upar);
  const int rc = post(m, p, CH_TRACES, upar);
  if (rc || !with_avg || !m->need_avg) return rc;
  return post(m, p, CH_AVG, apar);
}
// ... and the neighbours' arrives: into the ghost shards, or straight into the trace table the next stage will read
int recv_state(dflo_hip_multi *m, Part &p, int upar, int apar, bool with_avg) {
  if (!p.trace) {
    int rc = arrive(m, p, CH_CELLS, upar);
    if (rc) return rc;
    MENG(m, p, dflo_hip_unpack_ghost_cells(p.eng, p.recv_u[upar]));
    return mark_used(m, p, CH_CELLS, upar);
  }
  int rc = arrive(m, p, CH_TRACES, upar);
  if (rc) return rc;
  MENG(m, p, dflo_hip_use_ghost_traces(p.eng, upar));
  if (!with_avg || !m->need_avg) return DFLO_OK;
  if ((rc = arrive(m, p, CH_AVG, apar))) return rc;
  MENG(m, p, dflo_hip_unpack_ghost_avg(p.eng, p.recv_a[apar]));
  return mark_used(m, p, CH_AVG, apar);
}

// update_ghost_values() outside the overlapped stage (after set-up calls, in the KXRCF path), exchange number n: a part packs
// and sends ...
int ex_send(dflo_hip_multi *m, Part &p, int64_t n) {
  if (p.peers.empty()) return DFLO_OK;
  Group &g = m->groups[p.group];
  MHIP(m, hipSetDevice(p.device));
  MHIP(m, hipEventRecord(g.ev_open, g.M));
  MHIP(m, hipStreamWaitEvent(g.C, g.ev_open, 0));
  MENG(m, p, dflo_hip_set_stream(p.eng, g.C));
  // (a part that reads ghost cells by their traces: the table of the exchange before this one has been read by kernels on M
  //  that ev_open covers)
  int rc = mark_used(m, p, CH_TRACES, (int)(n & 1));
  if (!rc) rc = send_state(m, p, (int)((1 + n) & 1), (int)(n & 1), true);
  MENG(m, p, dflo_hip_set_stream(p.eng, g.M));
  return rc;
}
// ... and takes in what its neighbours sent
int ex_recv(dflo_hip_multi *m, Part &p, int64_t n) {
  if (p.peers.empty()) return DFLO_OK;
  Group &g = m->groups[p.group];
  MHIP(m, hipSetDevice(p.device));
  MENG(m, p, dflo_hip_set_stream(p.eng, g.C));
  int rc = recv_state(m, p, (int)((1 + n) & 1), (int)(n & 1), true);
  MENG(m, p, dflo_hip_set_stream(p.eng, g.M));
  if (rc) return rc;
  MHIP(m, hipEventRecord(g.ev_unpack, g.C));
  MHIP(m, hipStreamWaitEvent(g.M, g.ev_unpack, 0));
  return DFLO_OK;
}
int exchange_solution(dflo_hip_multi *m) {   // calling thread, all parts
  const int64_t n = m->nx++;
  for (Part &p : m->parts) {
    const int rc = ex_send(m, p, n);
    if (rc) return rc;
  }
  for (Part &p : m->parts) {
    const int rc = ex_recv(m, p, n);
    if (rc) return rc;
  }
  return DFLO_OK;
}

// One RK stage of a group's parts, in phases.  The rim shards run on the comm stream C (high priority), the interior shards on
// the compute stream M, side by side: both read the previous stage, they write disjoint shards.
//   C: [wait: interior of the previous stage / the new time step]  update rim  (TVB: rim + ring; exchange the averages of the
//      rim cells; limit the rim)  -> ev_rim;  pack; send / receive; unpack into the ghost shards
//      (TVB, tvb_one: update rim + ring; the UNLIMITED cut cells with their neighbours' averages leave / arrive -- the stage's one
//      exchange --; unpack, limit rim and ghost cells in one launch, whose ghost wavefronts form the ghost traces  -> ev_rim)
//   M: [wait: ev_rim of the previous stage]  update interior  (TVB: all but rim + ring; wait for the ring; limit all but the
//      rim);  last stage: wait ev_rim, reductions of the step
// A phase is issued for every part of the group before the next one: every record a phase waits for has then been issued by an
// earlier phase (parts of this group) or is being issued by another group's thread (wait_count), and the two streams are
// ordered by one event per phase and group instead of one per part.
struct StageCtx {
  int rk;
  double dt;
  int64_t n;   // number of this exchange of the state
  bool last;
};
constexpr int kStagePhases = 6;

int stage_phase(dflo_hip_multi *m, Group &g, const StageCtx &s, int ph) {
  const int upar = (int)((1 + s.n) & 1), apar = (int)(s.n & 1);
  const int rim_update = m->tvb ? 3 : 1, int_update = m->tvb ? 4 : 2;
  MHIP(m, hipSetDevice(g.device));
  switch (ph) {
    case 0:   // open the stage (buffer roles; rk = 0: boundary programs on M); then C may start once M is here
      for (int i : g.parts) MENG(m, m->parts[i], dflo_hip_stage_open(m->parts[i].eng, s.rk, s.dt));
      // interior of the previous stage, the step's time step, boundary data: recorded here, unless the compute stream's last
      // kernel of the previous stage carries the event as its completion signal (phases 2 / 4)
      if (!g.open_attached || s.rk == 0) MHIP(m, hipEventRecord(g.ev_open, g.M));
      g.open_attached = false;
      MHIP(m, hipStreamWaitEvent(g.C, g.ev_open, 0));
      return DFLO_OK;
    case 1:   // the rim on C and what leaves first
      for (int i : g.parts) {
        Part &p = m->parts[i];
        MENG(m, p, dflo_hip_set_stream(p.eng, g.C));
        // What the compute stream waits for -- TVB: the UPDATE of rim + ring (their averages; not the averages' way to the
        // neighbours, which in rank mode is an RCCL kernel that beside a full-size interior launch takes as long as the launch,
        // profiles/r05/tl_c4_self_rccl_before.txt); otherwise the rim's update and limiter -- is the completion of the last part's
        // last kernel here: the event rides on that kernel (dflo_hip_attach_event), no record packet behind it
        const bool last_part = i == g.parts.back();
        // (tail_wait: nobody waits for that event -- the interior launch of this stage does the waiting itself)
        if (last_part && (m->tvb || !m->sep_limiter) && !m->tail_wait) MENG(m, p, dflo_hip_attach_event(p.eng, m->tvb ? g.ev_ring : g.ev_rim));
        MENG(m, p, dflo_hip_stage_update_part(p.eng, rim_update));
        if (m->tail_wait) {   // "the rim (+ ring) of stage n is updated": what the interior launch of this stage waits for before it ends.
          // Published by the pack kernel that follows on this stream (its first thread: the kernel boundary has released the rim's
          // stores) -- a kernel or a stream memory operation of its own would stand in front of the send, where the comm stream's
          // chain sets the pace (C3 over RCCL: 0.88 -> 0.79 / 0.74 with either; on a side stream behind the rim's event: worse)
          MENG(m, p, dflo_hip_pack_publish(p.eng, m->flags + tail_word_index(p.index), (uint64_t)++m->tail_count));
        }
        int rc = mark_used(m, p, CH_TRACES, apar);     // the rim kernel has read the trace table of the exchange before (area apar = upar ^ 1)
        if (rc) return rc;
        if (!m->tvb && m->sep_limiter) {
          if (last_part) MENG(m, p, dflo_hip_attach_event(p.eng, g.ev_rim));
          MENG(m, p, dflo_hip_stage_limit_part(p.eng, 1));
        }
        // (tvb_one: the rim's cells as the update left them, with the averages of their neighbours -- rim and ring are updated)
        rc = m->tvb ? (m->tvb_one ? post(m, p, CH_CELLS, upar) : post(m, p, CH_AVG, apar)) : send_state(m, p, upar, apar, true);
        MENG(m, p, dflo_hip_set_stream(p.eng, g.M));
        if (rc) return rc;
      }
      return DFLO_OK;
    case 2:   // the interior on M, next to it
      // the rim cells of the previous stage are the halo of the interior.  Not with TVB: there the update is split at rim + ring
      // | rest, and a shard of the rest has no neighbour in the rim (Plan::rim2_shards) -- its halo, ring and rest, was limited
      // by this stream's own pass; the comm stream's chain (averages -> limit rim -> traces, the long one) is off this stream's path
      if (g.rim_pending && !m->tvb && !m->tail_wait) MHIP(m, hipStreamWaitEvent(g.M, g.ev_rim_prev, 0));   // (tail_wait: the interior launch of the stage before saw the rim's word before it ended)
      for (int i : g.parts) {
        // nothing else of this stage follows on this stream (no limiter pass, no reductions): the next stage's "open" is the
        // completion of the last part's interior kernel
        if (i == g.parts.back() && !m->tvb && !m->sep_limiter && !s.last) {
          MENG(m, m->parts[i], dflo_hip_attach_event(m->parts[i].eng, g.ev_open));
          g.open_attached = true;
        }
        if (m->tail_wait) MENG(m, m->parts[i], dflo_hip_stage_tail_wait(m->parts[i].eng, m->flags + tail_word_index(m->parts[i].index), (uint64_t)m->tail_count));
        MENG(m, m->parts[i], dflo_hip_stage_update_part(m->parts[i].eng, int_update));
      }
      return DFLO_OK;
    case 3:   // TVB: the averages of the neighbours across the cut arrive: limit the rim, send its cells
      if (!m->tvb) return DFLO_OK;
      if (m->tvb_one) {   // the neighbours' unlimited cut cells arrive: limit them as their owners do, form their traces; limit the rim
        for (int i : g.parts) {
          Part &p = m->parts[i];
          int rc = arrive(m, p, CH_CELLS, upar);
          if (rc) return rc;
          MENG(m, p, dflo_hip_set_stream(p.eng, g.C));
          MENG(m, p, dflo_hip_limit_ghost_cells(p.eng, p.recv_u[upar], upar));   // (the rim's limiter pass takes them along: unpack, rim + ghost shards, traces)
          MENG(m, p, dflo_hip_stage_limit_part(p.eng, 1));
          if ((rc = mark_used(m, p, CH_CELLS, upar))) return rc;
          MENG(m, p, dflo_hip_set_stream(p.eng, g.M));
        }
        return DFLO_OK;   // (no record of ev_rim: with a TVB limiter nobody waits for it -- whoever joins the streams records ev_unpack)
      }
      for (int i : g.parts) {
        Part &p = m->parts[i];
        int rc = arrive(m, p, CH_AVG, apar);
        if (rc) return rc;
        MENG(m, p, dflo_hip_set_stream(p.eng, g.C));
        if (m->avg_in_place) {   // the rim limiter reads the averages where they arrived (nobody else reads a ghost average)
          MENG(m, p, dflo_hip_ghost_avg_source(p.eng, p.recv_a[apar]));
          MENG(m, p, dflo_hip_stage_limit_part(p.eng, 1));
          if ((rc = mark_used(m, p, CH_AVG, apar))) return rc;
        } else {
          MENG(m, p, dflo_hip_unpack_ghost_avg(p.eng, p.recv_a[apar]));
          if ((rc = mark_used(m, p, CH_AVG, apar))) return rc;
          MENG(m, p, dflo_hip_stage_limit_part(p.eng, 1));
        }
        rc = send_state(m, p, upar, apar, false);   // the averages have travelled already
        MENG(m, p, dflo_hip_set_stream(p.eng, g.M));
        if (rc) return rc;
      }
      return DFLO_OK;   // (ev_rim: see above)
    case 4:   // limiter of the other shards (TVB: it reads averages from the ring), the stage's reductions
      if (m->tvb && !m->tail_wait) MHIP(m, hipStreamWaitEvent(g.M, g.ev_ring, 0));
      if (m->tvb || m->sep_limiter)
        for (int i : g.parts) {
          if (i == g.parts.back() && !s.last) {   // ... or of the last part's limiter pass
            MENG(m, m->parts[i], dflo_hip_attach_event(m->parts[i].eng, g.ev_open));
            g.open_attached = true;
          }
          MENG(m, m->parts[i], dflo_hip_stage_limit_part(m->parts[i].eng, 2));
        }
      // the step's reductions take in the rim shards' partials: those of their UPDATE, which with TVB this stream has waited for
      // already (ev_ring) -- the limiter changes neither the averages nor the residual
      if (s.last && !m->tvb && !m->tail_wait) MHIP(m, hipStreamWaitEvent(g.M, g.ev_rim, 0));
      for (int i : g.parts) {
        MENG(m, m->parts[i], dflo_hip_stage_finish(m->parts[i].eng));
        // the event that opens the next stage rode on this stream's last stage / limiter kernel; if finish put a kernel of its own behind
        // it (boundary programs that no limiter pass took along write the table the next stage's rim reads), the next stage records
        // the plain way, behind everything on this stream
        if (g.open_attached && dflo_hip_finish_enqueued(m->parts[i].eng)) g.open_attached = false;
      }
      std::swap(g.ev_rim, g.ev_rim_prev);      // the next stage's interior waits for this stage's rim
      g.rim_pending = !s.last;                 // (after the last stage M has waited already)
      return DFLO_OK;
    default:  // the neighbours' cells arrive: into the ghost shards, ready for the next stage's rim (same stream)
      for (int i : g.parts) {
        Part &p = m->parts[i];
        if (m->tvb_one) {   // phase 3 has formed the ghost traces of this stage
          MENG(m, p, dflo_hip_use_ghost_traces(p.eng, upar));
          continue;
        }
        MENG(m, p, dflo_hip_set_stream(p.eng, g.C));
        const int rc = recv_state(m, p, upar, apar, !m->tvb);
        MENG(m, p, dflo_hip_set_stream(p.eng, g.M));
        if (rc) return rc;
      }
      g.unpack_pending = true;
      return DFLO_OK;
  }
}

// DFLO_RANK_TRANSPORT=ipc where nothing but face traces travels and no limiter pass sits between update and send (C2, C5): ONE launch
// over all shards per stage on the compute stream -- the workgroups of the shards on a cut deliver their traces themselves and the
// last of them publishes the exchange's number (dflo_hip_stage_deliver) --, then a one-wavefront kernel that waits for the
// neighbours' numbers.  No rim launch, no pack kernel, no comm stream, no event: what is left of the exchange on the compute
// stream is that wait kernel.  A neighbour's table of this exchange is free when its number of the exchange before has been seen:
// it is published by the last of ITS workgroups that read ghost traces (the shards that read them are the shards that deliver).
int fused_stage(dflo_hip_multi *m, Group &g, const StageCtx &s) {
  const int upar = (int)((1 + s.n) & 1);
  MHIP(m, hipSetDevice(g.device));
  bool await_in_kernel = false;
  if (m->pend) {   // the traces of the stage before: awaited by this stage's launch itself (its workgroups on the cut), or by a kernel
    m->pend = false;
    await_in_kernel = m->kwait;
    if (!await_in_kernel) {
      Part &p0 = m->parts[g.parts[0]];
      xt_begin(p0, g.M);
      const int rc = wait_words(m, g.M, CH_TRACES, m->pend_from, m->pend_seq);
      xt_end(p0, g.M);
      if (rc) return rc;
    }
  }
  for (int i : g.parts) {
    Part &p = m->parts[i];
    MENG(m, p, dflo_hip_stage_open(p.eng, s.rk, s.dt));
    if (await_in_kernel) MENG(m, p, dflo_hip_stage_await(p.eng, m->pend_seq));
    MENG(m, p, dflo_hip_stage_deliver(p.eng, upar, ++m->ipc_post[CH_TRACES]));
    MENG(m, p, dflo_hip_stage_update_part(p.eng, 0));
    MENG(m, p, dflo_hip_stage_finish(p.eng));
  }
  for (int i : g.parts) {
    Part &p = m->parts[i];
    std::vector<int> from;
    for (int q : p.peers)
      if (p.recvf_off[q + 1] > p.recvf_off[q]) from.push_back(q);
    // who waits for these traces: the next stage (its launch, or a wait kernel in front of it), or -- after the last stage -- the
    // time step's kernel (reduce_dt_rank), or whoever joins the streams first
    m->pend = true;
    m->pend_from = from;
    m->pend_seq = ++m->ipc_arr[CH_TRACES];
    MENG(m, p, dflo_hip_use_ghost_traces(p.eng, upar));
  }
  return DFLO_OK;
}

// The same with a TVB limiter between update and send (C3, C4; two exchanges per stage, src_mpi/claw.cc:793 and src_mpi/limiter.cc:232):
// still the single engine's two launches per stage and nothing else.  The stage kernel's workgroups on the cut deliver their cells'
// new AVERAGES; the limiter pass has one extra wavefront per shard on the cut, which waits for the neighbours' averages, limits
// the shard with them and delivers the TRACES of the limited state; the next stage kernel's workgroups on the cut wait for the
// neighbours' traces.  (Where the tables are plain device memory the two waits are one-wavefront kernels in front of the launches.)
int fused_tvb_stage(dflo_hip_multi *m, Group &g, const StageCtx &s) {
  const int upar = (int)((1 + s.n) & 1), apar = (int)(s.n & 1);
  MHIP(m, hipSetDevice(g.device));
  Part &p = m->parts[g.parts[0]];   // (one process per GPU: one part)
  bool await_in_kernel = false;
  if (m->pend) {
    m->pend = false;
    await_in_kernel = m->kwait;
    if (!await_in_kernel) {
      xt_begin(p, g.M);
      const int rc = wait_words(m, g.M, CH_TRACES, m->pend_from, m->pend_seq);
      xt_end(p, g.M);
      if (rc) return rc;
    }
  }
  MENG(m, p, dflo_hip_stage_open(p.eng, s.rk, s.dt));
  if (await_in_kernel) MENG(m, p, dflo_hip_stage_await(p.eng, m->pend_seq));
  MENG(m, p, dflo_hip_stage_deliver_averages(p.eng, apar, ++m->ipc_post[CH_AVG]));
  MENG(m, p, dflo_hip_stage_update_part(p.eng, 0));
  std::vector<int> from;
  for (int q : p.peers)
    if (p.recv_off[q + 1] > p.recv_off[q]) from.push_back(q);
  const unsigned long long aseq = ++m->ipc_arr[CH_AVG];
  if (!m->kwait) {
    xt_begin(p, g.M);
    const int rc = wait_words(m, g.M, CH_AVG, from, aseq);
    xt_end(p, g.M);
    if (rc) return rc;
  }
  MENG(m, p, dflo_hip_ghost_avg_source(p.eng, p.recv_a[apar]));
  MENG(m, p, dflo_hip_limit_exchange(p.eng, upar, ++m->ipc_post[CH_TRACES], aseq, m->kwait ? 1 : 0));
  MENG(m, p, dflo_hip_stage_limit(p.eng));   // the pass over all shards + the step's reductions behind the last stage
  from.clear();
  for (int q : p.peers)
    if (p.recvf_off[q + 1] > p.recvf_off[q]) from.push_back(q);
  m->pend = true;
  m->pend_from = from;
  m->pend_seq = ++m->ipc_arr[CH_TRACES];
  MENG(m, p, dflo_hip_use_ghost_traces(p.eng, upar));
  return DFLO_OK;
}

// the KXRCF indicator reads the neighbours' unlimited DoFs of the new stage: ghosts are refreshed between update and
// limiter as well (update_ghost_values before compute_shock_indicator in the MPI variant); no overlap on this path.
// Two exchanges per stage (numbers n and n + 1).
constexpr int kKxrcfPhases = 6;
int kxrcf_phase(dflo_hip_multi *m, Group &g, const StageCtx &s, int ph) {
  MHIP(m, hipSetDevice(g.device));
  for (int i : g.parts) {
    Part &p = m->parts[i];
    int rc = DFLO_OK;
    switch (ph) {
      case 0: MENG(m, p, dflo_hip_stage_update(p.eng, s.rk, s.dt)); break;
      case 1: rc = ex_send(m, p, s.n); break;
      case 2: rc = ex_recv(m, p, s.n); break;
      case 3: MENG(m, p, dflo_hip_stage_limit(p.eng)); break;
      case 4: rc = ex_send(m, p, s.n + 1); break;
      default: rc = ex_recv(m, p, s.n + 1); break;
    }
    if (rc) return rc;
  }
  return DFLO_OK;
}

bool any_peers(dflo_hip_multi *m) {
  for (Part &p : m->parts)
    if (!p.peers.empty()) return true;
  return false;
}
int exchanges_per_stage(dflo_hip_multi *m) { return any_peers(m) ? (m->kxrcf ? 2 : 1) : 0; }

// One RK stage of the listed groups, phase by phase (a group's thread with its own group, or the calling thread with all of
// them).  n: number of the stage's (first) exchange of the state.
int stage_groups(dflo_hip_multi *m, Group *gs, int ng, int rk, double dt, int64_t n, bool peers) {
  if (!peers) {   // nothing to exchange: the plain stage
    for (int k = 0; k < ng; ++k)
      for (int i : gs[k].parts) MENG(m, m->parts[i], dflo_hip_stage(m->parts[i].eng, rk, dt));
    return DFLO_OK;
  }
  const StageCtx s{rk, dt, n, rk == m->n_rk - 1};
  if (m->fused || m->fused_tvb) {
    for (int k = 0; k < ng; ++k) {
      const int rc = m->fused_tvb ? fused_tvb_stage(m, gs[k], s) : fused_stage(m, gs[k], s);
      if (rc) return rc;
    }
    return DFLO_OK;
  }
  const int nph = m->kxrcf ? kKxrcfPhases : kStagePhases;
  for (int ph = 0; ph < nph; ++ph) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < ng; ++k) {
      const int rc = m->kxrcf ? kxrcf_phase(m, gs[k], s, ph) : stage_phase(m, gs[k], s, ph);
      if (rc) return rc;
    }
    if (m->verbose && m->workers.empty()) m->t_phase[ph] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  return DFLO_OK;
}

// one RK stage on every part of this process, driven by the calling thread
int run_stage(dflo_hip_multi *m, int rk, double dt) {
  const bool peers = any_peers(m);
  const int64_t n = m->nx;
  m->nx += exchanges_per_stage(m);
  return stage_groups(m, m->groups.data(), (int)m->groups.size(), rk, dt, n, peers);
}

// Utilities::MPI::min(global_dt) (src_mpi/claw.cc:579) on the device-resident time step, after the last stage of step number
// `step` (counted since create).  One process per GPU: an 8-byte all-reduce on the comm stream.  One process: phase 0, every
// part records that its reductions -- which write its CFL minimum into every part's table -- are issued; phase 1, it waits for
// the others' records.  Nothing is launched: the consumers of the time step form it from the table (engine: step_dt).
int reduce_dt_rank(dflo_hip_multi *m) {
  // the reductions of the step just ended have left this rank's minimum in the slot the next step reads: all-reduce it in
  // place on the comm stream; the consumers apply the rules themselves (no kernel in between)
  Part &p = m->parts[0];
  if (m->ipc) {
    // the reductions have written this rank's minimum into every rank's table (dflo_hip_dt_exchange with the mapped tables): say
    // so in every rank's word for (CH_FIN, this rank), then wait on the compute stream for every rank's word -- no stream hop
    const unsigned long long seq = ++m->ipc_post[CH_FIN];
    SignalArgs sg{};
    std::vector<int> from;
    for (int q = 0; q < m->n_parts; ++q) {
      if (q == p.index && !m->self_halo) continue;
      sg.flag[sg.n++] = m->pmap[q].flags + flag_index(CH_FIN, p.index);
      from.push_back(q);
    }
    sg.seq = seq;
    WaitArgs w{}, w2{};
    int rc = fill_wait(m, w, CH_FIN, from, seq);
    if (!rc && m->pend) rc = fill_wait(m, w2, CH_TRACES, m->pend_from, m->pend_seq);
    m->pend = false;
    if (rc) return rc;
    if (!sg.n && !w.n && !w2.n) return DFLO_OK;
    xt_begin(p, p.M);
    hipLaunchKernelGGL(signal_wait_kernel, dim3(1), dim3(64), 0, p.M, sg, w, w2);
    xt_end(p, p.M);
    MHIP(m, hipGetLastError());
    return DFLO_OK;
  }
  // On the compute stream, where its producer (the step's reductions) and its consumers (the next step's kernels) are: no hop to
  // the comm stream and back (two event waits, ~10 us each, per step: LAB R6.12).  DFLO_DT_ON_COMM=1: round 5's route.
  const bool hop = m->dt_on_comm;
  hipStream_t st = hop ? p.C : p.M;
  if (hop) {
    MHIP(m, hipEventRecord(p.ev_fin[0], p.M));
    MHIP(m, hipStreamWaitEvent(p.C, p.ev_fin[0], 0));
  }
  void *slot = nullptr;
  MENG(m, p, dflo_hip_dt_slot(p.eng, &slot));
  if (m->x_allreduce) {
    if (m->x_allreduce(m->x_user, (double *)slot, 1, DFLO_REDUCE_MIN, (void *)st)) { set_err(m, "the host program's all-reduce callback failed"); return DFLO_ERR_COMM; }
  } else {
    MNCCL(m, g_rccl.AllReduce(slot, slot, 1, ncclDouble, ncclMin, m->comm, st));
  }
  if (hop) {
    MHIP(m, hipEventRecord(p.ev_dt, p.C));
    MHIP(m, hipStreamWaitEvent(p.M, p.ev_dt, 0));
  }
  return DFLO_OK;
}
int reduce_dt_phase(dflo_hip_multi *m, Part &p, int64_t step, int ph) {
  const int par = (int)(step & 1);
  MHIP(m, hipSetDevice(p.device));
  if (ph == 0) {
    bool foreign = false;
    for (Part &q : m->parts) foreign |= q.M != p.M;
    if (foreign) MHIP(m, hipEventRecord(p.ev_fin[par], p.M));
    p.sy->fin.store(++p.n_fin, std::memory_order_release);
    return DFLO_OK;
  }
  // the peers' reductions have written their minima into this part's table: the compute stream waits for them, and with it
  // everything of the next step (its comm stream starts behind the compute stream's "open")
  for (Part &q : m->parts) {
    if (&q == &p) continue;
    const int rc = wait_count(m, q.sy->fin, p.n_fin);
    if (rc) return rc;
    if (q.M != p.M) MHIP(m, hipStreamWaitEvent(p.M, q.ev_fin[par], 0));   // (same stream: q's reductions are ahead of this point)
  }
  return DFLO_OK;
}
int reduce_dt(dflo_hip_multi *m) {   // calling thread, all parts
  if (m->n_parts == 1 && !m->self_halo) return DFLO_OK;   // finalize_kernel has applied the rules already
  if (m->rank_mode) return reduce_dt_rank(m);
  const int64_t step = m->n_steps_fin++;
  for (int ph = 0; ph < 2; ++ph)
    for (Part &p : m->parts) {
      const int rc = reduce_dt_phase(m, p, step, ph);
      if (rc) return rc;
    }
  return DFLO_OK;
}

// one whole time step of a group's parts (the group's own thread): stages, old_solution = current_solution, the time step
// of the next
int group_step(dflo_hip_multi *m, Group &g, double dt, int64_t n0, int64_t step, bool peers) {
  const int xs = m->kxrcf ? 2 : 1;
  for (int rk = 0; rk < m->n_rk; ++rk) {
    const int rc = stage_groups(m, &g, 1, rk, dt, n0 + (int64_t)rk * xs, peers);
    if (rc) return rc;
  }
  for (int i : g.parts) MENG(m, m->parts[i], dflo_hip_end_step(m->parts[i].eng));
  if (m->n_parts == 1 && !m->self_halo) return DFLO_OK;
  for (int ph = 0; ph < 2; ++ph)
    for (int i : g.parts) {
      const int rc = reduce_dt_phase(m, m->parts[i], step, ph);
      if (rc) return rc;
    }
  return DFLO_OK;
}

// the compute stream catches up with the comm stream (before the state is read, or a call outside the overlapped stage)
int join_all(dflo_hip_multi *m) {
  for (Group &g : m->groups) {
    MHIP(m, hipSetDevice(g.device));
    if (m->pend) {   // fused delivery: a last stage's traces nobody has waited for yet
      m->pend = false;
      const int rc = wait_words(m, g.M, CH_TRACES, m->pend_from, m->pend_seq);
      if (rc) return rc;
    }
    if (g.unpack_pending) {
      MHIP(m, hipEventRecord(g.ev_unpack, g.C));
      MHIP(m, hipStreamWaitEvent(g.M, g.ev_unpack, 0));
      g.unpack_pending = false;
    }
    g.rim_pending = false;
  }
  return DFLO_OK;
}

int sync_all(dflo_hip_multi *m) {
  int rc = join_all(m);
  if (rc) return rc;
  for (Group &g : m->groups) {
    MHIP(m, hipSetDevice(g.device));
    MHIP(m, hipStreamSynchronize(g.C));
    MHIP(m, hipStreamSynchronize(g.M));
  }
  if (m->ipc_fail_host && *m->ipc_fail_host) {
    set_err(m, "a neighbour's records did not arrive within DFLO_IPC_TIMEOUT_S (DFLO_RANK_TRANSPORT=ipc: a rank died or fell out of step)");
    return DFLO_ERR_COMM;
  }
  return DFLO_OK;
}

// rank mode: all-reduce of a few host doubles through the device scratch (set-up and reporting calls only)
int host_allreduce(dflo_hip_multi *m, double *v, int n, ncclRedOp_t op) {
  if (!m->rank_mode || m->n_parts == 1) return DFLO_OK;
  Part &p = m->parts[0];
  MHIP(m, hipSetDevice(p.device));
  double *buf = m->scal;
  if (n > 8) MHIP(m, hipMalloc((void **)&buf, (size_t)n * sizeof(double)));   // (set-up only: the handle exchange)
  struct Scratch { double *b, *keep; ~Scratch() { if (b != keep) hipFree(b); } } scratch{buf, m->scal};
  MHIP(m, hipMemcpyAsync(buf, v, n * sizeof(double), hipMemcpyHostToDevice, p.C));
  if (m->x_allreduce) {
    const int xop = op == ncclMin ? DFLO_REDUCE_MIN : (op == ncclSum ? DFLO_REDUCE_SUM : DFLO_REDUCE_MAX);
    if (m->x_allreduce(m->x_user, buf, n, xop, (void *)p.C)) { set_err(m, "the host program's all-reduce callback failed"); return DFLO_ERR_COMM; }
  } else {
    MNCCL(m, g_rccl.AllReduce(buf, buf, n, ncclDouble, op, m->comm, p.C));
  }
  MHIP(m, hipMemcpyAsync(v, buf, n * sizeof(double), hipMemcpyDeviceToHost, p.C));
  MHIP(m, hipStreamSynchronize(p.C));
  return DFLO_OK;
}

// ---- DFLO_RANK_TRANSPORT=ipc: set-up.  Every rank exports its receive areas, its time-step table, its sequence words and its
// receive offsets; the records meet through one sum all-reduce of bytes spread over doubles (each rank fills its own stretch of
// a zeroed array: any transport that can sum doubles can carry it); every rank then opens what it will write.
// What is exported are TWO windows per rank, each one allocation of whole 2 MB blocks: the data window (ghost-trace tables, average
// and cell areas, the time-step table -- plain or, DFLO_PEER_FINEGRAINED=1, fine-grained device memory) and the sync window (the
// sequence words; always fine-grained).  Not the individual buffers: the runtime serves small allocations as fragments of shared
// 2 MB blocks, and a fragment can be exported only if it happens to start and end on 4 KB (found by tools/fuzz_ranks.py: "invalid
// argument" from hipIpcGetMemHandle for a 2.8 KB trace table, and -- worse -- a wrong state where the export went through).  The
// engine is pointed at its share of the window (dflo_hip_set_ghost_trace_buffers / _set_dt_table_buffer).
struct IpcExport {
  hipIpcMemHandle_t data, sync;
  uint64_t off[7];             // tg0 tg1 recv_a0 recv_a1 recv_u0 recv_u1 dt_table, bytes from the start of the data window
  int32_t has_tg;
  int32_t ro[17], rfo[17];     // this rank's receive offsets by source rank (cells / face traces)
};
constexpr size_t kWindowBlock = 2u << 20;
int alloc_flags(dflo_hip_multi *m) {
  Part &p = m->parts[0];
  MHIP(m, hipSetDevice(p.device));
  // fine-grained, not uncached (hipDeviceMallocUncached): the protocol runs as well on uncached words, but an exported uncached
  // block that is freed poisons the allocations of this process that follow (the first engine created after the driver computes a
  // wrong state; leaking the block instead of freeing it: clean) -- profiles/LAB.md R5.15
  // (DFLO_IPC_WORDS=uncached, developer switch: that configuration, to hold the destroy order of round 6 against it)
  MHIP(m, hipExtMallocWithFlags(&m->win_sync, 2 * kWindowBlock, dflo::read_tunables().ipc_words_uncached ? hipDeviceMallocUncached : hipDeviceMallocFinegrained));
  MHIP(m, hipMemset(m->win_sync, 0, 2 * kWindowBlock));
  m->flags = (unsigned long long *)m->win_sync;
  void *fh = nullptr, *fd = nullptr;
  MHIP(m, hipHostMalloc(&fh, sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
  MHIP(m, hipHostGetDevicePointer(&fd, fh, 0));
  m->ipc_fail_host = (volatile int *)fh;
  m->ipc_fail = (int *)fd;
  *m->ipc_fail_host = 0;
  m->ipc_ticks = (long long)dflo::read_tunables().ipc_timeout_s * 100000000LL;
  return DFLO_OK;
}
int setup_ipc(dflo_hip_multi *m) {
  Part &p = m->parts[0];
  if (m->n_parts > 16) { set_err(m, "DFLO_RANK_TRANSPORT=ipc: at most 16 ranks"); return DFLO_ERR_UNSUPPORTED; }
  // Every rank leaves this function with the same verdict: a rank whose LOCAL steps fail (an allocation, hipIpcGetMemHandle, a
  // neighbour's window that cannot be opened) still takes part in the two all-reduces below and says so in them, so that its
  // peers bail out with it instead of waiting in a later collective for a rank that has gone.
  IpcExport mine;
  std::memset(&mine, 0, sizeof(mine));
  auto export_windows = [&]() -> int {
  int rc = alloc_flags(m);
  if (rc) return rc;
  {   // the data window and this rank's buffers inside it
    auto up = [](size_t b) { return (b + 4095) & ~(size_t)4095; };
    const size_t ng = std::max(p.n_ghost, 1);
    const size_t ntg = p.trace ? up((size_t)std::max(dflo_hip_n_ghost_traces(p.eng), 1) * 4 * m->N * sizeof(double)) : 0;
    const size_t na = up(ng * 4 * sizeof(double)), nu = up(ng * (m->ndof + 4) * sizeof(double));
    const size_t sizes[7] = {ntg, ntg, na, na, nu, nu, 4096};
    size_t total = 0;
    for (int i = 0; i < 7; ++i) { mine.off[i] = total; total += sizes[i]; }
    total = std::max(((total + kWindowBlock - 1) / kWindowBlock) * kWindowBlock, 2 * kWindowBlock);
    // fine-grained by default: what a neighbour stores is then coherent at every access of this device, not only behind a kernel
    // boundary -- nothing rests on how this part treats remote writes into plain device memory (DFLO_PEER_FINEGRAINED=0: plain)
    m->ipc_fine = dflo::read_tunables().ipc_finegrained;
    if (m->ipc_fine) MHIP(m, hipExtMallocWithFlags(&m->win_data, total, hipDeviceMallocFinegrained));
    else MHIP(m, hipMalloc(&m->win_data, total));
    MHIP(m, hipMemset(m->win_data, 0, total));
    char *w = (char *)m->win_data;
    if (p.trace) {
      MENG(m, p, dflo_hip_set_ghost_trace_buffers(p.eng, w + mine.off[0], w + mine.off[1]));
      p.tg[0] = w + mine.off[0];
      p.tg[1] = w + mine.off[1];
      mine.has_tg = 1;
    }
    for (int i = 0; i < 2; ++i) {
      hipFree(p.recv_a[i]);
      hipFree(p.recv_u[i]);
      p.recv_a[i] = (double *)(w + mine.off[2 + i]);
      p.recv_u[i] = (double *)(w + mine.off[4 + i]);
    }
    m->recv_in_window = true;
    MENG(m, p, dflo_hip_set_dt_table_buffer(p.eng, w + mine.off[6]));
    p.dt_table = w + mine.off[6];
  }
  hipError_t e = hipIpcGetMemHandle(&mine.data, m->win_data);
  if (e == hipSuccess) e = hipIpcGetMemHandle(&mine.sync, m->win_sync);
  if (e != hipSuccess) { set_err(m, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e)); return DFLO_ERR_HIP; }
  for (int q = 0; q <= m->n_parts; ++q) {
    mine.ro[q] = p.recv_off[q];
    mine.rfo[q] = p.trace ? p.recvf_off[q] : 0;
  }
  return DFLO_OK;
  };
  const int rc_export = export_windows();
  const size_t nb = sizeof(IpcExport);
  std::vector<double> spread((size_t)m->n_parts * nb + m->n_parts, 0.0);   // the handles, and behind them one status word per rank
  const unsigned char *mb = (const unsigned char *)&mine;
  if (!rc_export)
    for (size_t i = 0; i < nb; ++i) spread[(size_t)p.index * nb + i] = (double)mb[i];
  spread[(size_t)m->n_parts * nb + p.index] = rc_export ? 1.0 : 0.0;
  int rc = host_allreduce(m, spread.data(), (int)spread.size(), ncclSum);
  if (rc) return rc;
  if (rc_export) return rc_export;
  for (int q = 0; q < m->n_parts; ++q)
    if (spread[(size_t)m->n_parts * nb + q] != 0.0) {
      set_err(m, "DFLO_RANK_TRANSPORT=ipc: rank " + std::to_string(q) + " could not export its windows");
      return DFLO_ERR_COMM;
    }
  m->pmap.assign(m->n_parts, PeerMap{});
  auto open_windows = [&]() -> int {
  hipError_t e = hipSuccess;
  for (int q = 0; q < m->n_parts; ++q) {
    if (q == p.index) continue;
    IpcExport theirs;
    unsigned char *tb = (unsigned char *)&theirs;
    for (size_t i = 0; i < nb; ++i) tb[i] = (unsigned char)spread[(size_t)q * nb + i];
    PeerMap &pm = m->pmap[q];
    const bool peer = std::find(p.peers.begin(), p.peers.end(), q) != p.peers.end();
    void *wd = nullptr, *ws = nullptr;
    e = hipIpcOpenMemHandle(&wd, theirs.data, hipIpcMemLazyEnablePeerAccess);
    if (e == hipSuccess) { pm.opened[0] = wd; e = hipIpcOpenMemHandle(&ws, theirs.sync, hipIpcMemLazyEnablePeerAccess); }
    if (e != hipSuccess) { set_err(m, std::string("hipIpcOpenMemHandle (rank ") + std::to_string(q) + "): " + hipGetErrorString(e)); return DFLO_ERR_COMM; }
    pm.opened[1] = ws;
    char *w = (char *)wd;
    if (theirs.has_tg) { pm.tg[0] = (double *)(w + theirs.off[0]); pm.tg[1] = (double *)(w + theirs.off[1]); }
    for (int i = 0; i < 2; ++i) {
      pm.recv_a[i] = (double *)(w + theirs.off[2 + i]);
      pm.recv_u[i] = (double *)(w + theirs.off[4 + i]);
    }
    pm.dt_table = (double *)(w + theirs.off[6]);
    pm.flags = (unsigned long long *)ws;
    pm.ro = theirs.ro[p.index];
    pm.rfo = theirs.rfo[p.index];
    if (peer && (theirs.ro[p.index + 1] - theirs.ro[p.index] != p.send_off[q + 1] - p.send_off[q] ||
                 (p.trace && theirs.rfo[p.index + 1] - theirs.rfo[p.index] != p.sendf_off[q + 1] - p.sendf_off[q] ) || (p.trace && !theirs.has_tg))) {
      set_err(m, "partition: rank " + std::to_string(q) + " expects another number of records than rank " + std::to_string(p.index) + " sends");
      return DFLO_ERR_COMM;
    }
  }
  return DFLO_OK;
  };
  const int rc_open = open_windows();
  double bad = rc_open ? 1.0 : 0.0;
  if ((rc = host_allreduce(m, &bad, 1, ncclMax))) return rc;
  if (rc_open) return rc_open;
  if (bad != 0.0) { set_err(m, "DFLO_RANK_TRANSPORT=ipc: another rank could not map its neighbours' windows"); return DFLO_ERR_COMM; }
  // every rank's CFL minimum goes into every rank's table (the engine's reductions write them, FinalArgs::peer_mins)
  void *tables[16] = {};
  for (int q = 0; q < m->n_parts; ++q) tables[q] = q == p.index ? p.dt_table : (void *)m->pmap[q].dt_table;
  MENG(m, p, dflo_hip_dt_exchange(p.eng, p.index, m->n_parts, tables));
  m->ipc = true;
  return DFLO_OK;
}
// IPC transport, after the mappings are known: can the stage kernel deliver by itself?  (only face traces travel -- no LxF flux, no
// TVB limiter reading ghost averages, no whole cells --, and no limiter pass sits between the update and the send)
int setup_fused(dflo_hip_multi *m) {
  Part &p = m->parts[0];
  m->fused = m->fused_tvb = false;
  if (!m->ipc || !dflo::read_tunables().ipc_fused || !p.trace || m->kxrcf || p.peers.empty()) return DFLO_OK;
  // with a TVB limiter: the pass has to walk the list of marked shards (so that the shards on a cut can be taken off it) and read
  // the ghost averages where they arrive; the LxF flux would want them in the engine's array as well
  const bool tvb_ok = m->tvb && m->avg_in_place && m->prm.flux_type != DFLO_FLUX_LXF && dflo_hip_limiter_walks_list(p.eng) && (int)m->parts.size() == 1;
  if (m->tvb ? !tvb_ok : (m->need_avg || m->sep_limiter)) return DFLO_OK;
  for (int par = 0; par < 2; ++par) {
    int32_t first[17];
    void *dst[16], *fl[16];
    int nseg = 0;
    for (int q : p.peers) {
      const int n = p.sendf_off[q + 1] - p.sendf_off[q];
      if (!n) continue;
      if (nseg == 16) return DFLO_OK;   // (more neighbours than the kernel's list holds: the two-stream schedule)
      first[nseg] = p.sendf_off[q];
      dst[nseg] = m->pmap[q].tg[par] + (size_t)m->pmap[q].rfo * 4 * m->N;
      fl[nseg++] = m->pmap[q].flags + flag_index(CH_TRACES, p.index);
    }
    if (!nseg) return DFLO_OK;
    first[nseg] = p.sendf_off[m->n_parts];
    MENG(m, p, dflo_hip_set_deliver(p.eng, par, nseg, first, dst, fl));
  }
  {   // this rank's own words for the neighbours' traces: the stage kernel's workgroups on the cut can wait for them themselves where
      // what they then read is coherent inside a running kernel -- fine-grained tables, or tables this device writes itself
    void *words[16];
    int nw = 0;
    for (int q : p.peers)
      if (p.recvf_off[q + 1] > p.recvf_off[q]) words[nw++] = m->flags + flag_index(CH_TRACES, q);
    MENG(m, p, dflo_hip_set_arrival_words(p.eng, nw, words, m->ipc_fail));
    m->kwait = dflo::read_tunables().ipc_kwait && (m->self_halo || m->ipc_fine);
    // (self-halo: the tables are this engine's own plain allocations, and it is its own, later kernels that read them)
    MENG(m, p, dflo_hip_deliver_to_plain_memory(p.eng, ((!m->self_halo && !m->ipc_fine) || dflo::read_tunables().ipc_strict) ? 1 : 0));
  }
  if (m->tvb) {   // the averages' way: into the neighbours' average areas, words of their own
    void *words[16];
    int nw = 0;
    for (int q : p.peers)
      if (p.recv_off[q + 1] > p.recv_off[q]) words[nw++] = m->flags + flag_index(CH_AVG, q);
    for (int par = 0; par < 2; ++par) {
      int32_t first[17];
      void *dst[16], *fl[16];
      int nseg = 0;
      for (int q : p.peers) {
        const int n = p.send_off[q + 1] - p.send_off[q];
        if (!n) continue;
        first[nseg] = p.send_off[q];
        dst[nseg] = m->pmap[q].recv_a[par] + (size_t)m->pmap[q].ro * 4;
        fl[nseg++] = m->pmap[q].flags + flag_index(CH_AVG, p.index);
      }
      first[nseg] = p.send_off[m->n_parts];
      MENG(m, p, dflo_hip_set_deliver_averages(p.eng, par, nseg, first, dst, fl, nw, words, m->ipc_fail));
    }
    m->fused_tvb = true;
    return DFLO_OK;
  }
  m->fused = true;
  return DFLO_OK;
}

// One part per process on the two-stream schedule: the compute stream is ordered behind the comm stream's rim launch by a word the
// interior launch waits for before it ends (dflo_hip_multi::tail_wait), not by a wait packet.  Where the interior launch exists (a part
// whose every shard is rim has none) and the rim's last kernel before the send is its update (not where a separate limiter pass follows).
int setup_tail_wait(dflo_hip_multi *m) {
  m->tail_wait = false;
  if (!dflo::read_tunables().tail_wait || m->parts.size() != 1 || !(m->rank_mode || m->self_halo) || m->fused || m->fused_tvb) return DFLO_OK;
  Part &p = m->parts[0];
  if (p.peers.empty() || m->basis != DFLO_BASIS_QK || m->kxrcf || (m->sep_limiter && !m->tvb)) return DFLO_OK;
  if (dflo_hip_n_part_shards(p.eng, m->tvb ? 4 : 2) == 0) return DFLO_OK;
  // (measured: the one-exchange TVB stage at k = 1 -- C3 against itself over RCCL -- is 7 % SLOWER with it, 120 300 -> 111 900 MDoF/s,
  //  while its two-exchange form gains 8 % and every other configuration 1.5-6 %.  That run is host-bound -- 126 us of host time for
  //  140 us of device time per step -- and the driver's calls to the comm stream take the host longer when the compute stream's
  //  kernels end later: LAB R6.16.  Off there unless forced.)
  if (m->tvb_one && m->N == 2 && !std::getenv("DFLO_TAIL_WAIT")) return DFLO_OK;
  if (!m->flags) {
    const int rc = alloc_flags(m);
    if (rc) return rc;
    MENG(m, p, dflo_hip_set_arrival_words(p.eng, 0, nullptr, m->ipc_fail));   // (the failure word of the waiting workgroup)
  }
  m->tail_wait = true;
  return DFLO_OK;
}

// a barrier of the ranks on the host (IPC transport: nobody may write into a receive area whose owner still reads an earlier run)
int ipc_barrier(dflo_hip_multi *m) {
  if (!m->ipc || m->n_parts == 1) return DFLO_OK;
  double one = 1.0;
  const int rc = host_allreduce(m, &one, 1, ncclSum);
  if (!rc) std::memcpy(m->ipc_post_met, m->ipc_post, sizeof(m->ipc_post_met));
  return rc;
}

// One process per GPU: every rank leaves a call of the driver with the same status.  `rc` is what this rank found (a limiter
// flag, a refused parameter, 0); the ranks exchange it and return the gravest.  A rank whose device or transport has failed
// cannot take part in that exchange: it tears the communicator down instead, which ends its peers' pending collectives with an
// error (where the host program's own transport is in use, it has to do the same for its peers -- MPI_Abort in dflo).
int agree(dflo_hip_multi *m, int rc) {
  if (!m->rank_mode || m->n_parts == 1) return rc;
  if (rc == DFLO_ERR_HIP || rc == DFLO_ERR_COMM || rc == DFLO_ERR_NOMEM) {
    if (m->comm && g_rccl.CommAbort) { g_rccl.CommAbort(m->comm); m->comm = nullptr; }
    m->fatal = true;
    return rc;
  }
  // [0] negative mean state, [1] positivity root failure, [2] any other status (as a positive number: the maximum is the gravest)
  double v[3] = {rc == DFLO_ERR_NEGATIVE_MEAN_STATE ? 1.0 : 0.0, rc == DFLO_ERR_POSITIVITY_NO_ROOT ? 1.0 : 0.0,
                 (rc && rc != DFLO_ERR_NEGATIVE_MEAN_STATE && rc != DFLO_ERR_POSITIVITY_NO_ROOT) ? (double)(-rc) : 0.0};
  const int rc2 = host_allreduce(m, v, 3, ncclMax);
  if (rc2) return rc2;
  if (v[2] > 0.0) {
    const int other = -(int)v[2];
    if (rc != other) set_err(m, "another rank stopped with status " + std::to_string(other));
    return other;
  }
  if (v[0] > 0.0) { if (rc != DFLO_ERR_NEGATIVE_MEAN_STATE) set_err(m, "Fatal: Negative states (on another rank)"); return DFLO_ERR_NEGATIVE_MEAN_STATE; }
  if (v[1] > 0.0) { if (rc != DFLO_ERR_POSITIVITY_NO_ROOT) set_err(m, "Problem in positivity limiter (on another rank)"); return DFLO_ERR_POSITIVITY_NO_ROOT; }
  return DFLO_OK;
}

// failure flags of all local parts
int check_local(dflo_hip_multi *m, bool synchronise) {
  int worst = DFLO_OK;
  for (Part &p : m->parts) {
    int rc;
    if (synchronise) rc = dflo_hip_check(p.eng);
    else { int64_t st = -1; dflo_hip_failure_step(p.eng, &st); rc = st >= 0 ? dflo_hip_check(p.eng) : DFLO_OK; }
    if (rc && rc != DFLO_ERR_NEGATIVE_MEAN_STATE && rc != DFLO_ERR_POSITIVITY_NO_ROOT) { set_err(m, dflo_hip_last_error(p.eng)); return rc; }
    if (rc && (!worst || rc > worst)) { worst = rc; set_err(m, std::string("part ") + std::to_string(p.index) + ": " + dflo_hip_last_error(p.eng)); }
  }
  return worst;
}
// ... and, in rank mode, of all ranks: every caller gets the same answer
int check_all(dflo_hip_multi *m, bool synchronise) { return agree(m, check_local(m, synchronise)); }

// the device groups of the local parts and their streams (parts[i].index / .device are set)
int make_groups(dflo_hip_multi *m) {
  const int mode = dflo::read_tunables().group;
  const int per_device = mode == 1 ? 1 << 20 : (mode == 2 ? 1 : 2);   // groups a device may have
  for (size_t i = 0; i < m->parts.size(); ++i) {
    Part &p = m->parts[i];
    int gi = -1, on_dev = 0, fewest = 1 << 30;
    for (size_t k = 0; k < m->groups.size(); ++k)
      if (m->groups[k].device == p.device) ++on_dev;
    if (on_dev >= per_device)   // the device has its groups: join the one with the fewest parts
      for (size_t k = 0; k < m->groups.size(); ++k)
        if (m->groups[k].device == p.device && (int)m->groups[k].parts.size() < fewest) { gi = (int)k; fewest = (int)m->groups[k].parts.size(); }
    if (gi < 0) {
      gi = (int)m->groups.size();
      m->groups.emplace_back();
      Group &g = m->groups.back();
      g.device = p.device;
      MHIP(m, hipSetDevice(g.device));
      MHIP(m, hipStreamCreate(&g.M));
      // the comm stream outranks the compute stream: its small kernels (rim shards, pack, unpack) go ahead of the queued
      // workgroups of the interior launch
      int lo = 0, hi = 0;
      MHIP(m, hipDeviceGetStreamPriorityRange(&lo, &hi));
      MHIP(m, hipStreamCreateWithPriority(&g.C, hipStreamDefault, dflo::read_tunables().comm_priority ? hi : lo));
      hipEvent_t *evs[] = {&g.ev_open, &g.ev_rim, &g.ev_rim_prev, &g.ev_ring, &g.ev_unpack, &g.ev_chunk[0], &g.ev_chunk[1]};
      for (hipEvent_t *ev : evs) MHIP(m, hipEventCreateWithFlags(ev, hipEventDisableTiming));
    }
    m->groups[gi].parts.push_back((int)i);
    p.group = gi;
    p.M = m->groups[gi].M;
    p.C = m->groups[gi].C;
  }
  return DFLO_OK;
}

int setup_part(dflo_hip_multi *m, Part &p, const dflo_mesh_t *mesh, const dflo_params_t *prm, int method) {
  int rc = m->self_halo ? dflo_mesh_partition_self(mesh, m->self_virtual, method, &p.sub, &p.send_cells, &p.send_off, &p.recv_off)
                         : dflo_mesh_partition_ex(mesh, m->n_parts, p.index, method, &p.sub, &p.send_cells, &p.send_off, &p.recv_off);
  if (rc) { m->err = dflo_mesh_last_error(); return rc; }
  p.n_cells = p.sub->n_cells;
  p.n_owned = p.sub->n_owned_cells;
  p.n_ghost = p.n_cells - p.n_owned;
  p.n_send = p.send_off[m->n_parts];
  for (int q = 0; q < m->n_parts; ++q)
    if ((q != p.index || m->self_halo) && (p.send_off[q + 1] > p.send_off[q] || p.recv_off[q + 1] > p.recv_off[q])) p.peers.push_back(q);
  {   // the cell size of a lattice of squares as the single engine's plan would find it on the whole mesh (plan.h: build_plan's h_hint)
    double hmin = 0.0;
    if (mesh->mapping == DFLO_MAP_CARTESIAN) {
      hmin = 1.0e300;
      for (int32_t c = 0; c < mesh->n_cells; ++c) hmin = std::min(hmin, mesh->cell_vertices[(size_t)c * 8 + 2] - mesh->cell_vertices[(size_t)c * 8]);
    }
    // (will the limiter pass take the exchange along?  what setup_fused decides later, as far as it can be known here)
    const bool pass_x = m->want_ipc && dflo::read_tunables().ipc_fused && m->tvb && !m->kxrcf && prm->flux_type != DFLO_FLUX_LXF &&
                        dflo::read_tunables().avg_in_place && mesh->basis == DFLO_BASIS_QK;
    rc = dflo_hip_create_with_cell_size(p.sub, prm, p.device, &p.eng, hmin, pass_x);
  }
  if (rc) { m->err = dflo_hip_last_error(nullptr); return rc; }
  MHIP(m, hipSetDevice(p.device));
  MENG(m, p, dflo_hip_set_stream(p.eng, p.M));   // (the streams are the group's, made by make_groups)
  hipEvent_t *evs[] = {&p.ev_dt, &p.ev_fin[0], &p.ev_fin[1]};
  for (hipEvent_t *e : evs) MHIP(m, hipEventCreateWithFlags(e, hipEventDisableTiming));
  for (int k = 0; k < 3; ++k)
    for (int i = 0; i < 2; ++i) {
      MHIP(m, hipEventCreateWithFlags(&p.ev_sent[k][i], hipEventDisableTiming));
      MHIP(m, hipEventCreateWithFlags(&p.ev_used[k][i], hipEventDisableTiming));
    }
  MENG(m, p, dflo_hip_set_send_cells(p.eng, p.n_send, p.send_cells));
  const size_t ns = std::max(p.n_send, 1), ng = std::max(p.n_ghost, 1);
  MHIP(m, hipMalloc((void **)&p.send_u, ns * (m->ndof + 20) * sizeof(double)));   // DoFs + cell average per cell (+ the neighbours' averages: tvb_one)
  MHIP(m, hipMalloc((void **)&p.send_a, ns * 4 * sizeof(double)));
  {   // the receive areas are written by the neighbours' kernels (stores over xGMI): plain device memory, coherent at the kernel
      // boundaries the schedule provides, or -- DFLO_PEER_FINEGRAINED=1, like the engine's trace and time-step tables -- fine-grained
    const bool fine = dflo::read_tunables().peer_finegrained;
    for (int i = 0; i < 2; ++i) {
      if (fine) {
        MHIP(m, hipExtMallocWithFlags((void **)&p.recv_u[i], ng * (m->ndof + 20) * sizeof(double), hipDeviceMallocFinegrained));
        MHIP(m, hipExtMallocWithFlags((void **)&p.recv_a[i], ng * 4 * sizeof(double), hipDeviceMallocFinegrained));
      } else {
        MHIP(m, hipMalloc((void **)&p.recv_u[i], ng * (m->ndof + 20) * sizeof(double)));
        MHIP(m, hipMalloc((void **)&p.recv_a[i], ng * 4 * sizeof(double)));
      }
    }
  }
  p.trace = dflo_hip_halo_traces(p.eng) != 0;
  if (p.trace) {
    // Which faces travel: (owned cell c, face f) whose neighbour across f is a ghost cell, i.e. owned by a peer -- listed
    // by peer, then cell, then face: the order in which the peer's plan numbers the traces of its ghost cells (ghost
    // cells are sorted by owner and global id, owned cells keep the global order)
    auto owner_of_ghost = [&](int g) {   // g: local index of a ghost cell
      int q = 0;
      while (p.recv_off[q + 1] <= g - p.n_owned) ++q;
      return q;
    };
    std::vector<std::vector<int32_t>> sc(m->n_parts), sf(m->n_parts);
    std::vector<int64_t> gkeys;   // (ghost cell, its face) pairs this part looks at
    for (int c = 0; c < p.n_owned; ++c)
      for (int f = 0; f < 4; ++f) {
        const int nb = p.sub->cell_face_neighbor[(size_t)c * 4 + f];
        if (nb < p.n_owned) continue;   // boundary, none, or an owned neighbour
        const int q = owner_of_ghost(nb);
        sc[q].push_back(c);
        sf[q].push_back(f);
        gkeys.push_back((int64_t)nb * 4 + (p.sub->cell_face_neighbor_face[(size_t)c * 4 + f] & 3));
      }
    std::sort(gkeys.begin(), gkeys.end());
    gkeys.erase(std::unique(gkeys.begin(), gkeys.end()), gkeys.end());
    if ((int)gkeys.size() != dflo_hip_n_ghost_traces(p.eng)) { m->err = "ghost traces: the driver and the engine's plan count differently"; return DFLO_ERR_COMM; }
    p.sendf_off.assign(m->n_parts + 1, 0);
    p.recvf_off.assign(m->n_parts + 1, 0);
    std::vector<int32_t> cells, faces;
    for (int q = 0; q < m->n_parts; ++q) {
      p.sendf_off[q + 1] = p.sendf_off[q] + (int32_t)sc[q].size();
      cells.insert(cells.end(), sc[q].begin(), sc[q].end());
      faces.insert(faces.end(), sf[q].begin(), sf[q].end());
    }
    for (int64_t k : gkeys) ++p.recvf_off[owner_of_ghost((int)(k / 4)) + 1];
    for (int q = 0; q < m->n_parts; ++q) p.recvf_off[q + 1] += p.recvf_off[q];
    MENG(m, p, dflo_hip_set_send_faces(p.eng, (int32_t)cells.size(), cells.data(), faces.data()));
    MHIP(m, hipMalloc((void **)&p.send_t, std::max<size_t>(cells.size(), 1) * 4 * m->N * sizeof(double)));
    MENG(m, p, dflo_hip_ghost_trace_buffer(p.eng, 0, &p.tg[0]));
    MENG(m, p, dflo_hip_ghost_trace_buffer(p.eng, 1, &p.tg[1]));
  }
  MENG(m, p, dflo_hip_scalar_ptrs(p.eng, &p.dt_ptr, &p.res_ptr));
  MENG(m, p, dflo_hip_dt_table(p.eng, &p.dt_table));
  if (m->rank_mode && (m->n_parts > 1 || m->self_halo)) MENG(m, p, dflo_hip_dt_exchange(p.eng, 0, 1, nullptr));   // one slot, all-reduced in place
  // the engine's boundary faces -> their numbers in the undivided mesh
  const int nb = dflo_hip_n_boundary_faces(p.eng);
  std::vector<int32_t> bc(std::max(nb, 1)), bf(std::max(nb, 1));
  MENG(m, p, dflo_hip_boundary_faces(p.eng, bc.data(), bf.data(), nullptr, nullptr));
  std::unordered_map<int64_t, int32_t> where;
  where.reserve(m->gb_cell.size() * 2);
  for (size_t b = 0; b < m->gb_cell.size(); ++b) where[(int64_t)m->gb_cell[b] * 4 + m->gb_face[b]] = (int32_t)b;
  p.bface_global.resize(nb);
  for (int b = 0; b < nb; ++b) {
    auto it = where.find(p.sub->cell_global_id[bc[b]] * 4 + bf[b]);
    if (it == where.end()) { m->err = "boundary face of a part not found in the undivided mesh"; return DFLO_ERR_COMM; }
    p.bface_global[b] = it->second;
  }
  return DFLO_OK;
}

int create_common(const dflo_mesh_t *mesh, const dflo_params_t *prm, dflo_hip_multi *m) {
  if (mesh->n_owned_cells != mesh->n_cells) { m->err = "the mesh handed to the multi-device driver must be the undivided one"; return DFLO_ERR_BAD_PARAM; }
  if (mesh->degree < 0 || mesh->degree > DFLO_MAX_DEGREE) { m->err = "degree must be 0..5"; return DFLO_ERR_BAD_PARAM; }
  m->prm = *prm;
  if (mesh->degree == 0) {   // the limiters return at once for piecewise constants (as in the engines)
    m->prm.limiter_type = DFLO_LIMITER_NONE;
    m->prm.pos_lim = 0;
    m->prm.shock_indicator = DFLO_IND_LIMITER;
  }
  prm = &m->prm;
  m->degree = mesh->degree;
  m->basis = mesh->basis;
  m->N = mesh->degree + 1;
  m->ndof = 4 * (mesh->basis == DFLO_BASIS_PK ? m->N * (m->N + 1) / 2 : m->N * m->N);
  m->n_cells_global = mesh->n_cells;
  m->tvb = prm->limiter_type == DFLO_LIMITER_TVB;
  m->kxrcf = m->tvb && prm->shock_indicator != DFLO_IND_LIMITER;
  m->limited = m->tvb || prm->pos_lim;
  // boundary faces of the undivided mesh in MeshWorker order, and their quadrature points (src/assemble_explicit.cc:163-165)
  const dflo::BasisTables bt = dflo::make_basis(mesh->degree);
  for (int32_t c = 0; c < mesh->n_cells; ++c)
    for (int f = 0; f < 4; ++f) {
      const int32_t nb = mesh->cell_face_neighbor[(size_t)c * 4 + f];
      if (nb >= 0 || nb == DFLO_NBR_NONE) continue;
      m->gb_cell.push_back(c);
      m->gb_face.push_back(f);
      m->gb_id.push_back(DFLO_NBR_BOUNDARY_ID(nb));
      const double *v = &mesh->cell_vertices[(size_t)c * 8];
      for (int q = 0; q < m->N; ++q) {
        const double s = bt.x[q];
        const double xi = f == 0 ? 0.0 : (f == 1 ? 1.0 : s), eta = f == 2 ? 0.0 : (f == 3 ? 1.0 : s);
        for (int d = 0; d < 2; ++d)
          m->gb_xy.push_back((1 - xi) * (1 - eta) * v[d] + xi * (1 - eta) * v[2 + d] + (1 - xi) * eta * v[4 + d] + xi * eta * v[6 + d]);
      }
    }
  return DFLO_OK;
}

// Does every cell on a cut border on ONE other part only?  (What the one-exchange TVB stage needs: a ghost cell's neighbours are
// then cells of the receiver or of the ghost's owner, whose averages the owner can send along.  A function of the mesh, the number
// of parts and the partitioner alone: every rank finds the same answer without asking the others.)
void find_one_neighbour(dflo_hip_multi *m, const dflo_mesh_t *mesh, int method) {
  m->one_neighbour = true;
  if (m->self_halo || m->n_parts < 2 || !m->tvb) return;
  std::vector<int32_t> owner(mesh->n_cells);
  if (dflo_mesh_partition_owners(mesh, m->n_parts, method, owner.data())) { m->one_neighbour = false; return; }
  for (int32_t c = 0; c < mesh->n_cells && m->one_neighbour; ++c) {
    int other = -1;
    for (int f = 0; f < 4; ++f) {
      const int32_t nb = mesh->cell_face_neighbor[(size_t)c * 4 + f];
      if (nb < 0 || owner[nb] == owner[c]) continue;
      if (other >= 0 && owner[nb] != other) m->one_neighbour = false;
      other = owner[nb];
    }
  }
}

void finish_setup(dflo_hip_multi *m) {
  m->n_rk = dflo_hip_n_rk(m->parts[0].eng);
  // who reads the average of a ghost cell: the LxF flux (lambda from the cell averages, src/equation.h:357-359) and the TVB
  // limiter's differences (src/limiter.cc:284-317); without either the 4-double average message is not sent at all
  m->need_avg = m->prm.flux_type == DFLO_FLUX_LXF || m->tvb;
  m->avg_in_place = m->tvb && !m->kxrcf && m->prm.flux_type != DFLO_FLUX_LXF && dflo::read_tunables().avg_in_place;
  for (Part &p : m->parts) m->avg_in_place = m->avg_in_place && p.trace;
  // is there a limiter pass of its own between update and pack?  (positivity alone on Qk is applied inside the stage kernel
  // unless DFLO_FUSE_POS=0 says otherwise)
  const bool fused = m->prm.pos_lim && !m->tvb && m->basis == DFLO_BASIS_QK && dflo::read_tunables().fuse_pos;
  m->sep_limiter = m->limited && !fused;
  m->dt_on_comm = dflo::read_tunables().dt_on_comm;
  {
    const int want = dflo::read_tunables().tvb_one_exchange;   // unset: where an exchange is a library call
    m->tvb_one = m->tvb && !m->kxrcf && !m->ipc && !m->want_ipc && m->one_neighbour && (want < 0 ? (m->rank_mode || m->loopback) : want != 0);
  }
  for (Part &p : m->parts) m->tvb_one = m->tvb_one && p.trace;
  if (m->tvb_one) m->avg_in_place = false;   // (the ghost averages come out of the records into the engine's array)
}

}  // namespace

extern "C" {

const char *dflo_hip_multi_last_error(dflo_hip_multi_handle m) { return m ? m->err.c_str() : g_multi_error.c_str(); }

int dflo_hip_multi_destroy(dflo_hip_multi_handle m) {
  if (!m) return DFLO_OK;
  stop_workers(m);
  for (Group &g : m->groups) {
    hipSetDevice(g.device);
    if (g.C) hipStreamSynchronize(g.C);

    if (g.M) hipStreamSynchronize(g.M);
  }
  // IPC transport: a neighbour's last launches may still be storing into this rank's windows (its time-step minimum and the word of
  // the step that has just ended are awaited only by a next step).  Nobody unmaps or frees before everybody's streams are idle.
  // (a collective like the create call: skipped where this rank has seen a failure -- its peers have been told or are lost anyway)
  const bool meet = m->ipc && !m->self_halo && m->n_parts > 1 && m->created && !m->fatal && !(m->ipc_fail_host && *m->ipc_fail_host) &&
                    (m->comm || m->x_allreduce) &&
                    std::memcmp(m->ipc_post_met, m->ipc_post, sizeof(m->ipc_post_met)) != 0;   // (a driver that has never sent anything -- some rank's create failed -- just leaves)
  if (meet) ipc_barrier(m);
  if (!m->self_halo)
    for (PeerMap &pm : m->pmap)
      for (void *o : pm.opened)
        if (o) hipIpcCloseMemHandle(o);
  // ... and nobody FREES a window before every neighbour has closed its mapping of it: freeing exported memory that an importer still
  // maps is undefined (the IPC contract of the runtime: close in the importing process first).  Round 5 freed without this second
  // meeting.  (NOT the cause of LAB R5.15 -- uncached sequence words fail with either order, profiles/r06/r515_free_order.txt, LAB R6.3 --
  // but a violation all the same; tests/test_gpu_multi_ranks.py: 200 create / run / destroy cycles, each followed by a single engine
  // that takes the freed blocks.)  DFLO_IPC_FREE_EARLY=1 (developer switch): round 5's order.
  if (meet && !dflo::read_tunables().ipc_free_early) ipc_barrier(m);
  if (m->comm) g_rccl.CommDestroy(m->comm);
  if (m->ipc_fail_host) hipHostFree((void *)m->ipc_fail_host);
  for (Part &p : m->parts) {
    hipSetDevice(p.device);
    if (p.eng) dflo_hip_destroy(p.eng);
    hipFree(p.send_u); hipFree(p.send_a); hipFree(p.send_t);
    if (!m->recv_in_window)
      for (int i = 0; i < 2; ++i) { hipFree(p.recv_u[i]); hipFree(p.recv_a[i]); }
    hipEvent_t evs[] = {p.ev_dt, p.ev_fin[0], p.ev_fin[1]};
    for (hipEvent_t e : evs) if (e) hipEventDestroy(e);
    for (int k = 0; k < 3; ++k)
      for (int i = 0; i < 2; ++i) {
        if (p.ev_sent[k][i]) hipEventDestroy(p.ev_sent[k][i]);
        if (p.ev_used[k][i]) hipEventDestroy(p.ev_used[k][i]);
      }
    if (p.sub) dflo_mesh_free(p.sub);
    for (auto &e : p.x_pool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  }
  for (Group &g : m->groups) {
    hipSetDevice(g.device);
    hipEvent_t evs[] = {g.ev_open, g.ev_rim, g.ev_rim_prev, g.ev_ring, g.ev_unpack, g.ev_chunk[0], g.ev_chunk[1]};
    for (hipEvent_t e : evs) if (e) hipEventDestroy(e);
    if (g.C) hipStreamDestroy(g.C);

    if (g.M) hipStreamDestroy(g.M);
  }
  if (m->scal) hipFree(m->scal);
  if (m->win_data) hipFree(m->win_data);   // (behind the engines, which were pointed into it)
  if (m->win_sync && std::getenv("DFLO_IPC_SCRUB")) {   // (developer switch, LAB R6.3: hand the words back as zeros)
    hipMemset(m->win_sync, 0, 2 * kWindowBlock);
    hipDeviceSynchronize();
  }
  if (m->win_sync) hipFree(m->win_sync);
  for (int i = 0; i < 2; ++i) if (m->ev_chunk[i]) hipEventDestroy(m->ev_chunk[i]);
  delete m;
  return DFLO_OK;
}

int dflo_hip_multi_create(const dflo_mesh_t *mesh, const dflo_params_t *params, int n_devices, const int *device_ids,
                          int partitioner, dflo_hip_multi_handle *out) {
  if (!mesh || !params || !out || n_devices < 1 || n_devices > 16 || !device_ids) { g_multi_error = "bad arguments"; return DFLO_ERR_BAD_PARAM; }
  *out = nullptr;
  dflo_hip_multi *m = new dflo_hip_multi;
  auto bail = [&](int rc) { g_multi_error = m->err; dflo_hip_multi_destroy(m); return rc; };
  m->n_parts = n_devices;
  int rc = create_common(mesh, params, m);
  if (rc) return bail(rc);
  find_one_neighbour(m, mesh, partitioner);
  const dflo::Tunables tun = dflo::read_tunables();
  m->loopback = tun.loopback;
  m->strict = tun.strict;
  m->parts.resize(n_devices);
  m->sync.reset(new Sync[n_devices]);
  for (int i = 0; i < n_devices; ++i) {
    m->parts[i].index = i;
    m->parts[i].device = device_ids[i];
    m->parts[i].sy = &m->sync[i];
  }
  m->direct = !tun.copy;
  if ((rc = make_groups(m))) return bail(rc);
  for (int i = 0; i < n_devices; ++i)
    if ((rc = setup_part(m, m->parts[i], mesh, params, partitioner))) return bail(rc);
  // what one part sends is what the other expects, and peers are mutual (the receive areas' reuse rests on it, see post)
  for (Part &p : m->parts)
    for (int q : p.peers) {
      Part &o = m->parts[q];
      if (std::find(o.peers.begin(), o.peers.end(), p.index) == o.peers.end()) { m->err = "partition: part " + std::to_string(p.index) + " lists part " + std::to_string(q) + " as a neighbour but not the other way round"; return bail(DFLO_ERR_COMM); }
      if (p.send_off[q + 1] - p.send_off[q] != o.recv_off[p.index + 1] - o.recv_off[p.index]) { m->err = "partition: send and receive counts disagree"; return bail(DFLO_ERR_COMM); }
      if (p.trace && p.sendf_off[q + 1] - p.sendf_off[q] != o.recvf_off[p.index + 1] - o.recvf_off[p.index]) { m->err = "partition: face-trace send and receive counts disagree"; return bail(DFLO_ERR_COMM); }
    }
  // the engines read each other's time-step slots (and hipMemcpyPeerAsync goes direct) over xGMI
  for (Part &p : m->parts)
    for (Part &q : m->parts) {
      if (p.device == q.device) continue;
      int can = 0;
      hipDeviceCanAccessPeer(&can, p.device, q.device);
      if (!can) { m->err = "devices " + std::to_string(p.device) + " and " + std::to_string(q.device) + " are not peer-accessible"; return bail(DFLO_ERR_HIP); }
      hipSetDevice(p.device);
      const hipError_t e = hipDeviceEnablePeerAccess(q.device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { m->err = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e); return bail(DFLO_ERR_HIP); }
      (void)hipGetLastError();
    }
  if (n_devices > 1) {   // every part writes its CFL minimum into every part's table
    for (Part &p : m->parts) {
      void *tables[16] = {};
      for (Part &q : m->parts) tables[q.index] = q.dt_table;
      if ((rc = dflo_hip_dt_exchange(p.eng, p.index, n_devices, tables))) { m->err = dflo_hip_last_error(p.eng); return bail(rc); }
    }
  }
  if (m->loopback) {
    if (!load_rccl(m->err)) return bail(DFLO_ERR_COMM);
    ncclUniqueId id;
    hipSetDevice(m->parts[0].device);
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) { m->err = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return bail(DFLO_ERR_COMM); }
    r = g_rccl.CommInitRank(&m->comm, 1, id, 0);
    if (r != ncclSuccess) { m->err = std::string("ncclCommInitRank (loopback): ") + g_rccl.GetErrorString(r); return bail(DFLO_ERR_COMM); }
  }
  finish_setup(m);
  {  // one host thread per device group (DFLO_MULTI_THREADS=0: the calling thread drives them all); RCCL group calls on one
     // communicator must not come from several threads at once, so the loopback test transport stays on the calling thread
    if (m->groups.size() > 1 && !m->loopback && tun.threads) {
      for (size_t i = 0; i < m->groups.size(); ++i) {
        m->workers.emplace_back(new Worker);
        Worker *w = m->workers.back().get();
        w->th = std::thread(worker_main, w);
      }
    }
  }
  *out = m;
  return DFLO_OK;
}

int dflo_hip_comm_unique_id(void *id128) {
  if (!id128) return DFLO_ERR_BAD_PARAM;
  if (!load_rccl(g_multi_error)) return DFLO_ERR_COMM;
  static_assert(sizeof(ncclUniqueId) == DFLO_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  const ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) { g_multi_error = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return DFLO_ERR_COMM; }
  std::memcpy(id128, &id, sizeof(id));
  return DFLO_OK;
}

static int create_rank_impl(const dflo_mesh_t *mesh, const dflo_params_t *params, int device_id, int rank, int n_ranks,
                            const void *unique_id, dflo_exchange_fn xf, dflo_allreduce_fn af, void *user, int partitioner,
                            dflo_hip_multi_handle *out) {
  if (!mesh || !params || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks || (n_ranks > 1 && !unique_id && !(xf && af))) { g_multi_error = "bad arguments"; return DFLO_ERR_BAD_PARAM; }
  *out = nullptr;
  dflo_hip_multi *m = new dflo_hip_multi;
  m->x_exchange = xf;
  m->x_allreduce = af;
  m->x_user = user;
  auto bail = [&](int rc) { g_multi_error = m->err; dflo_hip_multi_destroy(m); return rc; };
  m->n_parts = n_ranks;
  m->rank = rank;
  m->rank_mode = true;
  int rc = create_common(mesh, params, m);
  if (rc) return bail(rc);
  find_one_neighbour(m, mesh, partitioner);
  m->parts.resize(1);
  m->sync.reset(new Sync[1]);
  m->parts[0].index = rank;
  m->parts[0].device = device_id;
  m->parts[0].sy = &m->sync[0];
  m->want_ipc = n_ranks > 1 && dflo::read_tunables().rank_transport == 1;
  if ((rc = make_groups(m))) return bail(rc);
  if ((rc = setup_part(m, m->parts[0], mesh, params, partitioner))) return bail(rc);
  {  // peers are mutual here too: what this rank sends to q, q expects, and the other way round -- both follow from one
     // deterministic partition of the same mesh on every rank; this rank can check its own half
    Part &p = m->parts[0];
    for (int q : p.peers)
      if ((p.send_off[q + 1] > p.send_off[q]) != (p.recv_off[q + 1] > p.recv_off[q])) { m->err = "partition: rank " + std::to_string(rank) + " and rank " + std::to_string(q) + " are not mutual neighbours"; return bail(DFLO_ERR_COMM); }
  }
  if (n_ranks > 1) {
    if (hipSetDevice(device_id) != hipSuccess) { m->err = "hipSetDevice failed"; return bail(DFLO_ERR_HIP); }
    if (!xf) {
      if (!load_rccl(m->err)) return bail(DFLO_ERR_COMM);
      ncclUniqueId id;
      std::memcpy(&id, unique_id, sizeof(id));
      const ncclResult_t r = g_rccl.CommInitRank(&m->comm, n_ranks, id, rank);
      if (r != ncclSuccess) { m->err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); return bail(DFLO_ERR_COMM); }
    }
    if (hipMalloc((void **)&m->scal, 8 * sizeof(double)) != hipSuccess) { m->err = "hipMalloc(scratch) failed"; return bail(DFLO_ERR_NOMEM); }
    if (dflo::read_tunables().rank_transport == 1 && (rc = setup_ipc(m))) return bail(rc);
  }
  finish_setup(m);
  if ((rc = setup_fused(m)) || (rc = setup_tail_wait(m))) return bail(rc);
  m->created = true;
  *out = m;
  return DFLO_OK;
}

int dflo_hip_multi_create_rank(const dflo_mesh_t *mesh, const dflo_params_t *params, int device_id, int rank, int n_ranks,
                               const void *unique_id, int partitioner, dflo_hip_multi_handle *out) {
  return create_rank_impl(mesh, params, device_id, rank, n_ranks, unique_id, nullptr, nullptr, nullptr, partitioner, out);
}

int dflo_hip_multi_create_rank_custom(const dflo_mesh_t *mesh, const dflo_params_t *params, int device_id, int rank, int n_ranks,
                                      dflo_exchange_fn exchange, dflo_allreduce_fn allreduce, void *user, int partitioner,
                                      dflo_hip_multi_handle *out) {
  if (!exchange || !allreduce) { g_multi_error = "both callbacks are needed"; return DFLO_ERR_BAD_PARAM; }
  return create_rank_impl(mesh, params, device_id, rank, n_ranks, nullptr, exchange, allreduce, user, partitioner, out);
}

int dflo_hip_multi_create_self(const dflo_mesh_t *mesh, const dflo_params_t *params, int device_id, int n_virtual, int partitioner,
                               int transport, dflo_hip_multi_handle *out) {
  if (!mesh || !params || !out || n_virtual < 1 || transport < DFLO_SELF_DIRECT || transport > DFLO_SELF_IPC) { g_multi_error = "bad arguments"; return DFLO_ERR_BAD_PARAM; }
  *out = nullptr;
  dflo_hip_multi *m = new dflo_hip_multi;
  auto bail = [&](int rc) { g_multi_error = m->err; dflo_hip_multi_destroy(m); return rc; };
  m->n_parts = 1;
  m->self_halo = true;
  m->self_virtual = n_virtual;
  m->rank_mode = transport == DFLO_SELF_RCCL || transport == DFLO_SELF_IPC;
  m->direct = transport != DFLO_SELF_COPY;
  m->want_ipc = transport == DFLO_SELF_IPC;
  m->strict = dflo::read_tunables().strict;
  int rc = create_common(mesh, params, m);
  if (rc) return bail(rc);
  m->parts.resize(1);
  m->sync.reset(new Sync[1]);
  m->parts[0].index = 0;
  m->parts[0].device = device_id;
  m->parts[0].sy = &m->sync[0];
  if ((rc = make_groups(m))) return bail(rc);
  if ((rc = setup_part(m, m->parts[0], mesh, params, partitioner))) return bail(rc);
  Part &p = m->parts[0];
  if (p.send_off[1] != p.recv_off[1] || (p.trace && p.sendf_off[1] != p.recvf_off[1])) { m->err = "self-halo: send and receive counts disagree"; return bail(DFLO_ERR_COMM); }
  if (transport == DFLO_SELF_IPC) {   // the sequence-word transport against itself: the "mapped" areas are its own
    if ((rc = alloc_flags(m))) return bail(rc);
    m->pmap.assign(1, PeerMap{});
    PeerMap &pm = m->pmap[0];
    for (int i = 0; i < 2; ++i) { pm.tg[i] = (double *)p.tg[i]; pm.recv_a[i] = p.recv_a[i]; pm.recv_u[i] = p.recv_u[i]; }
    pm.dt_table = (double *)p.dt_table;
    pm.flags = m->flags;
    pm.ro = pm.rfo = 0;
    void *tables[16] = {};
    tables[0] = p.dt_table;
    if ((rc = dflo_hip_dt_exchange(p.eng, 0, 1, tables))) { m->err = dflo_hip_last_error(p.eng); return bail(rc); }
    m->ipc = true;
  } else if (m->rank_mode) {   // the communicator of one rank: ncclSend / ncclRecv to itself, ncclAllReduce over itself
    if (hipSetDevice(device_id) != hipSuccess) { m->err = "hipSetDevice failed"; return bail(DFLO_ERR_HIP); }
    if (!load_rccl(m->err)) return bail(DFLO_ERR_COMM);
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) { m->err = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return bail(DFLO_ERR_COMM); }
    r = g_rccl.CommInitRank(&m->comm, 1, id, 0);
    if (r != ncclSuccess) { m->err = std::string("ncclCommInitRank (self-halo): ") + g_rccl.GetErrorString(r); return bail(DFLO_ERR_COMM); }
  } else {   // one-process schedule: the engine forms the time step from a table of one slot, its own
    void *tables[16] = {};
    tables[0] = p.dt_table;
    if ((rc = dflo_hip_dt_exchange(p.eng, 0, 1, tables))) { m->err = dflo_hip_last_error(p.eng); return bail(rc); }
  }
  finish_setup(m);
  if ((rc = setup_fused(m)) || (rc = setup_tail_wait(m))) return bail(rc);
  m->created = true;
  *out = m;
  return DFLO_OK;
}

int dflo_hip_multi_n_local(dflo_hip_multi_handle m) { return m ? (int)m->parts.size() : 0; }
int dflo_hip_multi_n_parts(dflo_hip_multi_handle m) { return m ? m->n_parts : 0; }
dflo_hip_handle dflo_hip_multi_engine(dflo_hip_multi_handle m, int i) { return (m && i >= 0 && i < (int)m->parts.size()) ? m->parts[i].eng : nullptr; }
int64_t dflo_hip_multi_n_dofs(dflo_hip_multi_handle m) { return m ? m->n_cells_global * m->ndof : 0; }
int64_t dflo_hip_multi_n_owned_dofs(dflo_hip_multi_handle m) {
  int64_t n = 0;
  if (m) for (Part &p : m->parts) n += (int64_t)p.n_owned * m->ndof;
  return n;
}
int32_t dflo_hip_multi_n_rk(dflo_hip_multi_handle m) { return m ? m->n_rk : 0; }

int dflo_hip_multi_part_cells(dflo_hip_multi_handle m, int i, int32_t *n_owned, int32_t *n_ghost, const int64_t **global_ids) {
  if (!m || i < 0 || i >= (int)m->parts.size()) return DFLO_ERR_BAD_PARAM;
  if (n_owned) *n_owned = m->parts[i].n_owned;
  if (n_ghost) *n_ghost = m->parts[i].n_ghost;
  if (global_ids) *global_ids = m->parts[i].sub->cell_global_id;
  return DFLO_OK;
}

int dflo_hip_multi_set_solution(dflo_hip_multi_handle m, const double *u) {
  if (!m || !u) return DFLO_ERR_BAD_PARAM;
  int rc = sync_all(m);
  if (rc) return rc;
  std::vector<double> loc;
  for (Part &p : m->parts) {
    loc.resize((size_t)p.n_cells * m->ndof);
    for (int c = 0; c < p.n_cells; ++c)   // owned and ghost cells alike: the ghosts start with their owners' values
      std::memcpy(&loc[(size_t)c * m->ndof], &u[(size_t)p.sub->cell_global_id[c] * m->ndof], m->ndof * sizeof(double));
    MENG(m, p, dflo_hip_set_solution(p.eng, loc.data()));
  }
  if ((rc = ipc_barrier(m))) return rc;   // (every rank has drained: nobody still reads what the first exchange will overwrite)
  // the engines read their trace table 0 now: the first exchange fills table 1.  Nothing is in flight (every stream has been
  // drained): the exchange counters start again.
  m->nx = 0;
  for (Part &p : m->parts) {
    for (int k = 0; k < 3; ++k) p.n_post[k] = p.n_arr[k] = 0;
    p.sy->reset();
    p.sy->fin.store(p.n_fin);
  }
  return DFLO_OK;
}

const dflo_mesh_t *dflo_hip_multi_part_mesh(dflo_hip_multi_handle m, int i) {
  return (m && i >= 0 && i < (int)m->parts.size()) ? m->parts[i].sub : nullptr;
}

int dflo_hip_multi_set_part_solution(dflo_hip_multi_handle m, int i, const double *u_part) {
  if (!m || !u_part || i < 0 || i >= (int)m->parts.size()) return DFLO_ERR_BAD_PARAM;
  int rc = sync_all(m);
  if (rc) return rc;
  // The state of ONE part (owned and ghost cells) is replaced; the exchange count and with it the roles of the receive areas
  // are those of the whole driver and stay.  The engine has filled both of its trace tables from the ghost cells' DoFs; it is
  // pointed at the one the schedule reads now.  Ghost copies of this part's cells held by OTHER parts are not touched: a
  // caller that changes the state sets every part before the next step (as bench.py's ranks do).
  MENG(m, m->parts[i], dflo_hip_set_solution(m->parts[i].eng, u_part));
  MENG(m, m->parts[i], dflo_hip_use_ghost_traces(m->parts[i].eng, (int)(m->nx & 1)));
  return ipc_barrier(m);
}

int dflo_hip_multi_get_solution(dflo_hip_multi_handle m, double *u) {
  if (!m || !u) return DFLO_ERR_BAD_PARAM;
  int rc = sync_all(m);
  if (rc) return rc;
  std::vector<double> loc;
  for (Part &p : m->parts) {
    loc.resize((size_t)p.n_cells * m->ndof);
    MENG(m, p, dflo_hip_get_solution(p.eng, loc.data()));
    for (int c = 0; c < p.n_owned; ++c)
      std::memcpy(&u[(size_t)p.sub->cell_global_id[c] * m->ndof], &loc[(size_t)c * m->ndof], m->ndof * sizeof(double));
  }
  return DFLO_OK;
}

int dflo_hip_multi_get_cell_average(dflo_hip_multi_handle m, double *avg) {
  if (!m || !avg) return DFLO_ERR_BAD_PARAM;
  int rc = sync_all(m);
  if (rc) return rc;
  std::vector<double> loc;
  for (Part &p : m->parts) {
    loc.resize((size_t)p.n_cells * 4);
    MENG(m, p, dflo_hip_get_cell_average(p.eng, loc.data()));
    for (int c = 0; c < p.n_owned; ++c) std::memcpy(&avg[(size_t)p.sub->cell_global_id[c] * 4], &loc[(size_t)c * 4], 4 * sizeof(double));
  }
  return DFLO_OK;
}

int32_t dflo_hip_multi_n_boundary_faces(dflo_hip_multi_handle m) { return m ? (int32_t)m->gb_cell.size() : 0; }

int dflo_hip_multi_boundary_faces(dflo_hip_multi_handle m, int32_t *cell, int32_t *face, int32_t *boundary_id, double *xy) {
  if (!m) return DFLO_ERR_BAD_PARAM;
  const size_t n = m->gb_cell.size();
  if (cell) std::memcpy(cell, m->gb_cell.data(), n * sizeof(int32_t));
  if (face) std::memcpy(face, m->gb_face.data(), n * sizeof(int32_t));
  if (boundary_id) std::memcpy(boundary_id, m->gb_id.data(), n * sizeof(int32_t));
  if (xy) std::memcpy(xy, m->gb_xy.data(), m->gb_xy.size() * sizeof(double));
  return DFLO_OK;
}

int dflo_hip_multi_set_boundary_values(dflo_hip_multi_handle m, int which, const double *values) {
  if (!m || !values || which < 0 || which > 1) return DFLO_ERR_BAD_PARAM;
  const size_t row = (size_t)m->N * 4;
  std::vector<double> loc;
  for (Part &p : m->parts) {
    if (p.bface_global.empty()) continue;
    loc.resize(p.bface_global.size() * row);
    for (size_t b = 0; b < p.bface_global.size(); ++b) std::memcpy(&loc[b * row], &values[(size_t)p.bface_global[b] * row], row * sizeof(double));
    MENG(m, p, dflo_hip_set_boundary_values(p.eng, which, loc.data()));
  }
  return DFLO_OK;
}

int dflo_hip_multi_set_boundary_program(dflo_hip_multi_handle m, int32_t boundary_id, int32_t component, int32_t n_ops, const int32_t *ops,
                                        int32_t n_consts, const double *consts) {
  if (!m) return DFLO_ERR_BAD_PARAM;
  for (Part &p : m->parts) MENG(m, p, dflo_hip_set_boundary_program(p.eng, boundary_id, component, n_ops, ops, n_consts, consts));
  return DFLO_OK;
}

int dflo_hip_multi_compute_dt(dflo_hip_multi_handle m, double elapsed_time, double *dt) {
  if (!m || !dt) return DFLO_ERR_BAD_PARAM;
  int rc = join_all(m);
  if (rc) return rc;
  double best = 1.0e300;
  int lrc = DFLO_OK;
  for (Part &p : m->parts) {
    double d = 0.0;
    if ((lrc = dflo_hip_compute_dt(p.eng, elapsed_time, &d))) { set_err(m, std::string("part ") + std::to_string(p.index) + ": " + dflo_hip_last_error(p.eng)); break; }
    best = std::min(best, d);   // the rules (cap by time_step, clip to final_time) are monotone: they commute with the minimum
  }
  if (m->rank_mode && m->n_parts > 1) {   // a rank that failed still takes part in the reduction (a negative value tells the others)
    if (lrc == DFLO_ERR_HIP || lrc == DFLO_ERR_COMM || lrc == DFLO_ERR_NOMEM) return agree(m, lrc);
    if (lrc) best = -1.0;
    if ((rc = host_allreduce(m, &best, 1, ncclMin))) return rc;
    if (best < 0.0) {
      if (!lrc) set_err(m, "the time step could not be formed on another rank");
      return lrc ? lrc : DFLO_ERR_COMM;
    }
  } else if (lrc) {
    return lrc;
  }
  *dt = best;
  return DFLO_OK;
}

// results of a step the driver reports: ||rhs|| of the first and of the last stage, summed over parts and ranks
static int step_norms(dflo_hip_multi *m, double *res_norm0, double *res_norm) {
  double tot[4] = {0, 0, 0, 0};
  for (Part &p : m->parts) {
    double r[4];
    MHIP(m, hipSetDevice(p.device));
    MHIP(m, hipMemcpy(r, p.res_ptr, sizeof(r), hipMemcpyDeviceToHost));
    for (int i = 0; i < 3; ++i) tot[i] += r[i];
  }
  const int rc = host_allreduce(m, tot, 3, ncclSum);
  if (rc) return rc;
  if (res_norm0) *res_norm0 = std::sqrt(tot[0]);
  if (res_norm) *res_norm = std::sqrt(tot[m->n_rk - 1]);
  return DFLO_OK;
}

static int step_body(dflo_hip_multi *m, double dt, double *res_norm0, double *res_norm) {
  int rc;
  if (!m->workers.empty()) {   // every part's own thread issues its step
    const bool peers = any_peers(m);
    const int64_t n0 = m->nx, step = m->n_steps_fin;
    rc = for_groups(m, [&](Group &g) { return group_step(m, g, dt, n0, step, peers); });
    m->nx += (int64_t)exchanges_per_stage(m) * m->n_rk;
    ++m->n_steps_fin;
    if (rc) return rc;
  } else {
    for (int rk = 0; rk < m->n_rk; ++rk)
      if ((rc = run_stage(m, rk, dt))) return rc;
    for (Part &p : m->parts) MENG(m, p, dflo_hip_end_step(p.eng));
    if ((rc = reduce_dt(m))) return rc;
  }
  if ((rc = sync_all(m))) return rc;
  if (res_norm0 || res_norm) rc = step_norms(m, res_norm0, res_norm);
  return rc;
}

int dflo_hip_multi_step(dflo_hip_multi_handle m, double dt, double *res_norm0, double *res_norm) {
  if (!m) return DFLO_ERR_BAD_PARAM;
  // rank mode: whatever this rank runs into, the ranks agree on the status before anyone returns (a rank that left early would
  // leave the others waiting in their next collective)
  const int rc = step_body(m, dt, res_norm0, res_norm);
  return agree(m, rc ? rc : check_local(m, false));
}

// The host looks at the failure flags every kCheckEvery steps without draining the device (see dflo_hip_advance); with one
// process per GPU the ranks must leave the loop together, so there the flags are read at the end only.
constexpr int kCheckEvery = 32;

// device-resident steps of a group's parts on the group's thread.  The threads leave the loop at the same step: one that finds
// a failure flag at a chunk boundary names the boundary two chunks on (`stop_at`), which no thread has passed yet -- the host
// threads are never more than a few stages apart (each waits for its neighbours' records of the same stage).
static int group_advance(dflo_hip_multi *m, Group &g, int n_steps, double dt0, int64_t n0, int64_t step0, bool peers) {
  MHIP(m, hipSetDevice(g.device));
  const int64_t xs = (int64_t)exchanges_per_stage(m) * m->n_rk;
  int chunk = 0, s = 0;
  for (; s < n_steps; ++s) {
    if (s > 0 && s % kCheckEvery == 0) {
      if (s >= m->stop_at.load(std::memory_order_acquire)) break;
      MHIP(m, hipEventRecord(g.ev_chunk[chunk & 1], g.M));
      ++chunk;
      if (chunk >= 2) {
        MHIP(m, hipEventSynchronize(g.ev_chunk[chunk & 1]));
        bool failed = false;
        for (int i : g.parts) { int64_t st = -1; dflo_hip_failure_step(m->parts[i].eng, &st); failed |= st >= 0; }
        if (failed) {
          int64_t want = (int64_t)s + 2 * kCheckEvery, cur = m->stop_at.load();
          while (want < cur && !m->stop_at.compare_exchange_weak(cur, want)) {}
        }
      }
    }
    const int rc = group_step(m, g, s == 0 ? dt0 : -1.0, n0 + (int64_t)s * xs, step0 + s, peers);
    if (rc) return rc;
  }
  for (int i : g.parts) m->parts[i].steps_run = s;
  return DFLO_OK;
}

static int advance_body(dflo_hip_multi *m, int n_steps, double dt0) {
  int rc;
  Part &p0 = m->parts[0];
  if (!m->workers.empty()) {
    const bool peers = any_peers(m);
    const int64_t n0 = m->nx, step0 = m->n_steps_fin;
    m->stop_at.store(INT64_MAX);
    rc = for_groups(m, [&](Group &g) { return group_advance(m, g, n_steps, dt0, n0, step0, peers); });
    if (rc) return rc;
    const int done = p0.steps_run;
    for (Part &p : m->parts)
      if (p.steps_run != done) { set_err(m, "multi-device advance: the parts' threads left the loop at different steps"); return DFLO_ERR_COMM; }
    m->nx += (int64_t)exchanges_per_stage(m) * m->n_rk * done;
    m->n_steps_fin += done;
    return DFLO_OK;
  }
  if (!m->rank_mode) {
    MHIP(m, hipSetDevice(p0.device));
    for (int i = 0; i < 2; ++i)
      if (!m->ev_chunk[i]) MHIP(m, hipEventCreateWithFlags(&m->ev_chunk[i], hipEventDisableTiming));
  }
  int chunk = 0;
  for (int s = 0; s < n_steps; ++s) {
    if (!m->rank_mode && s > 0 && s % kCheckEvery == 0) {
      MHIP(m, hipSetDevice(p0.device));
      MHIP(m, hipEventRecord(m->ev_chunk[chunk & 1], p0.M));
      ++chunk;
      if (chunk >= 2) {
        MHIP(m, hipEventSynchronize(m->ev_chunk[chunk & 1]));
        bool failed = false;
        for (Part &p : m->parts) { int64_t st = -1; dflo_hip_failure_step(p.eng, &st); failed |= st >= 0; }
        if (failed) break;
      }
    }
    for (int rk = 0; rk < m->n_rk; ++rk)
      if ((rc = run_stage(m, rk, s == 0 ? dt0 : -1.0))) return rc;
    for (Part &p : m->parts) MENG(m, p, dflo_hip_end_step(p.eng));
    if ((rc = reduce_dt(m))) return rc;
  }
  return DFLO_OK;
}

int dflo_hip_multi_advance(dflo_hip_multi_handle m, int n_steps, double *elapsed_time_inout) {
  if (!m || n_steps < 0 || !elapsed_time_inout) return DFLO_ERR_BAD_PARAM;
  if (m->n_parts == 1 && !m->self_halo) {   // one part: the engine's own device-resident loop (nothing to exchange, nothing to reduce)
    Part &p = m->parts[0];
    int rc1 = join_all(m);
    if (!rc1 && (rc1 = dflo_hip_advance(p.eng, n_steps, elapsed_time_inout))) set_err(m, dflo_hip_last_error(p.eng));
    return rc1;
  }
  double dt0 = 0.0;
  int rc = dflo_hip_multi_compute_dt(m, *elapsed_time_inout, &dt0);   // host value for the first step only
  if (rc) return rc;   // (compute_dt ends in a collective of its own: the ranks fail or pass together)
  m->verbose = dflo::read_tunables().multi_verbose;
  for (double &t : m->t_phase) t = 0.0;
  const auto t_issue0 = std::chrono::steady_clock::now();
  rc = advance_body(m, n_steps, dt0);
  const auto t_issue1 = std::chrono::steady_clock::now();
  if (!rc) rc = sync_all(m);
  if (m->verbose) {   // how far ahead of the devices the host runs
    const auto t2 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "dflo_hip_multi_advance: %d steps, host issued them in %.3f ms, devices done after %.3f ms\n", n_steps,
                 std::chrono::duration<double, std::milli>(t_issue1 - t_issue0).count(), std::chrono::duration<double, std::milli>(t2 - t_issue0).count());
    std::fprintf(stderr, "dflo_hip_multi_advance: host ms by phase: open %.2f rim %.2f interior %.2f tvb %.2f finish %.2f recv %.2f\n", m->t_phase[0], m->t_phase[1],
                 m->t_phase[2], m->t_phase[3], m->t_phase[4], m->t_phase[5]);
  }
  if (!rc) {
    double tt[4];
    Part &p0 = m->parts[0];
    if (hipSetDevice(p0.device) != hipSuccess || hipMemcpy(tt, p0.dt_ptr, sizeof(tt), hipMemcpyDeviceToHost) != hipSuccess) {
      set_err(m, "reading the clock back failed");
      rc = DFLO_ERR_HIP;
    } else {
      *elapsed_time_inout = tt[1];
    }
  }
  return agree(m, rc ? rc : check_local(m, false));
}

int dflo_hip_multi_apply_limiter(dflo_hip_multi_handle m) {
  if (!m) return DFLO_ERR_BAD_PARAM;
  int rc = join_all(m);
  if (rc) return rc;
  if (m->kxrcf && (rc = exchange_solution(m))) return rc;
  for (Part &p : m->parts) MENG(m, p, dflo_hip_apply_limiter(p.eng));
  return exchange_solution(m);
}

int dflo_hip_multi_apply_positivity_limiter(dflo_hip_multi_handle m) {
  if (!m) return DFLO_ERR_BAD_PARAM;
  int rc = join_all(m);
  if (rc) return rc;
  int worst = DFLO_OK;
  for (Part &p : m->parts) {
    const int r = dflo_hip_apply_positivity_limiter(p.eng);
    if (r) { worst = r; set_err(m, dflo_hip_last_error(p.eng)); }
  }
  if ((rc = exchange_solution(m))) return rc;
  const int all = check_all(m, true);
  return all ? all : worst;
}

int dflo_hip_multi_residual(dflo_hip_multi_handle m, int which, double *rhs) {
  if (!m || !rhs) return DFLO_ERR_BAD_PARAM;
  int rc = join_all(m);
  if (rc) return rc;
  std::vector<double> loc;
  for (Part &p : m->parts) {
    loc.resize((size_t)p.n_cells * m->ndof);
    MENG(m, p, dflo_hip_residual(p.eng, which, loc.data()));
    for (int c = 0; c < p.n_owned; ++c)
      std::memcpy(&rhs[(size_t)p.sub->cell_global_id[c] * m->ndof], &loc[(size_t)c * m->ndof], m->ndof * sizeof(double));
  }
  return DFLO_OK;
}

int dflo_hip_multi_check(dflo_hip_multi_handle m) { return m ? check_all(m, true) : DFLO_ERR_BAD_PARAM; }
int dflo_hip_multi_synchronize(dflo_hip_multi_handle m) { return m ? sync_all(m) : DFLO_ERR_BAD_PARAM; }

int dflo_hip_multi_stage_timing(dflo_hip_multi_handle m, int enable, double *avg_ms, int64_t *n) {
  if (!m) return DFLO_ERR_BAD_PARAM;
  int rc = sync_all(m);
  if (rc) return rc;
  double worst = 0.0;
  int64_t cnt = 0;
  for (Part &p : m->parts) {   // the slowest part sets the pace
    double ms = 0.0;
    int64_t k = 0;
    MENG(m, p, dflo_hip_stage_timing(p.eng, enable, &ms, &k));
    if (ms > worst) worst = ms;
    cnt = std::max(cnt, k);
  }
  if (avg_ms) *avg_ms = worst;
  if (n) *n = cnt;
  return DFLO_OK;
}

int dflo_hip_multi_exchange_timing(dflo_hip_multi_handle m, int enable, double *avg_us, int64_t *n) {
  if (!m) return DFLO_ERR_BAD_PARAM;
  int rc = sync_all(m);
  if (rc) return rc;
  double tot = 0.0;
  int64_t cnt = 0;
  for (Part &p : m->parts) {
    MHIP(m, hipSetDevice(p.device));
    for (size_t i = 0; i < p.x_used; ++i) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, p.x_pool[i].first, p.x_pool[i].second) == hipSuccess) { tot += ms; ++cnt; }
    }
    p.x_used = 0;
    p.x_seen = 0;
    p.x_on = enable != 0;
  }
  if (avg_us) *avg_us = cnt ? tot * 1000.0 / (double)cnt : 0.0;
  if (n) *n = cnt;
  return DFLO_OK;
}

int dflo_hip_multi_comm_info(dflo_hip_multi_handle m, int32_t *comm_count, int32_t *comm_rank, char *transport, int32_t transport_len) {
  if (!m) return DFLO_ERR_BAD_PARAM;
  int cnt = m->n_parts, rk = m->rank_mode ? m->rank : -1;
  std::string t;
  if (m->self_halo) {
    t = std::string("self-halo (one part, its own neighbour across ") + (m->self_virtual == 1 ? "the periodic seam in x" : "a cut through the middle") + "): ";
    if (m->ipc && m->fused_tvb) t += "the single engine's two launches per stage: the stage kernel's workgroups on the cut deliver their averages, the limiter pass's their traces, into the own areas + sequence words (the IPC transport against itself, delivery by the kernels)";
    else if (m->ipc) t += m->fused ? "one launch per stage, its workgroups on the cut store their traces into the own table + sequence words polled by a wait kernel (the IPC transport against itself, delivery by the stage kernel)"
                              : "rank schedule, pack kernels storing into the own receive areas + sequence words polled by a wait kernel (the IPC transport against itself)";
    else if (m->rank_mode && m->comm) {
      t += "rank schedule, grouped ncclSend/ncclRecv to itself + ncclAllReduce(min) on a one-rank RCCL communicator";
      cnt = rk = -1;
      if (g_rccl.CommCount) g_rccl.CommCount(m->comm, &cnt);
      if (g_rccl.CommUserRank) g_rccl.CommUserRank(m->comm, &rk);
    } else t += m->direct ? "one-process schedule, pack kernels storing into the own trace table" : "one-process schedule, staging buffer + hipMemcpyPeerAsync";
  } else if (m->rank_mode && m->ipc) {
    t = std::string(m->fused_tvb ? "IPC: two launches per stage, the stage kernel's workgroups on the cut store their averages, the limiter pass's their traces, into the neighbours' hipIpc-mapped areas"
                    : m->fused ? "IPC: one launch per stage, its workgroups on the cut store their traces into the neighbours' hipIpc-mapped tables"
                             : "IPC: pack kernels storing into the neighbours' hipIpc-mapped receive areas") +
        " + sequence words polled by a wait kernel; time step through the mapped tables (bootstrap: " + (m->comm ? "RCCL" : "host callbacks") + ")";
    if (m->comm) { cnt = rk = -1; if (g_rccl.CommCount) g_rccl.CommCount(m->comm, &cnt); if (g_rccl.CommUserRank) g_rccl.CommUserRank(m->comm, &rk); }
  } else if (m->rank_mode && m->x_exchange) t = "callbacks of the host program (dflo_hip_multi_create_rank_custom)";
  else if (m->rank_mode && m->comm) {
    t = "RCCL: grouped ncclSend/ncclRecv + ncclAllReduce(min) on the driver's own communicator";
    cnt = rk = -1;   // as the communicator itself reports them, or -1
    if (g_rccl.CommCount) g_rccl.CommCount(m->comm, &cnt);
    if (g_rccl.CommUserRank) g_rccl.CommUserRank(m->comm, &rk);
  } else if (m->rank_mode) t = "none (one rank)";
  else if (m->loopback) t = "one process: one-rank RCCL loopback (test transport)";
  else if (m->n_parts == 1) t = "none (one part)";
  else t = m->direct ? "one process: pack kernels storing into the peers' receive areas (xGMI peer access)" : "one process: staging buffer + hipMemcpyPeerAsync";
  if (m->ipc && !m->self_halo ? m->ipc_fine : dflo::read_tunables().peer_finegrained) t += "; peer-written buffers in fine-grained memory";
  if (m->strict) t += "; strict (senders wait for the receivers' consumed events)";
  if (m->tail_wait) t += "; the compute stream ordered behind the rim launch by a word its interior launch waits for (no wait packet)";
  if (m->tvb && !m->fused_tvb && any_peers(m))
    t += m->tvb_one ? "; TVB: one exchange per stage (unlimited cut cells + their neighbours' averages; ghost cells limited by the receiver)"
                    : "; TVB: two exchanges per stage (averages, then the limited state)";
  if (comm_count) *comm_count = cnt;
  if (comm_rank) *comm_rank = rk;
  if (transport && transport_len > 0) {
    std::strncpy(transport, t.c_str(), (size_t)transport_len - 1);
    transport[transport_len - 1] = 0;
  }
  return DFLO_OK;
}

}  // extern "C"
