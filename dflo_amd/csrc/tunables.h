// tunables.h -- the run-time switches of the engine and of the multi-device driver, read from the environment in ONE place.
// Every switch defaults to the path that measured fastest; the alternatives are kept because tests run both sides of each
// (a fused path against the separate pass it replaced, a transport against the other).  Measured-and-lost kernel variants
// are not switches: they live as patches under tools/variants/.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace dflo {

struct Tunables {
  // ---- engine (read at dflo_hip_create)
  bool graph = false;      // DFLO_GRAPH=1       dflo_hip_advance replays a captured hipGraph of two steps (no gain measured: opt-in)
  bool sweep = true;       // DFLO_SWEEP=0       every launch walks the shards forward (default: against the previous launch's direction)
  int stream = -1;         // DFLO_STREAM=0|1    forbid / force streaming stores of the new state (default: when no pass over all cells follows)
  bool fuse_dtq = true;    // DFLO_FUSE_DTQ=0    bilinear cells: compute_time_step_q by the separate pass instead of the last stage kernel
  bool fuse_pos = true;    // DFLO_FUSE_POS=0    positivity without TVB on Qk: separate limiter pass instead of inside the stage kernel
  bool fuse_fin = true;    // DFLO_FUSE_FIN=0    TVB on Qk squares: finalize_kernel as its own launch instead of inside the limiter pass that ends the step
  bool bc_fuse = true;     // DFLO_BC_FUSE=0     boundary programs always by bc_eval_kernel (default: the limiter pass behind stage 0 takes the later
                           //                    stages' table along, stage 0 reads the table the previous step's later stages used)
  bool mfma = false;       // DFLO_MFMA=1        degree 3: the dense per-element contractions on the matrix pipe -- Q3 (squares and bilinear cells): the
                           //                    eta-derivative as one v_mfma_f64_4x4x4_4b per cell; P3 on squares: modal -> nodal and nodal -> modal as
                           //                    v_mfma_f64_16x16x4 with 16 cells along the columns (default: the vector units, which measured faster --
                           //                    fp64 matrix and vector instructions share one issue pipe on MI355X, tools/mfma_f64_16x16_probe.hip)
  bool lazy_avg = true;    // DFLO_LAZY_AVG=0    store the cell averages of every stage (default: only when somebody reads them)
  bool lxf_from_dofs = true;   // DFLO_LXF_FROM_DOFS=0   the LxF flux on squares reads the arrays of cell averages (default: (u, v, c) of the averages from the DoFs)
  bool lim_list = true;    // DFLO_LIM_LIST=0    with marks: one wavefront per shard looks at its word instead of a short grid walking the list of marked shards
  int lim_grid = 1024;     // DFLO_LIM_GRID=n    wavefronts of that short grid
  int lim_mask = -1;       // DFLO_LIM_MASK=0|1  TVB on squares: forbid / force the stage kernel's marks for the limiter pass (default: degree >= 2; degree 1 without ghost cells)
  bool halo_cells = false; // DFLO_HALO_CELLS=1  multi-device: ghost cells as whole cells instead of face traces
  bool verbose = false;    // DFLO_VERBOSE=1     print the LDS footprint and the resident workgroups of the stage kernel
  int plan_refine = 8;     // DFLO_PLAN_REFINE=n swap-refinement passes of the shard plan on unstructured meshes
  bool rim_first = true;   // DFLO_PLAN_RIM_FIRST=0  unstructured shards keep the Hilbert order of their cells (default: rim cells first, by neighbouring shard)
  // ---- multi-device driver (read at dflo_hip_multi_create*)
  int group = 0;           // DFLO_MULTI_GROUP=part|device   1: a stream pair + host thread per part, 2: per device (default 0: per part,
                           //                                at most two per device)
  bool threads = true;     // DFLO_MULTI_THREADS=0           the calling thread drives every group
  bool strict = false;     // DFLO_MULTI_STRICT=1            a sender waits for the receiver's explicit "consumed" event
  bool copy = false;       // DFLO_MULTI_COPY=1              staging buffer + hipMemcpyPeerAsync instead of pack kernels that write remotely
  bool loopback = false;   // DFLO_MULTI_TRANSPORT=rccl_loopback   (test hook) the copies through a one-rank RCCL communicator
  bool multi_verbose = false;   // DFLO_MULTI_VERBOSE=1      dflo_hip_multi_advance reports how far the host ran ahead of the devices
  bool comm_priority = true;    // DFLO_MULTI_PRIORITY=0     the comm stream at the compute stream's priority (default: highest)
  bool peer_finegrained = false;  // DFLO_PEER_FINEGRAINED=1   every buffer a peer's kernel writes (ghost-trace tables, time-step tables,
                                  //                           receive areas) in fine-grained device memory (hipDeviceMallocFinegrained)
  bool ipc_finegrained = true;    // DFLO_PEER_FINEGRAINED=0   ... the window the IPC transport exports is fine-grained unless this says 0 (measured on
                                  //                           one device: no cost, profiles/LAB.md R5.11), the one-process driver's buffers only if it says 1
  int rank_transport = 0;         // DFLO_RANK_TRANSPORT=rccl|ipc   one process per GPU: 0 grouped ncclSend/ncclRecv + ncclAllReduce (default),
                                  //                           1 pack kernels storing into the peers' IPC-mapped receive areas + sequence flags
  bool ipc_fused = true;          // DFLO_IPC_FUSED=0         IPC transport: rim launch + pack kernel on the comm stream even where the stage kernel could
                                  //                           deliver its cut faces' traces itself (one launch per stage, one stream)
  bool ipc_kwait = true;          // DFLO_IPC_KWAIT=0         ... and a one-wavefront kernel in front of every stage waits for the neighbours' traces
                                  //                           (default: the stage kernel's workgroups on the cut wait themselves)
  int ipc_timeout_s = 120;        // DFLO_IPC_TIMEOUT_S=n     IPC transport: seconds a wait kernel / a polling workgroup waits for a neighbour's sequence word
                                  //                           before it raises DFLO_ERR_COMM (0: for ever, as an MPI or RCCL receive would).  Ranks must
                                  //                           enter advance() / step() within that time of each other (the calls are collective)
  bool ipc_strict = false;        // DFLO_IPC_STRICT=1        IPC transport, delivery by the stage kernel / the limiter pass: every delivering workgroup
                                  //                           fences at system scope before it counts itself (release; the publishing workgroup's fence
                                  //                           acquires) -- the formally complete protocol, whatever kind of memory the window is; default:
                                  //                           write-through stores + s_waitcnt on a fine-grained window, the fence only on a plain one
  bool ipc_words_uncached = false;  // DFLO_IPC_WORDS=uncached  (developer switch) the sequence words in hipDeviceMallocUncached memory: LAB R5.15's configuration
  bool ipc_free_early = false;      // DFLO_IPC_FREE_EARLY=1    (developer switch) dflo_hip_multi_destroy frees the exported windows without waiting for the
                                    //                           neighbours to close their mappings: round 5's order (against the IPC contract; NOT what LAB R5.15 saw: LAB R6.3)
  bool dt_on_comm = false;      // DFLO_DT_ON_COMM=1         rank mode, RCCL / callbacks: the all-reduce of the time step on the comm stream (two stream
                                //                           hops per step) instead of the compute stream
  bool tail_wait = true;        // DFLO_TAIL_WAIT=0          one process per GPU, two-stream schedule: the compute stream waits for the rim launch's event
                                //                           in front of its next kernel (a wait packet: 8.4 us of the stream) instead of inside the
                                //                           interior launch's last moments (dflo_hip_stage_tail_wait)
  int tvb_one_exchange = -1;    // DFLO_TVB_ONE_EXCHANGE=0|1 TVB stages of the multi-device driver (not the fused IPC form): 0 the reference's two
                                //                           exchanges per stage (averages, then the limited state), 1 one (the cut cells unlimited
                                //                           with their neighbours' averages; the receiver limits its ghost cells itself) wherever
                                //                           the partition allows; unset: one where an exchange is a library call (RCCL, the host
                                //                           program's callbacks: +16 % on self-halo C4), two where it is a kernel's stores (-1.6 %)
  bool avg_in_place = true;     // DFLO_MULTI_AVG_UNPACK=1   TVB: unpack the received ghost averages into the engine's array before the rim
                                //                           limiter (default: the limiter reads them where they arrived)
};

inline Tunables read_tunables() {
  Tunables t;
  auto flag = [](const char *name, bool dflt) {
    const char *e = std::getenv(name);
    return e ? std::atoi(e) != 0 : dflt;
  };
  auto tri = [](const char *name) {
    const char *e = std::getenv(name);
    return (e && e[0]) ? (e[0] != '0' ? 1 : 0) : -1;   // (set but empty counts as unset)
  };
  // an integer switch: a value that is not a number >= lo is reported once and ignored (the default stays)
  auto count = [](const char *name, int dflt, int lo) {
    const char *e = std::getenv(name);
    if (!e) return dflt;
    char *end = nullptr;
    const long v = std::strtol(e, &end, 10);
    if (end == e || *end != '\0' || v < lo || v > (1 << 30)) {
      std::fprintf(stderr, "dflo_hip: %s=%s ignored (an integer >= %d is expected)\n", name, e, lo);
      return dflt;
    }
    return (int)v;
  };
  t.graph = flag("DFLO_GRAPH", false);
  t.sweep = flag("DFLO_SWEEP", true);
  t.stream = tri("DFLO_STREAM");
  t.fuse_dtq = flag("DFLO_FUSE_DTQ", true);
  t.fuse_pos = flag("DFLO_FUSE_POS", true);
  t.fuse_fin = flag("DFLO_FUSE_FIN", true);
  t.bc_fuse = flag("DFLO_BC_FUSE", true);
  t.mfma = flag("DFLO_MFMA", false);
  t.lazy_avg = flag("DFLO_LAZY_AVG", true);
  t.lxf_from_dofs = flag("DFLO_LXF_FROM_DOFS", true);
  t.lim_mask = tri("DFLO_LIM_MASK");
  t.lim_list = flag("DFLO_LIM_LIST", true);
  t.lim_grid = count("DFLO_LIM_GRID", t.lim_grid, 1);
  t.halo_cells = flag("DFLO_HALO_CELLS", false);
  t.verbose = flag("DFLO_VERBOSE", false);
  t.plan_refine = count("DFLO_PLAN_REFINE", t.plan_refine, 0);
  t.rim_first = flag("DFLO_PLAN_RIM_FIRST", true);
  if (const char *e = std::getenv("DFLO_MULTI_GROUP")) t.group = std::strcmp(e, "part") == 0 ? 1 : (std::strcmp(e, "device") == 0 ? 2 : 0);
  t.threads = flag("DFLO_MULTI_THREADS", true);
  t.strict = flag("DFLO_MULTI_STRICT", false);
  t.copy = flag("DFLO_MULTI_COPY", false);
  if (const char *e = std::getenv("DFLO_MULTI_TRANSPORT")) t.loopback = std::strcmp(e, "rccl_loopback") == 0;
  t.multi_verbose = flag("DFLO_MULTI_VERBOSE", false);
  t.comm_priority = flag("DFLO_MULTI_PRIORITY", true);
  t.avg_in_place = !flag("DFLO_MULTI_AVG_UNPACK", false);
  t.tvb_one_exchange = tri("DFLO_TVB_ONE_EXCHANGE");
  t.tail_wait = flag("DFLO_TAIL_WAIT", true);
  t.dt_on_comm = flag("DFLO_DT_ON_COMM", false);
  t.peer_finegrained = flag("DFLO_PEER_FINEGRAINED", false);
  t.ipc_finegrained = flag("DFLO_PEER_FINEGRAINED", true);
  t.ipc_fused = flag("DFLO_IPC_FUSED", true);
  t.ipc_kwait = flag("DFLO_IPC_KWAIT", true);
  t.ipc_timeout_s = count("DFLO_IPC_TIMEOUT_S", t.ipc_timeout_s, 0);
  t.ipc_strict = flag("DFLO_IPC_STRICT", false);
  if (const char *e = std::getenv("DFLO_IPC_WORDS")) t.ipc_words_uncached = std::strcmp(e, "uncached") == 0;
  t.ipc_free_early = flag("DFLO_IPC_FREE_EARLY", false);
  if (const char *e = std::getenv("DFLO_RANK_TRANSPORT")) t.rank_transport = std::strcmp(e, "ipc") == 0 ? 1 : 0;
  return t;
}

}  // namespace dflo
