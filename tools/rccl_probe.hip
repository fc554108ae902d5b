// Developer probe: what do the RCCL calls of the rank schedule (dflo_amd/csrc/multi.hip: post / reduce_dt_rank) cost on the comm
// stream, measured where a box has ONE GPU -- on a one-rank communicator, the rank sending to and receiving from itself?
//
//   1. grouped ncclSend + ncclRecv (self) of 8 B .. 1 MB: the halo message of a cut (C4: 1000 faces x 96 B = 96 KB; C2: 1024 x 96 B)
//   2. ncclAllReduce(min) of one double in place: Utilities::MPI::min(global_dt), src_mpi/claw.cc:579
//   3. the same calls while a full-size C2 stage kernel (1024^2 Q2 HLLC, the engine of this repository through its C ABI) runs
//      on the compute stream beside them: how much does an RCCL kernel slow the interior launch, and the launch the RCCL kernel?
// per call: device time between two events on the comm stream around ONE call (isolated: the stream was idle), device time per
// call of a back-to-back train (throughput), host time to enqueue.  NCCL_MAX_NCHANNELS / NCCL_MIN_NCHANNELS are read by RCCL at
// communicator creation: run the probe once per setting (tools/rccl_probe.sh).
//
// What a one-rank communicator cannot show: the xGMI hop and the peer's progress; a self send/recv is RCCL's local copy path.
// The figures are a LOWER bound of the real exchange, and an honest measure of the fixed cost (kernel launch of the RCCL
// kernel, its channel setup, the group bookkeeping on the host) that DESIGN 6.1 had to assume before.
//
//   hipcc --offload-arch=gfx950 -O2 -o scratch/probe/rccl_probe tools/rccl_probe.hip -Ldflo_amd -ldflo_hip -Wl,-rpath,$PWD/dflo_amd -ldl
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/dflo_hip.h"

#define CK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { std::printf("%s: %s\n", #c, hipGetErrorString(e_)); return 1; } } while (0)
#define NK(c) do { ncclResult_t r_ = (c); if (r_ != ncclSuccess) { std::printf("%s: %s\n", #c, R.GetErrorString(r_)); return 1; } } while (0)

struct Rccl {
  ncclResult_t (*GetUniqueId)(ncclUniqueId *);
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*GroupStart)();
  ncclResult_t (*GroupEnd)();
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  const char *(*GetErrorString)(ncclResult_t);
} R;

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double median(std::vector<double> v) {
  std::sort(v.begin(), v.end());
  return v.empty() ? 0.0 : v[v.size() / 2];
}

struct Stat { double iso_med, iso_min, train, host; };

// fn(stream) enqueues ONE call on the stream
template <class F>
static int measure(F fn, hipStream_t C, int reps, Stat &st) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 10; ++i) if (fn(C)) return 1;   // warm-up (first calls build RCCL's kernels / channels)
  CK(hipStreamSynchronize(C));
  std::vector<double> iso, host;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a, C));
    const double t0 = now_us();
    if (fn(C)) return 1;
    host.push_back(now_us() - t0);
    CK(hipEventRecord(b, C));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    iso.push_back(ms * 1e3);
  }
  CK(hipEventRecord(a, C));
  for (int r = 0; r < reps; ++r) if (fn(C)) return 1;
  CK(hipEventRecord(b, C));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  st.iso_med = median(iso);
  st.iso_min = *std::min_element(iso.begin(), iso.end());
  st.train = ms * 1e3 / reps;
  st.host = median(host);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return 0;
}

int main(int argc, char **argv) {
  const int nx = argc > 1 ? std::atoi(argv[1]) : 1024;
  void *lib = nullptr;
  {   // the RCCL next to the HIP runtime this process runs on (as multi.hip does)
    Dl_info info;
    std::vector<std::string> names;
    if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
      std::string dir(info.dli_fname);
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) { dir.resize(slash + 1); names.push_back(dir + "librccl.so.1"); names.push_back(dir + "librccl.so"); }
    }
    names.push_back("librccl.so.1");
    names.push_back("/opt/rocm/lib/librccl.so.1");
    for (auto &n : names) if ((lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL))) { std::printf("# RCCL: %s\n", n.c_str()); break; }
    if (!lib) { std::printf("cannot load RCCL: %s\n", dlerror()); return 1; }
  }
#define SYM(f, n) *(void **)(&R.f) = dlsym(lib, n); if (!R.f) { std::printf("no symbol %s\n", n); return 1; }
  SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy") SYM(GroupStart, "ncclGroupStart")
  SYM(GroupEnd, "ncclGroupEnd") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(AllReduce, "ncclAllReduce") SYM(GetErrorString, "ncclGetErrorString")
  CK(hipSetDevice(0));
  for (const char *e : {"NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "NCCL_PROTO", "NCCL_ALGO", "RCCL_MSCCL_ENABLE", "HSA_ENABLE_IPC_MODE_LEGACY"})
    std::printf("# %s=%s\n", e, std::getenv(e) ? std::getenv(e) : "(unset)");
  ncclUniqueId id;
  ncclComm_t comm;
  const double tc0 = now_us();
  NK(R.GetUniqueId(&id));
  NK(R.CommInitRank(&comm, 1, id, 0));
  std::printf("# ncclCommInitRank(1 rank): %.1f ms\n", (now_us() - tc0) * 1e-3);
  hipStream_t M, C;
  CK(hipStreamCreate(&M));
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&C, hipStreamDefault, hi));
  const size_t maxb = 1 << 20;
  double *sb, *rb, *slot;
  CK(hipMalloc((void **)&sb, maxb));
  CK(hipMalloc((void **)&rb, maxb));
  CK(hipMalloc((void **)&slot, 64));
  CK(hipMemset(sb, 0, maxb));
  CK(hipMemset(slot, 0, 64));
  const int reps = 200;
  std::printf("\n== 1. grouped ncclSend + ncclRecv to itself, comm stream otherwise idle (us) ==\n");
  std::printf("%10s %10s %10s %12s %10s\n", "bytes", "iso_median", "iso_min", "train/call", "host/call");
  const size_t sizes[] = {8, 4096, 32768, 98304, 196608, 1 << 20};
  Stat halo96{};
  for (size_t bytes : sizes) {
    auto fn = [&](hipStream_t s) -> int {
      NK(R.GroupStart());
      NK(R.Send(sb, bytes / 8, ncclDouble, 0, comm, s));
      NK(R.Recv(rb, bytes / 8, ncclDouble, 0, comm, s));
      NK(R.GroupEnd());
      return 0;
    };
    Stat st{};
    if (measure(fn, C, reps, st)) return 1;
    if (bytes == 98304) halo96 = st;
    std::printf("%10zu %10.2f %10.2f %12.2f %10.2f\n", bytes, st.iso_med, st.iso_min, st.train, st.host);
  }
  {   // two messages in one group (an interior rank: one per neighbour)
    auto fn = [&](hipStream_t s) -> int {
      NK(R.GroupStart());
      for (int k = 0; k < 2; ++k) {
        NK(R.Send(sb + k * 16384, 98304 / 8, ncclDouble, 0, comm, s));
        NK(R.Recv(rb + k * 16384, 98304 / 8, ncclDouble, 0, comm, s));
      }
      NK(R.GroupEnd());
      return 0;
    };
    Stat st{};
    if (measure(fn, C, reps, st)) return 1;
    std::printf("%10s %10.2f %10.2f %12.2f %10.2f   (two 96 KB pairs in one group)\n", "2x98304", st.iso_med, st.iso_min, st.train, st.host);
  }
  {   // for scale: the same bytes with hipMemcpyAsync device-to-device
    auto fn = [&](hipStream_t s) -> int { CK(hipMemcpyAsync(rb, sb, 98304, hipMemcpyDeviceToDevice, s)); return 0; };
    Stat st{};
    if (measure(fn, C, reps, st)) return 1;
    std::printf("%10s %10.2f %10.2f %12.2f %10.2f   (hipMemcpyAsync d2d of 96 KB, for scale)\n", "memcpy", st.iso_med, st.iso_min, st.train, st.host);
  }
  std::printf("\n== 2. ncclAllReduce(min) of one double in place (us) ==\n");
  Stat ar{};
  {
    auto fn = [&](hipStream_t s) -> int { NK(R.AllReduce(slot, slot, 1, ncclDouble, ncclMin, comm, s)); return 0; };
    if (measure(fn, C, reps, ar)) return 1;
    std::printf("%10d %10.2f %10.2f %12.2f %10.2f\n", 8, ar.iso_med, ar.iso_min, ar.train, ar.host);
  }

  // ---- 3. beside a full-size stage kernel
  std::printf("\n== 3. beside the C2 stage kernel (%d x %d Q2 HLLC periodic, one launch per stage on the compute stream) ==\n", nx, nx);
  dflo_mesh_t *mesh = nullptr;
  const int32_t bids[4] = {-1, -1, -1, -1};
  if (dflo_mesh_cartesian(nx, nx, -5.0, -5.0, 10.0 / nx, bids, 2, &mesh)) { std::printf("mesh: %s\n", dflo_mesh_last_error()); return 1; }
  dflo_params_t prm;
  std::memset(&prm, 0, sizeof(prm));
  prm.flux_type = DFLO_FLUX_HLLC;
  prm.char_lim = 1;
  prm.global_time_step = 1;
  prm.cfl = 0.9;
  prm.final_time = 1.0e20;
  prm.beta = 2.0;
  dflo_hip_handle eng = nullptr;
  if (dflo_hip_create(mesh, &prm, 0, &eng)) { std::printf("engine: %s\n", dflo_hip_last_error(nullptr)); return 1; }
  {
    const int64_t nd = dflo_hip_n_dofs(eng);
    const int ndof = dflo_hip_dofs_per_cell(eng), ns = ndof / 4;
    std::vector<double> u((size_t)nd);
    for (int64_t c = 0; c < nd / ndof; ++c)
      for (int j = 0; j < ns; ++j) {
        const double w = 1.0 + 0.05 * std::sin(0.001 * (double)c + 0.3 * j);
        u[c * ndof + 0 * ns + j] = 0.5 * w;
        u[c * ndof + 1 * ns + j] = 0.3 * w;
        u[c * ndof + 2 * ns + j] = w;
        u[c * ndof + 3 * ns + j] = 2.5 + 0.5 * (0.25 + 0.09) * w;
      }
    if (dflo_hip_set_solution(eng, u.data())) { std::printf("set_solution: %s\n", dflo_hip_last_error(eng)); return 1; }
  }
  dflo_hip_set_stream(eng, M);
  double dt = 0;
  if (dflo_hip_compute_dt(eng, 0.0, &dt)) { std::printf("compute_dt: %s\n", dflo_hip_last_error(eng)); return 1; }
  dt *= 0.1;
  auto step = [&]() -> int {
    for (int rk = 0; rk < 3; ++rk) if (dflo_hip_stage(eng, rk, dt)) { std::printf("stage: %s\n", dflo_hip_last_error(eng)); return 1; }
    return dflo_hip_end_step(eng);
  };
  for (int i = 0; i < 30; ++i) if (step()) return 1;
  CK(hipStreamSynchronize(M));
  hipEvent_t a, b, ca, cb;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&ca)); CK(hipEventCreate(&cb));
  const int steps = 60;
  auto run = [&](int mode, double &step_us, double &call_us) -> int {
    // mode 0: stages alone; 1: + one 96 KB self send/recv per stage on C (not ordered against M: pure interference);
    // 2: + the 8-byte all-reduce per stage on C; 3: both
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, M));
    CK(hipEventRecord(ca, C));
    for (int s = 0; s < steps; ++s) {
      for (int rk = 0; rk < 3; ++rk) {
        if (dflo_hip_stage(eng, rk, dt)) return 1;
        if (mode & 1) { NK(R.GroupStart()); NK(R.Send(sb, 98304 / 8, ncclDouble, 0, comm, C)); NK(R.Recv(rb, 98304 / 8, ncclDouble, 0, comm, C)); NK(R.GroupEnd()); }
        if (mode & 2) NK(R.AllReduce(slot, slot, 1, ncclDouble, ncclMin, comm, C));
      }
      dflo_hip_end_step(eng);
    }
    CK(hipEventRecord(b, M));
    CK(hipEventRecord(cb, C));
    CK(hipEventSynchronize(b));
    CK(hipEventSynchronize(cb));
    float ms = 0, cms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventElapsedTime(&cms, ca, cb));
    step_us = ms * 1e3 / steps;
    call_us = cms * 1e3 / (steps * 3);
    return 0;
  };
  double base = 0, dummy = 0;
  for (int rep = 0; rep < 2; ++rep) {
    for (int mode = 0; mode < 4; ++mode) {
      double su = 0, cu = 0;
      if (run(mode, su, cu)) return 1;
      if (mode == 0) base = su;
      std::printf("rep %d  %-46s step %8.2f us  (x%.4f of alone)%s\n", rep,
                  mode == 0 ? "stages alone" : (mode == 1 ? "+ 96 KB self send/recv per stage on C" : (mode == 2 ? "+ 8 B all-reduce per stage on C" : "+ both")),
                  su, su / base, mode ? (std::string("   comm stream busy until ") + std::to_string(cu) + " us/stage").c_str() : "");
    }
  }
  (void)dummy;
  // ordered as the schedule orders them: the send/recv behind a small kernel that follows the stage's start is not reproducible
  // here without the engine's rim split -- that is what bench.py --self-halo measures.
  std::printf("\n# summary_json {\"sendrecv_96k_iso_us\": %.2f, \"sendrecv_96k_train_us\": %.2f, \"sendrecv_96k_host_us\": %.2f, \"allreduce_8b_iso_us\": %.2f, "
              "\"allreduce_8b_train_us\": %.2f, \"allreduce_8b_host_us\": %.2f, \"c2_step_alone_us\": %.2f}\n",
              halo96.iso_med, halo96.train, halo96.host, ar.iso_med, ar.train, ar.host, base);
  dflo_hip_destroy(eng);
  dflo_mesh_free(mesh);
  R.CommDestroy(comm);
  return 0;
}
