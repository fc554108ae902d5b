// Developer probe: what do the packets between two kernels of the compute stream cost in the rim || interior schedule
// (dflo_amd/csrc/multi.hip: stage_phase), and does hipExtLaunchKernelGGL's stopEvent -- the kernel's own completion signal as the
// event -- save the record packet?  Per iteration ("stage"): a ~100 us kernel on M (the interior), a ~15 us kernel on C (the rim);
// the rim waits for M's previous kernel, the interior for C's previous kernel.
//   mode 0: no dependencies at all (two independent chains)              -- the floor
//   mode 1: hipEventRecord + hipStreamWaitEvent on both streams         -- what the driver does
//   mode 2: the events ride on the launches (stopEvent), waits as before
//   mode 3: as 2, and the waits of M on C's event of TWO iterations back (always long satisfied) -- cost of a satisfied wait alone
//   hipcc --offload-arch=gfx950 -O2 -o scratch/probe/stop_event_probe tools/stop_event_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void spin(double *x, long n) {
  double v = x[threadIdx.x + 64 * blockIdx.x];
  for (long i = 0; i < n; ++i) v = v * 1.0000001 + 1e-9;
  x[threadIdx.x + 64 * blockIdx.x] = v;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("%s: %s\n", #c, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  double *x;
  CK(hipMalloc(&x, 1 << 22));
  CK(hipMemset(x, 0, 1 << 22));
  hipStream_t M, C;
  CK(hipStreamCreate(&M));
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&C, hipStreamDefault, hi));
  const int R = 300;
  hipEvent_t em[4], ec[4];
  for (int i = 0; i < 4; ++i) { CK(hipEventCreateWithFlags(&em[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ec[i], hipEventDisableTiming)); }
  long n = 20000;
  for (int it = 0; it < 6; ++it) {   // calibrate the big kernel to ~100 us
    double t0 = now();
    spin<<<2048, 64, 0, M>>>(x, n);
    CK(hipStreamSynchronize(M));
    double dt = now() - t0;
    if (it < 5) n = (long)(n * 100.0 / dt) + 1;
    else printf("big kernel: n=%ld, %.1f us with launch + sync\n", n, dt);
  }
  const long nsmall = n / 7;
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
      CK(hipDeviceSynchronize());
      const double t0 = now();
      for (int r = 0; r < R; ++r) {
        const int p = r & 3, q = (r + 3) & 3, q2 = (r + 2) & 3;   // q: previous iteration, q2: two back
        // C: the rim -- behind M's previous kernel
        if (mode && r > 0) CK(hipStreamWaitEvent(C, em[q], 0));
        if (mode >= 2) hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, C, nullptr, ec[p], 0, x + (1 << 18), nsmall);
        else { spin<<<256, 64, 0, C>>>(x + (1 << 18), nsmall); if (mode) CK(hipEventRecord(ec[p], C)); }
        // M: the interior -- behind C's previous kernel
        if (mode && mode < 3 && r > 0) CK(hipStreamWaitEvent(M, ec[q], 0));
        if (mode == 3 && r > 1) CK(hipStreamWaitEvent(M, ec[q2], 0));
        if (mode >= 2) hipExtLaunchKernelGGL(spin, dim3(2048), dim3(64), 0, M, nullptr, em[p], 0, x, n);
        else { spin<<<2048, 64, 0, M>>>(x, n); if (mode) CK(hipEventRecord(em[p], M)); }
      }
      CK(hipStreamSynchronize(M));
      CK(hipStreamSynchronize(C));
      printf("rep %d mode %d: %.2f us per iteration\n", rep, mode, (now() - t0) / R);
    }
  return 0;
}
