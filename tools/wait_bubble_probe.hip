// Developer probe: what does a cross-stream wait cost the WAITING stream when the awaited work finished long ago on the device
// (but not yet when the host issued the wait)?  A chain of ~50 us kernels on stream M; stream C runs a tiny kernel per iteration
// (behind M's previous kernel) that M's next-but-one kernel formally depends on.
//   mode 0: no cross-stream dependency on M (C still waits for M)      mode 1: hipStreamWaitEvent(M, event of C)
//   mode 2: hipStreamWaitValue64(M, flag >= iteration), C: hipStreamWriteValue64 behind its kernel
// (hipcc --offload-arch=gfx950 -O2 -o scratch/probe/wait_bubble_probe tools/wait_bubble_probe.hip)
#include <hip/hip_runtime.h>
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
int main(int argc, char **argv) {
  double *x;
  CK(hipMalloc(&x, 1 << 20));
  CK(hipMemset(x, 0, 1 << 20));
  hipStream_t M, C;
  CK(hipStreamCreate(&M));
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&C, hipStreamDefault, hi));
  hipEvent_t em[2], ec[2];
  for (int i = 0; i < 2; ++i) { CK(hipEventCreateWithFlags(&em[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ec[i], hipEventDisableTiming)); }
  uint64_t *flag = nullptr;
  hipError_t fe = hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory);
  if (fe != hipSuccess) { printf("hipMallocSignalMemory: %s -- plain hipMalloc\n", hipGetErrorString(fe)); CK(hipMalloc((void **)&flag, 8)); }
  CK(hipMemset(flag, 0, 8));
  long n = 20000;
  for (int it = 0; it < 6; ++it) {   // calibrate a kernel of ~50 us
    double t0 = now();
    spin<<<256, 64, 0, M>>>(x, n);
    CK(hipStreamSynchronize(M));
    double dt = now() - t0;
    if (it == 5) printf("kernel of n=%ld takes %.1f us (host clock, with launch + sync)\n", n, dt);
    else n = (long)(n * 50.0 / dt) + 1;
  }
  const int R = 400;
  for (int mode = 0; mode < 3; ++mode) {
    CK(hipMemset(flag, 0, 8));
    CK(hipDeviceSynchronize());
    double t0 = now();
    for (int r = 0; r < R; ++r) {
      const int p = r & 1;
      // C: behind M's previous kernel, a tiny kernel
      if (r > 0) CK(hipStreamWaitEvent(C, em[p ^ 1], 0));
      spin<<<4, 64, 0, C>>>(x + 65536, 50);
      if (mode == 1) CK(hipEventRecord(ec[p], C));
      if (mode == 2) CK(hipStreamWriteValue64(C, flag, (uint64_t)(r + 1), 0));
      // M: depends on C's kernel of the PREVIOUS iteration (finished ~50 us ago on the device)
      if (r > 0 && mode == 1) CK(hipStreamWaitEvent(M, ec[p ^ 1], 0));
      if (r > 0 && mode == 2) CK(hipStreamWaitValue64(M, flag, (uint64_t)r, hipStreamWaitValueGte, ~0ull));
      spin<<<256, 64, 0, M>>>(x, n);
      CK(hipEventRecord(em[p], M));
    }
    double t1 = now();
    CK(hipDeviceSynchronize());
    double t2 = now();
    printf("mode %d: %.2f us per iteration on the device (host issued in %.2f us per iteration)\n", mode, (t2 - t0) / R, (t1 - t0) / R);
  }
  return 0;
}
