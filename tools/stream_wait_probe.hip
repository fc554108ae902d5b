// Developer probe: does the host block when one stream is made to wait for an event another stream has not reached yet?
// (hipcc --offload-arch=gfx950 -O2 -o scratch/probe/stream_wait_probe tools/stream_wait_probe.hip)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(double *x, long n) {
  double v = x[threadIdx.x];
  for (long i = 0; i < n; ++i) v = v * 1.0000001 + 1e-9;
  x[threadIdx.x] = v;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const int flags = argc > 1 ? atoi(argv[1]) : 1;   // 1: hipEventDisableTiming
  const int prio = argc > 2 ? atoi(argv[2]) : 1;
  double *x;
  hipMalloc(&x, 4096);
  hipMemset(x, 0, 4096);
  hipStream_t A, B;
  hipStreamCreate(&A);
  int lo, hi;
  hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (prio) hipStreamCreateWithPriority(&B, hipStreamDefault, hi); else hipStreamCreate(&B);
  hipEvent_t e1, e2;
  hipEventCreateWithFlags(&e1, flags ? hipEventDisableTiming : hipEventDefault);
  hipEventCreateWithFlags(&e2, flags ? hipEventDisableTiming : hipEventDefault);
  spin<<<1, 64, 0, A>>>(x, 1000);
  spin<<<1, 64, 0, B>>>(x + 64, 1000);
  hipDeviceSynchronize();
  // calibrate: a kernel of ~200 us
  long n = 100000;
  for (int it = 0; it < 6; ++it) {
    double t0 = now();
    spin<<<1, 64, 0, A>>>(x, n);
    hipStreamSynchronize(A);
    double dt = now() - t0;
    if (it == 5) printf("kernel of n=%ld takes %.1f us\n", n, dt);
    else n = (long)(n * 200.0 / dt);
  }
  const int R = 50;
  double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double T0 = now();
  for (int r = 0; r < R; ++r) {
    double a = now();
    spin<<<1, 64, 0, A>>>(x, n);            // long kernel on A
    double b = now();
    hipEventRecord(e1, A);
    double c = now();
    hipStreamWaitEvent(B, e1, 0);           // B waits for A's kernel, which has not finished
    double d = now();
    spin<<<1, 64, 0, B>>>(x + 64, 100);      // small kernel on B
    double e = now();
    hipEventRecord(e2, B);
    double f = now();
    hipStreamWaitEvent(A, e2, 0);
    double g = now();
    t[0] += b - a; t[1] += c - b; t[2] += d - c; t[3] += e - d; t[4] += f - e; t[5] += g - f;
  }
  double T1 = now();
  hipDeviceSynchronize();
  double T2 = now();
  printf("flags=%d prio=%d per iteration (us): launch A %.1f | record A %.1f | B waits %.1f | launch B %.1f | record B %.1f | A waits %.1f\n", flags, prio,
         t[0] / R, t[1] / R, t[2] / R, t[3] / R, t[4] / R, t[5] / R);
  printf("host issued %d iterations in %.0f us, device done after %.0f us\n", R, T1 - T0, T2 - T0);
  return 0;
}
