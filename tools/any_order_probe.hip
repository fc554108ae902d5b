// Developer probe: does hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch) let a kernel start beside its predecessor in the SAME
// stream (no barrier bit on its packet)?  Per iteration: a small kernel (256 workgroups, ~20 us alone: the rim) and a big one
// (~100 us: the interior), which are independent of each other.
//   mode 0: both plain launches on one stream (serialised)         mode 1: the big one with hipExtAnyOrderLaunch
//   mode 2: two streams, no dependencies (what the driver has now)  mode 3: as 1, plus a tiny barrier kernel per iteration (the wait kernel)
//   hipcc --offload-arch=gfx950 -O2 -o scratch/probe/any_order_probe tools/any_order_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(double *x, long n) {
  double v = x[threadIdx.x + 64 * blockIdx.x];
  for (long i = 0; i < n; ++i) v = v * 1.0000001 + 1e-9;
  x[threadIdx.x + 64 * blockIdx.x] = v;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("%s: %s\n", #c, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  double *x;
  CK(hipMalloc(&x, 1 << 24));
  CK(hipMemset(x, 0, 1 << 24));
  hipStream_t M, C;
  CK(hipStreamCreate(&M));
  CK(hipStreamCreate(&C));
  long n = 20000;
  for (int it = 0; it < 6; ++it) {
    double t0 = now();
    spin<<<2048, 64, 0, M>>>(x, n);
    CK(hipStreamSynchronize(M));
    double dt = now() - t0;
    if (it < 5) n = (long)(n * 100.0 / dt) + 1;
  }
  const long ns = n / 5;
  const int R = 300;
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
      CK(hipDeviceSynchronize());
      const double t0 = now();
      for (int r = 0; r < R; ++r) {
        if (mode == 2) {
          spin<<<256, 64, 0, C>>>(x + (1 << 18), ns);
          spin<<<2048, 64, 0, M>>>(x, n);
          continue;
        }
        spin<<<256, 64, 0, M>>>(x + (1 << 18), ns);
        if (mode == 0) spin<<<2048, 64, 0, M>>>(x, n);
        else hipExtLaunchKernelGGL(spin, dim3(2048), dim3(64), 0, M, nullptr, nullptr, hipExtAnyOrderLaunch, x, n);
        if (mode == 3) spin<<<1, 64, 0, M>>>(x + (1 << 19), 10);
      }
      CK(hipStreamSynchronize(M));
      CK(hipStreamSynchronize(C));
      printf("rep %d mode %d: %.2f us per iteration\n", rep, mode, (now() - t0) / R);
    }
  return 0;
}
