// hbm_probe -- what a plain streaming kernel reaches on this GPU, in the stage kernel's own access pattern:
// 2 reads + 1 write of fp64 vectors of C2's size (37 748 736 doubles = 302 MB each), and 1R+1W / read-only for reference.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/hbm_probe tools/hbm_probe.hip && scratch/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

template <int MODE, int V>   // MODE 0: c = a + s b (2R 1W), 1: c = a (1R 1W), 2: sum(a) (1R)
__global__ __launch_bounds__(256) void stream_kernel(const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ c, size_t n, double s) {
  typedef double vec __attribute__((ext_vector_type(V)));
  const size_t nv = n / V;
  vec acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    const vec x = ((const vec *)a)[i];
    if (MODE == 0) ((vec *)c)[i] = x + s * ((const vec *)b)[i];
    else if (MODE == 1) ((vec *)c)[i] = x;
    else acc += x;
  }
  if (MODE == 2) { double t = 0; for (int k = 0; k < V; ++k) t += acc[k]; if (t == 1.2345e300) c[0] = t; }
}

template <int MODE, int V>
double run(const double *a, const double *b, double *c, size_t n, int grid, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((stream_kernel<MODE, V>), dim3(grid), dim3(256), 0, 0, a, b, c, n, 0.5);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<MODE, V>), dim3(grid), dim3(256), 0, 0, a, b, c, n, 0.5);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (MODE == 0 ? 3.0 : (MODE == 1 ? 2.0 : 1.0)) * n * 8.0;
  return bytes * reps / (ms * 1e-3) / 1e9;
}

int main() {
  const size_t n = 37748736;
  double *a, *b, *c;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&c, n * 8));
  CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8)); CK(hipMemset(c, 0, n * 8));
  // settle the clocks
  for (int i = 0; i < 400; ++i) hipLaunchKernelGGL((stream_kernel<0, 2>), dim3(4096), dim3(256), 0, 0, a, b, c, n, 0.5);
  CK(hipDeviceSynchronize());
  const int grids[] = {1024, 2048, 4096, 8192, 16384, 73728};
  std::printf("{\"n_doubles\": %zu, \"rows\": [\n", n);
  bool first = true;
  for (int g : grids) {
    const double r[6] = {run<0, 1>(a, b, c, n, g, 100), run<0, 2>(a, b, c, n, g, 100), run<1, 1>(a, b, c, n, g, 100),
                         run<1, 2>(a, b, c, n, g, 100), run<2, 1>(a, b, c, n, g, 100), run<2, 2>(a, b, c, n, g, 100)};
    std::printf("%s  {\"grid\": %d, \"2r1w_f64\": %.0f, \"2r1w_f64x2\": %.0f, \"1r1w_f64\": %.0f, \"1r1w_f64x2\": %.0f, \"1r_f64\": %.0f, \"1r_f64x2\": %.0f}",
                first ? "" : ",\n", g, r[0], r[1], r[2], r[3], r[4], r[5]);
    first = false;
  }
  std::printf("\n], \"unit\": \"GB/s\"}\n");
  return 0;
}
