// Developer probe: fp64 throughput of v_mfma_f64_16x16x4_f64 against v_fma_f64 on gfx950 (hipcc --offload-arch=gfx950 -O3).
// Result on MI355X: 48 vs 67 TFLOP/s -- the reason the Q3 contraction stays on the vector ALUs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_ __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  double4_ c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ __launch_bounds__(256) void k_fma(double *out, int iters) {
  double x[16];
  for (int j = 0; j < 16; ++j) x[j] = threadIdx.x * 1e-3 + j;
  const double a = 1.0000001, b = 1e-9;
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = __builtin_fma(x[j], a, b);
  double s = 0;
  for (int j = 0; j < 16; ++j) s += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double *d; (void)hipMalloc(&d, 256 * 2048 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000, blocks = 2048;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // one 16x16x4 MFMA = 16*16*4 FMAs = 2048 flop per wave instruction; 4 per iteration; 4 waves per block
    double tf = (double)blocks * 4 * iters * 4 * 2048.0 / (ms * 1e-3) / 1e12;
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms2; (void)hipEventElapsedTime(&ms2, e0, e1);
    double tf2 = (double)blocks * 256 * iters * 16 * 2.0 / (ms2 * 1e-3) / 1e12;
    printf("v_mfma_f64_16x16x4: %.1f TFLOP/s (%.2f ms)   v_fma_f64: %.1f TFLOP/s (%.2f ms)\n", tf, ms, tf2, ms2);
  }
}
