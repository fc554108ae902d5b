// Developer probe: which lane holds which element of A, B and D in v_mfma_f64_4x4x4_4b_f64 (gfx950)?
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_4x4_layout_probe.hip -o /tmp/probe && /tmp/probe
// One-hot A at lane la, one-hot B at lane lb: D = sum_k A_b[i][k] B_b[k][n] is non-zero in exactly one lane iff la and
// lb share the block b and the index k.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int *out) {   // out[la*64 + lb] = D lane + 1, or 0
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) out[la * 64 + lb] = m ? __ffsll((long long)m) : 0;
    }
}
int main() {
  int *d;
  hipMalloc(&d, 64 * 64 * sizeof(int));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  std::vector<int> h(64 * 64);
  hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb]) printf("  B%-2d->D%-2d", lb, h[la * 64 + lb] - 1);
    printf("\n");
  }
  return 0;
}
