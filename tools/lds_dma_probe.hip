// Developer probe (round 6, LAB R6.14): where do global_load_lds_dword / _dwordx3 / _dwordx4 put the bytes of lane i?
//   hipcc --offload-arch=gfx950 -O2 tools/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using G = const __attribute__((address_space(1))) void *;
using L = __attribute__((address_space(3))) void *;
template <int SIZE>
__global__ void probe(const unsigned *g, unsigned *out, int base_words) {
  extern __shared__ unsigned lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  if constexpr (SIZE == 4) __builtin_amdgcn_global_load_lds((G)((const char *)g + 4 * lane), (L)(lds + base_words), 4, 0, 0);
  else if constexpr (SIZE == 12) __builtin_amdgcn_global_load_lds((G)((const char *)g + 12 * lane), (L)(lds + base_words), 12, 0, 0);
  else __builtin_amdgcn_global_load_lds((G)((const char *)g + 16 * lane), (L)(lds + base_words), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
}
template <int SIZE>
void run(const unsigned *dg, unsigned *dout, int base_words) {
  hipLaunchKernelGGL(probe<SIZE>, dim3(1), dim3(64), 4096, 0, dg, dout, base_words);
  std::vector<unsigned> h(1024);
  (void)hipMemcpy(h.data(), dout, 4096, hipMemcpyDeviceToHost);
  // expected: word base + k holds source word k for k < 64 * SIZE / 4
  int bad = 0, first_bad = -1;
  for (int k = 0; k < 64 * SIZE / 4; ++k)
    if (h[base_words + k] != (unsigned)k) { if (first_bad < 0) first_bad = k; ++bad; }
  int touched = 0;
  for (int i = 0; i < 1024; ++i) touched += h[i] != 0xdeadbeefu;
  printf("size %2d base %3d words: %d of %d words not where 'lane i at base + i * size' puts them (first %d), %d words written in all\n", SIZE, base_words,
         bad, 64 * SIZE / 4, first_bad, touched);
  if (bad) {
    printf("  words 0..23 behind the base:");
    for (int k = 0; k < 24; ++k) printf(" %x", h[base_words + k]);
    printf("\n");
  }
}
int main() {
  unsigned *dg, *dout;
  std::vector<unsigned> src(1024);
  for (int i = 0; i < 1024; ++i) src[i] = i;
  (void)hipMalloc(&dg, 4096);
  (void)hipMalloc(&dout, 4096);
  (void)hipMemcpy(dg, src.data(), 4096, hipMemcpyHostToDevice);
  run<4>(dg, dout, 0);
  run<4>(dg, dout, 64);
  run<12>(dg, dout, 0);
  run<12>(dg, dout, 192);
  run<16>(dg, dout, 0);
  run<16>(dg, dout, 256);
  return 0;
}
