// Developer probe: how do the bits of hipExtStreamCreateWithCUMask map onto the 8 XCDs x 32 CUs of an MI355X, and what does a
// kernel on a masked stream cost / leave free?  For a handful of masks: which (XCC_ID, SE, CU) the workgroups of a 4096-workgroup
// launch landed on.  Use: reserve a CU per XCD for the comm stream (RCCL's kernel waits ~100 us for a slot beside a full-size stage
// kernel, tools/rccl_probe.hip section 3) without unbalancing the XCDs, whose workgroup share is fixed by the round-robin dispatch.
//   hipcc --offload-arch=gfx950 -O2 -o scratch/probe/cumask_probe tools/cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
#define CK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("%s: %s\n", #c, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void where(unsigned *out, long spin) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  double v = threadIdx.x;
  for (long i = 0; i < spin; ++i) v = v * 1.0000001 + 1e-9;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
  if (v == 12345.678) out[0] = 0;
}
int main() {
  const int nwg = 4096;
  unsigned *d;
  CK(hipMalloc(&d, nwg * 2 * sizeof(unsigned)));
  std::vector<unsigned> h(nwg * 2);
  struct M { const char *name; uint32_t w[8]; };
  std::vector<M> masks;
  masks.push_back({"all", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}});
  masks.push_back({"bits 0-7 only", {0xFFu, 0, 0, 0, 0, 0, 0, 0}});
  masks.push_back({"bits 0-31 only", {~0u, 0, 0, 0, 0, 0, 0, 0}});
  masks.push_back({"bit 0 of every word", {1, 1, 1, 1, 1, 1, 1, 1}});
  masks.push_back({"all but bits 0-7", {~0xFFu, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}});
  masks.push_back({"all but bits 248-255", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, 0x00FFFFFFu}});
  masks.push_back({"all but bits 0-15", {~0xFFFFu, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}});
  for (const M &m : masks) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m.w);
    if (e != hipSuccess) { printf("%-24s hipExtStreamCreateWithCUMask: %s\n", m.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
    CK(hipMemsetAsync(d, 0xFF, nwg * 2 * sizeof(unsigned), s));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    where<<<nwg, 256, 0, s>>>(d, 2000);
    CK(hipEventRecord(a, s));
    where<<<nwg, 256, 0, s>>>(d, 20000);
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipMemcpy(h.data(), d, nwg * 2 * sizeof(unsigned), hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> cus;   // xcc -> set of (se, sh, cu)
    std::map<unsigned, int> wgs;
    for (int i = 0; i < nwg; ++i) {
      const unsigned xcc = h[2 * i] & 0xF, hw = h[2 * i + 1];
      const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      cus[xcc].insert((se << 8) | (sh << 4) | cu);
      ++wgs[xcc];
    }
    int total = 0;
    printf("%-24s %7.1f us  ", m.name, ms * 1e3);
    for (auto &kv : cus) { printf(" xcc%u: %zu CUs / %d wgs", kv.first, kv.second.size(), wgs[kv.first]); total += (int)kv.second.size(); }
    printf("   total %d CUs\n", total);
    if (total <= 40) {
      for (auto &kv : cus) { printf("      xcc%u:", kv.first); for (unsigned c : kv.second) printf(" se%u.sh%u.cu%u", c >> 8, (c >> 4) & 1, c & 0xF); printf("\n"); }
    }
    CK(hipStreamDestroy(s));
  }
  return 0;
}
