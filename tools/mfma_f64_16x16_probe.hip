// Developer probe (round 6): v_mfma_f64_16x16x4_f64 on gfx950 -- (1) which lane / register holds which element of A, B, D;
// (2) issue rate of the two fp64 matrix instructions against v_fma_f64, alone and interleaved in one wavefront, and with the
// two kinds of work in different wavefronts of a SIMD (does the matrix pipe run beside the vector pipe?).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_16x16_probe.hip -o /tmp/probe16 && /tmp/probe16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_ __attribute__((ext_vector_type(4)));

// one-hot A at lane la (k-slice = its only register), one-hot B at lane lb: D non-zero in exactly one (lane, register)
__global__ void layout(int *out) {   // out[la * 64 + lb] = 1 + lane * 4 + reg, or 0
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      double4_ c = {0, 0, 0, 0};
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
      int hit = 0;
      for (int r = 0; r < 4; ++r) {
        const unsigned long long m = __ballot(c[r] != 0.0);
        if (m) hit = 1 + (__ffsll((long long)m) - 1) * 4 + r;
      }
      if (lane == 0) out[la * 64 + lb] = hit;
    }
}

// MODE 0: MFMA 16x16x4 only, 1: 4x4x4 only, 2: FMA only, 3: 16x16x4 + FMA interleaved, 4: 4x4x4 + FMA interleaved,
// 5: even waves 16x16x4 / odd waves FMA, 6: even waves 4x4x4 / odd waves FMA, 7: fp32 FMA only, 8: even waves 16x16x4 / odd waves fp32 FMA,
// 9: 32-bit integer multiply-add only, 10: even waves 16x16x4 / odd waves integer
template <int MODE>
__global__ __launch_bounds__(512) void rate(double *out, int iters) {
  const int wave = threadIdx.x >> 6;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  double4_ c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
  double x[16];
  for (int j = 0; j < 16; ++j) x[j] = threadIdx.x * 1e-3 + j;
  const double fa = 1.0000001, fb = 1e-9;
  const bool mf16 = MODE == 0 || MODE == 3 || (MODE == 5 && !(wave & 1));
  const bool mf4 = MODE == 1 || MODE == 4 || (MODE == 6 && !(wave & 1));
  const bool fm = MODE == 2 || MODE == 3 || MODE == 4 || ((MODE == 5 || MODE == 6) && (wave & 1));
  const bool f32 = MODE == 7 || (MODE == 8 && (wave & 1)), i32 = MODE == 9 || (MODE == 10 && (wave & 1));
  const bool mf16b = (MODE == 8 || MODE == 10) && !(wave & 1);
  float y[16];
  unsigned z[16];
  for (int j = 0; j < 16; ++j) { y[j] = threadIdx.x * 1e-3f + j; z[j] = threadIdx.x + j; }
  for (int i = 0; i < iters; ++i) {
    if (f32) {
#pragma unroll
      for (int j = 0; j < 16; ++j) y[j] = __builtin_fmaf(y[j], 1.0000001f, 1e-9f);
    }
    if (i32) {
#pragma unroll
      for (int j = 0; j < 16; ++j) z[j] = z[j] * 1664525u + 1013904223u;
    }
    if (mf16 || mf16b) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
    }
    if (mf4) {
      d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, a, d2, 0, 0, 0);
      d3 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, b, d3, 0, 0, 0);
    }
    if (fm) {
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = __builtin_fma(x[j], fa, fb);
    }
  }
  double s = c0[0] + c1[1] + c2[2] + c3[3] + d0 + d1 + d2 + d3;
  for (int j = 0; j < 16; ++j) s += x[j] + y[j] + z[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(double *d, int blocks, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(512), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  int *dl;
  (void)hipMalloc(&dl, 64 * 64 * sizeof(int));
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dl);
  std::vector<int> h(64 * 64);
  (void)hipMemcpy(h.data(), dl, h.size() * sizeof(int), hipMemcpyDeviceToHost);
  // A[i][k] in lane i + 16 k, B[k][n] in lane n + 16 k, D[i][n] in lane n + 16 (i % 4), register i / 4   (MI355X, round 6: confirmed)
  int bad = 0;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const int i = la & 15, ka = la >> 4, n = lb & 15, kb = lb >> 4;
      const int want = ka == kb ? 1 + (n + 16 * (i & 3)) * 4 + (i >> 2) : 0;
      if (h[la * 64 + lb] != want) {
        if (bad < 10) printf("layout mismatch: A lane %d B lane %d -> got %d want %d\n", la, lb, h[la * 64 + lb], want);
        ++bad;
      }
    }
  printf("16x16x4 layout (A[i][k] lane i+16k, B[k][n] lane n+16k, D[i][n] lane n+16(i%%4) reg i/4): %s (%d mismatches)\n",
         bad ? "WRONG" : "confirmed", bad);
  if (bad) {
    for (int la = 0; la < 64; la += 5) {
      printf("A lane %2d:", la);
      for (int lb = 0; lb < 64; ++lb)
        if (h[la * 64 + lb]) printf(" B%d->L%d.r%d", lb, (h[la * 64 + lb] - 1) / 4, (h[la * 64 + lb] - 1) % 4);
      printf("\n");
    }
  }
  double *d;
  (void)hipMalloc(&d, 512 * 2048 * 8);
  const int iters = 10000, blocks = 1024;   // 1024 blocks x 8 waves: 2 waves per SIMD on 256 CUs, 2 rounds
  const float t0 = run<0>(d, blocks, iters), t1 = run<1>(d, blocks, iters), t2 = run<2>(d, blocks, iters), t3 = run<3>(d, blocks, iters),
              t4 = run<4>(d, blocks, iters), t5 = run<5>(d, blocks, iters), t6 = run<6>(d, blocks, iters);
  const double waves = (double)blocks * 8;
  printf("mfma16x16x4 alone : %8.2f ms  %6.1f TFLOP/s  (%.1f cycles per instruction per SIMD at 2.4 GHz, 2 waves/SIMD)\n", t0,
         waves * iters * 4 * 2048.0 / (t0 * 1e-3) / 1e12, t0 * 1e-3 * 2.4e9 / (iters * 4.0 * 2 * (blocks / 256.0) / 1.0) );
  printf("mfma4x4x4_4b alone: %8.2f ms  %6.1f TFLOP/s\n", t1, waves * iters * 4 * 512.0 / (t1 * 1e-3) / 1e12);
  printf("v_fma_f64 alone   : %8.2f ms  %6.1f TFLOP/s\n", t2, waves * 64 * iters * 16 * 2.0 / (t2 * 1e-3) / 1e12);
  printf("same wave, 16x16x4 + fma interleaved: %8.2f ms (sum of the two alone %.2f, max %.2f)\n", t3, t0 + t2, t0 > t2 ? t0 : t2);
  printf("same wave, 4x4x4  + fma interleaved: %8.2f ms (sum %.2f, max %.2f)\n", t4, t1 + t2, t1 > t2 ? t1 : t2);
  printf("even waves 16x16x4, odd waves fma   : %8.2f ms (each alone at half the waves: %.2f / %.2f)\n", t5, t0 / 2, t2 / 2);
  printf("even waves 4x4x4,  odd waves fma   : %8.2f ms (each alone at half the waves: %.2f / %.2f)\n", t6, t1 / 2, t2 / 2);
  const float t7 = run<7>(d, blocks, iters), t8 = run<8>(d, blocks, iters), t9 = run<9>(d, blocks, iters), t10 = run<10>(d, blocks, iters);
  printf("v_fma_f32 alone   : %8.2f ms;  even waves 16x16x4, odd waves v_fma_f32: %8.2f ms (each alone at half the waves: %.2f / %.2f)\n", t7, t8, t0 / 2, t7 / 2);
  printf("v_mad_u32 alone   : %8.2f ms;  even waves 16x16x4, odd waves integer  : %8.2f ms (each alone at half the waves: %.2f / %.2f)\n", t9, t10, t0 / 2, t9 / 2);
  return 0;
}
