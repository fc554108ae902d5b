// Developer probe (round 6, LAB R6.3): what LAB R5.15 ran into, without the engine.  Two processes on one device.  The parent
// exports a block of device memory (uncached | finegrained | plain) through hipIpcGetMemHandle, the child maps it and stores into
// it; then either
//   early : the parent frees the block while the child still maps it (round 5's dflo_hip_multi_destroy), or
//   proper: the child closes its mapping first, the parent frees afterwards (round 6's order),
// and the parent at once allocates working buffers of the same size class, fills them by a kernel and verifies them over and over
// while the child (early: now) closes its mapping.  Any word of the parent's NEW buffers that changes under it is counted.
//   hipcc --offload-arch=gfx950 -O2 tools/ipc_free_order.hip -o /tmp/ipc_free_order
//   /tmp/ipc_free_order <uncached|finegrained|plain> <early|proper> [iterations = 200]
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(call)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); std::exit(2); } \
  } while (0)

constexpr size_t kBlock = 4u << 20;   // the size of the driver's window of sequence words
constexpr int kWords = (int)(kBlock / 8);

__global__ void store_all(unsigned long long *p, unsigned long long v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) __hip_atomic_store(p + i, v + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void fill(unsigned long long *p, unsigned long long v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v * 0x9E3779B97F4A7C15ull + i;
}
// (a workgroup verifies what the NEXT workgroup of the fill wrote -- workgroups go round the 8 XCDs, so every word is checked from another
//  XCD than the one that wrote it: a translation or a cache line that is stale in one XCD only would show; shift 0: the same XCD)
__global__ void verify(const unsigned long long *p, unsigned long long v, int n, unsigned int *bad, int shift) {
  const int i = (int)((blockIdx.x + shift) % gridDim.x) * blockDim.x + threadIdx.x;
  if (i < n && p[i] != v * 0x9E3779B97F4A7C15ull + i) atomicAdd(bad, 1u);
}
// ... and a kernel that, like a stage kernel, reads its neighbours' words and rewrites its own (q <- f(q, left, right)): the engine's
// data flow across XCDs on the recycled range, compared with the same recurrence on a buffer allocated before any window existed
__global__ void relax(const unsigned long long *in, unsigned long long *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long l = in[(i + n - 257) % n], r = in[(i + 257) % n];
  out[i] = in[i] * 6364136223846793005ull + (l ^ (r >> 7)) + 1442695040888963407ull;
}
__global__ void compare(const unsigned long long *a, const unsigned long long *b, int n, unsigned int *bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && a[i] != b[i]) atomicAdd(bad, 1u);
}

static void put(int fd, const void *b, size_t n) { if (write(fd, b, n) != (ssize_t)n) std::exit(3); }
static void get(int fd, void *b, size_t n) {
  size_t got = 0;
  while (got < n) {
    const ssize_t r = read(fd, (char *)b + got, n - got);
    if (r <= 0) std::exit(3);
    got += (size_t)r;
  }
}

int main(int argc, char **argv) {
  const char *kind = argc > 1 ? argv[1] : "uncached";
  const bool early = argc > 2 && std::strcmp(argv[2], "early") == 0;
  const int iters = argc > 3 ? std::atoi(argv[3]) : 200;
  int p2c[2], c2p[2];
  if (pipe(p2c) || pipe(c2p)) return 3;
  const pid_t pid = fork();   // before the first HIP call of either process
  if (pid == 0) {             // ---- the neighbour: maps, stores, closes when told
    CHK(hipSetDevice(0));
    for (int it = 0; it < iters; ++it) {
      hipIpcMemHandle_t h;
      get(p2c[0], &h, sizeof(h));
      void *w = nullptr;
      CHK(hipIpcOpenMemHandle(&w, h, hipIpcMemLazyEnablePeerAccess));
      hipLaunchKernelGGL(store_all, dim3(kWords / 256), dim3(256), 0, 0, (unsigned long long *)w, (unsigned long long)it << 32, kWords);
      CHK(hipDeviceSynchronize());
      char c = 'o';
      put(c2p[1], &c, 1);
      get(p2c[0], &c, 1);                       // "close now"
      usleep(200 + (std::rand() % 3000));       // (early: the parent is already working in its new buffers)
      CHK(hipIpcCloseMemHandle(w));
      c = 'c';
      put(c2p[1], &c, 1);
    }
    return 0;
  }
  CHK(hipSetDevice(0));
  unsigned int *bad = nullptr;
  CHK(hipMalloc((void **)&bad, 4));
  CHK(hipMemset(bad, 0, 4));
  unsigned long long *ref[2];   // allocated before any window exists: never in a recycled range
  CHK(hipMalloc((void **)&ref[0], kBlock));
  CHK(hipMalloc((void **)&ref[1], kBlock));
  long total_bad = 0, bad_iters = 0;
  for (int it = 0; it < iters; ++it) {
    void *blk = nullptr;
    if (!std::strcmp(kind, "uncached")) CHK(hipExtMallocWithFlags(&blk, kBlock, hipDeviceMallocUncached));
    else if (!std::strcmp(kind, "finegrained")) CHK(hipExtMallocWithFlags(&blk, kBlock, hipDeviceMallocFinegrained));
    else CHK(hipMalloc(&blk, kBlock));
    CHK(hipMemset(blk, 0, kBlock));
    hipIpcMemHandle_t h;
    CHK(hipIpcGetMemHandle(&h, blk));
    put(p2c[1], &h, sizeof(h));
    char c;
    get(c2p[0], &c, 1);                         // the neighbour has mapped the block and stored into it
    if (early) {
      CHK(hipFree(blk));                        // ... while the neighbour still maps it
      c = 'x';
      put(p2c[1], &c, 1);
    } else {
      c = 'x';
      put(p2c[1], &c, 1);
      get(c2p[0], &c, 1);                       // closed over there
      CHK(hipFree(blk));
    }
    // the next allocations of this process: what the first engine after the driver would take
    std::vector<unsigned long long *> bufs;
    for (int k = 0; k < 6; ++k) {
      unsigned long long *b = nullptr;
      CHK(hipMalloc((void **)&b, kBlock));
      bufs.push_back(b);
      hipLaunchKernelGGL(fill, dim3(kWords / 256), dim3(256), 0, 0, b, (unsigned long long)(it * 8 + k), kWords);
    }
    CHK(hipDeviceSynchronize());
    for (int rep = 0; rep < 40; ++rep) {
      for (int k = 0; k < 6; ++k)
        hipLaunchKernelGGL(verify, dim3(kWords / 256), dim3(256), 0, 0, bufs[k], (unsigned long long)(it * 8 + k), kWords, bad, rep % 9);
      CHK(hipDeviceSynchronize());
      usleep(100);
    }
    // the recurrence: 30 sweeps ping-ponging between two of the new buffers, against the same on the two old ones
    hipLaunchKernelGGL(fill, dim3(kWords / 256), dim3(256), 0, 0, ref[0], 7ull, kWords);
    hipLaunchKernelGGL(fill, dim3(kWords / 256), dim3(256), 0, 0, bufs[0], 7ull, kWords);
    for (int sw = 0; sw < 30; ++sw) {
      hipLaunchKernelGGL(relax, dim3(kWords / 256), dim3(256), 0, 0, ref[sw & 1], ref[(sw + 1) & 1], kWords);
      hipLaunchKernelGGL(relax, dim3(kWords / 256), dim3(256), 0, 0, bufs[sw & 1], bufs[(sw + 1) & 1], kWords);
    }
    hipLaunchKernelGGL(compare, dim3(kWords / 256), dim3(256), 0, 0, ref[0], bufs[0], kWords, bad);
    CHK(hipDeviceSynchronize());
    unsigned int nb = 0;
    CHK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
    if (nb) { ++bad_iters; total_bad += nb; CHK(hipMemset(bad, 0, 4)); }
    for (auto *b : bufs) CHK(hipFree(b));
    if (early) get(c2p[0], &c, 1);              // (the neighbour's close of this round)
  }
  int st = 0;
  waitpid(pid, &st, 0);
  std::printf("%s block, %s free: %d iterations, %ld with words of the NEW buffers changed (%ld word reads in all); neighbour exit status %d\n", kind,
              early ? "EARLY (freed while the neighbour maps it)" : "proper (neighbour closes first)", iters, bad_iters, total_bad, WEXITSTATUS(st));
  return bad_iters ? 1 : 0;
}
