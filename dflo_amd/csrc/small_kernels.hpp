// small_kernels.hpp -- boundary programs, layout conversion, halo pack / unpack, averages, time step, reductions
// Part of the device side of engine.hip (see there for the layout of the data and of a stage).
#pragma once
#include "limiter_kernels.hpp"

namespace dflo {

// accuracy probe of the reciprocal / square-root forms used by the flux functions
__global__ void debug_math_kernel(const double *x, double *rcp, double *sq, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    rcp[i] = frcp(x[i]);
    sq[i] = fsqrt(x[i]);
  }
}

// fexp_neg against the library's exp()
__global__ void debug_exp_kernel(const double *x, double *lib, double *fast, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    lib[i] = exp(x[i]);
    fast[i] = fexp_neg(x[i]);
  }
}

// ------------------------------------------------------------------ small kernels
constexpr int kMaxSegs = 16;   // peers of one part (dflo_hip_pack_send_to)
// quadrature weight of node j for the cell average: w_a w_b (squares) or w_a w_b det J / |K| (bilinear cells)
__device__ __forceinline__ double avg_weight(const KBasis &kb, int N, int j, const double *vert, int n_slots, int slot,
                                             double inv_area) {
  if (N < 0) return j == 0 ? 1.0 : 0.0;  // Pk (flagged by N < 0): the average is mode 0
  const double ww = kb.w[j % N] * kb.w[j / N];
  if (!vert) return ww;
  double v[8];
  for (int k = 0; k < 8; ++k) v[k] = vert[(size_t)k * n_slots + slot];
  const double xi = kb.x[j % N], eta = kb.x[j / N];
  const double xxi = (v[2] - v[0]) + eta * ((v[6] - v[4]) - (v[2] - v[0])), yxi = (v[3] - v[1]) + eta * ((v[7] - v[5]) - (v[3] - v[1]));
  const double xeta = (v[4] - v[0]) + xi * ((v[6] - v[2]) - (v[4] - v[0])), yeta = (v[5] - v[1]) + xi * ((v[7] - v[3]) - (v[5] - v[1]));
  return ww * (xxi * yeta - xeta * yxi) * inv_area;
}
__device__ __forceinline__ double cell_inv_area(const double *vert, int n_slots, int slot) {
  if (!vert) return 1.0;
  double v[8];
  for (int k = 0; k < 8; ++k) v[k] = vert[(size_t)k * n_slots + slot];
  return 1.0 / (0.5 * fabs((v[0] * v[3] - v[2] * v[1]) + (v[2] * v[7] - v[6] * v[3]) + (v[6] * v[5] - v[4] * v[7]) +
                           (v[4] * v[1] - v[0] * v[5])));
}
// user (dflo) layout <-> shard SoA layout
__global__ void scatter_kernel(const double *user, double *U, const int32_t *user_of, int n_slots, int ndof) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n_slots * ndof) return;
  const int slot = (int)(t / ndof), d = (int)(t - (long long)slot * ndof);
  const int uc = user_of[slot];
  const int ns = ndof / 4;
  // padding slots hold a harmless state (rho = 1, E = 1)
  const double v = uc >= 0 ? user[(size_t)uc * ndof + d] : ((d / ns) >= 2 ? 1.0 : 0.0);
  U[((size_t)(slot >> 6) * ndof + d) * 64 + (slot & 63)] = v;
}
__global__ void gather_kernel(double *user, const double *U, const int32_t *iid, int n_cells, int ndof) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n_cells * ndof) return;
  const int c = (int)(t / ndof), d = (int)(t - (long long)c * ndof);
  const int slot = iid[c];
  user[t] = U[((size_t)(slot >> 6) * ndof + d) * 64 + (slot & 63)];
}
// A pack kernel is the first kernel of the comm stream behind the rim launch: where the multi-device schedule wants "the rim launch
// has ended" published (dflo_hip_pack_publish; the kernel boundary has released the rim's stores), its first thread does that
// before anything else -- no kernel or stream operation of its own in front of the send.
__device__ __forceinline__ void pack_publish(unsigned long long *word, unsigned long long seq) {
  if (word && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// pack listed cells (internal slots) cell-major: buf[k][ndof]
__global__ void pack_kernel(double *buf, const double *U, const int32_t *slots, int n, int ndof, unsigned long long *pub, unsigned long long pub_seq) {
  pack_publish(pub, pub_seq);
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * ndof) return;
  const int k = (int)(t / ndof), d = (int)(t - (long long)k * ndof);
  const int slot = slots[k];
  buf[t] = U[((size_t)(slot >> 6) * ndof + d) * 64 + (slot & 63)];
}
// The same with the owner's cell average behind the DoFs of every cell, buf[k][ndof + 4]: a ghost cell then carries the
// very bits its owner holds (an average formed again from the DoFs would differ from the stage kernel's in the last
// place, and the LxF flux and the TVB differences read it)
__global__ void pack_cells_kernel(double *buf, const double *U, const double *avg, const int32_t *slots, int n, int ndof, unsigned long long *pub,
                                  unsigned long long pub_seq) {
  pack_publish(pub, pub_seq);
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int w = ndof + 4;
  if (t >= (long long)n * w) return;
  const int k = (int)(t / w), d = (int)(t - (long long)k * w);
  const int slot = slots[k];
  buf[t] = d < ndof ? U[((size_t)(slot >> 6) * ndof + d) * 64 + (slot & 63)]
                    : avg[((size_t)(slot >> 6) * 4 + (d - ndof)) * 64 + (slot & 63)];
}
__global__ void unpack_cells_kernel(const double *buf, double *U, double *avg, int first_slot, int n_ghost, int ndof) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int w = ndof + 4;
  if (t >= (long long)n_ghost * w) return;
  const int g = (int)(t / w), d = (int)(t - (long long)g * w);
  const int slot = first_slot + g;
  if (d < ndof) U[((size_t)(slot >> 6) * ndof + d) * 64 + (slot & 63)] = buf[t];
  else avg[((size_t)(slot >> 6) * 4 + (d - ndof)) * 64 + (slot & 63)] = buf[t];
}
// One exchange per stage where a TVB limiter sits between update and send (src_mpi/limiter.cc:232 and src_mpi/claw.cc:793 merged):
// the cut cells leave UNLIMITED, and with each of them what its owner's limiter would read -- buf[k][ndof + 4 + 16]: the DoFs, the
// cell's average, the averages of its four face neighbours as the owner holds them (0 where there is none; the entry of a face
// that leads to the receiver is not read there -- the receiver has that cell).  The receiver limits the ghost cell as the owner
// limits the original (limiter_kernel over the ghost shards) and forms the traces itself.
constexpr int kFatExtra = 20;
__device__ __forceinline__ double fat_value(const double *U, const double *avg, const int32_t *lrbt, int slot, int d, int ndof) {
  if (d < ndof) return U[((size_t)(slot >> 6) * ndof + d) * 64 + (slot & 63)];
  d -= ndof;
  if (d < 4) return avg[((size_t)(slot >> 6) * 4 + d) * 64 + (slot & 63)];
  d -= 4;
  const int nb = lrbt[((size_t)(slot >> 6) * 4 + (d >> 2)) * 64 + (slot & 63)];
  return nb >= 0 ? avg[((size_t)(nb >> 6) * 4 + (d & 3)) * 64 + (nb & 63)] : 0.0;
}
__global__ void pack_fat_kernel(double *buf, const double *U, const double *avg, const int32_t *lrbt, const int32_t *slots, int n, int ndof,
                                unsigned long long *pub, unsigned long long pub_seq) {
  pack_publish(pub, pub_seq);
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int w = ndof + kFatExtra;
  if (t >= (long long)n * w) return;
  const int k = (int)(t / w), d = (int)(t - (long long)k * w);
  buf[t] = fat_value(U, avg, lrbt, slots[k], d, ndof);
}
// ... into the ghost shards, the averages' array and gnb[g][4 faces][4]
__global__ void unpack_fat_kernel(const double *buf, double *U, double *avg, double *gnb, int first_slot, int n_ghost, int ndof) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int w = ndof + kFatExtra;
  if (t >= (long long)n_ghost * w) return;
  const int g = (int)(t / w), d = (int)(t - (long long)g * w);
  const int slot = first_slot + g;
  if (d < ndof) U[((size_t)(slot >> 6) * ndof + d) * 64 + (slot & 63)] = buf[t];
  else if (d < ndof + 4) avg[((size_t)(slot >> 6) * 4 + (d - ndof)) * 64 + (slot & 63)] = buf[t];
  else gnb[(size_t)g * 16 + (d - ndof - 4)] = buf[t];
}
// Face-trace halo records (N*4 doubles per cut face instead of the whole cell): record k = the trace of the
// listed cell on the listed face, out[k][4][N], formed exactly as the stage kernel's halo gather forms it.  Used to pack
// what a peer needs (owned cells on the cut) and to initialise the ghost traces from the ghost cells' DoFs after set_solution.
template <int N>
__global__ void face_trace_kernel(double *out, const double *U, const int32_t *slots, const int32_t *faces, int n, unsigned long long *pub,
                                  unsigned long long pub_seq) {
  pack_publish(pub, pub_seq);
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * 4 * N) return;
  const int k = (int)(t / (4 * N)), r = (int)(t - (long long)k * 4 * N);
  out[t] = cell_face_trace<N>(U, slots[k], faces[k], r / N, r % N);
}
// The same three packers delivering as they pack (one process, several engines): record k of the send list goes to the
// receive area of the peer it is meant for, seg.dst[i] + (k - seg.first[i]) * width for the segment i that holds k -- a
// plain store on this device, a store over xGMI peer access on another -- so that no copy per peer follows the kernel.
struct SendSegs {
  int n;
  int first[kMaxSegs + 1];
  double *dst[kMaxSegs];
  // one process per GPU, receive areas mapped through IPC handles: the kernel also tells the receivers.  The workgroup that
  // finishes last (counter `done` of this engine, zero between launches) stores `seq` into the receivers' sequence words, behind
  // a system-scope release: every record of this launch is visible to a peer that has seen the word.
  unsigned long long *flag[kMaxSegs];
  unsigned long long seq;
  unsigned int *done;
  unsigned long long *pub;       // pack_publish
  unsigned long long pub_seq;
};
__device__ __forceinline__ void seg_signal(const SendSegs &s) {
  if (!s.done) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wavefront's stores have left the compute unit before the barrier
  __syncthreads();                     // ... of all of this workgroup's wavefronts ...
  if (threadIdx.x != 0) return;
  __threadfence_system();              // ... and are visible to the other devices
  if (atomicAdd(s.done, 1u) != gridDim.x - 1) return;
  __threadfence_system();              // (acquire side: the other workgroups' releases)
  __hip_atomic_store(s.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch on this stream finds it at zero
  for (int i = 0; i < s.n; ++i)
    if (s.flag[i]) __hip_atomic_store(s.flag[i], s.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double *seg_dst(const SendSegs &s, int k, int width) {
  int i = 0;
  while (i + 1 < s.n && k >= s.first[i + 1]) ++i;
  return s.dst[i] + (size_t)(k - s.first[i]) * width;
}
// kind 0: DoFs + average, width ndof + 4; kind 1: averages (U unused), width 4
__global__ void pack_to_kernel(const SendSegs seg, const double *U, const double *avg, const int32_t *slots, int n, int ndof, int with_dofs) {
  pack_publish(seg.pub, seg.pub_seq);
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int w = with_dofs ? ndof + 4 : 4;
  if (t < (long long)n * w) {
    const int k = (int)(t / w), d = (int)(t - (long long)k * w);
    const int slot = slots[k];
    const int nd = with_dofs ? ndof : 0;
    const double v = d < nd ? U[((size_t)(slot >> 6) * ndof + d) * 64 + (slot & 63)]
                            : avg[((size_t)(slot >> 6) * 4 + (d - nd)) * 64 + (slot & 63)];
    seg_dst(seg, k, w)[d] = v;
  }
  seg_signal(seg);
}
__global__ void pack_fat_to_kernel(const SendSegs seg, const double *U, const double *avg, const int32_t *lrbt, const int32_t *slots, int n, int ndof) {
  pack_publish(seg.pub, seg.pub_seq);
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int w = ndof + kFatExtra;
  if (t < (long long)n * w) {
    const int k = (int)(t / w), d = (int)(t - (long long)k * w);
    seg_dst(seg, k, w)[d] = fat_value(U, avg, lrbt, slots[k], d, ndof);
  }
  seg_signal(seg);
}
template <int N>
__global__ void face_trace_to_kernel(const SendSegs seg, const double *U, const int32_t *slots, const int32_t *faces, int n) {
  pack_publish(seg.pub, seg.pub_seq);
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (long long)n * 4 * N) {
    const int k = (int)(t / (4 * N)), r = (int)(t - (long long)k * 4 * N);
    seg_dst(seg, k, 4 * N)[r] = cell_face_trace<N>(U, slots[k], faces[k], r / N, r % N);
  }
  seg_signal(seg);
}
// ghost cells: staging buffer [g][ndof] -> ghost shards, and their cell averages
// one thread per (ghost cell, component): its DoFs travel buffer -> ghost shard (the buffer is read with unit stride
// along the thread's own run of n_s values, the shard rows are written 64 cells wide) and their average is formed
__global__ void unpack_ghost_kernel(const double *buf, double *U, double *avg, int first_slot, int n_ghost, int ndof,
                                    KBasis kb, int N, const double *vert, int n_slots) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_ghost * 4) return;
  const int g = t >> 2, c = t & 3;
  const int slot = first_slot + g, ns = ndof / 4;
  const double ia = cell_inv_area(vert, n_slots, slot);
  double m = 0;
  for (int j = 0; j < ns; ++j) {
    const double v = buf[(size_t)g * ndof + c * ns + j];
    U[((size_t)(slot >> 6) * ndof + c * ns + j) * 64 + (slot & 63)] = v;
    m += avg_weight(kb, N, j, vert, n_slots, slot, ia) * v;
  }
  avg[((size_t)(slot >> 6) * 4 + c) * 64 + (slot & 63)] = m;
}
__global__ void unpack_ghost_avg_kernel(const double *buf, double *avg, int first_slot, int n_ghost) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_ghost) return;
  const int slot = first_slot + g;
  for (int c = 0; c < 4; ++c) avg[((size_t)(slot >> 6) * 4 + c) * 64 + (slot & 63)] = buf[(size_t)g * 4 + c];
}
// compute_cell_average (src/claw.cc:562-597) for all slots (owned and ghost shards)
__global__ void average_kernel(const double *U, double *avg, int ndof, KBasis kb, int N, const double *vert, int n_slots) {
  const int shard = blockIdx.x;
  const int lane = threadIdx.x;
  const int ns = ndof / 4, slot = shard * 64 + lane;
  const double ia = cell_inv_area(vert, n_slots, slot);
  for (int c = 0; c < 4; ++c) {
    double m = 0;
    for (int j = 0; j < ns; ++j)
      m += avg_weight(kb, N, j, vert, n_slots, slot, ia) * U[((size_t)shard * ndof + c * ns + j) * 64 + lane];
    avg[((size_t)shard * 4 + c) * 64 + lane] = m;
  }
}
// The same for Qk on squares with the sums in the order of the stage kernels' epilogue (cell_average_rows): an average formed
// here -- after set_solution, or on demand for a stage that kept its averages to itself -- carries the bits the epilogue would
// have stored, and the bits the LxF flux forms from the DoFs when the arrays of averages are off its path (stage_kernel AF).
template <int N>
__global__ __launch_bounds__(64) void average_rows_kernel(const double *U, double *avg) {
  constexpr int NS = N * N;
  const int shard = blockIdx.x, lane = threadIdx.x;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double u[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) u[j] = U[((size_t)shard * 4 * NS + c * NS + j) * 64 + lane];
    avg[((size_t)shard * 4 + c) * 64 + lane] = cell_average_rows<N>(u);
  }
}
// Pk on bilinear cells: the two quantities the stage kernel does not leave behind there, from the modes -- the cell average
// (quadrature of the expansion over the mapped cell, src/claw.cc:589-593; `avg` non-null: all slots) and / or the time step of
// compute_time_step_q (`shard_dtmin` non-null: owned shards; the 4 x 4 trapezoid points are reached through the values at the
// Gauss nodes, which carry the P_k function exactly).  Lane = cell.
template <int N>
__global__ __launch_bounds__(64) void modal_geo_kernel(const double *U, double *avg, double *shard_dtmin, const double *cell_h,
                                                       const double *vert, int n_slots, const int32_t *shard_count, KBasis kb,
                                                       double cfl, int degree, double *dt_cell) {
  constexpr int NS = N * N, NM = N * (N + 1) / 2;
  const int shard = blockIdx.x, lane = threadIdx.x;
  const size_t slot = (size_t)shard * 64 + lane;
  double v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = vert[(size_t)k * n_slots + slot];
  double u[4 * NS];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double um[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) um[m] = U[((size_t)shard * 4 * NM + c * NM + m) * 64 + lane];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      double t = 0.0;
#pragma unroll
      for (int m = 0; m < NM; ++m) t += PB<N>::t.T[j][m] * um[m];
      u[c * NS + j] = t;
    }
  }
  if (avg) {
    const double ax = v[2] - v[0], bx = (v[6] - v[4]) - ax, ay = v[3] - v[1], by = (v[7] - v[5]) - ay;
    const double cx = v[4] - v[0], dx = (v[6] - v[2]) - cx, cy = v[5] - v[1], dy = (v[7] - v[3]) - cy;
    const double area = 0.5 * fabs((v[0] * v[3] - v[2] * v[1]) + (v[2] * v[7] - v[6] * v[3]) + (v[6] * v[5] - v[4] * v[7]) + (v[4] * v[1] - v[0] * v[5]));
    const double ia = 1.0 / area;
    double A[4] = {0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < N; ++b)
#pragma unroll
      for (int aa = 0; aa < N; ++aa) {
        const double det = (ax + CB<N>::t.x[b] * bx) * (cy + CB<N>::t.x[aa] * dy) - (cx + CB<N>::t.x[aa] * dx) * (ay + CB<N>::t.x[b] * by);
        const double jxw = CB<N>::t.w[aa] * CB<N>::t.w[b] * det;
#pragma unroll
        for (int c = 0; c < 4; ++c) A[c] += u[c * NS + aa + N * b] * jxw;
      }
#pragma unroll
    for (int c = 0; c < 4; ++c) avg[((size_t)shard * 4 + c) * 64 + lane] = A[c] * ia;
  }
  if (shard_dtmin) {
    double dtmin = 1.0e20;
    if (lane < shard_count[shard]) {
      dtmin = dt_q_cell<N>(u, kb, cell_h[slot], cfl, degree);
      if (dt_cell) dt_cell[slot] = dtmin;
    }
    dtmin = wave_min(dtmin);
    if (lane == 0) shard_dtmin[shard] = dtmin;
  }
}
// compute_time_step_q (src/claw.cc:520-557): max of |v| + c over the 4 x 4 points of QIterated(QTrapez,3),
// dt = cfl h / lambda / (2k+1) with h = diameter / sqrt(2); per-shard minimum.  Lane = cell; the
// interpolation to the 16 points is sum-factorised (xi first, then eta).
template <int N>
__global__ __launch_bounds__(64) void dt_q_kernel(const double *U, const double *cell_h, const int32_t *shard_count,
                                                  double *shard_dtmin, KBasis kb, double cfl, int degree, double *dt_cell) {
  constexpr int NS = N * N, NDOF = 4 * NS;
  const int shard = blockIdx.x;
  const int lane = threadIdx.x;
  double dtmin = 1.0e20;
  if (lane < shard_count[shard]) {
    double u[NDOF];
#pragma unroll
    for (int j = 0; j < NDOF; ++j) u[j] = U[((size_t)shard * NDOF + j) * 64 + lane];
    dtmin = dt_q_cell<N>(u, kb, cell_h[(size_t)shard * 64 + lane], cfl, degree);
    if (dt_cell) dt_cell[(size_t)shard * 64 + lane] = dtmin;
  }
  dtmin = wave_min(dtmin);
  if (lane == 0) shard_dtmin[shard] = dtmin;
}
// compute_time_step_cartesian (src/claw.cc:486-511): per-shard minimum from the stored cell averages
__global__ void dt_kernel(const double *avg, const double *cell_h, double h_uniform, int uniform_h,
                          const int32_t *shard_count, double *shard_dtmin, double cfl, int degree, double *dt_cell) {
  const int shard = blockIdx.x;
  const int lane = threadIdx.x;
  double dtmin = 1.0e20;
  if (lane < shard_count[shard]) {
    double A[4];
    for (int c = 0; c < 4; ++c) A[c] = avg[((size_t)shard * 4 + c) * 64 + lane];
    const double h = uniform_h ? h_uniform : cell_h[(size_t)shard * 64 + lane];
    dtmin = cfl_dt(A, h, cfl, degree);
    if (dt_cell) dt_cell[(size_t)shard * 64 + lane] = dtmin;  // "time step type = local": dt(c), src/claw.cc:506
  }
  dtmin = wave_min(dtmin);
  if (lane == 0) shard_dtmin[shard] = dtmin;
}

__global__ __launch_bounds__(256) void finalize_kernel(const FinalArgs a) {
  __shared__ double sred[4][4];
  __shared__ int is_last;
  // Two levels, both with a fixed assignment and a fixed combination order -> deterministic sums:
  // workgroup b reduces the shards [b chunk, (b+1) chunk) of every array (the residual partials of all stages
  // of the step and the CFL minima), the workgroup that finishes last combines the kFinBlocks partials in index order.
  const int n = a.n_shards, t = threadIdx.x, b = blockIdx.x;
  const int chunk = ((n + kFinBlocks - 1) / kFinBlocks + 255) & ~255;
  const int lo = b * chunk, hi = min(n, lo + chunk);
  double rs[3] = {0.0, 0.0, 0.0}, m = 1.0e20;
  // four passes of the loop at a time: their loads are in flight together (one trip to memory instead of four)
  for (int s0 = lo + t; s0 < hi; s0 += 4 * 256) {
    double v[4][3], d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = s0 + j * 256;
      const bool in = s < hi;
#pragma unroll
      for (int st = 0; st < 3; ++st) v[j][st] = (in && st < a.n_stages) ? a.shard_res[(size_t)st * a.res_stride + s] : 0.0;
      d[j] = (in && a.do_dt) ? a.shard_dtmin[s] : 1.0e20;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int st = 0; st < 3; ++st) rs[st] += v[j][st];
      m = fmin(m, d[j]);
    }
  }
  for (int st = 0; st < 3; ++st) {
    const double r = wave_sum(rs[st]);
    if ((t & 63) == 0) sred[st][t >> 6] = r;
  }
  m = wave_min(m);
  if ((t & 63) == 0) sred[3][t >> 6] = m;
  __syncthreads();
  if (t == 0) {
    for (int st = 0; st < 3; ++st) a.partial[b * 4 + st] = (sred[st][0] + sred[st][1]) + (sred[st][2] + sred[st][3]);
    a.partial[b * 4 + 3] = fmin(fmin(sred[3][0], sred[3][1]), fmin(sred[3][2], sred[3][3]));
    __threadfence();
    is_last = atomicAdd(a.counter, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  __shared__ double spart[kFinBlocks * 4];
  if (t < (int)gridDim.x * 4) spart[t] = ((const volatile double *)a.partial)[t];  // one round trip for all partials
  __syncthreads();
  if (t >= 64) return;
  // the partials of workgroup i in lane i of the first wavefront; butterfly sums (a fixed order) instead of one thread walking
  // through them (a chain of dependent LDS reads: it was half of this kernel's duration)
  const bool have = t < (int)gridDim.x;
  double tot[3], dt = have ? spart[t * 4 + 3] : 1.0e20;
  for (int st = 0; st < 3; ++st) tot[st] = wave_sum(have ? spart[t * 4 + st] : 0.0);
  dt = wave_min(dt);
  if (t != 0) return;
  finalize_publish(a, tot, dt);
}

}  // namespace dflo
