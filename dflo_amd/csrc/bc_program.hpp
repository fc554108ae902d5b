// bc_program.hpp -- boundary functions on the device: postfix programs over (x, y, t), their interpreter, the kernel that fills the
// boundary-value tables and the one-wavefront form other kernels take along.
// Part of the device side of engine.hip (see there for the layout of the data and of a stage).
#pragma once
#include "kernels_common.hpp"

namespace dflo {

// ------------------------------------------------------------------ boundary functions on the device
// The boundary values of integrate_boundary_term_explicit (FunctionParser::vector_value_list at the face
// quadrature points with set_time(bc_time), src/assemble_explicit.cc:161-165, src/claw.cc:736-745) evaluated by
// the device from postfix programs (dflo_hip_set_boundary_program): no host round trip per step for
// time-dependent boundary data (C4's moving shock on the top wall).
struct BcArgs {
  const int32_t *ops;       // [n][2] (dflo_expr_op, constant index)
  const double *consts;
  const int32_t *prog;      // [DFLO_MAX_BOUNDARIES][4][2] (first op, number of ops); 0 ops = values stay as uploaded
  const int32_t *bface_id;  // [n_bfaces]
  const double *bxy;        // [n_bfaces][N][2]
  double *bval0, *bval1;    // [n_bfaces][N][4] tables of RK stage 0 (time t) and of the later stages (t + dt)
  const double *dt_dev;     // [0] dt, [1] elapsed time
  DtSrc dts;                // several engines: see step_dt
  double dt_host;
  const int32_t *faces;     // [n_faces] the boundary faces whose id has a program
  const double *pts;        // [n_faces * N][4] per listed point: x, y, its index in the value tables, its boundary id (one coalesced
                            // load instead of face -> (id, coordinates): two trips to memory less on a kernel that is all latency)
  int n_faces, N, n_ops, n_consts;
};
constexpr int kBcLdsOps = 1024, kBcLdsConsts = 256;  // programs up to this size are interpreted out of LDS
constexpr int kExprStack = 16;
constexpr int kBcThreads = 512;   // 8 wavefronts: (component, table) pairs over 64 face points
// Postfix interpreter.  Every lane of the wave runs the SAME program (the caller loops over the programs and masks
// the store), so the opcode is wave-uniform: it is moved to a scalar register and the switch becomes one scalar
// jump -- with a per-lane opcode the compiler walks through all forty cases under an execution mask.  The top of
// the stack lives in a register, the rest in LDS (st[depth][thread]; a private array indexed by the stack pointer
// would be placed in scratch memory).
// RICH: with the transcendental functions; a set of programs that uses none of them runs the lean instantiation (a fifth of
// the code, 97 registers instead of 256, no scratch memory)
template <bool RICH, int STRIDE = kBcThreads>
__device__ __forceinline__ double run_program(const int32_t *ops, int n, const double *consts, double x, double y, double t, double *st) {
  double tos = 0.0;
  int sp = 0;   // entries below the top, kept in st[0 .. sp)
  for (int i = 0; i < n; ++i) {
    const int op = __builtin_amdgcn_readfirstlane(ops[2 * i]);
    if (op <= DFLO_OP_T) {  // pushes
      st[(sp++) * STRIDE] = tos;
      tos = op == DFLO_OP_CONST ? consts[__builtin_amdgcn_readfirstlane(ops[2 * i + 1])] : (op == DFLO_OP_X ? x : (op == DFLO_OP_Y ? y : t));
      continue;
    }
    if (op == DFLO_OP_SEL) {
      const double b = tos, a = st[(--sp) * STRIDE], c = st[(--sp) * STRIDE];
      tos = c != 0.0 ? a : b;
      continue;
    }
    const bool binary = (op >= DFLO_OP_ADD && op <= DFLO_OP_OR) || op == DFLO_OP_MIN || op == DFLO_OP_MAX || op == DFLO_OP_ATAN2;
    double a = tos, b = 0.0;
    if (binary) {
      b = tos;
      a = st[(--sp) * STRIDE];
    }
    double r;
    switch (op) {
      case DFLO_OP_NEG: r = -a; break;
      case DFLO_OP_ADD: r = a + b; break;
      case DFLO_OP_SUB: r = a - b; break;
      case DFLO_OP_MUL: r = a * b; break;
      case DFLO_OP_DIV: r = a / b; break;
      case DFLO_OP_POW: if constexpr (RICH) r = pow(a, b); else r = __builtin_nan(""); break;
      case DFLO_OP_LT: r = a < b ? 1.0 : 0.0; break;
      case DFLO_OP_LE: r = a <= b ? 1.0 : 0.0; break;
      case DFLO_OP_GT: r = a > b ? 1.0 : 0.0; break;
      case DFLO_OP_GE: r = a >= b ? 1.0 : 0.0; break;
      case DFLO_OP_EQ: r = a == b ? 1.0 : 0.0; break;
      case DFLO_OP_NE: r = a != b ? 1.0 : 0.0; break;
      case DFLO_OP_AND: r = (a != 0.0 && b != 0.0) ? 1.0 : 0.0; break;
      case DFLO_OP_OR: r = (a != 0.0 || b != 0.0) ? 1.0 : 0.0; break;
      case DFLO_OP_SIN: if constexpr (RICH) r = sin(a); else r = __builtin_nan(""); break;
      case DFLO_OP_COS: if constexpr (RICH) r = cos(a); else r = __builtin_nan(""); break;
      case DFLO_OP_TAN: if constexpr (RICH) r = tan(a); else r = __builtin_nan(""); break;
      case DFLO_OP_EXP: if constexpr (RICH) r = exp(a); else r = __builtin_nan(""); break;
      case DFLO_OP_LOG: if constexpr (RICH) r = log(a); else r = __builtin_nan(""); break;
      case DFLO_OP_SQRT: r = sqrt(a); break;
      case DFLO_OP_ABS: r = fabs(a); break;
      case DFLO_OP_MIN: r = fmin(a, b); break;
      case DFLO_OP_MAX: r = fmax(a, b); break;
      case DFLO_OP_ATAN2: if constexpr (RICH) r = atan2(a, b); else r = __builtin_nan(""); break;
      case DFLO_OP_TANH: if constexpr (RICH) r = tanh(a); else r = __builtin_nan(""); break;
      case DFLO_OP_SINH: if constexpr (RICH) r = sinh(a); else r = __builtin_nan(""); break;
      case DFLO_OP_COSH: if constexpr (RICH) r = cosh(a); else r = __builtin_nan(""); break;
      case DFLO_OP_ASIN: if constexpr (RICH) r = asin(a); else r = __builtin_nan(""); break;
      case DFLO_OP_ACOS: if constexpr (RICH) r = acos(a); else r = __builtin_nan(""); break;
      case DFLO_OP_ATAN: if constexpr (RICH) r = atan(a); else r = __builtin_nan(""); break;
      case DFLO_OP_FLOOR: r = floor(a); break;
      case DFLO_OP_CEIL: r = ceil(a); break;
      case DFLO_OP_SIGN: r = a > 0.0 ? 1.0 : (a < 0.0 ? -1.0 : 0.0); break;
      case DFLO_OP_LOG10: if constexpr (RICH) r = log10(a); else r = __builtin_nan(""); break;
      case DFLO_OP_ERF: if constexpr (RICH) r = erf(a); else r = __builtin_nan(""); break;
      case DFLO_OP_ERFC: if constexpr (RICH) r = erfc(a); else r = __builtin_nan(""); break;
      default: r = __builtin_nan(""); break;
    }
    tos = r;
  }
  return tos;
}
template <bool RICH>
__global__ __launch_bounds__(kBcThreads) __attribute__((flatten)) void bc_eval_kernel(const BcArgs a) {
  // the programs are a few hundred words: interpret them out of LDS, not with a dependent global load per opcode
  __shared__ int32_t s_ops[2 * kBcLdsOps];
  __shared__ double s_consts[kBcLdsConsts];
  __shared__ int32_t s_prog[DFLO_MAX_BOUNDARIES * 4 * 2];
  __shared__ double s_stack[(kExprStack + 1) * kBcThreads];
  const bool in_lds = a.n_ops <= kBcLdsOps && a.n_consts <= kBcLdsConsts;
  // a block takes 64 (listed face, point) pairs; wavefront w evaluates component w & 3 for the table w >> 2, so the
  // eight short programs of a point run side by side instead of one after the other in a single thread.
  // (The point's data are requested first: their two dependent trips to memory overlap with the copy of the programs.)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = wave & 3, which = wave >> 2;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const bool valid = j < a.n_faces * a.N;
  const double *pt = a.pts + 4 * (size_t)(valid ? j : 0);
  const double x = pt[0], y = pt[1];
  const int i = (int)pt[2], id = (int)pt[3];
  const double t = a.dt_dev[1] + (which ? (a.dt_host >= 0.0 ? a.dt_host : step_dt(a.dts, a.dt_dev)) : 0.0);
  double *bval = which ? a.bval1 : a.bval0;
  if (in_lds) {
    for (int k = threadIdx.x; k < 2 * a.n_ops; k += blockDim.x) s_ops[k] = a.ops[k];
    for (int k = threadIdx.x; k < a.n_consts; k += blockDim.x) s_consts[k] = a.consts[k];
  }
  for (int k = threadIdx.x; k < DFLO_MAX_BOUNDARIES * 4 * 2; k += blockDim.x) s_prog[k] = a.prog[k];
  __syncthreads();
  for (int b = 0; b < DFLO_MAX_BOUNDARIES; ++b) {
    const int first = s_prog[(b * 4 + c) * 2], n = s_prog[(b * 4 + c) * 2 + 1];   // wave-uniform
    if (n == 0) continue;
    const double v = in_lds ? run_program<RICH>(s_ops + 2 * first, n, s_consts, x, y, t, s_stack + threadIdx.x)
                            : run_program<RICH>(a.ops + 2 * first, n, a.consts, x, y, t, s_stack + threadIdx.x);
    if (valid && id == b) bval[(size_t)i * 4 + c] = v;
  }
}

// The same evaluation as ONE wavefront's share, for kernels of 64 threads that take the boundary programs along (the limiter
// pass behind stage 0, limiter_kernel): unit u = (block of 64 listed points, component), table `which` only.  lds: the caller's
// (kBcWaveLds bytes).  Programs beyond the LDS budget of that form are not fused (the engine then launches bc_eval_kernel).
constexpr int kBcWaveOps = 256, kBcWaveConsts = 64;
constexpr int kBcWaveLds = 2 * kBcWaveOps * 4 + kBcWaveConsts * 8 + DFLO_MAX_BOUNDARIES * 4 * 2 * 4 + (kExprStack + 1) * 64 * 8;
__device__ __forceinline__ void bc_eval_wave(const BcArgs &a, int unit, int which, unsigned char *lds) {
  int32_t *s_ops = (int32_t *)lds;
  double *s_consts = (double *)(lds + 2 * kBcWaveOps * 4);
  int32_t *s_prog = (int32_t *)(lds + 2 * kBcWaveOps * 4 + kBcWaveConsts * 8);
  double *s_stack = (double *)(lds + 2 * kBcWaveOps * 4 + kBcWaveConsts * 8 + DFLO_MAX_BOUNDARIES * 4 * 2 * 4);
  const int c = unit & 3, j = (unit >> 2) * 64 + (int)threadIdx.x;
  const bool valid = j < a.n_faces * a.N;
  const double *pt = a.pts + 4 * (size_t)(valid ? j : 0);
  const double x = pt[0], y = pt[1];
  const int i = (int)pt[2], id = (int)pt[3];
  const double t = a.dt_dev[1] + (which ? (a.dt_host >= 0.0 ? a.dt_host : step_dt(a.dts, a.dt_dev)) : 0.0);
  double *bval = which ? a.bval1 : a.bval0;
  for (int k = threadIdx.x; k < 2 * a.n_ops; k += 64) s_ops[k] = a.ops[k];
  for (int k = threadIdx.x; k < a.n_consts; k += 64) s_consts[k] = a.consts[k];
  for (int k = threadIdx.x; k < DFLO_MAX_BOUNDARIES * 4 * 2; k += 64) s_prog[k] = a.prog[k];
  __syncthreads();
  for (int b = 0; b < DFLO_MAX_BOUNDARIES; ++b) {
    const int first = s_prog[(b * 4 + c) * 2], n = s_prog[(b * 4 + c) * 2 + 1];   // wave-uniform
    if (n == 0) continue;
    const double v = run_program<false, 64>(s_ops + 2 * first, n, s_consts, x, y, t, s_stack + threadIdx.x);
    if (valid && id == b) bval[(size_t)i * 4 + c] = v;
  }
}

}  // namespace dflo
