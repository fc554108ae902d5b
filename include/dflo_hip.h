/*
 * dflo_hip.h -- C ABI of the MI355X-native explicit DG residual + SSP-RK engine
 * that replaces dflo's assemble_system / solve / iterate_explicit seam.
 *
 * The reference (cpraveen/dflo) has no plugin or FFI interface for this path:
 * the path is a set of private member functions of ConservationLaw<2> working
 * on member dealii::Vector<double>s.  Every entry point below names the
 * reference function(s) it replaces (paths relative to the reference root).
 * INTEGRATION.md shows the patch a dflo maintainer would apply to call them.
 *
 * This header is the CONTRACT: what a dflo build calls (INTEGRATION.md sections 1-3) and nothing else.  Beside it:
 *   dflo_mesh.h           host-side construction / partition of the flat mesh (the stand-alone driver and the tests; dflo
 *                         itself fills dflo_mesh_t from its Triangulation, include/dflo_hip_dealii.hpp)
 *   dflo_hip_transport.h  the halo / delivery / sequence-word seams the multi-device driver (dflo_amd/csrc/multi.hip) is written
 *                         against, and inspection of a multi-device handle -- for a transport of the host program's own
 *   dflo_hip_diag.h       diagnostics and test hooks (timing, counters, debug math): not part of the contract, not installed
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every function returns
 *     DFLO_OK (0) or a negative dflo_status; dflo_hip_last_error() gives text.
 *     (reference convention replaced: AssertThrow -> catch in main ->
 *     exit code 1, src/main.cc:56-78)
 *   - host arrays belong to the caller, the engine copies what it needs;
 *     device buffers belong to the handle.
 *   - a handle is driven by one host thread (as dflo's time loop is,
 *     src/claw.cc:1026-1129).
 *   - state vectors use dflo's layout: component order [x-mom, y-mom, density,
 *     energy] (src/equation.h:25-28); DoF index = cell*ndof + comp*n_s + node
 *     with node = a + N*b (x fastest) for Qk at Gauss points
 *     (FE_DGQArbitraryNodes(QGauss<1>(k+1)), src/main.cc:40) and
 *     node = modal index for Pk (FE_DGP, src/claw.cc:107-113).
 */
#ifndef DFLO_HIP_H
#define DFLO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFLO_N_COMP 4
#define DFLO_MAX_BOUNDARIES 10 /* Parameters::AllParameters::max_n_boundaries, src/parameters.h:370 */
#define DFLO_MAX_DEGREE 5

typedef enum {
  DFLO_OK = 0,
  DFLO_ERR_BAD_PARAM = -1,           /* consistency checks, src/parameters.cc:536-550 */
  DFLO_ERR_NONSQUARE_CELL = -2,      /* "Cell is not square", src/claw.cc:219 */
  DFLO_ERR_NEGATIVE_MEAN_STATE = -3, /* "Fatal: Negative states", src/positivity.cc:28-38 */
  DFLO_ERR_POSITIVITY_NO_ROOT = -4,  /* exit(0) in src/positivity.cc:160-169 */
  DFLO_ERR_HIP = -5,                 /* HIP runtime failure / no device */
  DFLO_ERR_COMM = -6,                /* halo plumbing misuse */
  DFLO_ERR_UNSUPPORTED = -7,         /* feature outside the hot-path scope */
  DFLO_ERR_NOMEM = -8
} dflo_status;

/* Parameters::Flux::FluxType, src/parameters.h:229 */
typedef enum { DFLO_FLUX_LXF = 0, DFLO_FLUX_SW = 1, DFLO_FLUX_KFVS = 2, DFLO_FLUX_ROE = 3, DFLO_FLUX_HLLC = 4 } dflo_flux;
/* EulerEquations::BoundaryKind, src/equation.h:862-869 */
typedef enum { DFLO_BC_INFLOW = 0, DFLO_BC_OUTFLOW = 1, DFLO_BC_SLIP = 2, DFLO_BC_PRESSURE = 3, DFLO_BC_FARFIELD = 4 } dflo_bc_kind;
/* Parameters::Limiter::LimiterType, src/parameters.h:241 */
typedef enum { DFLO_LIMITER_NONE = 0, DFLO_LIMITER_TVB = 1 } dflo_limiter;
/* Parameters::Limiter::ShockIndType, src/parameters.h:244 ("limiter" marks every cell; u2 belongs to the MOOD
 * scheme and is not part of the explicit TVB path: DFLO_ERR_UNSUPPORTED) */
typedef enum { DFLO_IND_LIMITER = 0, DFLO_IND_DENSITY = 1, DFLO_IND_ENERGY = 2, DFLO_IND_U2 = 3 } dflo_shock_indicator;
/* AllParameters::BasisType / MappingType, src/parameters.h:384-387 */
typedef enum { DFLO_BASIS_QK = 0, DFLO_BASIS_PK = 1 } dflo_basis;
typedef enum { DFLO_MAP_Q1 = 0, DFLO_MAP_Q2 = 1, DFLO_MAP_CARTESIAN = 2 } dflo_mapping;
/* how dflo_hip_multi_create* (and dflo_mesh_partition_ex, dflo_mesh.h) cut a mesh: x-slabs of the cell order after a coordinate sort
 * (C4: the 4001 x 1000 lattice) or recursive coordinate bisection of the cell centres (compact blocks on unstructured meshes, C5;
 * the MPI variant gets Morton-order blocks from p4est, src_mpi/claw.h:220) */
typedef enum { DFLO_PART_SLAB = 0, DFLO_PART_RCB = 1 } dflo_partitioner;

/* face neighbour encoding: >=0 neighbour cell, DFLO_NBR_BOUNDARY(id) on a
 * physical boundary, DFLO_NBR_NONE for a face of a ghost cell that leads
 * outside the local halo (never integrated). */
#define DFLO_NBR_BOUNDARY(id) (-1 - (int32_t)(id))
#define DFLO_NBR_BOUNDARY_ID(n) (-1 - (int32_t)(n))
#define DFLO_NBR_NONE (-1000000)

/*
 * Flat description of what dflo's Triangulation + DoFHandler hold
 * (src/claw.h:178-185).  deal.II reference-cell conventions: vertices
 * lexicographic v0=(0,0) v1=(1,0) v2=(0,1) v3=(1,1); faces 0:x=0 1:x=1 2:y=0
 * 3:y=1; face quadrature points ordered by the increasing free coordinate.
 */
typedef struct dflo_mesh {
  int32_t n_cells;        /* owned + ghost cells; owned cells come first */
  int32_t n_owned_cells;  /* == n_cells on a single device */
  int32_t degree;         /* k, 0..DFLO_MAX_DEGREE (0: piecewise constants, one RK stage, the limiters return at once,
                             src/claw.cc:141-145, src/positivity.cc:19, src/limiter.cc:379) */
  int32_t basis;          /* dflo_basis */
  int32_t mapping;        /* dflo_mapping; q2 is taken as q1: on straight-edged cells -- all this structure can describe, and all
                             the reference has, src/claw.cc:976-979 -- MappingQ(2) is the bilinear map */
  const double *cell_vertices;            /* [n_cells][4][2] */
  const int32_t *cell_face_neighbor;      /* [n_cells][4] */
  const int32_t *cell_face_neighbor_face; /* [n_cells][4]: face number seen from the neighbour, +4 if the
                                             face points run in the opposite direction on the two sides,
                                             +8 if the link is a periodic identification (the MPI variant
                                             integrates those from both sides, src_mpi/assemble_explicit.cc:186-260;
                                             the engine treats them as interior faces) */
  const int64_t *cell_global_id;          /* [n_cells] or NULL: id in the undecomposed mesh; the face
                                             integrating side is the smaller id (MeshWorker visits an
                                             interior face once, from the earlier cell) */
} dflo_mesh_t;

/* Scalars of Parameters::AllParameters the path reads (src/parameters.h:363-411). */
typedef struct dflo_params {
  int32_t flux_type;      /* dflo_flux */
  int32_t limiter_type;   /* dflo_limiter */
  int32_t char_lim;       /* "characteristic limiter" */
  int32_t pos_lim;        /* "positivity limiter" */
  int32_t global_time_step; /* 1: "time step type = global", 0: local */
  int32_t n_rk;           /* 0: by degree as src/claw.cc:141-159; else override */
  double gravity;         /* src/parameters.cc:346-348 */
  double cfl;
  double time_step;
  double final_time;
  double M;               /* TVB constant */
  double beta;
  int32_t bc_kind[DFLO_MAX_BOUNDARIES]; /* dflo_bc_kind per boundary id */
  int32_t shock_indicator; /* dflo_shock_indicator: "shock indicator" of subsection limiter, src/parameters.cc:228-239 */
  int32_t conserve_angular_momentum; /* "conserve angular momentum" of subsection limiter: the correction of the limited
                                        slopes in apply_limiter_TVB_Pk, src/limiter.cc:453,496-500 (Pk basis only) */
} dflo_params_t;

/* Opcodes of the postfix programs of dflo_hip_set_boundary_program: what deal.II's FunctionParser evaluates for the
 * "w_i value" entries of a boundary subsection, in the variables x, y, t (src/parameters.cc:441-477). Comparisons and
 * logical operators yield 1.0 / 0.0; SEL pops (condition, a, b) and keeps a if condition != 0, else b. */
typedef enum {
  DFLO_OP_CONST = 0, DFLO_OP_X = 1, DFLO_OP_Y = 2, DFLO_OP_T = 3, DFLO_OP_NEG = 4, DFLO_OP_ADD = 5, DFLO_OP_SUB = 6,
  DFLO_OP_MUL = 7, DFLO_OP_DIV = 8, DFLO_OP_POW = 9, DFLO_OP_LT = 10, DFLO_OP_LE = 11, DFLO_OP_GT = 12, DFLO_OP_GE = 13,
  DFLO_OP_EQ = 14, DFLO_OP_NE = 15, DFLO_OP_AND = 16, DFLO_OP_OR = 17, DFLO_OP_SEL = 18, DFLO_OP_SIN = 19, DFLO_OP_COS = 20,
  DFLO_OP_TAN = 21, DFLO_OP_EXP = 22, DFLO_OP_LOG = 23, DFLO_OP_SQRT = 24, DFLO_OP_ABS = 25, DFLO_OP_MIN = 26,
  DFLO_OP_MAX = 27, DFLO_OP_ATAN2 = 28, DFLO_OP_TANH = 29, DFLO_OP_SINH = 30, DFLO_OP_COSH = 31, DFLO_OP_ASIN = 32,
  DFLO_OP_ACOS = 33, DFLO_OP_ATAN = 34, DFLO_OP_FLOOR = 35, DFLO_OP_CEIL = 36, DFLO_OP_SIGN = 37, DFLO_OP_LOG10 = 38,
  DFLO_OP_ERF = 39, DFLO_OP_ERFC = 40, DFLO_OP_COUNT = 41
} dflo_expr_op;

typedef struct dflo_hip_engine *dflo_hip_handle;

/* ---------------------------------------------------------------- lifetime */

/* Replaces setup_system() + compute_inv_mass_matrix() + neighbour maps
 * (src/claw.cc:229-258, 271-386) and setup_mesh_worker (src/claw.cc:417-437).
 * device_id: HIP device ordinal. */
int dflo_hip_create(const dflo_mesh_t *mesh, const dflo_params_t *params, int device_id, dflo_hip_handle *out);
int dflo_hip_destroy(dflo_hip_handle h);
const char *dflo_hip_last_error(dflo_hip_handle h); /* h may be NULL: error of the last failed create */

/* -------------------------------------------------------------- state I/O */

int64_t dflo_hip_n_dofs(dflo_hip_handle h);     /* n_cells * ndof */
int32_t dflo_hip_n_rk(dflo_hip_handle h);       /* stages per step, src/claw.cc:141-159 */

/* current_solution = old_solution = u  (src/ic.cc:118-120); also recomputes
 * cell averages as run() does after the IC (src/claw.cc:997). */
int dflo_hip_set_solution(dflo_hip_handle h, const double *u);
int dflo_hip_get_solution(dflo_hip_handle h, double *u);
/* cell_average, [n_cells][4] (src/claw.cc:562-597).  After an intermediate stage of a run in which nothing on the device reads
 * the averages (no LxF flux, limiter, indicator or local time step) they are formed by this call, not by the stage. */
int dflo_hip_get_cell_average(dflo_hip_handle h, double *avg);

/* Boundary faces in the order MeshWorker meets them (cell ascending, face
 * ascending).  xy: [n_bfaces][N][2] quadrature points = what
 * fe_v.get_quadrature_points() hands to FunctionParser::vector_value_list
 * (src/assemble_explicit.cc:163-165). Any pointer may be NULL. */
int32_t dflo_hip_n_boundary_faces(dflo_hip_handle h);
int dflo_hip_boundary_faces(dflo_hip_handle h, int32_t *cell, int32_t *face, int32_t *boundary_id, double *xy);
/* values: [n_bfaces][N][4] boundary function values. which=0: used by RK stage 0
 * (bc_time = t), which=1: used by later stages (bc_time = t+dt), src/claw.cc:736-745. */
int dflo_hip_set_boundary_values(dflo_hip_handle h, int which, const double *values);

/* ------------------------------------------------------------- hot path */

/* Boundary functions evaluated on the device instead of being uploaded: a postfix program per (boundary id, component)
 * over x, y, t (ops: [n_ops][2] = (dflo_expr_op, constant index), stack depth <= 16; n_ops = 0 removes the program).
 * At the start of every time step the engine then evaluates the programmed components at the face quadrature points
 * at t (table of RK stage 0) and t + dt (later stages) -- FunctionParser::set_time + vector_value_list of
 * src/claw.cc:736-745, src/assemble_explicit.cc:161-165 -- from its device-resident clock (set by dflo_hip_compute_dt,
 * advanced by every step); components without a program keep the uploaded values. */
int dflo_hip_set_boundary_program(dflo_hip_handle h, int32_t boundary_id, int32_t component, int32_t n_ops, const int32_t *ops,
                                  int32_t n_consts, const double *consts);

/* assemble_system(IntegratorExplicit&) (src/assemble_explicit.cc:433-452): rhs of the current
 * solution with the current cell averages and boundary set `which`; rhs_out [n_dofs], dflo layout. */
int dflo_hip_residual(dflo_hip_handle h, int which, double *rhs_out);

/* compute_time_step() (src/claw.cc:444-557): global dt from the current cell
 * averages (cartesian) or point values (q1), time_step cap and final_time clip. */
int dflo_hip_compute_dt(dflo_hip_handle h, double elapsed_time, double *dt);

/* iterate_explicit() (src/claw.cc:726-772) + old_solution = current_solution
 * (src/claw.cc:1110): all RK stages, each = residual, dt*M^-1, SSP combine,
 * cell average, TVB limiter, positivity limiter.  res_norm0/res_norm: l2 norm of
 * the rhs in the first/last stage (src/claw.cc:749-750); either may be NULL. */
int dflo_hip_step(dflo_hip_handle h, double dt, double *res_norm0, double *res_norm);

/* One RK stage `rk` of iterate_explicit (fine-grained seam for multi-device
 * drivers that exchange halos between stages). dt<0: use the device-resident dt. */
int dflo_hip_stage(dflo_hip_handle h, int rk, double dt);
/* old_solution = current_solution (src/claw.cc:1110).  MANDATORY after the last stage of every time step driven through
 * dflo_hip_stage / _stage_update / _stage_limit / the halo seam: it also advances the engine's step count, whose parity selects
 * the step-index slot of the failure flags and the row of the time-step table the next step's kernels read (a caller that
 * leaves it out keeps reading the row of the step before: failure_step stays 0 and a device-resident dt never updates). */
int dflo_hip_end_step(dflo_hip_handle h);

/* n_steps x { compute_time_step ; iterate_explicit ; elapsed_time += dt } with
 * dt and time resident on the device (no host round trip inside the loop);
 * the production loop of run() (src/claw.cc:1026-1110) without output. */
int dflo_hip_advance(dflo_hip_handle h, int n_steps, double *elapsed_time_inout);

/* compute_cell_average / apply_limiter / apply_positivity_limiter as separate
 * calls (src/claw.cc:562, src/limiter.cc:36, src/positivity.cc:17) -- what run()
 * does once after the initial condition (src/claw.cc:997-1001). */
int dflo_hip_compute_cell_average(dflo_hip_handle h);
int dflo_hip_apply_limiter(dflo_hip_handle h);   /* compute_shock_indicator(); apply_limiter(); */
int dflo_hip_apply_positivity_limiter(dflo_hip_handle h);

/* compute_shock_indicator (src/indicator.cc:17-198) from the current solution and cell averages, and the
 * shock_indicator vector (n_cells; 1e20 everywhere for "shock indicator = limiter"). Inside a stage the
 * engine runs it between compute_cell_average and apply_limiter as src/claw.cc:762-764 does. */
int dflo_hip_compute_shock_indicator(dflo_hip_handle h);
int dflo_hip_get_shock_indicator(dflo_hip_handle h, double *shock_indicator);

/* Device-side failure flags raised by kernels, checked here (no mid-kernel abort):
 * returns DFLO_OK, DFLO_ERR_NEGATIVE_MEAN_STATE or DFLO_ERR_POSITIVITY_NO_ROOT. */
int dflo_hip_check(dflo_hip_handle h);
int dflo_hip_synchronize(dflo_hip_handle h);

/* ------------------------------------------------ several devices behind one handle */
/* The native multi-device driver (dflo_amd/csrc/multi.hip): partitions the undivided mesh, owns one engine per part and
 * runs the stage schedule of the MPI variant -- update_ghost_values() after the update and after the limiter
 * (src_mpi/claw.cc:793, src_mpi/limiter.cc:232), Utilities::MPI::min(global_dt) (src_mpi/claw.cc:579), the summed
 * ||rhs|| (src_mpi/claw.cc:777) -- with the rim shards of every part advanced first and their cells in flight while the
 * interior shards are computed.  State, boundary data and results cross this boundary in the numbering of the
 * UNDIVIDED mesh (cells, DoFs, boundary faces in MeshWorker order), exactly as for a single engine.
 *
 *   dflo_hip_multi_create       one process (dflo's serial tree, src/): n_devices engines; the driver keeps one host thread
 *                               and one stream pair per device (per pair of parts where device_ids repeat) to issue the
 *                               launches, so the caller still calls from ONE thread -- the handle is not thread-safe;
 *                               halos are written into the peers' receive areas by the pack kernels over xGMI peer access,
 *                               the time-step minimum comes from peer reads of device slots.  device_ids may repeat
 *                               (several parts on one device).
 *   dflo_hip_multi_create_rank  one process per GPU (dflo's MPI tree, src_mpi/): this process is part `rank` of n_ranks;
 *                               halos by grouped ncclSend/ncclRecv, the time step by an 8-byte ncclAllReduce(min) on the
 *                               device.  unique_id: DFLO_COMM_ID_BYTES bytes obtained on one rank with
 *                               dflo_hip_comm_unique_id and handed to the others by the host program's own means
 *                               (MPI_Bcast where src_mpi/main.cc has MPI; a torch.distributed broadcast in bench.py).
 *                               set_solution / boundary values take global arrays and use this rank's cells; the get_*
 *                               calls fill this rank's owned cells and leave the rest of the array alone.
 * partitioner: dflo_partitioner (above: 0 = slabs, 1 = RCB).
 * One process per GPU: create_rank*, set_solution, advance / step / compute_dt, residual, the limiter calls and destroy are
 * COLLECTIVE over the ranks (every rank makes the same calls in the same order, as the MPI variant's do); with
 * DFLO_RANK_TRANSPORT=ipc the ranks also meet inside set_solution and destroy (nobody frees or rewrites a window a neighbour's
 * kernel may still store into), so a rank that leaves early must still call destroy. */
#define DFLO_COMM_ID_BYTES 128
typedef struct dflo_hip_multi *dflo_hip_multi_handle;
int dflo_hip_multi_create(const dflo_mesh_t *mesh, const dflo_params_t *params, int n_devices, const int *device_ids,
                          int partitioner, dflo_hip_multi_handle *out);
int dflo_hip_comm_unique_id(void *id_bytes);
int dflo_hip_multi_create_rank(const dflo_mesh_t *mesh, const dflo_params_t *params, int device_id, int rank, int n_ranks,
                               const void *unique_id, int partitioner, dflo_hip_multi_handle *out);
/* One process per GPU with the host program's own transport instead of RCCL (a cluster whose ranks talk MPI; the
 * repository's two-process tests on one GPU, which RCCL refuses).  Both callbacks get DEVICE pointers and the stream of the
 * driver the call belongs on (hipStream_t as void*: the comm stream for an exchange, the compute stream for the time step's
 * all-reduce); when they return, the transfer must be complete or enqueued on that stream in order.  With a TVB limiter the
 * driver calls `exchange` ONCE per stage (the reference's two, src_mpi/limiter.cc:232 and src_mpi/claw.cc:793, merged: the cut
 * cells travel unlimited with their neighbours' averages and the receiver limits its ghost cells itself).
 *   exchange:  for each of n_peers ranks, send_bytes[i] bytes at send_ptr[i] go to rank peer[i], recv_bytes[i] bytes from
 *              it arrive at recv_ptr[i] -- update_ghost_values(), src_mpi/claw.cc:793
 *   allreduce: n doubles at `values`, in place, op = dflo_reduce_op -- Utilities::MPI::min / sum, src_mpi/claw.cc:579,777
 * Return 0 on success. */
typedef enum { DFLO_REDUCE_MIN = 0, DFLO_REDUCE_SUM = 1, DFLO_REDUCE_MAX = 2 } dflo_reduce_op;
typedef int (*dflo_exchange_fn)(void *user, int n_peers, const int *peer, const void *const *send_ptr, const size_t *send_bytes,
                                void *const *recv_ptr, const size_t *recv_bytes, void *hip_stream);
typedef int (*dflo_allreduce_fn)(void *user, double *values, int n, int op, void *hip_stream);
int dflo_hip_multi_create_rank_custom(const dflo_mesh_t *mesh, const dflo_params_t *params, int device_id, int rank, int n_ranks,
                                      dflo_exchange_fn exchange, dflo_allreduce_fn allreduce, void *user, int partitioner,
                                      dflo_hip_multi_handle *out);
/* One process per GPU with DFLO_RANK_TRANSPORT=ipc: collective (the ranks meet before anybody frees a window its neighbours map). */
int dflo_hip_multi_destroy(dflo_hip_multi_handle m);
const char *dflo_hip_multi_last_error(dflo_hip_multi_handle m); /* m may be NULL: error of the last failed create */
/* the calls of a single engine, on the undivided mesh */
int dflo_hip_multi_set_solution(dflo_hip_multi_handle m, const double *u);
int dflo_hip_multi_get_solution(dflo_hip_multi_handle m, double *u);
int dflo_hip_multi_get_cell_average(dflo_hip_multi_handle m, double *avg);
int32_t dflo_hip_multi_n_boundary_faces(dflo_hip_multi_handle m);
int dflo_hip_multi_boundary_faces(dflo_hip_multi_handle m, int32_t *cell, int32_t *face, int32_t *boundary_id, double *xy);
int dflo_hip_multi_set_boundary_values(dflo_hip_multi_handle m, int which, const double *values);
int dflo_hip_multi_set_boundary_program(dflo_hip_multi_handle m, int32_t boundary_id, int32_t component, int32_t n_ops,
                                        const int32_t *ops, int32_t n_consts, const double *consts);
int dflo_hip_multi_residual(dflo_hip_multi_handle m, int which, double *rhs_out);
int dflo_hip_multi_compute_dt(dflo_hip_multi_handle m, double elapsed_time, double *dt);
int dflo_hip_multi_step(dflo_hip_multi_handle m, double dt, double *res_norm0, double *res_norm);
int dflo_hip_multi_advance(dflo_hip_multi_handle m, int n_steps, double *elapsed_time_inout);
int dflo_hip_multi_apply_limiter(dflo_hip_multi_handle m);
int dflo_hip_multi_apply_positivity_limiter(dflo_hip_multi_handle m);
int dflo_hip_multi_check(dflo_hip_multi_handle m);

#ifdef __cplusplus
}
#endif
#endif /* DFLO_HIP_H */

