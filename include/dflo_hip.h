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

/* Launch all work of this handle on an existing HIP stream (hipStream_t passed
 * as void*); NULL = the handle's own stream. */
int dflo_hip_set_stream(dflo_hip_handle h, void *hip_stream);

/* -------------------------------------------------------------- state I/O */

int64_t dflo_hip_n_dofs(dflo_hip_handle h);     /* n_cells * ndof */
int32_t dflo_hip_dofs_per_cell(dflo_hip_handle h);
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
/* The boundary-value table in use (which = 0: stage 0, 1: later stages), [n_boundary_faces][k+1][4]. */
int dflo_hip_get_boundary_values(dflo_hip_handle h, int which, double *values);

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
/* Index (0-based, counted from the last dflo_hip_set_solution) of the time step in which the first failure flag went
 * up, -1 if none: where the reference would have stopped (src/positivity.cc:28-38 throws, :160-169 exits, inside the
 * stage).  dflo_hip_advance looks at the flags every 32 steps and returns early with the error. */
int dflo_hip_failure_step(dflo_hip_handle h, int64_t *step);
/* Positivity limiter applied inside the stage kernel (pos_lim without TVB on Qk): counts[0] = cell-stages that failed the
 * cheap nodal-box bound and went through the limiter proper (src/positivity.cc:43-205), counts[1] = cell-stages it
 * changed (theta1 < 1 or theta2 < 1), summed since the last reset.  A diagnostic for bench.py's config.check. */
int dflo_hip_positivity_stats(dflo_hip_handle h, int64_t *counts, int reset);
int dflo_hip_synchronize(dflo_hip_handle h);

/* Average duration (ms) of the stage kernel launches since the last reset, measured with HIP events on the
 * engine's stream. enable = 1: every fifth stage is timed (each stage of a 2- or 3-stage step equally often, and the event
 * records stay out of the way of the others); enable = k > 1: every k-th (choose k coprime to 2 and 3; a timed launch costs
 * its stream a few microseconds of bubbles, so a long run samples sparsely); n receives the number of timed stages. */
int dflo_hip_stage_timing(dflo_hip_handle h, int enable, double *avg_ms, int64_t *n);
/* 1 if this engine's stage kernel forms its dense per-element basis contractions with matrix instructions (degree 3 with
 * DFLO_MFMA=1: the eta-derivative of the Qk kernel -- the dense ndof x n_q loops of src/assemble_explicit.cc:85-115 after sum
 * factorisation -- as v_mfma_f64_4x4x4_4b, the modal <-> nodal tables of FE_DGP, src/main.cc:44-48, as v_mfma_f64_16x16x4),
 * 0 if the vector units do (the default: measured faster).  A diagnostic for bench.py's roofline.mfma. */
int dflo_hip_uses_mfma(dflo_hip_handle h);

/* ------------------------------------------------ multi-device halo seam */
/* Replaces LA::distributed::Vector::update_ghost_values() of the MPI variant
 * (src_mpi/claw.cc:793, src_mpi/limiter.cc:232).  The engine owns cells
 * [0,n_owned) and reads ghost cells [n_owned,n_cells).  pack gathers the DoFs
 * ([n][ndof]) or the cell averages ([n][4]) of the listed owned cells into a
 * contiguous device buffer; unpack scatters a received buffer ([n_ghost][ndof] or
 * [n_ghost][4], ghost order) into the ghost cells.  The transport between the
 * two (RCCL send/recv through torch.distributed) is the caller's.
 * With a TVB limiter the ghost AVERAGES must be refreshed between the update and
 * the limiter of a stage (the MPI variant computes cell averages on owned+ghost
 * cells after the first update_ghost_values, src_mpi/claw.cc:793,653-669):
 *   dflo_hip_stage_update -> exchange averages -> dflo_hip_stage_limit -> exchange DoFs.
 * dflo_hip_stage == dflo_hip_stage_update + dflo_hip_stage_limit. */
int dflo_hip_stage_update(dflo_hip_handle h, int rk, double dt);
int dflo_hip_stage_limit(dflo_hip_handle h);
/* The same stage split by shard set, for overlapping the exchange with compute (what dflo_amd/csrc/multi.hip is written
 * against): part 1 = rim shards (those that read ghost cells), part 2 = interior shards, part 0 = all; for the update of a
 * stage that a TVB limiter follows also part 3 = rim shards + the ring of shards next to them (the limiter of a rim cell
 * reads the new averages of its neighbours there) and part 4 = the others.  dflo_hip_set_stream chooses the stream of
 * the following launches; launches of different parts of one stage may run side by side on different streams (they read
 * the previous stage and write disjoint shards); ordering between the streams is the caller's (events).
 *   open -> update_part(1) -> [limit_part(1) -> pack -> exchange -> unpack on a second stream]
 *        -> update_part(2) -> limit_part(2) -> finish (reductions, CFL minimum)                    */
int dflo_hip_stage_open(dflo_hip_handle h, int rk, double dt);
int dflo_hip_stage_update_part(dflo_hip_handle h, int part);
int dflo_hip_stage_limit_part(dflo_hip_handle h, int part);
int dflo_hip_stage_finish(dflo_hip_handle h);
int dflo_hip_n_rim_shards(dflo_hip_handle h);
int dflo_hip_n_ghost_cells(dflo_hip_handle h);
int dflo_hip_set_send_cells(dflo_hip_handle h, int32_t n, const int32_t *cells);
int dflo_hip_pack_send(dflo_hip_handle h, void *device_buffer);
int dflo_hip_pack_send_avg(dflo_hip_handle h, void *device_buffer);
int dflo_hip_unpack_ghost(dflo_hip_handle h, const void *device_buffer);     /* also recomputes ghost averages */
int dflo_hip_unpack_ghost_avg(dflo_hip_handle h, const void *device_buffer);
/* Instead of unpack_ghost_avg: the limiter passes that follow (Qk) read the ghost cells' averages straight from the received
 * buffer ([n_ghost][4], ghost order) -- one small kernel less between the arrival of the averages and the limiter of the rim
 * cells, the stretch of a TVB stage that every neighbour waits for.  NULL, or the next dflo_hip_unpack_ghost_avg /
 * dflo_hip_unpack_ghost_cells / dflo_hip_set_solution, returns to the averages held by the engine.  Not for runs whose stage
 * kernels read ghost averages too (LxF flux). */
int dflo_hip_ghost_avg_source(dflo_hip_handle h, const void *device_buffer);
/* DoFs and cell average of every listed cell in one record, [n][ndof + 4]: the ghost copy then holds the bits of its owner
 * (an average formed again from the DoFs differs from the stage kernel's in the last place; the LxF flux and the TVB
 * differences read it).  What the native multi-device driver ships. */
int dflo_hip_pack_send_cells(dflo_hip_handle h, void *device_buffer);
int dflo_hip_unpack_ghost_cells(dflo_hip_handle h, const void *device_buffer);
/* Face-trace records: when nothing needs more of a ghost cell than its trace on the cut faces and its average
 * -- Qk without the KXRCF indicator: dflo_hip_halo_traces() = 1 -- the stage kernels read the ghost cells from a table of
 * traces, [n_ghost_traces][4][k+1] doubles ordered by (ghost cell, face), and the halo message of a cut face shrinks from
 * the cell's (k+1)^2 * 4 doubles to (k+1) * 4 (Q2: 36 -> 12; the 4-double average travels with pack_send_avg).  The
 * sender lists its (owned cell, face) pairs in the receiver's order (set_send_faces) and packs their traces; the receiver
 * lets the transport write straight into one of the engine's two trace tables (ghost_trace_buffer) and switches the
 * stage kernels to it before the next stage (use_ghost_traces) -- no unpack kernel.  dflo_hip_set_solution fills both
 * tables from the ghost cells' DoFs.  DFLO_HALO_CELLS=1 keeps whole-cell records. */
int dflo_hip_halo_traces(dflo_hip_handle h);
int dflo_hip_n_ghost_traces(dflo_hip_handle h);
int dflo_hip_set_send_faces(dflo_hip_handle h, int32_t n, const int32_t *cells, const int32_t *faces);
int dflo_hip_pack_send_traces(dflo_hip_handle h, void *device_buffer);
int dflo_hip_ghost_trace_buffer(dflo_hip_handle h, int which, void **device_ptr);
int dflo_hip_use_ghost_traces(dflo_hip_handle h, int which);
/* The engine's two ghost-trace tables ([n_ghost_traces][4][k+1] doubles each) and its table of the parts' time-step minima
 * ([2][16] doubles) in memory of the CALLER's -- a window it exports to other processes as one allocation (the runtime serves small
 * allocations as fragments of shared blocks, which cannot be exported reliably one by one).  The current contents move along;
 * the buffers stay the caller's and must outlive the engine. */
int dflo_hip_set_ghost_trace_buffers(dflo_hip_handle h, void *table0, void *table1);
int dflo_hip_set_dt_table_buffer(dflo_hip_handle h, void *table);
/* The next stage or limiter kernel this engine launches (stage_update_part / stage_limit_part) carries `event` (a hipEvent_t)
 * as its completion signal -- hipExtLaunchKernel's stopEvent -- instead of the caller recording the event behind it: one packet
 * less between two kernels of a stream (the multi-device schedule orders its two streams with one such event per phase).
 * If that launch turns out to be empty the event is recorded the plain way. */
int dflo_hip_attach_event(dflo_hip_handle h, void *event);
/* Delivery by the stage kernel itself (one process per GPU over mapped tables; Qk, ghost cells by their traces).  set_deliver,
 * once per receive area (0 | 1): the records of the send list of set_send_faces go, segment by segment as in pack_send_to, to
 * dst[i] -- the neighbours' trace tables of that area --, and flags[i] are the neighbours' sequence words.  stage_deliver arms
 * the NEXT launch over all shards (stage_update_part(h, 0) / stage(h, ..)): every workgroup whose shard has cut faces forms the
 * traces of its new state on them (the bits face_trace / pack_send_traces would give) and stores them at their destination;
 * the last such workgroup publishes `seq` in the words.  No rim launch of its own, no pack kernel, no second stream:
 * update_ghost_values (src_mpi/claw.cc:793) is part of the kernel that produced the values. */
int dflo_hip_set_deliver(dflo_hip_handle h, int area, int n_segments, const int32_t *first, void *const *dst, void *const *flags);
int dflo_hip_stage_deliver(dflo_hip_handle h, int area, uint64_t seq);
/* ... and the arrival of the neighbours' traces of the stage BEFORE can be awaited inside that launch as well: set_arrival_words
 * names this engine's own sequence words (one per neighbour that sends; fine-grained memory) and a host-mapped failure word;
 * stage_await(seq), together with stage_deliver, makes the workgroups of the shards that read ghost traces poll the words behind
 * their own loads until they have reached seq (30 s, then the failure word).  The other workgroups wait for nothing.  Only where
 * the trace tables are fine-grained memory (or written by this device itself): the traces are read inside the running kernel. */
int dflo_hip_set_arrival_words(dflo_hip_handle h, int n, void *const *words, void *fail);
int dflo_hip_stage_await(dflo_hip_handle h, uint64_t seq);
/* With a TVB limiter between update and send (src_mpi/limiter.cc:232: a second update_ghost_values) the exchange rides in BOTH
 * kernels of a stage.  set_deliver_averages, once per receive area: the averages of the cells of set_send_cells go to dst[i] (the
 * neighbours' average areas: [4] doubles per cell), flags[i] are the neighbours' words for them, words[] this engine's own words
 * for the neighbours' averages.  stage_deliver_averages arms the next launch over all shards: the workgroups of the shards on a
 * cut deliver their cells' new averages (and the stage kernel keeps those shards off the list of marked shards).  limit_exchange
 * arms the next limiter pass over all shards (stage_limit): one extra wavefront per shard on a cut waits for the neighbours'
 * averages to reach average_seq (poll_in_kernel; else the caller has waited), limits the shard with them (ghost_avg_source) and
 * delivers the traces of the limited state into the neighbours' tables of trace_area, publishing trace_seq.  Needs a pass that
 * walks the list of marked shards (limiter_walks_list: TVB on Qk squares with marks). */
int dflo_hip_set_deliver_averages(dflo_hip_handle h, int area, int n_segments, const int32_t *first, void *const *dst, void *const *flags,
                                  int n_words, void *const *words, void *fail);
int dflo_hip_stage_deliver_averages(dflo_hip_handle h, int area, uint64_t seq);
int dflo_hip_limit_exchange(dflo_hip_handle h, int trace_area, uint64_t trace_seq, uint64_t average_seq, int poll_in_kernel);
int dflo_hip_limiter_walks_list(dflo_hip_handle h);
/* The kernels that deliver store their values at system scope -- written through where the destination is fine-grained memory
 * -- and wait for them; where the destinations are PLAIN device memory (of another process on this device: only a release writes
 * such stores back) every delivering workgroup also has to fence: plain = 1. */
int dflo_hip_deliver_to_plain_memory(dflo_hip_handle h, int plain);
/* Pack and deliver in one kernel (several engines in one process): records [first[i], first[i+1]) of the send list are
 * written at dst[i] -- the receive area of the i-th peer, on this device or on another one reached over xGMI peer access --
 * instead of into a staging buffer that a copy per peer then moves.  kind: 0 whole cells ([ndof + 4] doubles per record,
 * as pack_send_cells), 1 cell averages ([4], as pack_send_avg), 2 face traces ([4 (k+1)], as pack_send_traces; the send
 * list is that of set_send_faces).  n_segments <= 16. */
int dflo_hip_pack_send_to(dflo_hip_handle h, int kind, int n_segments, const int32_t *first, void *const *dst);
/* The same, and the kernel tells the receivers: once every record of the launch is visible system-wide, the workgroup that
 * finishes last stores `seq` (release, system scope) into the 64-bit words flags[i] -- sequence words in the receivers'
 * fine-grained memory, which a one-wavefront wait kernel on the receiver's comm stream polls.  One process per GPU without a
 * transport library on the per-stage path: the receive areas and the words are mapped through hipIpcGetMemHandle /
 * hipIpcOpenMemHandle once, at create (dflo_hip_multi_create_rank with DFLO_RANK_TRANSPORT=ipc).  Replaces the same
 * update_ghost_values (src_mpi/claw.cc:793, src_mpi/limiter.cc:232).  flags NULL: dflo_hip_pack_send_to. */
int dflo_hip_pack_send_to_signal(dflo_hip_handle h, int kind, int n_segments, const int32_t *first, void *const *dst,
                                 void *const *flags, uint64_t seq);
/* device address of the {dt, elapsed time, raw CFL minimum} and {res_norm_sq per stage} scalars (src_mpi/claw.cc:579,777) */
int dflo_hip_scalar_ptrs(dflo_hip_handle h, void **dt_ptr, void **res_ptr);
/* The time step of a run over several engines (Utilities::MPI::min(global_dt), src_mpi/claw.cc:579) without a host hop and
 * without a kernel of its own.  Every engine keeps a device table mins[2][16] of raw CFL minima: row p holds, slot by slot,
 * the minima of all parts for the step of parity p (steps counted since set_solution).  The reductions that end a step write
 * this engine's minimum into slot my_slot of the next step's row -- of its own table and of the peer_tables handed over here
 * (engines of the same process: plain stores over xGMI peer access; entry my_slot and null entries are skipped) -- and every
 * consumer of the time step (stage kernels, boundary programs, the clock) takes the minimum over the n_slots of its row and
 * applies the rules of src/claw.cc:468-476 itself.  The caller orders the streams: the next step's first kernel after the
 * peers' reductions.  One process per GPU: n_slots = 1 and an all-reduce(min) in place on dflo_hip_dt_slot (the slot the
 * next step to run reads) between the two.  n_slots = 0 (the default): one engine, the reductions apply the rules. */
int dflo_hip_dt_table(dflo_hip_handle h, void **table);
int dflo_hip_dt_exchange(dflo_hip_handle h, int my_slot, int n_slots, void *const *peer_tables);
int dflo_hip_dt_slot(dflo_hip_handle h, void **slot);

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
 * partitioner: dflo_partitioner (declared with dflo_mesh_partition_ex below: 0 = slabs, 1 = RCB). */
#define DFLO_COMM_ID_BYTES 128
typedef struct dflo_hip_multi *dflo_hip_multi_handle;
int dflo_hip_multi_create(const dflo_mesh_t *mesh, const dflo_params_t *params, int n_devices, const int *device_ids,
                          int partitioner, dflo_hip_multi_handle *out);
int dflo_hip_comm_unique_id(void *id_bytes);
int dflo_hip_multi_create_rank(const dflo_mesh_t *mesh, const dflo_params_t *params, int device_id, int rank, int n_ranks,
                               const void *unique_id, int partitioner, dflo_hip_multi_handle *out);
/* One process per GPU with the host program's own transport instead of RCCL (a cluster whose ranks talk MPI; the
 * repository's two-process tests on one GPU, which RCCL refuses).  Both callbacks get DEVICE pointers and the driver's comm
 * stream (hipStream_t as void*); when they return, the transfer must be complete or enqueued on that stream in order.
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
/* Self-halo: ONE part on ONE device that is its own neighbour across a virtual cut (dflo_mesh_partition_self below:
 * n_virtual = 1 cuts at the periodic faces in x, >= 2 between the virtual parts of `partitioner`), driven through the complete
 * stage schedule of a multi-device run -- rim shards beside the interior on two streams, pack, transport into the trace table,
 * the time-step reduction -- where a plain one-part handle issues the single engine's launches.  A measuring device for boxes
 * with one GPU: its rate over the plain engine's bounds the weak-scaling efficiency of a rank whose neighbours are as fast as
 * itself (update_ghost_values / Utilities::MPI::min of src_mpi/claw.cc:793, 579 and src_mpi/limiter.cc:232 all happen, against
 * itself).  Results are those of the single engine, bit for bit on the nodal basis.  transport: dflo_self_transport --
 * DIRECT the one-process schedule (pack kernels store into the own trace table), RCCL the one-process-per-GPU schedule on a
 * one-rank communicator (grouped ncclSend / ncclRecv to itself, ncclAllReduce(min)), COPY staging buffer + hipMemcpyPeerAsync,
 * IPC the one-process-per-GPU schedule with the sequence-word transport of DFLO_RANK_TRANSPORT=ipc against itself. */
typedef enum { DFLO_SELF_DIRECT = 0, DFLO_SELF_RCCL = 1, DFLO_SELF_COPY = 2, DFLO_SELF_IPC = 3 } dflo_self_transport;
int dflo_hip_multi_create_self(const dflo_mesh_t *mesh, const dflo_params_t *params, int device_id, int n_virtual, int partitioner,
                               int transport, dflo_hip_multi_handle *out);
/* One process per GPU with DFLO_RANK_TRANSPORT=ipc: collective (the ranks meet before anybody frees a window its neighbours map). */
int dflo_hip_multi_destroy(dflo_hip_multi_handle m);
const char *dflo_hip_multi_last_error(dflo_hip_multi_handle m); /* m may be NULL: error of the last failed create */
int dflo_hip_multi_n_parts(dflo_hip_multi_handle m);            /* parts of the partition */
int dflo_hip_multi_n_local(dflo_hip_multi_handle m);            /* parts (engines) this process owns */
dflo_hip_handle dflo_hip_multi_engine(dflo_hip_multi_handle m, int i); /* i-th local engine (timing, inspection) */
int dflo_hip_multi_part_cells(dflo_hip_multi_handle m, int i, int32_t *n_owned, int32_t *n_ghost, const int64_t **global_ids);
int64_t dflo_hip_multi_n_dofs(dflo_hip_multi_handle m);         /* of the undivided mesh */
int64_t dflo_hip_multi_n_owned_dofs(dflo_hip_multi_handle m);   /* owned by this process */
int32_t dflo_hip_multi_n_rk(dflo_hip_multi_handle m);
/* the calls of a single engine, on the undivided mesh */
int dflo_hip_multi_set_solution(dflo_hip_multi_handle m, const double *u);
/* The same for one local part in ITS numbering (owned cells first, then its ghost cells; dflo_hip_multi_part_mesh
 * gives the part's mesh, owned by the handle): a rank of a large run evaluates the initial data on its own cells only,
 * as VectorTools::interpolate does on the locally owned range (src_mpi/ic.cc). */
const dflo_mesh_t *dflo_hip_multi_part_mesh(dflo_hip_multi_handle m, int i);
int dflo_hip_multi_set_part_solution(dflo_hip_multi_handle m, int i, const double *u_part);
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
int dflo_hip_multi_synchronize(dflo_hip_multi_handle m);
int dflo_hip_multi_stage_timing(dflo_hip_multi_handle m, int enable, double *avg_ms, int64_t *n); /* slowest local part */
/* Reporting (bench.py's N > 1 line): the average time, in microseconds, that the comm stream of the local parts spent in an
 * exchange of halo records (every fifth exchange is bracketed by events: one process per GPU -- the grouped send / receive,
 * the rendezvous with the peers included; one process -- the wait for the peers' records), and what the transport is: the
 * number of ranks and this process's rank AS THE RCCL COMMUNICATOR REPORTS THEM (ncclCommCount / ncclCommUserRank; the
 * partition's numbers for the other transports, rank -1 in one process) and a description of the transport in use. */
int dflo_hip_multi_exchange_timing(dflo_hip_multi_handle m, int enable, double *avg_us, int64_t *n);
int dflo_hip_multi_comm_info(dflo_hip_multi_handle m, int32_t *comm_count, int32_t *comm_rank, char *transport, int32_t transport_len);

/* Test hook: evaluates the device reciprocal / square-root forms the flux functions use
 * (dflo_amd/csrc/physics.hpp) on n host doubles. */
int dflo_hip_debug_math(int n, const double *x, double *rcp_out, double *sqrt_out);
/* Test hook: exp() of the device library and the form the kinetic split fluxes use for their Gaussians (fexp_neg,
 * dflo_amd/csrc/physics.hpp; arguments <= 0), side by side. */
int dflo_hip_debug_exp(int n, const double *x, double *exp_library, double *exp_flux);

/* ------------------------------------------- host-side mesh construction */
/* What GridIn::read_msh + Triangulation hand to dflo (src/claw.cc:957-967),
 * flattened.  The returned mesh owns its arrays; free with dflo_mesh_free. */

/* nx x ny squares on [x0,x0+nx*h] x [y0,y0+ny*h], cell c = i + nx*j.
 * side_bc[4] = boundary id on the faces x=min, x=max, y=min, y=max, or -1 for
 * a periodic side (src_mpi semantics, src_mpi/assemble_explicit.cc:186-260). */
int dflo_mesh_cartesian(int32_t nx, int32_t ny, double x0, double y0, double h, const int32_t side_bc[4],
                        int32_t degree, dflo_mesh_t **out);
/* General conforming quad mesh: vertices [n_vertices][2], quads [n_quads][4]
 * (any consistent vertex order; re-ordered to deal.II's), boundary edges
 * [n_bedges][2] vertex pairs with ids.  mapping = DFLO_MAP_Q1. */
int dflo_mesh_from_quads(int32_t n_vertices, const double *vertices, int32_t n_quads, const int32_t *quads,
                         int32_t n_bedges, const int32_t *bedges, const int32_t *bedge_id, int32_t degree,
                         dflo_mesh_t **out);
/* Gmsh v2 ASCII .msh with quads + physical lines (what "gmsh -2 file.geo" writes, README.md:70-72). */
int dflo_mesh_read_gmsh(const char *path, int32_t degree, int32_t mapping, dflo_mesh_t **out);
/* Pair the boundary faces with ids id_first / id_second, offset along direction (0 = x, 1 = y), into periodic
 * neighbours in place: GridTools::collect_periodic_faces + add_periodicity for the "type = periodic", "pair",
 * "direction" entries of a boundary subsection (src_mpi/parameters.cc:397-410, src_mpi/claw.cc:156-200). */
int dflo_mesh_make_periodic(dflo_mesh_t *mesh, int32_t id_first, int32_t id_second, int32_t direction);
/* Sub-mesh of rank `rank` of `n_ranks` (contiguous slabs of the cell order after a
 * coordinate sort) with one layer of face-neighbour ghost cells -- the flat
 * equivalent of parallel::distributed::Triangulation's owned+ghost view
 * (src_mpi/claw.h:220).  send_cells/send_offsets (size n_ranks+1) list owned cells
 * to send per destination rank; recv_offsets the ghost ranges per source rank
 * (ghost cells are ordered by source rank). Arrays owned by the mesh. */
int dflo_mesh_partition(const dflo_mesh_t *mesh, int32_t n_ranks, int32_t rank, dflo_mesh_t **out,
                        const int32_t **send_cells, const int32_t **send_offsets, const int32_t **recv_offsets);
/* The same with a choice of partitioner (dflo_partitioner): DFLO_PART_SLAB as above (C4: x-slabs of the 4001 x 1000
 * lattice), DFLO_PART_RCB recursive coordinate bisection of the cell centres (compact blocks on unstructured meshes, C5;
 * the MPI variant gets Morton-order blocks from p4est, src_mpi/claw.h:220).  partition_owners writes the owner rank of
 * every cell ([n_cells]) without building a sub-mesh. */
typedef enum { DFLO_PART_SLAB = 0, DFLO_PART_RCB = 1 } dflo_partitioner;
int dflo_mesh_partition_ex(const dflo_mesh_t *mesh, int32_t n_ranks, int32_t rank, int32_t method, dflo_mesh_t **out,
                           const int32_t **send_cells, const int32_t **send_offsets, const int32_t **recv_offsets);
int dflo_mesh_partition_owners(const dflo_mesh_t *mesh, int32_t n_ranks, int32_t method, int32_t *owner_out);
/* Self-halo partition: ONE part that owns every cell and is its own neighbour across a virtual cut (n_virtual >= 2: the
 * non-periodic faces between the cells of different virtual owners of dflo_mesh_partition_owners; n_virtual == 1: the
 * periodic faces in x).  Every cell on the cut gets a ghost copy; send list = those cells, offsets for the one "peer" 0.
 * A measuring device (dflo_hip_multi_create_self): one full-size part runs the whole schedule that replaces
 * update_ghost_values / Utilities::MPI::min (src_mpi/claw.cc:793, 579) with itself as the neighbour. */
int dflo_mesh_partition_self(const dflo_mesh_t *mesh, int32_t n_virtual, int32_t method, dflo_mesh_t **out,
                             const int32_t **send_cells, const int32_t **send_offsets, const int32_t **recv_offsets);
void dflo_mesh_free(dflo_mesh_t *mesh);
const char *dflo_mesh_last_error(void);

/* Initial condition by nodal interpolation for Qk (VectorTools::interpolate, src/ic.cc:104-121):
 * xy [n_cells][n_s][2] = support point coordinates in dflo's DoF order. */
int dflo_mesh_support_points(const dflo_mesh_t *mesh, double *xy);

#ifdef __cplusplus
}
#endif
#endif /* DFLO_HIP_H */
