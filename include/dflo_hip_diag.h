/*
 * dflo_hip_diag.h -- diagnostics and test hooks of libdflo_hip.so: timing, counters, debug math.  NOT part of the contract
 * (dflo_hip.h) and not installed with it: bench.py and the tests use these, a dflo build does not.
 */
#ifndef DFLO_HIP_DIAG_H
#define DFLO_HIP_DIAG_H

#include "dflo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The boundary-value table in use (which = 0: stage 0, 1: later stages), [n_boundary_faces][k+1][4]. */
int dflo_hip_get_boundary_values(dflo_hip_handle h, int which, double *values);
/* Index (0-based, counted from the last dflo_hip_set_solution) of the time step in which the first failure flag went
 * up, -1 if none: where the reference would have stopped (src/positivity.cc:28-38 throws, :160-169 exits, inside the
 * stage).  dflo_hip_advance looks at the flags every 32 steps and returns early with the error. */
int dflo_hip_failure_step(dflo_hip_handle h, int64_t *step);
/* Positivity limiter applied inside the stage kernel (pos_lim without TVB on Qk): counts[0] = cell-stages that failed the
 * cheap nodal-box bound and went through the limiter proper (src/positivity.cc:43-205), counts[1] = cell-stages it
 * changed (theta1 < 1 or theta2 < 1), summed since the last reset.  A diagnostic for bench.py's config.check. */
int dflo_hip_positivity_stats(dflo_hip_handle h, int64_t *counts, int reset);
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
int dflo_hip_multi_stage_timing(dflo_hip_multi_handle m, int enable, double *avg_ms, int64_t *n); /* slowest local part */
/* device address of the {dt, elapsed time, raw CFL minimum} and {res_norm_sq per stage} scalars (src_mpi/claw.cc:579,777) */
int dflo_hip_scalar_ptrs(dflo_hip_handle h, void **dt_ptr, void **res_ptr);
/* Test hook: evaluates the device reciprocal / square-root forms the flux functions use
 * (dflo_amd/csrc/physics.hpp) on n host doubles. */
int dflo_hip_debug_math(int n, const double *x, double *rcp_out, double *sqrt_out);
/* Test hook: exp() of the device library and the form the kinetic split fluxes use for their Gaussians (fexp_neg,
 * dflo_amd/csrc/physics.hpp; arguments <= 0), side by side. */
int dflo_hip_debug_exp(int n, const double *x, double *exp_library, double *exp_flux);
/* Test hook (host only, no device): what the shard plan of a part (owned + ghost sub-mesh of dflo_mesh_partition*) knows of its
 * ghost cells' face neighbours -- the table the limiter pass over the ghost shards of a one-exchange TVB stage reads
 * (dflo_hip_limit_ghost_cells) --, table[n_ghost][4] in the sub-mesh's own cell numbering: the index of the neighbour where it is an
 * OWNED cell of the part, -1 at a physical boundary, -2 where the neighbour lives with the ghost's owner (its average comes with the
 * ghost's record). */
int dflo_hip_plan_ghost_neighbours(const dflo_mesh_t *part_mesh, int32_t *table);

#ifdef __cplusplus
}
#endif
#endif /* DFLO_HIP_DIAG_H */
