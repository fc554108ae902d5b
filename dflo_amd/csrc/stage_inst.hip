// stage_inst.hip -- the stage kernels of ONE polynomial degree: compiled by build() with -DDFLO_STAGE_N=1..6
// (5 fluxes x 3 modes x 2 geometries x the limiter variants, Qk and Pk, take most of the compile time; the translation
// units are built in parallel with engine.hip, which holds everything else and reaches the kernels through stage_of_N).
// For N = 5, 6 (degrees 4, 5: four minutes of compile time per degree) the unit is cut once more, by flux: with
// -DDFLO_STAGE_FLUX=f it holds the kernels of that flux only (stage_of_N_f / stage_pk_of_N_f).
// -DDFLO_STAGE_ONLY=1 (nodal, Qk) / 2 (modal, Pk) cuts a unit in two by element: the two halves of a degree may want
// different instruction schedulers (build(): measured per degree and element on MI355X).
#include "stage_kernels.hpp"

#ifndef DFLO_STAGE_N
#error "compile with -DDFLO_STAGE_N=1 .. 6"
#endif
#define DFLO_CAT_(a, b) a##b
#define DFLO_CAT(a, b) DFLO_CAT_(a, b)
#define DFLO_CAT4(a, b, c, d) DFLO_CAT(DFLO_CAT(a, b), DFLO_CAT(c, d))

namespace dflo {
#ifdef DFLO_STAGE_MF   // the matrix-pipe variants of degree 3 (-DDFLO_STAGE_N=4 -DDFLO_STAGE_MF=1 -DDFLO_STAGE_ONLY=1|2)
#if DFLO_STAGE_N != 4
#error "the matrix-pipe variants exist for N = 4 only"
#endif
#if DFLO_STAGE_ONLY == 1
stage_fn stage_mf_of_4(int flux, int mode, int geo, int pos, int nt) { return pick_stage_n<4, 1>(flux, mode, geo, pos, nt); }
#else
stage_fn stage_pk_mf_of_4(int flux, int mode, int nt) { return pick_pk_n<4, 1>(flux, mode, nt); }
#endif
#elif defined(DFLO_STAGE_FLUX)
stage_fn DFLO_CAT4(stage_of_, DFLO_STAGE_N, _f, DFLO_STAGE_FLUX)(int mode, int geo, int pos, int nt) { return pick_stage_m<DFLO_STAGE_N, DFLO_STAGE_FLUX>(mode, geo, pos, nt); }
stage_fn DFLO_CAT4(stage_pk_of_, DFLO_STAGE_N, _f, DFLO_STAGE_FLUX)(int mode, int nt) { return pick_pk_m<DFLO_STAGE_N, DFLO_STAGE_FLUX>(mode, nt); }
#else
#if !defined(DFLO_STAGE_ONLY) || DFLO_STAGE_ONLY == 1
stage_fn DFLO_CAT(stage_of_, DFLO_STAGE_N)(int flux, int mode, int geo, int pos, int nt) { return pick_stage_n<DFLO_STAGE_N>(flux, mode, geo, pos, nt); }
#endif
#if !defined(DFLO_STAGE_ONLY) || DFLO_STAGE_ONLY == 2
stage_fn DFLO_CAT(stage_pk_of_, DFLO_STAGE_N)(int flux, int mode, int nt) { return pick_pk_n<DFLO_STAGE_N>(flux, mode, nt); }
#endif
#endif
}  // namespace dflo
