// stage_inst.hip -- the stage kernels of ONE polynomial degree: compiled three times by build() with -DDFLO_STAGE_N=2|3|4
// (5 fluxes x 3 modes x 2 geometries x the limiter variants, Qk and Pk, take most of the compile time; the translation
// units are built in parallel with engine.hip, which holds everything else and reaches the kernels through stage_of_N).
#include "stage_kernels.hpp"

#ifndef DFLO_STAGE_N
#error "compile with -DDFLO_STAGE_N=1, 2, 3 or 4"
#endif
#define DFLO_CAT_(a, b) a##b
#define DFLO_CAT(a, b) DFLO_CAT_(a, b)

namespace dflo {
stage_fn DFLO_CAT(stage_of_, DFLO_STAGE_N)(int flux, int mode, int geo, int pos, int nt) { return pick_stage_n<DFLO_STAGE_N>(flux, mode, geo, pos, nt); }
stage_fn DFLO_CAT(stage_pk_of_, DFLO_STAGE_N)(int flux, int mode, int nt) { return pick_pk_n<DFLO_STAGE_N>(flux, mode, nt); }
}  // namespace dflo
