"""The scalars of Parameters::AllParameters the explicit path reads (src/parameters.h:363-411,
schema src/parameters.cc:316-551)."""
from . import _lib


class Parameters:
    def __init__(self, flux="lxf", limiter="none", char_lim=True, pos_lim=False, cfl=0.9, time_step=0.0,
                 final_time=1.0e20, M=0.0, beta=2.0, gravity=0.0, time_step_type="global", boundary=None, n_rk=0,
                 shock_indicator="limiter", conserve_angular_momentum=False):
        self.flux = flux                    # subsection flux / flux
        self.limiter = limiter              # subsection limiter / type
        self.char_lim = char_lim            # characteristic limiter
        self.pos_lim = pos_lim              # positivity limiter
        self.cfl = cfl
        self.time_step = time_step
        self.final_time = final_time
        self.M = M
        self.beta = beta
        self.gravity = gravity
        self.time_step_type = time_step_type
        self.boundary = dict(boundary or {})  # boundary id -> kind name ("slip", "inflow", ...)
        self.n_rk = n_rk
        self.shock_indicator = shock_indicator  # subsection limiter / shock indicator: limiter | density | energy
        self.conserve_angular_momentum = conserve_angular_momentum  # subsection limiter (TVB on the Pk basis, src/limiter.cc:496-500)

    def struct(self):
        p = _lib.ParamsStruct()
        p.flux_type = _lib.FLUX[self.flux]
        p.limiter_type = _lib.LIMITER[self.limiter]
        p.char_lim = int(bool(self.char_lim))
        p.pos_lim = int(bool(self.pos_lim))
        p.global_time_step = 1 if self.time_step_type == "global" else 0
        p.n_rk = self.n_rk
        p.gravity = self.gravity
        p.cfl = self.cfl
        p.time_step = self.time_step
        p.final_time = self.final_time
        p.M = self.M
        p.beta = self.beta
        p.shock_indicator = _lib.SHOCK_INDICATOR[self.shock_indicator]
        p.conserve_angular_momentum = int(bool(self.conserve_angular_momentum))
        for i in range(_lib.MAX_BOUNDARIES):
            p.bc_kind[i] = _lib.BC[self.boundary.get(i, "outflow")]
        return p
