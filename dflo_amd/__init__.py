"""dflo_amd -- MI355X-native explicit DG residual + SSP-RK engine for dflo (host-side mirror).

The compute path is the HIP library behind include/dflo_hip.h; this package only mirrors the
reference's interface for that path (mesh description, parameters, ConservationLaw driver).
"""
from ._lib import DfloError, FLUX, BC, LIMITER  # noqa: F401
from .mesh import Mesh  # noqa: F401
from .params import Parameters  # noqa: F401
from .claw import ConservationLaw  # noqa: F401
from .multi import MultiConservationLaw  # noqa: F401
