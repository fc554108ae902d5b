// abi.h -- everything libdflo_hip.so exports, for the translation units that define it: the contract (include/dflo_hip.h), the
// host-side mesh helpers (dflo_mesh.h), the seams below the contract (dflo_hip_transport.h) and the diagnostics (dflo_hip_diag.h).
#pragma once
#include "../../include/dflo_hip.h"
#include "../../include/dflo_mesh.h"
#include "../../include/dflo_hip_transport.h"
#include "../../include/dflo_hip_diag.h"
