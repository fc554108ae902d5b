"""ctypes binding of the C ABI in include/dflo_hip.h (libdflo_hip.so).

This is the only door into the product: there is no CPU or PyTorch fallback.  If the
shared library has not been built (``python -c 'import __graft_entry__ as g; g.build()'``)
importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdflo_hip.so")

N_COMP = 4
MAX_BOUNDARIES = 10

# enums of include/dflo_hip.h
FLUX = {"lxf": 0, "sw": 1, "kfvs": 2, "roe": 3, "hllc": 4}            # src/parameters.h:229
BC = {"inflow": 0, "outflow": 1, "slip": 2, "pressure": 3, "farfield": 4}  # src/equation.h:862-869
LIMITER = {"none": 0, "TVB": 1}
SHOCK_INDICATOR = {"limiter": 0, "density": 1, "energy": 2, "u2": 3}
BASIS = {"Qk": 0, "Pk": 1}
MAPPING = {"q1": 0, "q2": 1, "cartesian": 2}
NBR_NONE = -1000000

OK = 0
ERRORS = {
    -1: "DFLO_ERR_BAD_PARAM", -2: "DFLO_ERR_NONSQUARE_CELL", -3: "DFLO_ERR_NEGATIVE_MEAN_STATE",
    -4: "DFLO_ERR_POSITIVITY_NO_ROOT", -5: "DFLO_ERR_HIP", -6: "DFLO_ERR_COMM", -7: "DFLO_ERR_UNSUPPORTED",
    -8: "DFLO_ERR_NOMEM",
}


class DfloError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "DFLO_ERR"), code, msg))


class MeshStruct(C.Structure):
    _fields_ = [
        ("n_cells", C.c_int32), ("n_owned_cells", C.c_int32), ("degree", C.c_int32), ("basis", C.c_int32),
        ("mapping", C.c_int32),
        ("cell_vertices", C.POINTER(C.c_double)), ("cell_face_neighbor", C.POINTER(C.c_int32)),
        ("cell_face_neighbor_face", C.POINTER(C.c_int32)), ("cell_global_id", C.POINTER(C.c_int64)),
    ]


class ParamsStruct(C.Structure):
    _fields_ = [
        ("flux_type", C.c_int32), ("limiter_type", C.c_int32), ("char_lim", C.c_int32), ("pos_lim", C.c_int32),
        ("global_time_step", C.c_int32), ("n_rk", C.c_int32),
        ("gravity", C.c_double), ("cfl", C.c_double), ("time_step", C.c_double), ("final_time", C.c_double),
        ("M", C.c_double), ("beta", C.c_double),
        ("bc_kind", C.c_int32 * MAX_BOUNDARIES),
        ("shock_indicator", C.c_int32), ("conserve_angular_momentum", C.c_int32),
    ]


# every symbol include/dflo_hip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "dflo_hip_create", "dflo_hip_destroy", "dflo_hip_last_error", "dflo_hip_set_stream", "dflo_hip_n_dofs",
    "dflo_hip_dofs_per_cell", "dflo_hip_n_rk", "dflo_hip_set_solution", "dflo_hip_get_solution",
    "dflo_hip_get_cell_average", "dflo_hip_n_boundary_faces", "dflo_hip_boundary_faces",
    "dflo_hip_set_boundary_values", "dflo_hip_set_boundary_program", "dflo_hip_get_boundary_values", "dflo_hip_residual", "dflo_hip_compute_dt", "dflo_hip_step", "dflo_hip_stage",
    "dflo_hip_end_step", "dflo_hip_advance", "dflo_hip_compute_cell_average", "dflo_hip_apply_limiter",
    "dflo_hip_apply_positivity_limiter", "dflo_hip_compute_shock_indicator", "dflo_hip_get_shock_indicator", "dflo_hip_check", "dflo_hip_synchronize", "dflo_hip_stage_timing", "dflo_hip_uses_mfma",
    "dflo_hip_set_send_cells", "dflo_hip_pack_send", "dflo_hip_pack_send_avg", "dflo_hip_unpack_ghost",
    "dflo_hip_unpack_ghost_avg", "dflo_hip_n_ghost_cells", "dflo_hip_stage_update", "dflo_hip_stage_limit",
    "dflo_hip_stage_open", "dflo_hip_stage_update_part", "dflo_hip_stage_limit_part", "dflo_hip_stage_finish",
    "dflo_hip_n_rim_shards",
    "dflo_hip_scalar_ptrs", "dflo_hip_debug_math", "dflo_hip_debug_exp",
    "dflo_mesh_cartesian", "dflo_mesh_from_quads", "dflo_mesh_read_gmsh", "dflo_mesh_partition", "dflo_mesh_make_periodic", "dflo_mesh_free",
    "dflo_mesh_last_error", "dflo_mesh_support_points", "dflo_mesh_partition_ex", "dflo_mesh_partition_owners",
    "dflo_hip_failure_step", "dflo_hip_positivity_stats", "dflo_hip_dt_table", "dflo_hip_dt_exchange", "dflo_hip_dt_slot",
    "dflo_hip_multi_create", "dflo_hip_comm_unique_id", "dflo_hip_multi_create_rank", "dflo_hip_multi_create_rank_custom", "dflo_hip_multi_create_self", "dflo_mesh_partition_self", "dflo_hip_pack_send_to_signal", "dflo_hip_attach_event", "dflo_hip_finish_enqueued", "dflo_hip_set_deliver", "dflo_hip_stage_deliver", "dflo_hip_set_arrival_words", "dflo_hip_stage_await", "dflo_hip_set_deliver_averages", "dflo_hip_stage_deliver_averages", "dflo_hip_limit_exchange", "dflo_hip_limiter_walks_list", "dflo_hip_deliver_to_plain_memory", "dflo_hip_set_ghost_trace_buffers", "dflo_hip_set_dt_table_buffer", "dflo_hip_multi_destroy",
    "dflo_hip_multi_last_error", "dflo_hip_multi_n_parts", "dflo_hip_multi_n_local", "dflo_hip_multi_engine",
    "dflo_hip_multi_part_cells", "dflo_hip_multi_n_dofs", "dflo_hip_multi_n_owned_dofs", "dflo_hip_multi_n_rk",
    "dflo_hip_multi_set_solution", "dflo_hip_multi_get_solution", "dflo_hip_multi_get_cell_average",
    "dflo_hip_multi_n_boundary_faces", "dflo_hip_multi_boundary_faces", "dflo_hip_multi_set_boundary_values",
    "dflo_hip_multi_set_boundary_program", "dflo_hip_multi_residual", "dflo_hip_multi_compute_dt", "dflo_hip_multi_step",
    "dflo_hip_multi_advance", "dflo_hip_multi_apply_limiter", "dflo_hip_multi_apply_positivity_limiter",
    "dflo_hip_multi_check", "dflo_hip_multi_synchronize", "dflo_hip_multi_stage_timing", "dflo_hip_multi_exchange_timing", "dflo_hip_multi_comm_info",
    "dflo_hip_multi_part_mesh", "dflo_hip_multi_set_part_solution", "dflo_hip_pack_send_cells", "dflo_hip_unpack_ghost_cells",
    "dflo_hip_halo_traces", "dflo_hip_n_ghost_traces", "dflo_hip_set_send_faces", "dflo_hip_pack_send_traces", "dflo_hip_pack_send_to", "dflo_hip_ghost_avg_source",
    "dflo_hip_ghost_trace_buffer", "dflo_hip_use_ghost_traces", "dflo_hip_pack_send_cells_unlimited", "dflo_hip_limit_ghost_cells",
    "dflo_hip_plan_ghost_neighbours", "dflo_hip_stage_tail_wait", "dflo_hip_n_part_shards", "dflo_hip_pack_publish",
]
PARTITIONER = {"slab": 0, "rcb": 1}
COMM_ID_BYTES = 128

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "dflo_amd: %s is missing -- build the HIP extension first (__graft_entry__.build()); "
        "there is no CPU fallback" % LIB_PATH)

lib = C.CDLL(LIB_PATH)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_H = C.c_void_p
_MP = C.POINTER(MeshStruct)


def _sig(name, restype, *argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


_sig("dflo_hip_create", C.c_int, _MP, C.POINTER(ParamsStruct), C.c_int, C.POINTER(_H))
_sig("dflo_hip_destroy", C.c_int, _H)
_sig("dflo_hip_last_error", C.c_char_p, _H)
_sig("dflo_hip_set_stream", C.c_int, _H, C.c_void_p)
_sig("dflo_hip_n_dofs", C.c_int64, _H)
_sig("dflo_hip_dofs_per_cell", C.c_int32, _H)
_sig("dflo_hip_n_rk", C.c_int32, _H)
_sig("dflo_hip_set_solution", C.c_int, _H, _dp)
_sig("dflo_hip_get_solution", C.c_int, _H, _dp)
_sig("dflo_hip_get_cell_average", C.c_int, _H, _dp)
_sig("dflo_hip_n_boundary_faces", C.c_int32, _H)
_sig("dflo_hip_boundary_faces", C.c_int, _H, _ip, _ip, _ip, _dp)
_sig("dflo_hip_set_boundary_values", C.c_int, _H, C.c_int, _dp)
_sig("dflo_hip_set_boundary_program", C.c_int, _H, C.c_int32, C.c_int32, C.c_int32, _ip, C.c_int32, _dp)
_sig("dflo_hip_get_boundary_values", C.c_int, _H, C.c_int, _dp)
_sig("dflo_hip_residual", C.c_int, _H, C.c_int, _dp)
_sig("dflo_hip_compute_dt", C.c_int, _H, C.c_double, _dp)
_sig("dflo_hip_step", C.c_int, _H, C.c_double, _dp, _dp)
_sig("dflo_hip_stage", C.c_int, _H, C.c_int, C.c_double)
_sig("dflo_hip_end_step", C.c_int, _H)
_sig("dflo_hip_advance", C.c_int, _H, C.c_int, _dp)
_sig("dflo_hip_compute_cell_average", C.c_int, _H)
_sig("dflo_hip_apply_limiter", C.c_int, _H)
_sig("dflo_hip_apply_positivity_limiter", C.c_int, _H)
_sig("dflo_hip_compute_shock_indicator", C.c_int, _H)
_sig("dflo_hip_get_shock_indicator", C.c_int, _H, _dp)
_sig("dflo_hip_check", C.c_int, _H)
_sig("dflo_hip_synchronize", C.c_int, _H)
_sig("dflo_hip_stage_timing", C.c_int, _H, C.c_int, _dp, C.POINTER(C.c_int64))
_sig("dflo_hip_uses_mfma", C.c_int, _H)
_sig("dflo_hip_finish_enqueued", C.c_int, _H)
_sig("dflo_hip_set_send_cells", C.c_int, _H, C.c_int32, _ip)
_sig("dflo_hip_pack_send", C.c_int, _H, C.c_void_p)
_sig("dflo_hip_pack_send_avg", C.c_int, _H, C.c_void_p)
_sig("dflo_hip_unpack_ghost", C.c_int, _H, C.c_void_p)
_sig("dflo_hip_unpack_ghost_avg", C.c_int, _H, C.c_void_p)
_sig("dflo_hip_n_ghost_cells", C.c_int, _H)
_sig("dflo_hip_halo_traces", C.c_int, _H)
_sig("dflo_hip_n_ghost_traces", C.c_int, _H)
_sig("dflo_hip_set_send_faces", C.c_int, _H, C.c_int32, _ip, _ip)
_sig("dflo_hip_pack_send_traces", C.c_int, _H, C.c_void_p)
_sig("dflo_hip_ghost_avg_source", C.c_int, _H, C.c_void_p)
_sig("dflo_hip_pack_send_to", C.c_int, _H, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_void_p))
_sig("dflo_hip_ghost_trace_buffer", C.c_int, _H, C.c_int, C.POINTER(C.c_void_p))
_sig("dflo_hip_use_ghost_traces", C.c_int, _H, C.c_int)
_sig("dflo_hip_pack_send_cells", C.c_int, _H, C.c_void_p)
_sig("dflo_hip_unpack_ghost_cells", C.c_int, _H, C.c_void_p)
_sig("dflo_hip_pack_send_cells_unlimited", C.c_int, _H, C.c_void_p)
_sig("dflo_hip_limit_ghost_cells", C.c_int, _H, C.c_void_p, C.c_int)
_sig("dflo_hip_plan_ghost_neighbours", C.c_int, _MP, C.POINTER(C.c_int32))
_sig("dflo_hip_stage_tail_wait", C.c_int, _H, C.c_void_p, C.c_uint64)
_sig("dflo_hip_n_part_shards", C.c_int, _H, C.c_int)
_sig("dflo_hip_pack_publish", C.c_int, _H, C.c_void_p, C.c_uint64)
_sig("dflo_hip_stage_update", C.c_int, _H, C.c_int, C.c_double)
_sig("dflo_hip_stage_limit", C.c_int, _H)
_sig("dflo_hip_stage_open", C.c_int, _H, C.c_int, C.c_double)
_sig("dflo_hip_stage_update_part", C.c_int, _H, C.c_int)
_sig("dflo_hip_stage_limit_part", C.c_int, _H, C.c_int)
_sig("dflo_hip_stage_finish", C.c_int, _H)
_sig("dflo_hip_set_deliver", C.c_int, _H, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p))
_sig("dflo_hip_stage_deliver", C.c_int, _H, C.c_int, C.c_uint64)
_sig("dflo_hip_set_arrival_words", C.c_int, _H, C.c_int, C.POINTER(C.c_void_p), C.c_void_p)
_sig("dflo_hip_stage_await", C.c_int, _H, C.c_uint64)
_sig("dflo_hip_limiter_walks_list", C.c_int, _H)
_sig("dflo_hip_n_rim_shards", C.c_int, _H)
_sig("dflo_hip_scalar_ptrs", C.c_int, _H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p))
_sig("dflo_hip_debug_math", C.c_int, C.c_int, _dp, _dp, _dp)
_sig("dflo_hip_debug_exp", C.c_int, C.c_int, _dp, _dp, _dp)
_sig("dflo_mesh_cartesian", C.c_int, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, _ip, C.c_int32,
     C.POINTER(_MP))
_sig("dflo_mesh_from_quads", C.c_int, C.c_int32, _dp, C.c_int32, _ip, C.c_int32, _ip, _ip, C.c_int32, C.POINTER(_MP))
_sig("dflo_mesh_read_gmsh", C.c_int, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(_MP))
_sig("dflo_mesh_partition", C.c_int, _MP, C.c_int32, C.c_int32, C.POINTER(_MP), C.POINTER(_ip), C.POINTER(_ip),
     C.POINTER(_ip))
_sig("dflo_mesh_partition_ex", C.c_int, _MP, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_MP), C.POINTER(_ip), C.POINTER(_ip),
     C.POINTER(_ip))
_sig("dflo_mesh_partition_owners", C.c_int, _MP, C.c_int32, C.c_int32, _ip)
_sig("dflo_mesh_partition_self", C.c_int, _MP, C.c_int32, C.c_int32, C.POINTER(_MP), C.POINTER(_ip), C.POINTER(_ip), C.POINTER(_ip))
_sig("dflo_hip_failure_step", C.c_int, _H, C.POINTER(C.c_int64))
_sig("dflo_hip_positivity_stats", C.c_int, _H, C.POINTER(C.c_int64), C.c_int)
_sig("dflo_hip_dt_table", C.c_int, _H, C.POINTER(C.c_void_p))
_sig("dflo_hip_dt_exchange", C.c_int, _H, C.c_int, C.c_int, C.POINTER(C.c_void_p))
_sig("dflo_hip_dt_slot", C.c_int, _H, C.POINTER(C.c_void_p))
_sig("dflo_hip_multi_create", C.c_int, _MP, C.POINTER(ParamsStruct), C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(_H))
_sig("dflo_hip_comm_unique_id", C.c_int, C.c_void_p)
_sig("dflo_hip_multi_create_rank", C.c_int, _MP, C.POINTER(ParamsStruct), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
     C.POINTER(_H))
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                          C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p)
_sig("dflo_hip_multi_create_rank_custom", C.c_int, _MP, C.POINTER(ParamsStruct), C.c_int, C.c_int, C.c_int, EXCHANGE_FN, ALLREDUCE_FN,
     C.c_void_p, C.c_int, C.POINTER(_H))
_sig("dflo_hip_multi_create_self", C.c_int, _MP, C.POINTER(ParamsStruct), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_H))
SELF_TRANSPORT = {"direct": 0, "rccl": 1, "copy": 2, "ipc": 3}
_sig("dflo_hip_multi_destroy", C.c_int, _H)
_sig("dflo_hip_multi_last_error", C.c_char_p, _H)
_sig("dflo_hip_multi_n_parts", C.c_int, _H)
_sig("dflo_hip_multi_n_local", C.c_int, _H)
_sig("dflo_hip_multi_engine", _H, _H, C.c_int)
_sig("dflo_hip_multi_part_cells", C.c_int, _H, C.c_int, _ip, _ip, C.POINTER(C.POINTER(C.c_int64)))
_sig("dflo_hip_multi_n_dofs", C.c_int64, _H)
_sig("dflo_hip_multi_n_owned_dofs", C.c_int64, _H)
_sig("dflo_hip_multi_n_rk", C.c_int32, _H)
_sig("dflo_hip_multi_set_solution", C.c_int, _H, _dp)
_sig("dflo_hip_multi_part_mesh", _MP, _H, C.c_int)
_sig("dflo_hip_multi_set_part_solution", C.c_int, _H, C.c_int, _dp)
_sig("dflo_hip_multi_get_solution", C.c_int, _H, _dp)
_sig("dflo_hip_multi_get_cell_average", C.c_int, _H, _dp)
_sig("dflo_hip_multi_n_boundary_faces", C.c_int32, _H)
_sig("dflo_hip_multi_boundary_faces", C.c_int, _H, _ip, _ip, _ip, _dp)
_sig("dflo_hip_multi_set_boundary_values", C.c_int, _H, C.c_int, _dp)
_sig("dflo_hip_multi_set_boundary_program", C.c_int, _H, C.c_int32, C.c_int32, C.c_int32, _ip, C.c_int32, _dp)
_sig("dflo_hip_multi_residual", C.c_int, _H, C.c_int, _dp)
_sig("dflo_hip_multi_compute_dt", C.c_int, _H, C.c_double, _dp)
_sig("dflo_hip_multi_step", C.c_int, _H, C.c_double, _dp, _dp)
_sig("dflo_hip_multi_advance", C.c_int, _H, C.c_int, _dp)
_sig("dflo_hip_multi_apply_limiter", C.c_int, _H)
_sig("dflo_hip_multi_apply_positivity_limiter", C.c_int, _H)
_sig("dflo_hip_multi_check", C.c_int, _H)
_sig("dflo_hip_multi_synchronize", C.c_int, _H)
_sig("dflo_hip_multi_stage_timing", C.c_int, _H, C.c_int, _dp, C.POINTER(C.c_int64))
_sig("dflo_hip_multi_exchange_timing", C.c_int, _H, C.c_int, _dp, C.POINTER(C.c_int64))
_sig("dflo_hip_multi_comm_info", C.c_int, _H, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p, C.c_int32)
_sig("dflo_mesh_make_periodic", C.c_int, _MP, C.c_int32, C.c_int32, C.c_int32)
_sig("dflo_mesh_free", None, _MP)
_sig("dflo_mesh_last_error", C.c_char_p)
_sig("dflo_mesh_support_points", C.c_int, _MP, _dp)


def dptr(a):
    return a.ctypes.data_as(_dp)


def iptr(a):
    return a.ctypes.data_as(_ip)
