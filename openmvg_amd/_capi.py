"""ctypes binding of libmvgx_hip.so (the C ABI declared in include/mvgx.h).

The library is the product; there is NO CPU fallback: if it is missing the import fails loudly.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` (hipcc, gfx950).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVGX_LIB_PATH") or os.path.join(_HERE, "lib", "libmvgx_hip.so")   # (override: kernel-variant experiments)

MVGX_OK = 0
MVGX_ERR_ARG, MVGX_ERR_HIP, MVGX_ERR_NODEV, MVGX_ERR_STATE, MVGX_ERR_UNSUPPORTED, MVGX_ERR_NUMERIC, MVGX_ERR_STRUCTURE = 1, 2, 3, 4, 5, 6, 7
MVGX_BA_MAX_INTR_PARAMS = 8


class MvgxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mvgx error {code}: {msg}")
        self.code = code


class MatchStats(C.Structure):
    _fields_ = [
        ("n_pairs", C.c_uint64),
        ("n_desc_pairs", C.c_uint64),
        ("n_matches", C.c_uint64),
        ("n_kernel_launches", C.c_uint64),
        ("kernel_ms", C.c_double),
        ("total_ms", C.c_double),
        ("kernel_vgprs", C.c_uint32),
        ("variant", C.c_uint32),
    ]


class BaProblem(C.Structure):
    _fields_ = [
        ("n_poses", C.c_uint32),
        ("n_intrinsics", C.c_uint32),
        ("n_points", C.c_uint32),
        ("n_obs", C.c_uint64),
        ("poses", C.c_void_p),
        ("intrinsics", C.c_void_p),
        ("intr_model", C.c_void_p),
        ("points", C.c_void_p),
        ("obs_pose", C.c_void_p),
        ("obs_intr", C.c_void_p),
        ("obs_point", C.c_void_p),
        ("obs_xy", C.c_void_p),
        ("pose_const_mask", C.c_void_p),
        ("intr_const_mask", C.c_void_p),
        ("points_constant", C.c_uint8),
        ("huber_a", C.c_double),
        ("obs_weight", C.c_void_p),
        ("obs_is_control", C.c_void_p),
        ("point_const_mask", C.c_void_p),
        ("n_pose_priors", C.c_uint32),
        ("prior_pose", C.c_void_p),
        ("prior_center", C.c_void_p),
        ("prior_weight", C.c_void_p),
        ("prior_huber_a", C.c_double),
    ]


class BaOptions(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("initial_radius", C.c_double),
        ("max_radius", C.c_double),
        ("min_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
        ("max_consecutive_invalid_steps", C.c_int32),
        ("jacobi_scaling", C.c_int32),
        ("verbose", C.c_int32),
    ]


class BaSummary(C.Structure):
    _fields_ = [
        ("num_iterations", C.c_int32),
        ("num_successful_steps", C.c_int32),
        ("termination", C.c_int32),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("initial_rmse", C.c_double),
        ("final_rmse", C.c_double),
        ("total_ms", C.c_double),
        ("iter_ms_mean", C.c_double),
        ("jacobian_ms", C.c_double),
        ("schur_ms", C.c_double),
        ("solve_ms", C.c_double),
        ("backsub_ms", C.c_double),
        ("cost_ms", C.c_double),
    ]


class BaSolverInfo(C.Structure):
    _fields_ = [
        ("sparse", C.c_int32), ("n_columns", C.c_int32), ("n_padded", C.c_int32), ("n_parts", C.c_int32),
        ("n_border_blocks", C.c_int32), ("n_levels", C.c_int32), ("n_factor_tiles", C.c_int64), ("n_dense_tiles", C.c_int64),
        ("flops", C.c_double), ("n_point_groups", C.c_int32), ("n_grouped_points", C.c_int32),
    ]


class GeofilterOptions(C.Structure):
    _fields_ = [("precision", C.c_double), ("max_iterations", C.c_uint32)]


class GeofilterResult(C.Structure):
    _fields_ = [("F", C.c_double * 9), ("precision_robust", C.c_double), ("nfa", C.c_double), ("n_inliers", C.c_uint32), ("ok", C.c_uint32)]


class GeofilterStats(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("n_pairs_estimated", C.c_uint64), ("n_pairs_ok", C.c_uint64), ("n_inliers", C.c_uint64),
                ("kernel_ms", C.c_double), ("host_prepare_ms", C.c_double), ("total_ms", C.c_double),
                ("n_iterations", C.c_uint64), ("n_models", C.c_uint64), ("wave_clocks", C.c_uint64)]


MATCH_SINK = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32)
MATCH_BATCH_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))
HOST_ITEM_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint)
ALLREDUCE_F64 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p)
MVGX_REDUCE_SUM, MVGX_REDUCE_MAX = 0, 1

# name -> (restype, argtypes). tests/test_capi_symbols.py checks every one of these is exported.
class GuidedStats(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("n_matches", C.c_uint64), ("n_geometric_tests", C.c_uint64), ("n_geometric_passed", C.c_uint64),
                ("n_descriptor_stages", C.c_uint64), ("kernel_ms", C.c_double), ("total_ms", C.c_double)]


PROTOTYPES = {
    "mvgx_last_error": (C.c_char_p, []),
    "mvgx_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mvgx_abi_version": (C.c_int, []),
    "mvgx_match_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mvgx_match_create_multi": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "mvgx_match_destroy": (C.c_int, [C.c_void_p]),
    "mvgx_match_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "mvgx_match_set_regions": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32]),
    "mvgx_match_set_regions_device": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32]),
    "mvgx_match_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.POINTER(MatchStats)]),
    "mvgx_match_run_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, MATCH_BATCH_SINK, C.c_void_p,
                                        C.POINTER(MatchStats)]),
    "mvgx_match_results": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint32))]),
    "mvgx_match_pairs_u8_l2": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_void_p,
                                          C.c_uint64, C.c_float, C.c_int, MATCH_SINK, C.c_void_p]),
    "mvgx_hamming_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mvgx_hamming_destroy": (C.c_int, [C.c_void_p]),
    "mvgx_hamming_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "mvgx_hamming_set_regions": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32]),
    "mvgx_hamming_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.POINTER(MatchStats)]),
    "mvgx_hamming_results": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint32))]),
    "mvgx_l2f_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mvgx_l2f_destroy": (C.c_int, [C.c_void_p]),
    "mvgx_l2f_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "mvgx_l2f_set_regions": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32]),
    "mvgx_l2f_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.POINTER(MatchStats)]),
    "mvgx_l2f_results": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint32))]),
    "mvgx_l2u8_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mvgx_l2u8_destroy": (C.c_int, [C.c_void_p]),
    "mvgx_l2u8_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "mvgx_l2u8_set_regions": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32]),
    "mvgx_l2u8_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.POINTER(MatchStats)]),
    "mvgx_l2u8_results": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint32))]),
    "mvgx_cascade_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mvgx_cascade_destroy": (C.c_int, [C.c_void_p]),
    "mvgx_cascade_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "mvgx_cascade_set_regions": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "mvgx_cascade_set_regions_typed": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                 C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "mvgx_cascade_hash_regions": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_void_p,
                                            C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "mvgx_cascade_hash_regions_typed": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_void_p,
                                                  C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "mvgx_cascade_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.POINTER(MatchStats)]),
    "mvgx_cascade_results": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint32))]),
    "mvgx_ba_default_options": (None, [C.POINTER(BaOptions)]),
    "mvgx_ba_create": (C.c_int, [C.c_int, C.POINTER(BaProblem), C.POINTER(C.c_void_p)]),
    "mvgx_ba_create_multi": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(BaProblem), C.POINTER(C.c_void_p)]),
    "mvgx_ba_destroy": (C.c_int, [C.c_void_p]),
    "mvgx_ba_update": (C.c_int, [C.c_void_p, C.POINTER(BaProblem)]),
    "mvgx_ba_update_subset": (C.c_int, [C.c_void_p, C.POINTER(BaProblem), C.c_void_p]),
    "mvgx_host_parallel_for": (C.c_int, [C.c_uint64, C.c_uint, HOST_ITEM_FN, C.c_void_p]),
    "mvgx_ba_set_allreduce": (C.c_int, [C.c_void_p, ALLREDUCE_F64, C.c_void_p]),
    "mvgx_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mvgx_ba_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mvgx_ba_solve": (C.c_int, [C.c_void_p, C.POINTER(BaOptions), C.POINTER(BaSummary)]),
    "mvgx_ba_lm_iteration": (C.c_int, [C.c_void_p, C.POINTER(BaOptions), C.POINTER(BaSummary)]),
    "mvgx_ba_read_params": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mvgx_ba_evaluate": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mvgx_ba_residuals": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mvgx_ba_track_angles": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mvgx_ba_get_solver_info": (C.c_int, [C.c_void_p, C.POINTER(BaSolverInfo)]),
    "mvgx_ba_set_linear_solver": (C.c_int, [C.c_void_p, C.c_int]),
    "mvgx_geofilter_f_acransac": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(GeofilterOptions),
                                            C.c_void_p, C.c_void_p, C.POINTER(GeofilterStats)]),
    "mvgx_geofilter_f_acransac_indexed": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_uint64, C.POINTER(GeofilterOptions), C.c_void_p, C.c_void_p, C.POINTER(GeofilterStats)]),
    "mvgx_geofilter_h_acransac": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(GeofilterOptions),
                                            C.c_void_p, C.c_void_p, C.POINTER(GeofilterStats)]),
    "mvgx_geofilter_h_acransac_indexed": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_uint64, C.POINTER(GeofilterOptions), C.c_void_p, C.c_void_p, C.POINTER(GeofilterStats)]),
    "mvgx_geofilter_e_acransac": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                            C.POINTER(GeofilterOptions), C.c_void_p, C.c_void_p, C.POINTER(GeofilterStats)]),
    "mvgx_geofilter_e_acransac_indexed": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_uint64, C.POINTER(GeofilterOptions), C.c_void_p, C.c_void_p, C.POINTER(GeofilterStats)]),
    "mvgx_geofilter_eo_acransac": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(GeofilterOptions),
                                             C.c_void_p, C.c_void_p, C.POINTER(GeofilterStats)]),
    "mvgx_geofilter_eo_acransac_indexed": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_uint64, C.POINTER(GeofilterOptions), C.c_void_p, C.c_void_p, C.POINTER(GeofilterStats)]),
    "mvgx_geofilter_e_angular_acransac": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(GeofilterOptions),
                                                    C.c_void_p, C.c_void_p, C.POINTER(GeofilterStats)]),
    "mvgx_geofilter_e_angular_acransac_indexed": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                                            C.c_int, C.POINTER(GeofilterOptions), C.c_void_p, C.c_void_p, C.POINTER(GeofilterStats)]),
    "mvgx_guided_match_u8": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                       C.c_int, C.c_double, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(GuidedStats)]),
    "mvgx_guided_match": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_uint64, C.c_int, C.c_double, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(GuidedStats)]),
    "mvgx_host_free": (None, [C.c_void_p]),
}

_lib = None


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension has not been built "
                "(run `python -c \"import __graft_entry__ as g; g.build()\"`). "
                "openmvg_amd has no CPU fallback.")
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (restype, argtypes) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing: loud by design
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(rc):
    if rc != MVGX_OK:
        msg = lib().mvgx_last_error()
        raise MvgxError(rc, msg.decode("utf-8", "replace") if msg else "")


def device_count():
    n = C.c_int(0)
    check(lib().mvgx_device_count(C.byref(n)))
    return n.value
