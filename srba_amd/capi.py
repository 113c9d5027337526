"""ctypes mirror of include/srba_hip.h and srba_amd/csrc/engine_capi.h (plumbing for tests / bench; the product is the C ABI)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)

# enum srba_family
SE2_RELPOSE2D, SE2_RB2D, SE2_CART2D, SE3_STEREO, SE3_MONO, SE3_CART3D, SE3_RB3D, SE3_RELPOSE3D, SE2_STEREO = range(9)
SOLVER_SCHUR_DENSE, SOLVER_SCHUR_SPARSE, SOLVER_NO_SCHUR_SPARSE = range(3)
NOISE_IDENTITY, NOISE_MATRIX = range(2)
SENSOR_POSE_NONE, SENSOR_POSE_SE3 = range(2)
EXT_SCHUR_KEEPS_GRADIENT = 1   # enum srba_extensions
DIMS = {0: (3, 3, 3, 3), 1: (3, 2, 2, 3), 2: (3, 2, 2, 3), 3: (6, 3, 4, 12), 4: (6, 3, 2, 12), 5: (6, 3, 3, 12), 6: (6, 3, 3, 12), 7: (6, 6, 6, 12), 8: (3, 3, 4, 3)}  # P, L, O, PD
TRACE_LEN = 48

c_i32, c_u8, c_f64 = C.c_int32, C.c_uint8, C.c_double
PI32, PU8, PF64 = C.POINTER(c_i32), C.POINTER(c_u8), C.POINTER(c_f64)


class HipParams(C.Structure):
    _fields_ = [("family", c_i32), ("solver", c_i32), ("noise", c_i32), ("sensor_pose", c_i32),
                ("std_noise_observations", c_f64), ("lambda_", c_f64 * 36), ("sensor_pose_se3", c_f64 * 12),
                ("cam_left", c_f64 * 4), ("cam_right", c_f64 * 4), ("right_cam_pose", c_f64 * 7),
                ("max_iters", c_i32), ("use_robust_kernel", c_i32), ("kernel_param", c_f64),
                ("max_error_per_obs_to_stop", c_f64), ("max_rho", c_f64), ("max_lambda", c_f64),
                ("min_error_reduction_ratio_to_relinearize", c_f64), ("cov_recovery", c_i32), ("extensions", c_i32)]


class Capsule(C.Structure):
    _fields_ = [(n, c_i32) for n in ("n_edges", "n_unk_edges", "n_unk_lms", "n_known_lms", "n_pairs", "n_path", "n_obs", "n_valid", "n_bp", "n_bf",
                                     "n_hap", "n_hap_terms", "n_hf", "n_hf_terms", "n_hapf", "n_hapf_terms", "n_sch_terms", "reserved0")] + \
               [("edge_pose", PF64), ("ulm_pos", PF64), ("klm_pos", PF64),
                ("pair_path_off", PI32), ("path_edge", PI32), ("pair_needed", PU8), ("pose_required", PU8), ("pose", PF64),
                ("obs_pose", PI32), ("obs_lm", PI32), ("obs_valid", PI32), ("obs_z", PF64),
                ("bp_col", PI32), ("bp_res", PI32), ("bp_A", PI32), ("bp_D", PI32), ("bp_lm", PI32), ("bp_normal", PU8), ("colp_off", PI32),
                ("bf_col", PI32), ("bf_res", PI32), ("bf_pose", PI32), ("colf_off", PI32),
                ("hap_i", PI32), ("hap_j", PI32), ("hap_term_off", PI32), ("hap_t1", PI32), ("hap_t2", PI32),
                ("hf_i", PI32), ("hf_j", PI32), ("hf_term_off", PI32), ("hf_t1", PI32), ("hf_t2", PI32),
                ("hapf_i", PI32), ("hapf_j", PI32), ("hapf_term_off", PI32), ("hapf_t1", PI32), ("hapf_t2", PI32),
                ("hap_diag", PI32), ("hf_diag", PI32),
                ("sch_term_off", PI32), ("sch_b1", PI32), ("sch_b2", PI32), ("sch_lm", PI32), ("lm_hapf_off", PI32), ("lm_hapf_idx", PI32),
                ("ulm_inf", PF64), ("ulm_inf_valid", PU8)]


class LmResult(C.Structure):
    _fields_ = [(n, c_i32) for n in ("status", "num_iters", "num_trials", "num_not_pd", "num_accepted", "num_relinearized", "num_invalid_jacobs", "stop_reason",
                                     "num_observations", "num_jacobians", "num_span_tree_numeric_updates", "reserved")] + \
               [(n, c_f64) for n in ("total_sqr_error_init", "total_sqr_error_final", "obs_rmse", "lambda_init", "lambda_final")] + \
               [("trace_chi2", c_f64 * TRACE_LEN), ("trace_lambda", c_f64 * TRACE_LEN), ("trace_rho", c_f64 * TRACE_LEN), ("lambda_last_trial", c_f64)]


class BatchStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_problems", "n_edges", "n_unk_edges", "n_unk_lms", "n_pairs", "n_pairs_needed", "n_path", "n_path_needed", "n_obs", "n_bp", "n_bf",
                                         "n_hap", "n_hap_terms", "n_hf_terms", "n_hapf_terms", "n_sch_terms", "n_scalars", "n_chol_blocks", "n_chol_items", "device_bytes")]


class EngineConfig(C.Structure):
    _fields_ = [("family", c_i32), ("solver", c_i32), ("noise", c_i32), ("sensor_pose", c_i32),
                ("std_noise_observations", c_f64), ("lambda_", c_f64 * 36), ("sensor_pose_xyzypr", c_f64 * 6),
                ("cam_left", c_f64 * 4), ("cam_right", c_f64 * 4), ("right_cam_pose", c_f64 * 7),
                ("max_tree_depth", C.c_uint64), ("max_optimize_depth", C.c_uint64), ("submap_size", C.c_uint64), ("min_obs_to_loop_closure", C.c_uint64),
                ("optimize_new_edges_alone", c_i32), ("use_robust_kernel", c_i32), ("use_robust_kernel_stage1", c_i32), ("max_iters", c_i32),
                ("kernel_param", c_f64), ("max_error_per_obs_to_stop", c_f64), ("max_rho", c_f64), ("max_lambda", c_f64), ("min_error_reduction_ratio_to_relinearize", c_f64),
                ("cov_recovery", c_i32), ("run_local_optimization", c_i32), ("harvest", c_i32), ("verbose", c_i32), ("enable_profiler", c_i32), ("hip_device", c_i32), ("refresh_all_read_poses",
                        c_i32), ("ecp", c_i32)]


class KfInfo(C.Structure):
    _fields_ = [("kf_id", C.c_uint64), ("n_new_edges", c_i32), ("reserved", c_i32),
                ("edge_id", C.c_uint64 * 4), ("lc_observer", C.c_uint64 * 4), ("lc_base", C.c_uint64 * 4), ("edge_has_init", c_i32 * 4),
                ("num_observations", C.c_uint64), ("num_jacobians", C.c_uint64), ("num_k2k", C.c_uint64), ("num_k2f", C.c_uint64),
                ("chi2_init", c_f64), ("chi2_final", c_f64), ("obs_rmse", c_f64), ("lm", LmResult), ("lm_stage1", LmResult)]


BACKEND_FN = C.CFUNCTYPE(c_i32, C.POINTER(HipParams), C.POINTER(Capsule), C.POINTER(LmResult))
PCAP = C.POINTER(Capsule)

_libs = {}


def _load(name, path):
    if name not in _libs:
        if not os.path.exists(path):
            raise RuntimeError("%s not built (%s): run `python __graft_entry__.py`" % (name, path))
        _libs[name] = C.CDLL(path, mode=C.RTLD_GLOBAL)
    return _libs[name]


def hip_lib():
    """libsrba_hip.so: the product. Raises if missing -- there is no fallback."""
    lib = _load("hip", os.path.join(_HERE, "lib", "libsrba_hip.so"))
    if not getattr(lib, "_proto", False):
        lib.srba_family_dims.argtypes = [c_i32, PI32, PI32, PI32, PI32]
        lib.srba_hip_params_default.argtypes = [C.POINTER(HipParams), c_i32]; lib.srba_hip_params_default.restype = None
        lib.srba_hip_create.argtypes = [c_i32, C.POINTER(HipParams)]; lib.srba_hip_create.restype = C.c_void_p
        lib.srba_hip_destroy.argtypes = [C.c_void_p]
        lib.srba_hip_set_params.argtypes = [C.c_void_p, C.POINTER(HipParams)]
        lib.srba_hip_last_error.argtypes = [C.c_void_p]; lib.srba_hip_last_error.restype = C.c_char_p
        lib.srba_hip_upload_problems.argtypes = [C.c_void_p, PCAP, c_i32]
        lib.srba_hip_reset_state.argtypes = [C.c_void_p]
        for f in ("srba_hip_linearize", "srba_hip_apply_update", "srba_hip_rollback", "srba_hip_lm_run_async", "srba_hip_sync"):
            getattr(lib, f).argtypes = [C.c_void_p]
        lib.srba_hip_update_spantree.argtypes = [C.c_void_p, c_i32]
        lib.srba_hip_eval_residuals.argtypes = [C.c_void_p, PF64]
        lib.srba_hip_solve.argtypes = [C.c_void_p, PF64, PI32]
        lib.srba_hip_lm_run.argtypes = [C.c_void_p, C.POINTER(LmResult)]
        lib.srba_hip_optimize_capsule.argtypes = [C.c_void_p, PCAP, C.POINTER(LmResult)]
        lib.srba_hip_stream.argtypes = [C.c_void_p]; lib.srba_hip_stream.restype = C.c_void_p
        lib.srba_hip_download_state.argtypes = [C.c_void_p, PCAP, c_i32]
        lib.srba_hip_download_results.argtypes = [C.c_void_p, C.POINTER(LmResult), c_i32]
        lib.srba_hip_debug_size.argtypes = [C.c_void_p, c_i32]; lib.srba_hip_debug_size.restype = C.c_int64
        lib.srba_hip_debug_read.argtypes = [C.c_void_p, c_i32, PF64, C.c_int64]
        lib.srba_hip_debug_write.argtypes = [C.c_void_p, c_i32, PF64, C.c_int64]
        lib.srba_hip_hessian_from_jacobians.argtypes = [C.c_void_p]
        lib.srba_hip_big_path_stats.argtypes = [C.c_void_p, PF64]
        lib.srba_hip_big_path_stats2.argtypes = [C.c_void_p, PF64]
        lib.srba_hip_spec_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        lib.srba_hip_debug_assemble_records.argtypes = [PCAP, C.POINTER(C.c_uint32), C.c_int64]; lib.srba_hip_debug_assemble_records.restype = C.c_int64
        lib.srba_hip_launch_order.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]
        lib.srba_hip_batch_stats.argtypes = [C.c_void_p, C.POINTER(BatchStats)]
        lib.srba_hip_last_kernel_ms.argtypes = [C.c_void_p]; lib.srba_hip_last_kernel_ms.restype = c_f64
        lib.srba_hip_kernel_ms_history.argtypes = [C.c_void_p, PF64, c_i32]
        lib._proto = True
    return lib


def engine_lib():
    hip_lib()
    lib = _load("engine", os.path.join(_HERE, "lib", "libsrba_engine.so"))
    if not getattr(lib, "_proto", False):
        lib.srba_engine_config_default.argtypes = [C.POINTER(EngineConfig), c_i32]; lib.srba_engine_config_default.restype = None
        lib.srba_engine_create.argtypes = [C.POINTER(EngineConfig)]; lib.srba_engine_create.restype = C.c_void_p
        lib.srba_engine_destroy.argtypes = [C.c_void_p]; lib.srba_engine_destroy.restype = None
        lib.srba_engine_last_error.argtypes = [C.c_void_p]; lib.srba_engine_last_error.restype = C.c_char_p
        lib.srba_engine_profiler_mean.argtypes = [C.c_void_p, C.c_char_p]; lib.srba_engine_profiler_mean.restype = c_f64
        lib.srba_engine_set_overall_fn.argtypes = [C.c_void_p, C.c_void_p]
        lib.srba_engine_eval_overall_sqr_error.argtypes = [C.c_void_p, C.POINTER(c_f64)]
        lib.srba_engine_set_backend_fn.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
        lib.srba_engine_add_keyframe.argtypes = [C.c_void_p, c_i32, C.POINTER(C.c_uint64), PF64, PU8, PF64, C.POINTER(KfInfo)]
        lib.srba_engine_optimize_local_area.argtypes = [C.c_void_p, C.c_uint64, C.c_uint, C.POINTER(KfInfo)]
        lib.srba_engine_num_edges.argtypes = [C.c_void_p]; lib.srba_engine_num_edges.restype = C.c_int64
        PU64 = C.POINTER(C.c_uint64)
        lib.srba_engine_plan_sweep.argtypes = [C.c_void_p, PU64, C.c_int64, C.c_uint, PI32, C.POINTER(C.c_int64), C.POINTER(C.c_uint32), C.c_int64]; lib.srba_engine_plan_sweep.restype = C.c_int64
        lib.srba_engine_optimize_batch.argtypes = [C.c_void_p, PU64, C.c_int64, C.c_uint, C.POINTER(KfInfo)]
        lib.srba_engine_get_edge_poses.argtypes = [C.c_void_p, PU64, C.c_int64, PF64]; lib.srba_engine_set_edge_poses.argtypes = [C.c_void_p, PU64, C.c_int64, PF64]
        lib.srba_engine_plan_sweep_lms.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_uint32), C.c_int64]; lib.srba_engine_plan_sweep_lms.restype = C.c_int64
        lib.srba_engine_get_lm_positions.argtypes = [C.c_void_p, PU64, C.c_int64, PF64]; lib.srba_engine_set_lm_positions.argtypes = [C.c_void_p, PU64, C.c_int64, PF64]
        lib.srba_engine_get_edge.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), PF64]
        lib.srba_engine_num_unknown_lms.argtypes = [C.c_void_p]; lib.srba_engine_num_unknown_lms.restype = C.c_int64
        lib.srba_engine_get_unknown_lms.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), PF64]
        lib.srba_engine_st_dump.argtypes = [C.c_void_p, c_i32, C.POINTER(C.c_int64), C.c_int64]; lib.srba_engine_st_dump.restype = C.c_int64
        lib.srba_engine_get_rel_pose.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, PF64]
        lib.srba_engine_profiler_mean.argtypes = [C.c_void_p, C.c_char_p]; lib.srba_engine_profiler_mean.restype = c_f64
        lib.srba_engine_set_overall_fn.argtypes = [C.c_void_p, C.c_void_p]
        lib.srba_engine_eval_overall_sqr_error.argtypes = [C.c_void_p, C.POINTER(c_f64)]
        lib.srba_engine_alloc_keyframe.argtypes = [C.c_void_p]; lib.srba_engine_alloc_keyframe.restype = C.c_uint64
        lib.srba_engine_create_edge.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, PF64]; lib.srba_engine_create_edge.restype = C.c_int64
        lib.srba_engine_export_graphslam.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), PF64, C.c_int64, C.POINTER(C.c_uint64), PF64,
                C.c_int64]; lib.srba_engine_export_graphslam.restype = C.c_int64
        lib.srba_engine_harvest_count.argtypes = [C.c_void_p]; lib.srba_engine_harvest_count.restype = C.c_int64
        lib.srba_engine_harvest_capsules.argtypes = [C.c_void_p]; lib.srba_engine_harvest_capsules.restype = PCAP
        lib.srba_engine_harvest_kf.argtypes = [C.c_void_p, C.c_int64]; lib.srba_engine_harvest_kf.restype = C.c_uint64
        lib.srba_engine_harvest_clear.argtypes = [C.c_void_p]; lib.srba_engine_harvest_clear.restype = None
        lib.srba_engine_get_hip_params.argtypes = [C.c_void_p, C.POINTER(HipParams)]
        lib.srba_engine_harvest_save.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64]
        lib.srba_capsule_file_load.argtypes = [C.c_char_p]; lib.srba_capsule_file_load.restype = C.c_void_p
        lib.srba_capsule_file_count.argtypes = [C.c_void_p]; lib.srba_capsule_file_count.restype = C.c_int64
        lib.srba_capsule_file_capsules.argtypes = [C.c_void_p]; lib.srba_capsule_file_capsules.restype = PCAP
        lib.srba_capsule_file_params.argtypes = [C.c_void_p, C.POINTER(HipParams)]
        lib.srba_capsule_file_free.argtypes = [C.c_void_p]; lib.srba_capsule_file_free.restype = None
        lib.srba_capsule_clone.argtypes = [PCAP, C.c_int64, c_i32]; lib.srba_capsule_clone.restype = C.c_void_p
        lib.srba_capsule_from_blocks.argtypes = [c_i32, c_i32, c_i32, c_i32, PI32, PI32, c_i32]; lib.srba_capsule_from_blocks.restype = C.c_void_p
        lib._proto = True
    return lib
