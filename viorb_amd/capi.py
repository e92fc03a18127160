"""ctypes declarations of include/viorb.h (kept 1:1 with the header; tests/test_host_hooks.py checks that every
declared entry point is exported, has a signature here and that the argument counts agree)."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("VIORB_LIBRARY") or os.path.join(_HERE, "libviorb_hip.so")      # VIORB_LIBRARY: another build of the library (kernel experiments)

KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"),
                     ("response", "f4"), ("octave", "i4"), ("class_id", "i4")])

VIORB_OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5


class ViorbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("viorb error %d: %s" % (code, msg))
        self.code = code


class FrontendConfig(C.Structure):
    _fields_ = [("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("cam", C.c_double * 16), ("gravity", C.c_double * 3),
                ("scale_factors", C.c_float * 16), ("inv_level_sigma2", C.c_float * 16),
                ("nlevels", C.c_int32), ("check_orientation", C.c_int32),
                ("gyr_meas_cov", C.c_double), ("acc_meas_cov", C.c_double), ("acc_bias_rw2", C.c_double),
                ("dist_coef", C.c_float * 5), ("reserved0", C.c_int32)]


class LbaWindow(C.Structure):
    """viorb_lba_window (include/viorb.h): the arguments of one viorb_local_ba_navstate call + its status."""
    _fields_ = [("kfs", C.c_void_p), ("nk", C.c_int32), ("n_local", C.c_int32), ("prev_kf", C.c_int32), ("preint", C.c_void_p),
                ("points", C.c_void_p), ("np", C.c_int32), ("edge_idx", C.c_void_p), ("edge_obs", C.c_void_p), ("ne", C.c_int32),
                ("gw", C.c_void_p), ("cam", C.c_void_p), ("stop", C.c_void_p), ("kfs_out", C.c_void_p), ("points_out", C.c_void_p),
                ("erase", C.c_void_p), ("info", C.c_void_p), ("status", C.c_int32)]


class LbaSe3Window(C.Structure):
    """viorb_lba_se3_window (include/viorb.h)."""
    _fields_ = [("kfs", C.c_void_p), ("nk", C.c_int32), ("n_local", C.c_int32), ("points", C.c_void_p), ("np", C.c_int32),
                ("edge_idx", C.c_void_p), ("edge_obs", C.c_void_p), ("ne", C.c_int32), ("intr5", C.c_void_p), ("stop", C.c_void_p),
                ("kfs_out", C.c_void_p), ("points_out", C.c_void_p), ("erase", C.c_void_p), ("info", C.c_void_p), ("status", C.c_int32)]


class ExtractorParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


class VocabularyFlat(C.Structure):
    """viorb_vocabulary_flat (include/viorb.h)."""
    _fields_ = [("n_nodes", C.c_int32), ("k", C.c_int32), ("L", C.c_int32), ("n_words", C.c_int32), ("child_start", C.c_void_p),
                ("child_ids", C.c_void_p), ("word_id", C.c_void_p), ("desc", C.c_void_p), ("weight", C.c_void_p)]


class TrackerConfig(C.Structure):
    """viorb_tracker_config (include/viorb.h)."""
    _fields_ = [("extractor", ExtractorParams), ("frontend", FrontendConfig), ("width", C.c_int32), ("height", C.c_int32), ("batch", C.c_int32),
                ("device", C.c_int32), ("th_projection", C.c_float), ("track_local_map", C.c_int32), ("local_frames", C.c_int32),
                ("compute_marg", C.c_int32), ("max_steps_ahead", C.c_int32), ("synth_plane_z0", C.c_double)]


class TrackerInputs(C.Structure):
    """viorb_tracker_inputs (include/viorb.h)."""
    _fields_ = [("d_images", C.c_void_p), ("image_stride", C.c_int32), ("image_pitch_bytes", C.c_size_t), ("d_imu", C.c_void_p), ("n_imu", C.c_int32),
                ("d_t_cur", C.c_void_p), ("d_map_updated", C.c_void_p), ("d_recent_reloc", C.c_void_p), ("d_t_next_last", C.c_void_p),
                ("d_reset_ns", C.c_void_p), ("d_reset_marg", C.c_void_p), ("d_synth_pose12", C.c_void_p), ("h_images", C.c_void_p)]


class TrackerResults(C.Structure):
    """viorb_tracker_results (include/viorb.h)."""
    _fields_ = [("cap", C.c_int32)] + [(n, C.c_void_p) for n in (
        "state", "status", "nmatches", "n_map", "n_loc", "inliers", "n_obs", "n_obs2", "cur_match", "loc_match", "last_count",
        "info", "info2", "pred_ns", "ns_stage1", "ns_stage2", "final_ns", "final_marg", "last_ns",
        "outlier_cur", "outlier_cur2", "last_flags", "last_Pw", "last_pts_f", "extractor")]


vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
PP = C.POINTER
# name -> (restype, argtypes); mirrors include/viorb.h
SIGNATURES = {
    "viorb_abi_version": (i32, []),
    "viorb_last_error": (C.c_char_p, []),
    "viorb_device_count": (i32, []),
    "viorb_extractor_create": (i32, [PP(ExtractorParams), i32, i32, PP(vp)]),
    "viorb_extractor_destroy": (i32, [vp]),
    "viorb_extractor_tables": (i32, [vp, vp, vp, vp, vp, vp]),
    "viorb_extractor_max_keypoints": (i32, [vp, PP(i32)]),
    "viorb_extractor_max_keypoints_for": (i32, [vp, i32, i32, PP(i32)]),
    "viorb_extractor_fast_launch_images": (i32, [i32]),
    "viorb_extract": (i32, [vp, vp, i32, i32, i32, vp, vp, i32, PP(i32)]),
    "viorb_extract_batch_device": (i32, [vp, vp, i32, i32, i32, i32, sz, vp]),
    "viorb_extractor_results_device": (i32, [vp, PP(vp), PP(vp), PP(vp), PP(vp), PP(i32)]),
    "viorb_extractor_download": (i32, [vp, i32, vp, vp, i32, PP(i32)]),
    "viorb_extractor_level_device": (i32, [vp, i32, i32, i32, PP(vp), PP(i32), PP(i32), PP(i32)]),
    "viorb_extractor_level_download": (i32, [vp, i32, i32, i32, vp, PP(i32), PP(i32)]),
    "viorb_stereo_match_device": (i32, [vp, i32, vp, i32, i32, f32, f32, vp, vp, vp, vp]),
    "viorb_stereo_match": (i32, [vp, vp, f32, f32, vp, vp, i32, PP(i32)]),
    "viorb_extractor_debug_level_points": (i32, [vp, i32, i32, i32, vp, i32, PP(i32)]),
    "viorb_frontend_create": (i32, [PP(FrontendConfig), i32, i32, i32, PP(vp)]),
    "viorb_frontend_search_capacity": (i32, []),
    "viorb_frontend_destroy": (i32, [vp]),
    "viorb_frontend_grid_device": (i32, [vp, vp, vp, i32, vp, vp, vp]),
    "viorb_frontend_undistort_device": (i32, [vp, vp, vp, i32, vp, vp]),
    "viorb_undistort_points": (i32, [vp, i32, vp, vp, vp]),
    "viorb_image_bounds": (i32, [i32, i32, vp, vp, vp]),
    "viorb_frontend_imu_predict_device": (i32, [vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp]),
    "viorb_frontend_search_projection_device": (i32, [vp] * 12 + [f32, i32, vp, vp, vp, vp]),
    "viorb_frontend_search_projection_retry_device": (i32, [vp] * 12 + [f32, i32, i32, vp, vp, vp, vp]),
    "viorb_frontend_set_pose_shape": (i32, [i32, i32]),
    "viorb_frontend_search_local_points_device": (i32, [vp] * 11 + [i32, f32, f32, vp, i32, vp, vp, vp, vp, vp]),
    "viorb_frontend_search_local_points_stereo_device": (i32, [vp] * 5 + [f32] + [vp] * 7 + [i32, f32, f32, vp, i32, vp, vp, vp, vp, vp, vp]),
    "viorb_frontend_build_observations_device": (i32, [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp]),
    "viorb_frontend_pose_opt_device": (i32, [vp, i32, i32] + [vp] * 9 + [i32] + [vp] * 7),
    "viorb_frontend_pose_opt_se3_device": (i32, [vp, vp, vp, vp, C.c_double, i32, vp, vp, vp, vp]),
    "viorb_pose_opt_se3": (i32, [vp, vp, vp, i32, vp, vp, vp]),
    "viorb_local_ba_navstate": (i32, [vp, i32, i32, i32, vp, vp, i32, vp, vp, i32] + [vp] * 7),
    "viorb_local_ba_navstate_batch": (i32, [vp, i32, i32]),
    "viorb_local_ba_set_device": (i32, [i32]),
    "viorb_local_ba_se3_batch": (i32, [vp, i32, i32]),
    "viorb_local_ba_se3": (i32, [vp, i32, i32, vp, i32, vp, vp, i32] + [vp] * 6),
    "viorb_vocabulary_create": (i32, [i32, i32, vp, vp, vp, vp, vp, PP(vp)]),
    "viorb_vocabulary_destroy": (i32, [vp]),
    "viorb_vocabulary_read_file": (i32, [C.c_char_p, i32, PP(VocabularyFlat)]),
    "viorb_vocabulary_flat_free": (None, [PP(VocabularyFlat)]),
    "viorb_vocabulary_load_text": (i32, [C.c_char_p, PP(vp)]),
    "viorb_vocabulary_load_binary": (i32, [C.c_char_p, PP(vp)]),
    "viorb_vocabulary_save_text": (i32, [C.c_char_p, i32, i32, i32, vp, vp, vp, vp]),
    "viorb_vocabulary_save_binary": (i32, [C.c_char_p, i32, i32, i32, vp, vp, vp, vp]),
    "viorb_bow_transform_device": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "viorb_bow_transform": (i32, [vp, vp, i32, i32, vp, vp, vp]),
    "viorb_search_by_bow_device": (i32, [vp] * 9 + [i32, i32, f32, i32, vp, vp, vp]),
    "viorb_search_by_bow": (i32, [vp, vp, vp, vp, i32, vp, vp, vp, i32, f32, i32, vp, PP(i32)]),
    "viorb_match_bruteforce_device": (i32, [vp, vp, i32, vp, vp, i32, i32, vp, vp, vp, vp]),
    "viorb_match_bruteforce": (i32, [vp, i32, vp, i32, vp, vp, vp]),
    "viorb_frontend_discard_outliers_device": (i32, [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]),
    "viorb_frontend_pose_from_navstate_device": (i32, [vp, vp, i32, vp, vp]),
    "viorb_frontend_build_observations2_device": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]),
    "viorb_synth_local_points_device": (i32, [vp, vp, vp, vp, vp, i32, vp, vp]),
    "viorb_search_for_triangulation_device": (i32, [vp] * 18 + [i32, i32, i32, i32, i32, vp, vp, vp]),
    "viorb_search_for_triangulation": (i32, [vp] * 5 + [i32] + [vp] * 5 + [i32] + [vp] * 6 + [i32, i32, i32, vp, PP(i32)]),
    "viorb_frontend_fuse_device": (i32, [vp] * 12 + [i32, f32, f32, i32, vp, vp, vp]),
    "viorb_fuse": (i32, [vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, f32, vp, PP(i32)]),
    "viorb_synth_plane_points_device": (i32, [vp, vp, vp, vp, C.c_double, i32, vp, vp, vp, vp]),
    "viorb_frontend_roll_device": (i32, [vp] * 12 + [i32, i32] + [vp] * 7 + [i32, vp]),
    "viorb_memcpy_dtod_async": (i32, [vp, vp, sz, vp]),
    "viorb_memcpy_dtoh": (i32, [vp, vp, sz]),
    "viorb_memcpy_htod": (i32, [vp, vp, sz]),
    "viorb_frontend_pose_opt_select_device": (i32, [vp, vp, vp, i32] + [vp] * 9 + [i32] + [vp] * 7),
    "viorb_frontend_self_index_device": (i32, [vp, vp, vp, i32, vp, vp]),
    "viorb_tracker_create": (i32, [PP(TrackerConfig), PP(vp)]),
    "viorb_tracker_destroy": (i32, [vp]),
    "viorb_tracker_capacity": (i32, [vp, PP(i32)]),
    "viorb_tracker_bootstrap": (i32, [vp, vp, i32, sz, vp, vp, vp, vp, vp]),
    "viorb_tracker_set_last_points_device": (i32, [vp, vp, vp, vp, vp]),
    "viorb_tracker_step": (i32, [vp, PP(TrackerInputs), vp]),
    "viorb_tracker_sync": (i32, [vp]),
    "viorb_tracker_results_device": (i32, [vp, PP(TrackerResults)]),
    "viorb_tracker_host_stats": (i32, [vp, PP(C.c_double), PP(C.c_double), PP(C.c_longlong), i32]),
    "viorb_profile_enable": (i32, [i32]),
    "viorb_profile_reset": (i32, []),
    "viorb_profile_select": (i32, [C.c_char_p]),
    "viorb_profile_read": (i32, [C.c_char_p, i32, vp, vp, i32, PP(i32)]),
    "viorb_profile_timeline": (i32, [vp, vp, vp, i32, vp]),
    "viorb_descriptor_distance": (i32, [vp, vp]),
    "viorb_search_by_projection_frame": (i32, [vp, vp, i32, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, f32, i32, vp, PP(i32)]),
    "viorb_search_by_projection_frame_stereo": (i32, [vp, vp, vp, i32, vp, vp, vp, vp, f32, f32, vp, i32, vp, i32, vp, vp, vp, f32, i32, vp, PP(i32)]),
    "viorb_frontend_search_projection_stereo_device": (i32, [vp] * 14 + [f32, f32, f32, i32, i32, vp, vp, vp, vp]),
    "viorb_search_by_projection_points": (i32, [vp, vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, i32, f32, f32, vp, vp, PP(i32), vp]),
    "viorb_search_by_projection_points_stereo": (i32, [vp, vp, vp, f32, i32, vp, vp, vp, vp, i32, vp, vp, vp, i32, f32, f32, vp, vp, PP(i32), vp, vp]),
    "viorb_preintegrate": (i32, [vp, i32, vp, vp, C.c_double, C.c_double, vp]),
    "viorb_pose_opt_vi": (i32, [i32, i32] + [vp] * 8 + [i32, vp, i32] + [vp] * 6),
    "viorb_debug_pvr_edge": (None, [vp] * 7),
    "viorb_debug_proj_edge": (None, [vp] * 5),
    "viorb_debug_prior_edge": (None, [vp] * 5),
    "viorb_debug_update_ns": (None, [vp] * 6),
    "viorb_debug_preint_step": (None, [vp, vp, vp, C.c_double]),
    "viorb_debug_octree_host": (i32, [vp, i32, i32, i32, i32, vp, i32, PP(i32)]),
    "viorb_debug_fast_atan2": (f32, [f32, f32]),
    "viorb_debug_sincos": (None, [f32, PP(f32), PP(f32)]),
}

_lib = None


def lib():
    """Load libviorb_hip.so. Raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc, gfx950). viorb_amd has no CPU fallback." % SO_PATH)
        # One HIP runtime per process: PyTorch ships its own libamdhip64 (same SONAME as /opt/rocm's).
        # If torch is going to be used in this process it must be loaded FIRST so that libviorb_hip.so
        # binds to the runtime already in memory; loading ours first makes torch open a second runtime
        # that then sees no GPU. C/C++ callers without torch are unaffected.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != VIORB_OK:
        raise ViorbError(rc, lib().viorb_last_error().decode())
    return rc


def ptr(a):
    """void* of a numpy array or a torch tensor (device or host)."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(a.data_ptr())
