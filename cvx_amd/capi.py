"""ctypes view of libcvx_amd.so's C ABI (include/cvx_amd.h), for tests, bench.py and scripting.

The product is the shared library; this module only declares its entry points.  There is no
CPU fallback: if the library is missing the import fails, and without a HIP device every
compute entry point returns CVX_ERR_NO_DEVICE, which `check` raises as CvxError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CVX_AMD_LIB: a development build of the same library, e.g. one compiled with -DK2_PROF for scripts/k2_prof.py)
LIB_PATH = os.environ.get("CVX_AMD_LIB") or os.path.join(_HERE, "libcvx_amd.so")

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)
fp = C.POINTER(C.c_float)
TRACE_MAX = 1024
RCCL_ID_BYTES = 128

OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_DUPLICATE_MARKER, ERR_EXCHANGE, ERR_NUMERIC = range(8)


class CvxError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = lib().cvx_status_string(status).decode()
        last = lib().cvx_last_error().decode()
        super().__init__("%s: %s%s" % (what, msg, (" [" + last + "]") if last else ""))


def check(status, what=""):
    if status != OK:
        raise CvxError(status, what)


class LMOptions(C.Structure):
    _fields_ = [("strategy", C.c_int), ("max_iterations", C.c_int),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double),
                ("initial_radius", C.c_double), ("max_radius", C.c_double), ("min_radius", C.c_double),
                ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("jacobi_scaling", C.c_int), ("nonmonotonic", C.c_int),
                ("max_consecutive_nonmonotonic_steps", C.c_int),
                ("tau", C.c_double), ("eps1", C.c_double), ("eps2", C.c_double), ("eps3", C.c_double),
                ("inner_iterations", C.c_int), ("inner_iteration_tolerance", C.c_double)]


class LMSummary(C.Structure):
    _fields_ = [("iterations", C.c_int), ("num_successful", C.c_int), ("num_linear_solves", C.c_int),
                ("termination", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("trace_len", C.c_int),
                ("trace_cost", C.c_double * TRACE_MAX), ("trace_accepted", C.c_double * TRACE_MAX),
                ("trace_damping", C.c_double * TRACE_MAX), ("trace_gradient", C.c_double * TRACE_MAX),
                ("seconds_total", C.c_double), ("num_inner_iteration_steps", C.c_int)]

    def trace(self):
        n = self.trace_len
        return {"cost": np.array(self.trace_cost[:n]), "accepted": np.array(self.trace_accepted[:n]),
                "damping": np.array(self.trace_damping[:n]), "gradient": np.array(self.trace_gradient[:n])}


class CalibDesc(C.Structure):
    _fields_ = [("n_cameras", C.c_int), ("n_views", C.c_int), ("n_points", C.c_int), ("n_obs", C.c_int64),
                ("cam_of_view", ip), ("view_off", ip), ("point_id", ip), ("obs_x", dp), ("obs_y", dp),
                ("fix_k3", C.c_int), ("fix_intrinsics", C.c_int), ("fix_points", C.c_int),
                ("device", C.c_int), ("stream", C.c_void_p), ("obs_xf", fp), ("obs_yf", fp),
                ("prepare_inner_iterations", C.c_int)]


class ToolBADesc(C.Structure):
    _fields_ = [("n_views", C.c_int), ("n_points", C.c_int), ("n_obs", C.c_int64), ("view_off", ip), ("point_id", ip),
                ("obs_x", dp), ("obs_y", dp), ("trans", dp), ("intrinsics", C.c_double * 9), ("huber", C.c_double),
                ("device", C.c_int), ("stream", C.c_void_p)]


class RansacParams(C.Structure):
    """cvx_ransac_params = cvx::util::RANSAC::Parameters (ransac.hpp:12-23)"""
    _fields_ = [("max_error", C.c_double), ("outlier_probability", C.c_double), ("max_num_trials", C.c_int), ("min_inliers", C.c_int)]


class Damping(C.Structure):
    _fields_ = [("inv_radius", C.c_double), ("mu", C.c_double), ("min_diag", C.c_double), ("max_diag", C.c_double)]


class SynthCalibSpec(C.Structure):
    _fields_ = [("n_cameras", C.c_int), ("views_per_camera", C.c_int), ("board_w", C.c_int), ("board_h", C.c_int),
                ("tile", C.c_double), ("seed", C.c_uint64), ("pixel_noise", C.c_double), ("ragged", C.c_int)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, dp, C.c_size_t, C.c_int, C.c_void_p)

# name -> (restype, argtypes): every symbol include/cvx_amd.h declares
SIGNATURES = {
    "cvx_status_string": (C.c_char_p, [C.c_int]),
    "cvx_last_error": (C.c_char_p, []),
    "cvx_device_count": (C.c_int, []),
    "cvx_lm_options_refine": (None, [C.POINTER(LMOptions)]),
    "cvx_lm_options_handeye": (None, [C.POINTER(LMOptions)]),
    "cvx_lm_options_levmar": (None, [C.POINTER(LMOptions)]),
    "cvx_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "cvx_rccl_create": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "cvx_rccl_allreduce": (C.c_int, [C.c_void_p, dp, C.c_size_t, C.c_int, C.c_void_p]),
    "cvx_rccl_destroy": (None, [C.c_void_p]),
    "cvx_partition_views": (C.c_int, [C.c_int, ip, C.c_int, ip]),
    "cvx_calib_create": (C.c_int, [C.POINTER(CalibDesc), C.POINTER(C.c_void_p)]),
    "cvx_calib_prepare_inner_iterations": (C.c_int, [C.c_void_p]),
    "cvx_release_cached_memory": (C.c_int, []),
    "cvx_calib_destroy": (None, [C.c_void_p]),
    "cvx_calib_set_params": (C.c_int, [C.c_void_p, dp, dp, dp]),
    "cvx_calib_get_params": (C.c_int, [C.c_void_p, dp, dp, dp]),
    "cvx_calib_num_shared": (C.c_int, [C.c_void_p]),
    "cvx_calib_cost": (C.c_int, [C.c_void_p, dp]),
    "cvx_calib_inner_pass": (C.c_int, [C.c_void_p, dp]),
    "cvx_toolba_inner_pass": (C.c_int, [C.c_void_p, dp, dp, dp]),
    "cvx_calib_inner_ordering": (C.c_int, [C.c_int, C.c_int, C.c_int, ip, ip, ip, C.c_int, C.c_int, ip, ip]),
    "cvx_calib_evaluate": (C.c_int, [C.c_void_p, dp, dp, dp, dp, dp]),
    "cvx_calib_evaluate_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "cvx_calib_init_scaling": (C.c_int, [C.c_void_p, C.c_int]),
    "cvx_calib_linear_step": (C.c_int, [C.c_void_p, C.POINTER(Damping), dp, dp, dp, dp, dp, dp, dp]),
    "cvx_calib_set_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "cvx_calib_solve": (C.c_int, [C.c_void_p, C.POINTER(LMOptions), C.POINTER(LMSummary)]),
    "cvx_calib_refine_return": (C.c_double, [C.c_double, C.c_int64]),
    "cvx_calib_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "cvx_calib_phase_times": (C.c_int, [C.c_void_p, dp]),
    "cvx_calib_time_schur": (C.c_int, [C.c_void_p, C.c_int, dp]),
    "cvx_calib_geometry": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "cvx_calib_speculation_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "cvx_handeye_create": (C.c_int, [C.c_int, dp, dp, dp, dp, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "cvx_handeye_destroy": (None, [C.c_void_p]),
    "cvx_handeye_cost": (C.c_int, [C.c_void_p, dp, dp]),
    "cvx_handeye_normal_eq": (C.c_int, [C.c_void_p, dp, dp, dp, dp]),
    "cvx_handeye_evaluate": (C.c_int, [C.c_void_p, dp, dp, dp, dp]),
    "cvx_handeye_normal_eq_static": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_int, C.c_void_p]),
    "cvx_handeye_set_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "cvx_handeye_solve": (C.c_int, [C.c_void_p, dp, C.POINTER(LMOptions), C.POINTER(LMSummary)]),
    "cvx_rotation_to_angle_axis": (None, [dp, dp]),
    "cvx_angle_axis_to_rotation": (None, [dp, dp]),
    "cvx_rotation_to_quaternion": (None, [dp, dp]),
    "cvx_affine_to_quat_trans": (None, [dp, dp, dp]),
    "cvx_quat_trans_to_affine": (None, [dp, dp]),
    "cvx_handeye_pairs_fixed": (C.c_int, [C.c_int, dp, dp, dp, dp, C.c_int, ip]),
    "cvx_handeye_pairs_moving": (C.c_int, [C.c_int, dp, dp, dp, dp, C.c_int, ip]),
    "cvx_handeye_linear": (C.c_int, [C.c_int, C.c_int, dp, dp, dp]),
    "cvx_toolba_create": (C.c_int, [C.POINTER(ToolBADesc), C.POINTER(C.c_void_p)]),
    "cvx_toolba_destroy": (None, [C.c_void_p]),
    "cvx_lm_options_toolba": (None, [C.POINTER(LMOptions)]),
    "cvx_toolba_set_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "cvx_toolba_cost": (C.c_int, [C.c_void_p, dp, dp, dp]),
    "cvx_toolba_normal_eq": (C.c_int, [C.c_void_p, dp, dp, dp, dp, dp, dp, dp, dp]),
    "cvx_toolba_solve": (C.c_int, [C.c_void_p, dp, dp, C.POINTER(LMOptions), C.POINTER(LMSummary)]),
    "cvx_homography_create": (C.c_int, [C.c_int, ip, fp, fp, fp, fp, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "cvx_homography_destroy": (None, [C.c_void_p]),
    "cvx_homography_solve": (C.c_int, [C.c_void_p, dp, C.c_int, dp, dp, ip]),
    "cvx_homography_set": (C.c_int, [C.c_void_p, dp]),
    "cvx_homography_run": (C.c_int, [C.c_void_p, C.c_int, dp]),
    "cvx_homography_get": (C.c_int, [C.c_void_p, dp, dp, ip]),
    "cvx_homography_dlt": (C.c_int, [C.c_void_p, dp]),
    "cvx_homography_residuals": (C.c_int, [C.c_void_p, dp, dp]),
    "cvx_ransac_default_params": (None, [C.POINTER(RansacParams)]),
    "cvx_homography_ransac": (C.c_int, [C.c_void_p, C.POINTER(RansacParams), ip, C.c_ulonglong, dp, ip,
                                        C.POINTER(C.c_ubyte), ip, ip, C.POINTER(C.c_void_p)]),
    "cvx_calib_data_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "cvx_calib_data_free": (None, [C.c_void_p]),
    "cvx_calib_data_info": (C.c_int, [C.c_void_p, ip, ip, ip, ip, C.POINTER(C.c_int64)]),
    "cvx_calib_data_copy": (C.c_int, [C.c_void_p, dp, ip, ip, dp, dp]),
    "cvx_calib_data_path": (C.c_char_p, [C.c_void_p, C.c_int]),
    "cvx_calib_data_save": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, dp, C.c_int,
                                      C.POINTER(C.c_char_p), ip, ip, dp, dp]),
    "cvx_camera_yaml_read": (C.c_int, [C.c_char_p, ip, ip, dp, dp]),
    "cvx_camera_yaml_write": (C.c_int, [C.c_char_p, C.c_int, C.c_int, dp, dp]),
    "cvx_robot_config_read": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, dp, dp]),
    "cvx_robot_config_write_launch": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, dp]),
    "cvx_synth_calib_count": (C.c_int64, [C.POINTER(SynthCalibSpec)]),
    "cvx_synth_calib": (C.c_int, [C.POINTER(SynthCalibSpec), ip, ip, ip, dp, dp, dp, dp, dp, dp, dp, dp]),
    "cvx_synth_handeye": (C.c_int, [C.c_int, C.c_uint64, C.c_double, C.c_double, C.c_int, dp, dp, dp, dp]),
}

_LIB = None


def lib():
    """The loaded library (raises OSError when cvx_amd/libcvx_amd.so has not been built)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C cvx_amd/csrc`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def pd(a):
    return a.ctypes.data_as(dp) if a is not None else None


def pi(a):
    return a.ctypes.data_as(ip) if a is not None else None


def options(kind="refine"):
    o = LMOptions()
    {"refine": lib().cvx_lm_options_refine, "handeye": lib().cvx_lm_options_handeye,
     "levmar": lib().cvx_lm_options_levmar, "toolba": lib().cvx_lm_options_toolba}[kind](C.byref(o))
    return o
