"""ctypes bindings for oracle/libcvx_oracle.so (and oracle/_ref/liblevmar_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under cvx_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
TRACE_MAX = 1024


class CalibProblem(C.Structure):
    _fields_ = [("C", C.c_int), ("V", C.c_int), ("P", C.c_int), ("N", C.c_int),
                ("cam_of_view", c_int_p), ("view_off", c_int_p), ("point_id", c_int_p),
                ("obs_x", c_double_p), ("obs_y", c_double_p),
                ("fix_k3", C.c_int), ("fix_intrinsics", C.c_int), ("fix_points", C.c_int)]


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
                ("num_inner_iteration_steps", C.c_int)]

    def trace(self):
        n = self.trace_len
        return {"cost": np.array(self.trace_cost[:n]), "accepted": np.array(self.trace_accepted[:n]),
                "damping": np.array(self.trace_damping[:n]), "gradient": np.array(self.trace_gradient[:n])}


class HandEyeProblem(C.Structure):
    _fields_ = [("n", C.c_int), ("qa", c_double_p), ("ta", c_double_p), ("qb", c_double_p), ("tb", c_double_p)]


LM_FUNC = C.CFUNCTYPE(None, c_double_p, c_double_p, C.c_int, C.c_int, C.c_void_p)


class RansacParams(C.Structure):
    _fields_ = [("max_error", C.c_double), ("outlier_probability", C.c_double), ("max_num_trials", C.c_int), ("min_inliers", C.c_int)]


class ToolBAProblem(C.Structure):
    _fields_ = [("V", C.c_int), ("P", C.c_int), ("N", C.c_int), ("view_off", c_int_p), ("point_id", c_int_p),
                ("obs_x", c_double_p), ("obs_y", c_double_p), ("trans", c_double_p), ("intr", C.c_double * 9),
                ("huber", C.c_double)]


def build(force=False):
    """Compile the C restatement (and, when /root/reference exists, levmar into oracle/_ref)."""
    so = os.path.join(_HERE, "libcvx_oracle.so")
    src = os.path.join(_HERE, "cvx_oracle.c")
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src)
    ref_missing = os.path.isdir("/root/reference") and not os.path.exists(os.path.join(_HERE, "_ref", "liblevmar_ref.so"))
    if force or stale or ref_missing:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "all"])


def set_threads(n):
    """OpenMP threads of the calibration model's per-view loops (bench.py's all-cores baseline only; 1 = the
    checker's single-thread summation order, the default)."""
    lib().orc_set_threads(int(n))


def lib():
    global _LIB
    if _LIB is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "libcvx_oracle.so"))
        dp, ip = c_double_p, c_int_p
        L.orc_reproj_analytic.argtypes = [dp, dp, dp, C.c_double, C.c_double, dp, dp, dp, dp]
        L.orc_reproj_autodiff.argtypes = L.orc_reproj_analytic.argtypes
        L.orc_handeye_analytic.argtypes = [dp] * 9
        L.orc_handeye_autodiff.argtypes = [dp] * 9
        L.orc_angle_axis_to_rotation.argtypes = [dp, dp]
        L.orc_rotation_to_angle_axis.argtypes = [dp, dp]
        L.orc_quaternion_plus.argtypes = [dp, dp, dp]
        for f in (L.orc_lm_options_refine, L.orc_lm_options_handeye, L.orc_lm_options_levmar):
            f.argtypes = [C.POINTER(LMOptions)]
        pp = C.POINTER(CalibProblem)
        L.orc_calib_num_shared.argtypes = [pp]; L.orc_calib_num_shared.restype = C.c_int
        L.orc_calib_num_free.argtypes = [pp]; L.orc_calib_num_free.restype = C.c_int
        L.orc_calib_cost.argtypes = [pp, dp, dp, dp]; L.orc_calib_cost.restype = C.c_double
        L.orc_calib_eval.argtypes = [pp, dp, dp, dp, dp, dp, dp, dp]
        L.orc_calib_dense_jacobian.argtypes = [pp, dp, dp, dp, dp]
        L.orc_calib_normal_blocks.argtypes = [pp, dp, dp, dp, dp, dp, dp, dp, dp, dp]
        L.orc_calib_reduced_system.argtypes = [pp, dp, dp, dp, dp, dp, dp, dp, dp]
        L.orc_calib_solve.argtypes = [pp, dp, dp, dp, C.POINTER(LMOptions), C.POINTER(LMSummary)]
        L.orc_calib_solve.restype = C.c_int
        L.orc_calib_refine_return.argtypes = [C.c_double, C.c_int]; L.orc_calib_refine_return.restype = C.c_double
        L.orc_calib_inner_ordering.argtypes = [pp, ip]; L.orc_calib_inner_ordering.restype = C.c_int
        L.orc_calib_inner_pass.argtypes = [pp, dp, dp, dp]; L.orc_calib_inner_pass.restype = C.c_double
        hp = C.POINTER(HandEyeProblem)
        L.orc_handeye_cost.argtypes = [hp, dp]; L.orc_handeye_cost.restype = C.c_double
        L.orc_handeye_normal_eq.argtypes = [hp, dp, dp, dp, dp]
        L.orc_handeye_solve.argtypes = [hp, dp, C.POINTER(LMOptions), C.POINTER(LMSummary)]
        L.orc_handeye_solve.restype = C.c_int
        L.orc_lm_nielsen_der.argtypes = [LM_FUNC, LM_FUNC, dp, dp, C.c_int, C.c_int, C.c_int, dp, dp, C.c_void_p]
        L.orc_lm_nielsen_der.restype = C.c_int
        L.orc_cholesky_solve.argtypes = [C.c_int, dp, dp]; L.orc_cholesky_solve.restype = C.c_int
        L.orc_set_threads.argtypes = [C.c_int]; L.orc_set_threads.restype = None
        L.orc_get_threads.argtypes = []; L.orc_get_threads.restype = C.c_int
        tp = C.POINTER(ToolBAProblem)
        L.orc_toolba_analytic.argtypes = [dp, dp, dp, dp, dp, C.c_double, C.c_double, dp, dp, dp, dp]
        L.orc_toolba_autodiff.argtypes = L.orc_toolba_analytic.argtypes
        L.orc_huber.argtypes = [C.c_double, C.c_double, dp]
        L.orc_toolba_cost.argtypes = [tp, dp, dp]; L.orc_toolba_cost.restype = C.c_double
        L.orc_toolba_normal_eq.argtypes = [tp, dp, dp, dp, dp, dp]
        L.orc_lm_options_toolba.argtypes = [C.POINTER(LMOptions)]
        L.orc_toolba_solve.argtypes = [tp, dp, dp, C.POINTER(LMOptions), C.POINTER(LMSummary)]
        L.orc_toolba_solve.restype = C.c_int
        L.orc_toolba_inner_pass.argtypes = [tp, dp, dp]; L.orc_toolba_inner_pass.restype = C.c_double
        fp = C.POINTER(C.c_float)
        L.orc_homography_lm.argtypes = [C.c_void_p, C.c_int, fp, fp, fp, fp, dp, C.c_int, dp, dp]
        L.orc_homography_lm.restype = C.c_int
        L.orc_homography_errors.argtypes = [C.c_int, fp, fp, fp, fp, dp, dp]
        L.orc_homography_jacobian.argtypes = [C.c_int, fp, fp, dp, dp]
        L.orc_homography_dlt.argtypes = [C.c_int, fp, fp, fp, fp, dp]
        L.orc_homography_dlt.restype = C.c_int
        L.orc_homography_dlt_matrix.argtypes = [C.c_int, fp, fp, fp, fp, dp, dp, dp]
        L.orc_null_vector9.argtypes = [C.c_int, dp, dp]
        L.orc_null_vector9.restype = C.c_int
        L.orc_homography_residuals.argtypes = [C.c_int, fp, fp, fp, fp, dp, dp]
        L.orc_ransac_defaults.argtypes = [C.POINTER(RansacParams)]
        L.orc_rng_sample4.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_homography_ransac.argtypes = [C.c_int, fp, fp, fp, fp, C.POINTER(RansacParams), C.POINTER(C.c_int), dp,
                                            C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_homography_ransac.restype = C.c_int
        _LIB = L
    return _LIB


def levmar_ref():
    """The reference's own levmar 2.5 (built from /root/reference by oracle/Makefile), or None."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "liblevmar_ref.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.dlevmar_der.argtypes = [LM_FUNC, LM_FUNC, c_double_p, c_double_p, C.c_int, C.c_int, C.c_int,
                                  c_double_p, c_double_p, c_double_p, c_double_p, C.c_void_p]
        R.dlevmar_der.restype = C.c_int
        _REF = R
    return _REF


def _dp(a):
    return a.ctypes.data_as(c_double_p) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(c_int_p)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Calib:
    """Oracle view of one calibration problem (numpy in / numpy out)."""

    def __init__(self, cam_of_view, view_off, point_id, obs_x, obs_y, n_cameras, n_points,
                 fix_k3=True, fix_intrinsics=False, fix_points=False):
        self.cam_of_view = i32(cam_of_view); self.view_off = i32(view_off); self.point_id = i32(point_id)
        self.obs_x = f64(obs_x); self.obs_y = f64(obs_y)
        self.C, self.P = int(n_cameras), int(n_points)
        self.V, self.N = len(self.cam_of_view), len(self.point_id)
        self.pb = CalibProblem(self.C, self.V, self.P, self.N, _ip(self.cam_of_view), _ip(self.view_off),
                               _ip(self.point_id), _dp(self.obs_x), _dp(self.obs_y),
                               int(fix_k3), int(fix_intrinsics), int(fix_points))
        self.ns = lib().orc_calib_num_shared(C.byref(self.pb))
        self.nfree = lib().orc_calib_num_free(C.byref(self.pb))

    def cost(self, intr, poses, points):
        intr, poses, points = f64(intr), f64(poses), f64(points)
        return lib().orc_calib_cost(C.byref(self.pb), _dp(intr), _dp(poses), _dp(points))

    def eval(self, intr, poses, points):
        intr, poses, points = f64(intr), f64(poses), f64(points)
        r = np.zeros(2 * self.N); Jc = np.zeros((self.N, 2, 9)); Jp = np.zeros((self.N, 2, 6)); Jx = np.zeros((self.N, 2, 3))
        lib().orc_calib_eval(C.byref(self.pb), _dp(intr), _dp(poses), _dp(points), _dp(r), _dp(Jc), _dp(Jp), _dp(Jx))
        return r.reshape(self.N, 2), Jc, Jp, Jx

    def dense_jacobian(self, intr, poses, points):
        intr, poses, points = f64(intr), f64(poses), f64(points)
        J = np.zeros((2 * self.N, self.nfree))
        lib().orc_calib_dense_jacobian(C.byref(self.pb), _dp(intr), _dp(poses), _dp(points), _dp(J))
        return J

    def normal_blocks(self, intr, poses, points, with_W=True):
        intr, poses, points = f64(intr), f64(poses), f64(points)
        Vv = np.zeros((self.V, 6, 6)); gv = np.zeros((self.V, 6))
        U = np.zeros((self.ns, self.ns)); gs = np.zeros(self.ns)
        W = np.zeros((self.V, 6, self.ns)) if with_W else None
        cost = C.c_double()
        lib().orc_calib_normal_blocks(C.byref(self.pb), _dp(intr), _dp(poses), _dp(points), _dp(Vv), _dp(gv),
                                      _dp(U), _dp(gs), _dp(W), C.byref(cost))
        return Vv, gv, U, gs, W, cost.value

    def reduced_system(self, intr, poses, points, Dp, Ds):
        intr, poses, points, Dp, Ds = f64(intr), f64(poses), f64(points), f64(Dp), f64(Ds)
        S = np.zeros((self.ns, self.ns)); g = np.zeros(self.ns); cost = C.c_double()
        lib().orc_calib_reduced_system(C.byref(self.pb), _dp(intr), _dp(poses), _dp(points), _dp(Dp), _dp(Ds),
                                       _dp(S), _dp(g), C.byref(cost))
        return S, g, cost.value

    def inner_ordering(self):
        """(group per block [C + P + V], number of independent sets) of Ceres' automatic inner-iteration ordering"""
        g = np.zeros(self.C + self.P + self.V, dtype=np.int32)
        n = lib().orc_calib_inner_ordering(C.byref(self.pb), _ip(g))
        return g, n

    def inner_pass(self, intr, poses, points):
        """one coordinate-descent pass of the inner iterations; returns the new (intr, poses, points, cost)"""
        intr, poses, points = f64(intr).copy(), f64(poses).copy(), f64(points).copy()
        c = lib().orc_calib_inner_pass(C.byref(self.pb), _dp(intr), _dp(poses), _dp(points))
        return intr, poses, points, c

    def solve(self, intr, poses, points, options):
        intr, poses, points = f64(intr).copy(), f64(poses).copy(), f64(points).copy()
        s = LMSummary()
        lib().orc_calib_solve(C.byref(self.pb), _dp(intr), _dp(poses), _dp(points), C.byref(options), C.byref(s))
        return intr, poses, points, s


class HandEye:
    def __init__(self, qa, ta, qb, tb):
        self.qa, self.ta, self.qb, self.tb = f64(qa), f64(ta), f64(qb), f64(tb)
        self.n = len(self.qa)
        self.pb = HandEyeProblem(self.n, _dp(self.qa), _dp(self.ta), _dp(self.qb), _dp(self.tb))

    def cost(self, p):
        p = f64(p)
        return lib().orc_handeye_cost(C.byref(self.pb), _dp(p))

    def normal_eq(self, p):
        p = f64(p); H = np.zeros((6, 6)); g = np.zeros(6); cost = C.c_double()
        lib().orc_handeye_normal_eq(C.byref(self.pb), _dp(p), _dp(H), _dp(g), C.byref(cost))
        return H, g, cost.value

    def solve(self, p, options):
        p = f64(p).copy(); s = LMSummary()
        lib().orc_handeye_solve(C.byref(self.pb), _dp(p), C.byref(options), C.byref(s))
        return p, s


class ToolBA:
    """Oracle view of the tool-level bundle adjustment (hand_eye_registration.cpp:144-238)."""

    def __init__(self, view_off, point_id, obs_x, obs_y, trans, intr, n_points, huber=1.0):
        self.view_off = i32(view_off); self.point_id = i32(point_id)
        self.obs_x = f64(obs_x); self.obs_y = f64(obs_y); self.trans = f64(np.asarray(trans).reshape(-1, 12))
        self.V, self.N, self.P = len(self.view_off) - 1, len(self.point_id), int(n_points)
        self.pb = ToolBAProblem(self.V, self.P, self.N, _ip(self.view_off), _ip(self.point_id), _dp(self.obs_x),
                                _dp(self.obs_y), _dp(self.trans), (C.c_double * 9)(*[float(v) for v in intr]), float(huber))
        self.nf = 6 + 3 * self.P

    def cost(self, p, points):
        p, points = f64(p), f64(points)
        return lib().orc_toolba_cost(C.byref(self.pb), _dp(p), _dp(points))

    def normal_eq(self, p, points):
        p, points = f64(p), f64(points)
        H = np.zeros((self.nf, self.nf)); g = np.zeros(self.nf); cost = C.c_double()
        lib().orc_toolba_normal_eq(C.byref(self.pb), _dp(p), _dp(points), _dp(H), _dp(g), C.byref(cost))
        return H, g, cost.value

    def inner_pass(self, p, points):
        p, points = f64(p).copy(), f64(points).copy()
        c = lib().orc_toolba_inner_pass(C.byref(self.pb), _dp(p), _dp(points))
        return p, points, c

    def solve(self, p, points, options):
        p, points = f64(p).copy(), f64(points).copy(); s = LMSummary()
        lib().orc_toolba_solve(C.byref(self.pb), _dp(p), _dp(points), C.byref(options), C.byref(s))
        return p, points, s


def toolba_functor(intr, trans, q, t, X, ox, oy, autodiff=False):
    intr, trans, q, t, X = f64(intr), f64(trans), f64(q), f64(t), f64(X)
    r = np.zeros(2); Jq = np.zeros((2, 4)); Jt = np.zeros((2, 3)); Jx = np.zeros((2, 3))
    f = lib().orc_toolba_autodiff if autodiff else lib().orc_toolba_analytic
    f(_dp(intr), _dp(trans), _dp(q), _dp(t), _dp(X), float(ox), float(oy), _dp(r), _dp(Jq), _dp(Jt), _dp(Jx))
    return r, Jq, Jt, Jx


def options(kind="refine"):
    o = LMOptions()
    {"refine": lib().orc_lm_options_refine, "handeye": lib().orc_lm_options_handeye,
     "levmar": lib().orc_lm_options_levmar, "toolba": lib().orc_lm_options_toolba}[kind](C.byref(o))
    return o


def reproj(intr, pose, X, ox, oy, autodiff=False):
    intr, pose, X = f64(intr), f64(pose), f64(X)
    r = np.zeros(2); Jc = np.zeros((2, 9)); Jp = np.zeros((2, 6)); Jx = np.zeros((2, 3))
    f = lib().orc_reproj_autodiff if autodiff else lib().orc_reproj_analytic
    f(_dp(intr), _dp(pose), _dp(X), float(ox), float(oy), _dp(r), _dp(Jc), _dp(Jp), _dp(Jx))
    return r, Jc, Jp, Jx


def handeye(q, t, qa, ta, qb, tb, autodiff=False):
    a = [f64(v) for v in (q, t, qa, ta, qb, tb)]
    r = np.zeros(12); Jq = np.zeros((12, 4)); Jt = np.zeros((12, 3))
    f = lib().orc_handeye_autodiff if autodiff else lib().orc_handeye_analytic
    f(*[_dp(v) for v in a], _dp(r), _dp(Jq), _dp(Jt))
    return r, Jq, Jt


LEVMAR_OPTS = (1e-3, np.finfo(np.float64).eps, np.finfo(np.float64).eps, np.finfo(np.float64).eps)   # lm.hpp:43-46,61


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def homography_lm(x1, y1, x2, y2, h, itmax=100, opts=LEVMAR_OPTS, engine="restatement"):
    """HomographyEstimator::solveLM (homography.cpp:218-236) on one problem: h[8] -> (ret, h, info[10]).
    engine "restatement" = this repo's levmar restatement, "reference" = the reference's dlevmar_der compiled
    into oracle/_ref (None is returned when that library is absent, as on the GPU box if it did not travel)."""
    x1, y1, x2, y2 = (np.ascontiguousarray(a, dtype=np.float32) for a in (x1, y1, x2, y2))
    hh = np.array(h, dtype=np.float64).reshape(-1)[:8].copy()
    o = np.array(opts, dtype=np.float64); info = np.zeros(10)
    fn = None
    if engine == "reference":
        R = levmar_ref()
        if R is None:
            return None
        fn = C.cast(R.dlevmar_der, C.c_void_p)
    ret = lib().orc_homography_lm(fn, len(x1), _fp(x1), _fp(y1), _fp(x2), _fp(y2), _dp(hh), int(itmax), _dp(o), _dp(info))
    return ret, hh, info


def homography_dlt(x1, y1, x2, y2):
    """HomographyEstimator::solve (homography.cpp:46-93) on one problem: (sweeps of the SVD or -1, H[9] row-major)"""
    x1, y1, x2, y2 = (np.ascontiguousarray(a, dtype=np.float32) for a in (x1, y1, x2, y2))
    H = np.zeros(9)
    ret = lib().orc_homography_dlt(len(x1), _fp(x1), _fp(y1), _fp(x2), _fp(y2), _dp(H))
    return ret, H


def homography_dlt_matrix(x1, y1, x2, y2):
    """the constraint matrix (2n x 9) over the normalised points, and the two normalising transforms"""
    x1, y1, x2, y2 = (np.ascontiguousarray(a, dtype=np.float32) for a in (x1, y1, x2, y2))
    A = np.zeros((2 * len(x1), 9)); T1 = np.zeros(9); T2 = np.zeros(9)
    lib().orc_homography_dlt_matrix(len(x1), _fp(x1), _fp(y1), _fp(x2), _fp(y2), _dp(A), _dp(T1), _dp(T2))
    return A, T1.reshape(3, 3), T2.reshape(3, 3)


def null_vector9(A):
    A = np.array(A, dtype=np.float64, order="C")
    v = np.zeros(9)
    sweeps = lib().orc_null_vector9(A.shape[0], _dp(A), _dp(v))
    return v, sweeps


def homography_residuals(x1, y1, x2, y2, H):
    """HomographyEstimator::residuals (homography.cpp:95-110)"""
    x1, y1, x2, y2 = (np.ascontiguousarray(a, dtype=np.float32) for a in (x1, y1, x2, y2))
    H = np.ascontiguousarray(np.asarray(H, dtype=np.float64).reshape(-1))
    res = np.zeros(len(x1))
    lib().orc_homography_residuals(len(x1), _fp(x1), _fp(y1), _fp(x2), _fp(y2), _dp(H), _dp(res))
    return res


def ransac_params(**kw):
    """cvx::util::RANSAC::Parameters (ransac.hpp:12-23) with the defaults solveRansac runs on"""
    p = RansacParams()
    lib().orc_ransac_defaults(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def rng_sample4(N, draw):
    """RNG::sample(4, N, .) (rng.hpp:62-75) from the four integers its distributions return"""
    d = (C.c_int * 4)(*[int(x) for x in draw])
    out = (C.c_int * 4)()
    lib().orc_rng_sample4(int(N), d, out)
    return list(out)


def homography_ransac(x1, y1, x2, y2, subsets, params=None):
    """HomographyEstimator::solveRansac (homography.cpp:151-156) on a given sequence of subsets [trials][4]:
    (ok, H[9] row-major or None, inlier indices, trials run)"""
    x1, y1, x2, y2 = (np.ascontiguousarray(a, dtype=np.float32) for a in (x1, y1, x2, y2))
    p = params or ransac_params()
    sub = np.ascontiguousarray(subsets, dtype=np.int32).reshape(-1, 4)
    assert sub.shape[0] >= p.max_num_trials
    n = len(x1)
    H = np.zeros(9)
    inl = np.zeros(max(n, 1), dtype=np.int32)
    ni, tr = C.c_int(0), C.c_int(0)
    ok = lib().orc_homography_ransac(n, _fp(x1), _fp(y1), _fp(x2), _fp(y2), C.byref(p), sub.ctypes.data_as(C.POINTER(C.c_int)),
                                     _dp(H), inl.ctypes.data_as(C.POINTER(C.c_int)), C.byref(ni), C.byref(tr))
    return bool(ok), (H if ok else None), inl[:ni.value].copy(), tr.value

