"""numpy-facing wrappers of the two device handles of libcvx_amd.so, plus the synthetic
workloads of SURVEY.md section 8(d).  Thin: every method is one C-ABI call."""
import ctypes as C

import numpy as np

from . import capi
from .capi import check, f64, i32, lib, pd, pi


class Calib:
    """cvx_calib handle: the LM inner loop of CameraCalibration::refine
    (/root/reference/modules/calib/src/calibration.cpp:354-464) on one GPU."""

    def __init__(self, cam_of_view, view_off, point_id, obs_x, obs_y, n_cameras, n_points,
                 fix_k3=True, fix_intrinsics=False, fix_points=False, device=0, stream=None, prepare_inner=False):
        self.cam_of_view = i32(cam_of_view); self.view_off = i32(view_off); self.point_id = i32(point_id)
        # float32 observations (cv::Point2f as the reference holds them) go up as they are: half the upload
        f32 = getattr(obs_x, "dtype", None) == np.float32 and getattr(obs_y, "dtype", None) == np.float32
        self.obs_x = np.ascontiguousarray(obs_x) if f32 else f64(obs_x)
        self.obs_y = np.ascontiguousarray(obs_y) if f32 else f64(obs_y)
        self.C, self.P = int(n_cameras), int(n_points)
        self.V, self.N = len(self.cam_of_view), len(self.point_id)
        as_f = lambda a: a.ctypes.data_as(capi.fp)
        d = capi.CalibDesc(self.C, self.V, self.P, self.N, pi(self.cam_of_view), pi(self.view_off),
                           pi(self.point_id), None if f32 else pd(self.obs_x), None if f32 else pd(self.obs_y),
                           int(fix_k3), int(fix_intrinsics), int(fix_points), int(device),
                           C.c_void_p(stream) if stream else None, as_f(self.obs_x) if f32 else None,
                           as_f(self.obs_y) if f32 else None, int(prepare_inner))
        self.h = C.c_void_p()
        check(lib().cvx_calib_create(C.byref(d), C.byref(self.h)), "cvx_calib_create")
        self.ns = lib().cvx_calib_num_shared(self.h)
        self._hook = None

    def close(self):
        if self.h:
            lib().cvx_calib_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, intr, poses, points):
        intr, poses, points = f64(intr), f64(poses), f64(points)
        assert intr.size == 9 * self.C and poses.size == 6 * self.V and points.size == 3 * self.P
        check(lib().cvx_calib_set_params(self.h, pd(intr), pd(poses), pd(points)), "set_params")

    def prepare_inner_iterations(self):
        check(lib().cvx_calib_prepare_inner_iterations(self.h), "prepare_inner_iterations")

    def get_params(self):
        intr = np.zeros((self.C, 9)); poses = np.zeros((self.V, 6)); points = np.zeros((self.P, 3))
        check(lib().cvx_calib_get_params(self.h, pd(intr), pd(poses), pd(points)), "get_params")
        return intr, poses, points

    def cost(self):
        c = C.c_double()
        check(lib().cvx_calib_cost(self.h, C.byref(c)), "cost")
        return c.value

    def inner_pass(self):
        """one coordinate-descent pass of Ceres' inner iterations on the current parameters; returns the cost"""
        c = C.c_double()
        check(lib().cvx_calib_inner_pass(self.h, C.byref(c)), "inner_pass")
        return c.value

    def evaluate(self):
        r = np.zeros((self.N, 2)); Jc = np.zeros((self.N, 2, 9)); Jp = np.zeros((self.N, 2, 6)); Jx = np.zeros((self.N, 2, 3))
        c = C.c_double()
        check(lib().cvx_calib_evaluate(self.h, C.byref(c), pd(r), pd(Jc), pd(Jp), pd(Jx)), "evaluate")
        return c.value, r, Jc, Jp, Jx

    def evaluate_device(self):
        p = C.c_void_p()
        check(lib().cvx_calib_evaluate_device(self.h, C.byref(p)), "evaluate_device")
        return p.value

    def init_scaling(self, enable=True):
        check(lib().cvx_calib_init_scaling(self.h, int(enable)), "init_scaling")

    def linear_step(self, inv_radius=0.0, mu=0.0, min_diag=0.0, max_diag=0.0):
        d = capi.Damping(inv_radius, mu, min_diag, max_diag)
        ns = max(self.ns, 1)
        S = np.zeros((ns, ns)); g = np.zeros(ns); ds = np.zeros(ns); dpz = np.zeros((self.V, 6))
        cost = C.c_double(); cand = C.c_double(); mc = C.c_double()
        check(lib().cvx_calib_linear_step(self.h, C.byref(d), pd(S), pd(g), pd(ds), pd(dpz), C.byref(cost),
                                          C.byref(cand), C.byref(mc)), "linear_step")
        return {"S": S[:self.ns, :self.ns], "gred": g[:self.ns], "delta_shared": ds[:self.ns], "delta_poses": dpz,
                "cost": cost.value, "cand_cost": cand.value, "model_change": mc.value}

    def set_exchange(self, fn, rank, nranks, device_buffers, ctx=None):
        """fn: a capi.ALLREDUCE_FN instance, or a raw function address (e.g. cvx_rccl_allreduce)."""
        self._hook = fn
        addr = C.cast(fn, C.c_void_p) if not isinstance(fn, int) else C.c_void_p(fn)
        check(lib().cvx_calib_set_exchange(self.h, addr, ctx, rank, nranks, int(device_buffers)), "set_exchange")

    def solve(self, options=None):
        o = options if options is not None else capi.options("refine")
        s = capi.LMSummary()
        check(lib().cvx_calib_solve(self.h, C.byref(o), C.byref(s)), "solve")
        return s

    def enable_timing(self, on=True):
        check(lib().cvx_calib_enable_timing(self.h, int(on)), "enable_timing")

    def phase_times(self):
        ms = np.zeros(16)
        check(lib().cvx_calib_phase_times(self.h, pd(ms)), "phase_times")
        names = ["normal_eq", "schur", "assemble", "backsubst", "cost", "solve"]
        return {n: (ms[i], int(ms[8 + i])) for i, n in enumerate(names)}

    def time_schur(self, reps=20):
        ms = C.c_double()
        check(lib().cvx_calib_time_schur(self.h, int(reps), C.byref(ms)), "cvx_calib_time_schur")
        return ms.value

    def speculation_counts(self):
        """(speculated steps the loop used, speculated steps it switched off) since the handle was made"""
        g = (C.c_int64 * 2)()
        check(lib().cvx_calib_speculation_counts(self.h, g), "speculation counts")
        return int(g[0]), int(g[1])

    def geometry(self):
        g = (C.c_int64 * 8)()
        check(lib().cvx_calib_geometry(self.h, g), "geometry")
        return {"ZC": g[0], "ns": g[1], "n_groups": g[2], "n_chunks": g[3], "tile_pairs": g[4], "threads": g[5],
                "z_rows": g[6], "tile_columns": g[7]}


def inner_ordering(cam_of_view, view_off, point_id, n_cameras, n_points, fix_intrinsics=False, fix_points=False):
    """(group per block [cameras | points | poses], number of sets) of the automatic inner-iteration ordering"""
    cam_of_view, view_off, point_id = i32(cam_of_view), i32(view_off), i32(point_id)
    V = len(cam_of_view)
    g = np.zeros(int(n_cameras) + int(n_points) + V, dtype=np.int32); n = C.c_int()
    check(lib().cvx_calib_inner_ordering(int(n_cameras), V, int(n_points), pi(cam_of_view), pi(view_off), pi(point_id),
                                         int(fix_intrinsics), int(fix_points), pi(g), C.byref(n)), "inner_ordering")
    return g, n.value


class HandEye:
    """cvx_handeye handle: HandEyeCalibration::solveNonLinear
    (/root/reference/modules/calib/src/handeye.cpp:272-333) on one GPU."""

    def __init__(self, qa, ta, qb, tb, device=0, stream=None):
        self.qa, self.ta, self.qb, self.tb = f64(qa), f64(ta), f64(qb), f64(tb)
        self.n = self.qa.size // 4
        self.h = C.c_void_p()
        check(lib().cvx_handeye_create(self.n, pd(self.qa), pd(self.ta), pd(self.qb), pd(self.tb), int(device),
                                       C.c_void_p(stream) if stream else None, C.byref(self.h)), "cvx_handeye_create")
        self._hook = None

    def close(self):
        if self.h:
            lib().cvx_handeye_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def cost(self, p):
        p = f64(p); c = C.c_double()
        check(lib().cvx_handeye_cost(self.h, pd(p), C.byref(c)), "handeye_cost")
        return c.value

    def normal_eq(self, p):
        p = f64(p); H = np.zeros((6, 6)); g = np.zeros(6); c = C.c_double()
        check(lib().cvx_handeye_normal_eq(self.h, pd(p), pd(H), pd(g), C.byref(c)), "handeye_normal_eq")
        return H, g, c.value

    def evaluate(self, p):
        p = f64(p)
        r = np.zeros((self.n, 12)); Jq = np.zeros((self.n, 12, 4)); Jt = np.zeros((self.n, 12, 3))
        check(lib().cvx_handeye_evaluate(self.h, pd(p), pd(r), pd(Jq), pd(Jt)), "handeye_evaluate")
        return r, Jq, Jt

    def set_exchange(self, fn, rank, nranks, device_buffers, ctx=None):
        self._hook = fn
        addr = C.cast(fn, C.c_void_p) if not isinstance(fn, int) else C.c_void_p(fn)
        check(lib().cvx_handeye_set_exchange(self.h, addr, ctx, rank, nranks, int(device_buffers)), "set_exchange")

    def solve(self, p, options=None):
        o = options if options is not None else capi.options("handeye")
        p = f64(p).copy(); s = capi.LMSummary()
        check(lib().cvx_handeye_solve(self.h, pd(p), C.byref(o), C.byref(s)), "handeye_solve")
        return p, s


# ---------------------------------------------------------------------------------------
# host-only helpers (no device)
# ---------------------------------------------------------------------------------------
def synth_calib(n_cameras, views_per_camera, board_w, board_h, seed, tile=25.0, pixel_noise=0.1, ragged=0):
    """SURVEY.md 8(d) synthetic calibration workload -> dict of numpy arrays."""
    s = capi.SynthCalibSpec(n_cameras, views_per_camera, board_w, board_h, tile, seed, pixel_noise, ragged)
    N = lib().cvx_synth_calib_count(C.byref(s))
    V, P, Cn = n_cameras * views_per_camera, board_w * board_h, n_cameras
    out = {"cam_of_view": np.zeros(V, np.int32), "view_off": np.zeros(V + 1, np.int32),
           "point_id": np.zeros(N, np.int32), "obs_x": np.zeros(N), "obs_y": np.zeros(N),
           "intr_gt": np.zeros((Cn, 9)), "poses_gt": np.zeros((V, 6)), "points_gt": np.zeros((P, 3)),
           "intr_init": np.zeros((Cn, 9)), "poses_init": np.zeros((V, 6)), "points_init": np.zeros((P, 3))}
    check(lib().cvx_synth_calib(C.byref(s), pi(out["cam_of_view"]), pi(out["view_off"]), pi(out["point_id"]),
                                pd(out["obs_x"]), pd(out["obs_y"]), pd(out["intr_gt"]), pd(out["poses_gt"]),
                                pd(out["points_gt"]), pd(out["intr_init"]), pd(out["poses_init"]),
                                pd(out["points_init"])), "synth_calib")
    out.update(C=Cn, V=V, P=P, N=int(N))
    return out


def synth_handeye(n_poses, seed, rot_noise=1e-3, trans_noise=1e-4, moving=False):
    G = np.zeros((n_poses, 16)); T = np.zeros((n_poses, 16)); xg = np.zeros(7); x0 = np.zeros(7)
    check(lib().cvx_synth_handeye(n_poses, seed, rot_noise, trans_noise, int(moving), pd(G), pd(T), pd(xg), pd(x0)),
          "synth_handeye")
    return G, T, xg, x0


def handeye_pairs(G, T, moving=False):
    """Motion pairs of solveFixed / solveMoving (handeye.cpp:681-693, 738-746) as 4x4 column-major."""
    G, T = f64(G), f64(T)
    n = G.shape[0]
    f = lib().cvx_handeye_pairs_moving if moving else lib().cvx_handeye_pairs_fixed
    cnt = C.c_int()
    check(f(n, pd(G), pd(T), None, None, 0, C.byref(cnt)), "pairs count")
    A = np.zeros((max(cnt.value, 1), 16)); B = np.zeros((max(cnt.value, 1), 16))
    check(f(n, pd(G), pd(T), pd(A), pd(B), cnt.value, C.byref(cnt)), "pairs")
    return A[:cnt.value], B[:cnt.value]


def handeye_linear(A, B, method=0):
    """Closed-form initial X (4x4 column-major flat) from motion pairs: 0 dual quaternion, 1 Tsai, 2 Horaud
    (handeye.cpp:335-599)."""
    A, B = f64(A), f64(B)
    X = np.zeros(16)
    check(lib().cvx_handeye_linear(int(method), A.shape[0], pd(A), pd(B), pd(X)), "handeye_linear")
    return X


def pairs_to_quat_trans(A, B):
    n = A.shape[0]
    qa = np.zeros((n, 4)); ta = np.zeros((n, 3)); qb = np.zeros((n, 4)); tb = np.zeros((n, 3))
    for i in range(n):
        a = f64(A[i]); b = f64(B[i])
        lib().cvx_affine_to_quat_trans(pd(a), pd(qa[i]), pd(ta[i]))
        lib().cvx_affine_to_quat_trans(pd(b), pd(qb[i]), pd(tb[i]))
    return qa, ta, qb, tb


def partition_views(view_off, nranks):
    view_off = i32(view_off)
    first = np.zeros(nranks + 1, np.int32)
    check(lib().cvx_partition_views(len(view_off) - 1, pi(view_off), nranks, pi(first)), "partition_views")
    return first


def shard_calib(data, first, rank):
    """The slice of a synth_calib-style dict that rank `rank` owns (views [first[rank], first[rank+1]))."""
    v0, v1 = int(first[rank]), int(first[rank + 1])
    o0, o1 = int(data["view_off"][v0]), int(data["view_off"][v1])
    out = dict(data)
    out["cam_of_view"] = data["cam_of_view"][v0:v1].copy()
    out["view_off"] = (data["view_off"][v0:v1 + 1] - o0).astype(np.int32)
    for k in ("point_id", "obs_x", "obs_y"):
        out[k] = data[k][o0:o1].copy()
    for k in ("poses_gt", "poses_init"):
        out[k] = data[k][v0:v1].copy()
    out.update(V=v1 - v0, N=o1 - o0)
    return out


class HomographyBatch:
    """cvx_homography handle: HomographyEstimator::solveLM (homography.cpp:218-236) for many views at once.
    Point pairs are float32 (cv::Point2f), problem i owning [off[i], off[i+1])."""

    def __init__(self, off, x1, y1, x2, y2, device=0, stream=None):
        self.off = i32(off)
        self.pts = [np.ascontiguousarray(a, dtype=np.float32) for a in (x1, y1, x2, y2)]
        self.n = len(self.off) - 1
        fp = lambda a: a.ctypes.data_as(capi.fp)
        self.h = C.c_void_p()
        check(lib().cvx_homography_create(self.n, pi(self.off), *[fp(a) for a in self.pts], int(device),
                                          C.c_void_p(stream) if stream else None, C.byref(self.h)),
              "cvx_homography_create")

    def close(self):
        if self.h:
            lib().cvx_homography_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve(self, H, max_iter=100, opts=None):
        """H [n, 9] initial estimates -> (H refined, info [n, 10], ret [n])"""
        H = f64(np.array(H, dtype=np.float64).reshape(self.n, 9))
        info = np.zeros((self.n, 10)); ret = np.zeros(self.n, np.int32)
        o = f64(opts) if opts is not None else None
        check(lib().cvx_homography_solve(self.h, pd(H), int(max_iter), pd(o) if o is not None else None, pd(info),
                                         pi(ret)), "cvx_homography_solve")
        return H, info, ret

    def set(self, H):
        H = f64(np.array(H, dtype=np.float64).reshape(self.n, 9))
        check(lib().cvx_homography_set(self.h, pd(H)), "cvx_homography_set")

    def run(self, max_iter=100):
        check(lib().cvx_homography_run(self.h, int(max_iter), None), "cvx_homography_run")

    def get(self):
        H = np.zeros((self.n, 9)); info = np.zeros((self.n, 10)); ret = np.zeros(self.n, np.int32)
        check(lib().cvx_homography_get(self.h, pd(H), pd(info), pi(ret)), "cvx_homography_get")
        return H, info, ret

    def dlt(self, fetch=True):
        """HomographyEstimator::solve (homography.cpp:46-93) of every problem; the estimates stay on the device as the
        start of run()"""
        H = np.zeros((self.n, 9)) if fetch else None
        check(lib().cvx_homography_dlt(self.h, pd(H) if fetch else None), "cvx_homography_dlt")
        return H

    def residuals(self, H=None):
        """HomographyEstimator::residuals (homography.cpp:95-110) per point pair; H None = the handle's estimates"""
        res = np.zeros(int(self.off[-1]))
        if H is not None:
            H = np.ascontiguousarray(H, dtype=np.float64).reshape(self.n, 9)
        check(lib().cvx_homography_residuals(self.h, pd(H) if H is not None else None, pd(res)), "cvx_homography_residuals")
        return res


    def ransac(self, subsets=None, seed=0, params=None, want_subsets=False, keep_inliers=False):
        """HomographyEstimator::solveRansac (homography.cpp:151-156) of every problem.  subsets [n, max_num_trials, 4] = what
        RNG::sample draws trial after trial, or None: drawn on the device from `seed`.
        -> dict(H [n, 9], n_inliers [n], mask [total] uint8, trials_run [n], subsets (if asked), inliers = a
        HomographyBatch of the inliers with the refit as estimates (if asked: run() on it is solveLM over the inliers))"""
        p = capi.RansacParams()
        lib().cvx_ransac_default_params(C.byref(p))
        if params:
            for k, v in params.items():
                setattr(p, k, v)
        T = p.max_num_trials
        sub = None
        if subsets is not None:
            sub = i32(np.asarray(subsets).reshape(self.n, T, 4))
        H = np.zeros((self.n, 9)); ni = np.zeros(self.n, np.int32); tr = np.zeros(self.n, np.int32)
        mask = np.zeros(max(int(self.off[-1]), 1), np.uint8)
        used = np.zeros((self.n, T, 4), np.int32) if want_subsets else None
        nb = C.c_void_p()
        check(lib().cvx_homography_ransac(self.h, C.byref(p), pi(sub) if sub is not None else None, int(seed), pd(H), pi(ni),
                                          mask.ctypes.data_as(C.POINTER(C.c_ubyte)), pi(tr), pi(used) if want_subsets else None,
                                          C.byref(nb) if keep_inliers else None), "cvx_homography_ransac")
        out = dict(H=H, n_inliers=ni, mask=mask[:int(self.off[-1])], trials_run=tr)
        if want_subsets:
            out["subsets"] = used
        if keep_inliers:
            b = HomographyBatch.__new__(HomographyBatch)
            b.n = self.n
            b.off = np.concatenate([[0], np.cumsum(ni)]).astype(np.int32)
            b.pts = None
            b.h = nb
            out["inliers"] = b
        return out


class ToolBA:
    """cvx_toolba handle: bundleAdjustment of tools/hand_eye_registration.cpp:144-238 on one GPU."""

    def __init__(self, view_off, point_id, obs_x, obs_y, trans, intr, n_points, huber=1.0, device=0, stream=None):
        self.view_off = i32(view_off); self.point_id = i32(point_id)
        self.obs_x = f64(obs_x); self.obs_y = f64(obs_y)
        self.trans = f64(np.asarray(trans, dtype=np.float64).reshape(-1, 12))
        self.V, self.N, self.P = len(self.view_off) - 1, len(self.point_id), int(n_points)
        d = capi.ToolBADesc(self.V, self.P, self.N, pi(self.view_off), pi(self.point_id), pd(self.obs_x), pd(self.obs_y),
                            pd(self.trans), (C.c_double * 9)(*[float(v) for v in intr]), float(huber), int(device),
                            C.c_void_p(stream) if stream else None)
        self.h = C.c_void_p()
        check(lib().cvx_toolba_create(C.byref(d), C.byref(self.h)), "cvx_toolba_create")

    def close(self):
        if self.h:
            lib().cvx_toolba_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_exchange(self, fn, rank, nranks, device_buffers, ctx=None):
        self._hook = fn
        addr = C.cast(fn, C.c_void_p) if not isinstance(fn, int) else C.c_void_p(fn)
        check(lib().cvx_toolba_set_exchange(self.h, addr, ctx, rank, nranks, int(device_buffers)), "set_exchange")

    def cost(self, p, points):
        p, points = f64(p), f64(points); c = C.c_double()
        check(lib().cvx_toolba_cost(self.h, pd(p), pd(points), C.byref(c)), "cvx_toolba_cost")
        return c.value

    def normal_eq(self, p, points):
        """-> Hpp[6,6], gp[6], W[P,6,3], U[P,3,3], gx[P,3], cost"""
        p, points = f64(p), f64(points)
        Hpp = np.zeros((6, 6)); gp = np.zeros(6); W = np.zeros((self.P, 6, 3)); U = np.zeros((self.P, 3, 3))
        gx = np.zeros((self.P, 3)); c = C.c_double()
        check(lib().cvx_toolba_normal_eq(self.h, pd(p), pd(points), pd(Hpp), pd(gp), pd(W), pd(U), pd(gx), C.byref(c)),
              "cvx_toolba_normal_eq")
        return Hpp, gp, W, U, gx, c.value

    def inner_pass(self, p, points):
        """one pass of Ceres' inner iterations over the blocks q, t and every point -> (p, points, cost)"""
        p, points = f64(p).copy(), f64(points).copy(); c = C.c_double()
        check(lib().cvx_toolba_inner_pass(self.h, pd(p), pd(points), C.byref(c)), "cvx_toolba_inner_pass")
        return p, points, c.value

    def solve(self, p, points, options):
        p, points = f64(p).copy(), f64(points).copy(); s = capi.LMSummary()
        check(lib().cvx_toolba_solve(self.h, pd(p), pd(points), C.byref(options), C.byref(s)), "cvx_toolba_solve")
        return p, points, s
