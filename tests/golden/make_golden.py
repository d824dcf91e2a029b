#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/.  Run in the BUILD container only
(it needs /root/reference for part A/B and mpmath for part C):

    python tests/golden/make_golden.py

Part A  levmar_kat.json      the reference's vendored levmar 2.5 (compiled from /root/reference into
                             oracle/_ref/liblevmar_ref.so by oracle/Makefile, never copied) run on its own
                             demo problems (lmdemo.c:49-230, set-ups lmdemo.c:860-961; Osborne data also at
                             modules/util/test/test_lm.cpp:16-21, stated minimum test_lm.cpp:62).
Part B  levmar_calib_trace.json, levmar_calib_fixed_points_trace.json, levmar_handeye_trace.json
                             the same levmar driven by the oracle's residual/Jacobian callbacks on BASELINE
                             config 1 (20 views of a 9x6 board) and config 2 (200 poses, moving camera).
Part C  functor_mp.json      50-digit mpmath evaluation of the two cost functors and their Jacobians
                             (forward-mode duals over mpf on the literal expressions of
                             calibration.cpp:231-311 and handeye.cpp:228-261), incl. the theta -> 0 branch of
                             AngleAxisRotatePoint and a k3 != 0 camera.
The fixtures are data only (inputs + expected outputs).
"""
import ctypes as C
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
import cvx_amd  # noqa: E402  (host-only synthetic generators; no device work)

dp = C.POINTER(C.c_double)


# ---------------------------------------------------------------------------------------------
# Part A: classic problems (formulas restated from their textbook definitions)
# ---------------------------------------------------------------------------------------------
X33 = [8.44E-1, 9.08E-1, 9.32E-1, 9.36E-1, 9.25E-1, 9.08E-1, 8.81E-1, 8.5E-1, 8.18E-1, 7.84E-1, 7.51E-1,
       7.18E-1, 6.85E-1, 6.58E-1, 6.28E-1, 6.03E-1, 5.8E-1, 5.58E-1, 5.38E-1, 5.22E-1, 5.06E-1, 4.9E-1,
       4.78E-1, 4.67E-1, 4.57E-1, 4.48E-1, 4.38E-1, 4.31E-1, 4.24E-1, 4.2E-1, 4.14E-1, 4.11E-1, 4.06E-1]
MEYER = [34.780, 28.610, 23.650, 19.630, 16.370, 13.720, 11.540, 9.744, 8.261, 7.030, 6.005, 5.147, 4.427,
         3.820, 3.307, 2.872]


def f_ros(p, n):
    return [((1.0 - p[0]) * (1.0 - p[0]) + 105.0 * (p[1] - p[0] * p[0]) * (p[1] - p[0] * p[0]))] * n


def j_ros(p, n):
    return [[-2 + 2 * p[0] - 4 * 105.0 * (p[1] - p[0] * p[0]) * p[0], 2 * 105.0 * (p[1] - p[0] * p[0])]] * n


def f_modros(p, n):
    return [10 * (p[1] - p[0] * p[0]), 1.0 - p[0], 1e2]


def j_modros(p, n):
    return [[-20.0 * p[0], 10.0], [-1.0, 0.0], [0.0, 0.0]]


def f_powell(p, n):
    return [p[0], 10.0 * p[0] / (p[0] + 0.1) + 2 * p[1] * p[1]]


def j_powell(p, n):
    return [[1.0, 0.0], [1.0 / ((p[0] + 0.1) * (p[0] + 0.1)), 4.0 * p[1]]]


def f_meyer(p, n):
    return [p[0] * math.exp(10.0 * p[1] / (0.45 + 0.05 * i + p[2]) - 13.0) for i in range(n)]


def j_meyer(p, n):
    out = []
    for i in range(n):
        ui = 0.45 + 0.05 * i
        t = math.exp(10.0 * p[1] / (ui + p[2]) - 13.0)
        out.append([t, 10.0 * p[0] * t / (ui + p[2]), -10.0 * p[0] * p[1] * t / ((ui + p[2]) * (ui + p[2]))])
    return out


def f_osborne(p, n):
    return [p[0] + p[1] * math.exp(-p[3] * 10 * i) + p[2] * math.exp(-p[4] * 10 * i) for i in range(n)]


def j_osborne(p, n):
    out = []
    for i in range(n):
        t = 10 * i
        t1, t2 = math.exp(-p[3] * t), math.exp(-p[4] * t)
        out.append([1.0, t1, t2, -p[1] * t * t1, -p[2] * t * t2])
    return out


def f_helval(p, n):
    if p[0] < 0.0:
        th = math.atan(p[1] / p[0]) / (2.0 * math.pi) + 0.5
    elif 0.0 < p[0]:
        th = math.atan(p[1] / p[0]) / (2.0 * math.pi)
    else:
        th = 0.25 if p[1] >= 0 else -0.25
    return [10.0 * (p[2] - 10.0 * th), 10.0 * (math.sqrt(p[0] * p[0] + p[1] * p[1]) - 1.0), p[2]]


def j_helval(p, n):
    t = p[0] * p[0] + p[1] * p[1]
    return [[50.0 * p[1] / (math.pi * t), -50.0 * p[0] / (math.pi * t), 10.0],
            [10.0 * p[0] / math.sqrt(t), 10.0 * p[1] / math.sqrt(t), 0.0], [0.0, 0.0, 1.0]]


PROBLEMS = {
    "rosenbrock": (f_ros, j_ros, [-1.2, 1.0], [0.0, 0.0], [1.0, 1.0]),
    "modified_rosenbrock": (f_modros, j_modros, [-1.2, 1.0], [0.0, 0.0, 0.0], [1.0, 1.0]),
    "powell": (f_powell, j_powell, [3.0, 1.0], [0.0, 0.0], [0.0, 0.0]),
    "meyer": (f_meyer, j_meyer, [8.85, 4.0, 2.5], MEYER, [2.48, 6.18, 3.45]),
    "osborne": (f_osborne, j_osborne, [0.5, 1.5, -1.0, 1.0e-2, 2.0e-2], X33, [0.3754, 1.9358, -1.4647, 0.0129, 0.0221]),
    "helical_valley": (f_helval, j_helval, [-1.0, 0.0, 0.0], [0.0, 0.0, 0.0], [1.0, 0.0, 0.0]),
}
KAT_OPTS = [1e-3, 1e-15, 1e-15, 1e-20]   # lmdemo.c:816


def make_callbacks(f, j, x, evals):
    def func(p, hx, m, n, _):
        v = f([p[i] for i in range(m)], n)
        for i in range(n):
            hx[i] = v[i]
        evals.append(float(sum((x[i] - v[i]) ** 2 for i in range(n))))

    def jac(p, out, m, n, _):
        J = j([p[i] for i in range(m)], n)
        for i in range(n):
            for k in range(m):
                out[i * m + k] = J[i][k]
    return orc.LM_FUNC(func), orc.LM_FUNC(jac)


def run_ref_levmar(func, jac, p0, x, itmax, opts):
    ref = orc.levmar_ref()
    assert ref is not None, "oracle/_ref/liblevmar_ref.so missing: make -C oracle"
    p = np.array(p0, dtype=np.float64)
    xv = np.array(x, dtype=np.float64)
    o = np.array(list(opts) + [0.0], dtype=np.float64)
    info = np.zeros(10)
    ret = ref.dlevmar_der(func, jac, p.ctypes.data_as(dp), xv.ctypes.data_as(dp), len(p), len(xv), itmax,
                          o.ctypes.data_as(dp), info.ctypes.data_as(dp), None, None, None)
    return ret, p, info


def part_a():
    out = {}
    for name, (f, j, p0, x, stated) in PROBLEMS.items():
        evals = []
        func, jac = make_callbacks(f, j, x, evals)
        ret, p, info = run_ref_levmar(func, jac, p0, x, 1000, KAT_OPTS)
        out[name] = {"m": len(p0), "n": len(x), "p0": p0, "x": x, "opts": KAT_OPTS, "itmax": 1000,
                     "stated_minimum": stated, "ret": ret, "p": p.tolist(), "info": info.tolist(), "evals": evals}
        print("A %-20s ret=%d reason=%d |e|^2=%.6g p=%s" % (name, ret, int(info[6]), info[1], np.array2string(p, precision=6)))
    json.dump(out, open(os.path.join(HERE, "levmar_kat.json"), "w"), indent=1)


# ---------------------------------------------------------------------------------------------
# Part B: reference levmar on the two hot-path functors
# ---------------------------------------------------------------------------------------------
CALIB_SPEC = {"n_cameras": 1, "views_per_camera": 20, "board_w": 9, "board_h": 6, "seed": 20260001}
HANDEYE_SPEC = {"n_poses": 200, "seed": 20260002, "moving": True}
LMSOLVER_OPTS = [1e-3, 2.220446049250313e-16, 2.220446049250313e-16, 2.220446049250313e-16]   # lm.hpp:43-50,61


def part_b_fixed_points():
    """BASELINE config 1 with the board points held fixed: the gauge freedom of refine() (rigid motion + scale of the
    board, calibration.cpp:405 commented out) is gone, the minimiser is unique, and the dense-LU levmar run and a
    Schur-complement run walk the same path to rounding level -- the trace a GPU run can be held to tightly."""
    d = cvx_amd.synth_calib(**CALIB_SPEC)
    o = orc.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], d["C"], d["P"], fix_points=True)
    C_, V = d["C"], d["V"]
    m, n = o.nfree, 2 * d["N"]
    assert m == 8 * C_ + 6 * V
    cols = [0, 1, 2, 3, 4, 5, 7, 8]

    def unpack(p):
        intr = d["intr_init"].copy()
        for c in range(C_):
            intr[c, cols] = p[c * 8:(c + 1) * 8]
        return intr, np.array(p[C_ * 8:C_ * 8 + 6 * V]).reshape(V, 6), d["points_init"]

    evals = []

    def func(p, hx, m_, n_, _):
        r = o.eval(*unpack(np.ctypeslib.as_array(p, shape=(m_,))))[0].reshape(-1)
        np.ctypeslib.as_array(hx, shape=(n_,))[:] = r
        evals.append(float(np.dot(r, r)))

    def jac(p, out, m_, n_, _):
        J = o.dense_jacobian(*unpack(np.ctypeslib.as_array(p, shape=(m_,))))
        assert J.shape == (n_, m_)
        np.ctypeslib.as_array(out, shape=(n_ * m_,))[:] = J.reshape(-1)

    p0 = np.concatenate([d["intr_init"][:, cols].reshape(-1), d["poses_init"].reshape(-1)])
    ret, p, info = run_ref_levmar(orc.LM_FUNC(func), orc.LM_FUNC(jac), p0, np.zeros(n), 100, LMSOLVER_OPTS)
    print("B calib, points fixed: ret=%d reason=%d |e|^2 %.10g -> %.10g, %d evals" % (ret, int(info[6]), info[0], info[1], len(evals)))
    json.dump({"spec": CALIB_SPEC, "fix_points": True, "opts": LMSOLVER_OPTS, "itmax": 100, "m": m, "n": n, "ret": ret,
               "p": p.tolist(), "info": info.tolist(), "evals": evals},
              open(os.path.join(HERE, "levmar_calib_fixed_points_trace.json"), "w"), indent=1)


def part_b():
    d = cvx_amd.synth_calib(**CALIB_SPEC)
    o = orc.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], d["C"], d["P"])
    C_, V, P = d["C"], d["V"], d["P"]
    m, n = o.nfree, 2 * d["N"]
    nci = 8

    # free vector order of orc_calib_dense_jacobian: [intrinsics without k3 | poses | points]
    def unpack(p):
        intr = d["intr_init"].copy()
        cols = [0, 1, 2, 3, 4, 5, 7, 8]
        for c in range(C_):
            intr[c, cols] = p[c * nci:(c + 1) * nci]
        poses = np.array(p[C_ * nci:C_ * nci + 6 * V]).reshape(V, 6)
        pts = np.array(p[C_ * nci + 6 * V:]).reshape(P, 3)
        return intr, poses, pts

    def pack(intr, poses, pts):
        cols = [0, 1, 2, 3, 4, 5, 7, 8]
        return np.concatenate([intr[:, cols].reshape(-1), poses.reshape(-1), pts.reshape(-1)])

    evals = []

    def func(p, hx, m_, n_, _):
        pv = np.ctypeslib.as_array(p, shape=(m_,))
        r = o.eval(*unpack(pv))[0].reshape(-1)
        np.ctypeslib.as_array(hx, shape=(n_,))[:] = r
        evals.append(float(np.dot(r, r)))

    def jac(p, out, m_, n_, _):
        pv = np.ctypeslib.as_array(p, shape=(m_,))
        J = o.dense_jacobian(*unpack(pv))
        np.ctypeslib.as_array(out, shape=(n_ * m_,))[:] = J.reshape(-1)

    p0 = pack(d["intr_init"], d["poses_init"], d["points_init"])
    ret, p, info = run_ref_levmar(orc.LM_FUNC(func), orc.LM_FUNC(jac), p0, np.zeros(n), 100, LMSOLVER_OPTS)
    print("B calib: ret=%d reason=%d |e|^2 %.10g -> %.10g, %d evals" % (ret, int(info[6]), info[0], info[1], len(evals)))
    json.dump({"spec": CALIB_SPEC, "opts": LMSOLVER_OPTS, "itmax": 100, "m": m, "n": n, "ret": ret,
               "p": p.tolist(), "info": info.tolist(), "evals": evals},
              open(os.path.join(HERE, "levmar_calib_trace.json"), "w"), indent=1)

    part_b_fixed_points()

    # hand-eye, 7 global parameters (no parameterisation: this pins the functor and its global Jacobian)
    G, T, xg, x0 = cvx_amd.synth_handeye(HANDEYE_SPEC["n_poses"], HANDEYE_SPEC["seed"], moving=HANDEYE_SPEC["moving"])
    A, B = cvx_amd.handeye_pairs(G, T, moving=HANDEYE_SPEC["moving"])
    qa, ta, qb, tb = cvx_amd.pairs_to_quat_trans(A, B)
    npairs = len(A)
    evals2 = []

    def hfunc(p, hx, m_, n_, _):
        pv = np.ctypeslib.as_array(p, shape=(7,)).copy()
        out = np.ctypeslib.as_array(hx, shape=(n_,))
        for i in range(npairs):
            out[12 * i:12 * i + 12] = orc.handeye(pv[:4], pv[4:], qa[i], ta[i], qb[i], tb[i])[0]
        evals2.append(float(np.dot(out, out)))

    def hjac(p, outp, m_, n_, _):
        pv = np.ctypeslib.as_array(p, shape=(7,)).copy()
        out = np.ctypeslib.as_array(outp, shape=(n_, 7))
        for i in range(npairs):
            _, Jq, Jt = orc.handeye(pv[:4], pv[4:], qa[i], ta[i], qb[i], tb[i])
            out[12 * i:12 * i + 12, :4] = Jq
            out[12 * i:12 * i + 12, 4:] = Jt

    ret, p, info = run_ref_levmar(orc.LM_FUNC(hfunc), orc.LM_FUNC(hjac), x0, np.zeros(12 * npairs), 100, LMSOLVER_OPTS)
    print("B handeye: ret=%d reason=%d |e|^2 %.10g -> %.10g, %d evals, p=%s" % (ret, int(info[6]), info[0], info[1], len(evals2), p))
    json.dump({"spec": HANDEYE_SPEC, "opts": LMSOLVER_OPTS, "itmax": 100, "n_pairs": npairs, "ret": ret,
               "p": p.tolist(), "info": info.tolist(), "evals": evals2},
              open(os.path.join(HERE, "levmar_handeye_trace.json"), "w"), indent=1)


# ---------------------------------------------------------------------------------------------
# Part C: 50-digit functor values
# ---------------------------------------------------------------------------------------------
def part_c():
    import mpmath as mp
    mp.mp.dps = 50
    EPS = mp.mpf(2) ** -52

    class D:   # forward-mode dual over mpf
        __slots__ = ("v", "d")

        def __init__(self, v, d):
            self.v, self.d = v, d

        @staticmethod
        def const(c, n):
            return D(mp.mpf(c), [mp.mpf(0)] * n)

        @staticmethod
        def var(c, k, n):
            d = [mp.mpf(0)] * n
            d[k] = mp.mpf(1)
            return D(mp.mpf(c), d)

        def _c(self, o):
            return o if isinstance(o, D) else D.const(o, len(self.d))

        def __add__(self, o):
            o = self._c(o)
            return D(self.v + o.v, [a + b for a, b in zip(self.d, o.d)])
        __radd__ = __add__

        def __sub__(self, o):
            o = self._c(o)
            return D(self.v - o.v, [a - b for a, b in zip(self.d, o.d)])

        def __rsub__(self, o):
            return self._c(o) - self

        def __mul__(self, o):
            o = self._c(o)
            return D(self.v * o.v, [a * o.v + self.v * b for a, b in zip(self.d, o.d)])
        __rmul__ = __mul__

        def __truediv__(self, o):
            o = self._c(o)
            q = self.v / o.v
            return D(q, [(a - q * b) / o.v for a, b in zip(self.d, o.d)])

        def __rtruediv__(self, o):
            return self._c(o) / self

    def dsqrt(a):
        r = mp.sqrt(a.v)
        return D(r, [x / (2 * r) for x in a.d])

    def dsin(a):
        return D(mp.sin(a.v), [x * mp.cos(a.v) for x in a.d])

    def dcos(a):
        return D(mp.cos(a.v), [-x * mp.sin(a.v) for x in a.d])

    def rotate(w, pt):   # ceres::AngleAxisRotatePoint
        th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2]
        if th2.v > EPS:
            th = dsqrt(th2)
            c, s = dcos(th), dsin(th)
            ti = 1 / th
            a = [w[0] * ti, w[1] * ti, w[2] * ti]
            wx = [a[1] * pt[2] - a[2] * pt[1], a[2] * pt[0] - a[0] * pt[2], a[0] * pt[1] - a[1] * pt[0]]
            tmp = (a[0] * pt[0] + a[1] * pt[1] + a[2] * pt[2]) * (1 - c)
            return [pt[i] * c + wx[i] * s + a[i] * tmp for i in range(3)]
        wx = [w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]]
        return [pt[i] + wx[i] for i in range(3)]

    def reproj(intr, pose, X, ox, oy):
        n = 18
        i_ = [D.var(intr[k], k, n) for k in range(9)]
        rt = [D.var(pose[k], 9 + k, n) for k in range(6)]
        pt = [D.var(X[k], 15 + k, n) for k in range(3)]
        x = rotate(rt[:3], pt)
        x = [x[0] + rt[3], x[1] + rt[4], x[2] + rt[5]]
        xn, yn = x[0] / x[2], x[1] / x[2]
        fx, fy, cx, cy, k1, k2, k3, p1, p2 = i_
        r2 = xn * xn + yn * yn
        r4 = r2 * r2
        r6 = r4 * r2
        rc = 1 + k1 * r2 + k2 * r4 + k3 * r6
        xd = xn * rc + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
        yd = yn * rc + 2 * p2 * xn * yn + p1 * (r2 + 2 * yn * yn)
        return [fx * xd + cx - mp.mpf(ox), fy * yd + cy - mp.mpf(oy)]

    def s(v):
        return mp.nstr(v, 30, min_fixed=0, max_fixed=0)

    rng = np.random.default_rng(7)
    cams = [[1012.5, 1003.25, 641.5, 478.25, -0.12, 0.05, 0.0, 1e-3, -1.5e-3],
            [870.0, 880.5, 600.0, 500.0, -0.21, 0.09, -0.013, -2e-3, 7e-4]]
    poses = []
    for k in range(8):
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        ang = [0.0, 1e-9, 0.05, 0.3, 0.6, 1.2, 2.5, 3.1][k]   # exact zero, below the sqrt(eps) branch point, ...
        poses.append(list(ax * ang) + [float(rng.uniform(-150, 150)), float(rng.uniform(-100, 100)), float(rng.uniform(500, 900))])
    pts = [[0.0, 0.0, 0.0], [200.0, 125.0, 0.0], [-75.0, 50.0, 12.5], [37.5, -62.5, -20.0]]
    calib = {"cameras": cams, "poses": poses, "points": pts, "cam_of_view": [v % 2 for v in range(8)], "obs": []}
    for v in range(8):
        for p_ in range(4):
            ox, oy = float(np.float32(rng.uniform(100, 1180))), float(np.float32(rng.uniform(100, 860)))
            e = reproj(cams[v % 2], poses[v], pts[p_], ox, oy)
            calib["obs"].append({"view": v, "point": p_, "ox": ox, "oy": oy, "r": [s(e[0].v), s(e[1].v)],
                                 "J": [[s(x) for x in e[0].d], [s(x) for x in e[1].d]]})

    def q2R(q):
        w, x, y, z = q
        tx, ty, tz = 2 * x, 2 * y, 2 * z
        twx, twy, twz = tx * w, ty * w, tz * w
        txx, txy, txz = tx * x, ty * x, tz * x
        tyy, tyz, tzz = ty * y, tz * y, tz * z
        return [[1 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1 - (txx + tzz), tyz - twx],
                [txz - twy, tyz + twx, 1 - (txx + tyy)]]

    def mm(A, B):
        return [[A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j] for j in range(3)] for i in range(3)]

    def handeye(p, qa, ta, qb, tb):
        n = 7
        q = [D.var(p[k], k, n) for k in range(4)]
        t = [D.var(p[4 + k], 4 + k, n) for k in range(3)]
        Rx = q2R(q)
        Ra = q2R([D.const(v, n) for v in qa])
        Rb = q2R([D.const(v, n) for v in qb])
        A_, B_ = mm(Ra, Rx), mm(Rx, Rb)
        res = [A_[i][j] - B_[i][j] for i in range(3) for j in range(3)]
        for i in range(3):
            res.append(Ra[i][0] * t[0] + Ra[i][1] * t[1] + Ra[i][2] * t[2] + mp.mpf(ta[i])
                       - (Rx[i][0] * mp.mpf(tb[0]) + Rx[i][1] * mp.mpf(tb[1]) + Rx[i][2] * mp.mpf(tb[2])) - t[i])
        return res

    he = {"p": [0.97, 0.06, -0.11, 0.17, 0.05, -0.03, 0.12], "pairs": []}   # deliberately not unit norm
    for k in range(16):
        qa = rng.normal(size=4); qa /= np.linalg.norm(qa)
        qb = rng.normal(size=4); qb /= np.linalg.norm(qb)
        ta = rng.uniform(-0.5, 0.5, size=3); tb = rng.uniform(-0.5, 0.5, size=3)
        e = handeye(he["p"], qa.tolist(), ta.tolist(), qb.tolist(), tb.tolist())
        he["pairs"].append({"qa": qa.tolist(), "ta": ta.tolist(), "qb": qb.tolist(), "tb": tb.tolist(),
                            "r": [s(x.v) for x in e], "J": [[s(x) for x in r_.d] for r_ in e]})
    json.dump({"digits": 30, "calib": calib, "handeye": he}, open(os.path.join(HERE, "functor_mp.json"), "w"), indent=1)
    print("C wrote %d observations, %d pairs" % (len(calib["obs"]), len(he["pairs"])))


if __name__ == "__main__":
    which = sys.argv[1:] or ["a", "b", "c"]
    if "a" in which:
        part_a()
    if "b" in which:
        part_b()
    if "c" in which:
        part_c()
