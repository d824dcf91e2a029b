"""Shared helpers of the tests (not a test module)."""
import importlib.util
import os

import numpy as np

_g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py")


def _load_generator():
    spec = importlib.util.spec_from_file_location("make_golden", _g)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


_gen = _load_generator()
# name -> (f, jac) of the classic problems, shared with the fixture generator
KAT_FUNCS = {k: (v[0], v[1]) for k, v in _gen.PROBLEMS.items()}


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300)) if a.size else 0.0


def param_block_errs(got, exp):
    """Per-block relative error of a calibration's parameters, got / exp = (intr[C, 9], poses[V, 6], points[P, 3]):
    max |a - b| over the block / max |b| over the block, for the blocks whose entries share a unit and a scale --
    focal lengths + principal point (pixels), distortion coefficients (calibration.cpp:219-229 order: k1 k2 k3 p1 p2),
    rotations (rad), translations (board units), board points.  A whole-array norm would measure p1 ~ 1e-3 against
    fx ~ 1e3 and the rotations against the translations.  'dist_cols' holds the distortion block column by column
    (each coefficient over all cameras against its own largest value)."""
    (ia, pa, xa), (ib, pb, xb) = [tuple(np.asarray(t, dtype=np.float64) for t in x) for x in (got, exp)]
    ia, ib = ia.reshape(-1, 9), ib.reshape(-1, 9)
    pa, pb = pa.reshape(-1, 6), pb.reshape(-1, 6)
    e = {"focal_pp": rel_err(ia[:, :4], ib[:, :4]), "distortion": rel_err(ia[:, 4:], ib[:, 4:]),
         "rotation": rel_err(pa[:, :3], pb[:, :3]), "translation": rel_err(pa[:, 3:], pb[:, 3:]),
         "points": rel_err(xa, xb)}
    e["dist_cols"] = {n: rel_err(ia[:, 4 + k], ib[:, 4 + k]) for k, n in enumerate(("k1", "k2", "k3", "p1", "p2"))
                      if np.any(ib[:, 4 + k] != 0.0)}
    return e


# north_star's bar for the final intrinsics / extrinsics
PARAM_TOL_NORTH_STAR = {"focal_pp": 1e-9, "distortion": 1e-9, "rotation": 1e-9, "translation": 1e-9, "points": 1e-9}


def assert_params_close(got, exp, tol=None, what=""):
    """Every block of param_block_errs within its tolerance: tol maps block name -> bound (missing names: 1e-9,
    north_star's figure for intrinsics / extrinsics); a key of dist_cols (k1 .. p2) bounds that column on its own."""
    t = dict(PARAM_TOL_NORTH_STAR)
    t.update(tol or {})
    e = param_block_errs(got, exp)
    for k, bound in t.items():
        if k in e:
            v = e[k]
        elif k in e["dist_cols"]:
            v = e["dist_cols"][k]
        else:
            continue                    # a column that is zero everywhere (k3 fixed at 0)
        assert v <= bound, "%s block %s: %.3g > %.3g (%s)" % (what, k, v, bound, e)
    return e


def calib_from_fixture(g):
    """functor_mp.json 'calib' block -> arrays of a calibration problem (8 views x 4 points, 2 cameras)."""
    V, P = len(g["poses"]), len(g["points"])
    obs = sorted(g["obs"], key=lambda o: (o["view"], o["point"]))
    view_off = np.arange(V + 1, dtype=np.int32) * P
    point_id = np.array([o["point"] for o in obs], dtype=np.int32)
    ox = np.array([o["ox"] for o in obs]); oy = np.array([o["oy"] for o in obs])
    r = np.array([[float(s) for s in o["r"]] for o in obs])
    J = np.array([[[float(s) for s in row] for row in o["J"]] for o in obs])
    return {"C": len(g["cameras"]), "V": V, "P": P, "cam_of_view": np.array(g["cam_of_view"], dtype=np.int32),
            "view_off": view_off, "point_id": point_id, "obs_x": ox, "obs_y": oy,
            "intr": np.array(g["cameras"]), "poses": np.array(g["poses"]), "points": np.array(g["points"]),
            "r": r, "Jc": J[:, :, :9], "Jp": J[:, :, 9:15], "Jx": J[:, :, 15:]}


def homography_problem(rng, N, noise=0.3, perturb=1e-2):
    """N board points (12 x 9 grid of 25 mm tiles, pattern.hpp:53-55) seen through a random homography with
    pixel noise, float32 like cv::Point2f, and a perturbed initial estimate normalised to H(2,2) = 1."""
    bx, by = np.meshgrid(np.arange(12) * 25.0, np.arange(9) * 25.0)
    idx = rng.permutation(108)[:N] if N <= 108 else rng.integers(0, 108, N)
    x1 = bx.ravel()[idx].astype(np.float32); y1 = by.ravel()[idx].astype(np.float32)
    H = np.array([[3.5, 0.3, 400.0], [-0.2, 3.6, 300.0], [2e-4, -3e-4, 1.0]])
    H = H + rng.normal(0, 1, (3, 3)) * np.array([[0.1, 0.1, 20], [0.1, 0.1, 20], [1e-4, 1e-4, 0]])
    q = H @ np.stack([x1.astype(np.float64), y1.astype(np.float64), np.ones(N)])
    x2 = (q[0] / q[2] + rng.normal(0, noise, N)).astype(np.float32)
    y2 = (q[1] / q[2] + rng.normal(0, noise, N)).astype(np.float32)
    H0 = H * (1 + rng.normal(0, perturb, (3, 3)))
    H0 /= H0[2, 2]
    return x1, y1, x2, y2, H0.reshape(-1)


def homography_batch(rng, sizes, **kw):
    """concatenated problems -> off, x1, y1, x2, y2, H0[n, 9]"""
    parts = [homography_problem(rng, int(n), **kw) for n in sizes]
    off = np.concatenate([[0], np.cumsum([len(p[0]) for p in parts])]).astype(np.int32)
    cat = lambda k: np.concatenate([p[k] for p in parts]) if parts else np.zeros(0, np.float32)
    return off, cat(0), cat(1), cat(2), cat(3), np.stack([p[4] for p in parts])


def _rot(axis, angle):
    axis = np.asarray(axis, float); axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def _quat_of(R):
    w = np.sqrt(max(1e-30, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    return np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])


TOOLBA_INTR = np.array([1000.0, 1005.0, 640.0, 480.0, -0.12, 0.05, 0.01, 1e-3, -1.5e-3])   # fx fy cx cy k1 k2 k3 p1 p2


def toolba_problem(rng, V, bw=9, bh=6, tile=0.025, noise=0.3, outliers=0.03, ragged=0):
    """Synthetic input of the tool-level bundle adjustment (hand_eye_registration.cpp:144-238): one camera pose
    (base -> camera), V views each with a known rigid pre-transform (board -> base), board points in metres,
    float32 pixel observations with noise and a few gross outliers for the Huber loss to act on."""
    P = bw * bh
    gx, gy = np.meshgrid(np.arange(bw) * tile, np.arange(bh) * tile)
    X = np.stack([gx.ravel(), gy.ravel(), np.zeros(P)], 1).astype(np.float32).astype(np.float64)
    Rg = _rot([1, 2, 3], 0.4); tg = np.array([0.05, -0.03, 0.12])
    Tg = np.eye(4); Tg[:3, :3] = Rg; Tg[:3, 3] = tg                       # camera <- base
    Tgi = np.linalg.inv(Tg)
    fx, fy, cx, cy, k1, k2, k3, p1, p2 = TOOLBA_INTR
    trans = np.zeros((V, 12)); view_off = [0]; pid = []; ox = []; oy = []
    for v in range(V):
        Tb = np.eye(4)                                                    # camera <- board
        Tb[:3, :3] = _rot(rng.normal(size=3), rng.uniform(0, 0.6))
        Tb[:3, 3] = [rng.uniform(-0.15, 0.15) - gx.mean(), rng.uniform(-0.1, 0.1) - gy.mean(), rng.uniform(0.5, 0.9)]
        trans[v] = (Tgi @ Tb)[:3].reshape(-1)
        ids = np.arange(P)
        if ragged:
            ids = ids[rng.uniform(size=P) > 1.0 / ragged]
        xc = (Tb[:3, :3] @ X[ids].T).T + Tb[:3, 3]
        xn, yn = xc[:, 0] / xc[:, 2], xc[:, 1] / xc[:, 2]
        r2 = xn * xn + yn * yn
        rc = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
        u = fx * (xn * rc + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)) + cx + rng.normal(0, noise, len(ids))
        w = fy * (yn * rc + 2 * p2 * xn * yn + p1 * (r2 + 2 * yn * yn)) + cy + rng.normal(0, noise, len(ids))
        bad = rng.uniform(size=len(ids)) < outliers
        u = u + bad * rng.normal(0, 20, len(ids)); w = w + bad * rng.normal(0, 20, len(ids))
        pid.append(ids); ox.append(u.astype(np.float32)); oy.append(w.astype(np.float32))
        view_off.append(view_off[-1] + len(ids))
    q_gt = _quat_of(Rg)
    R0 = _rot(rng.normal(size=3), 0.02) @ Rg
    p0 = np.concatenate([_quat_of(R0), tg + rng.normal(0, 0.01, 3)])
    return {"V": V, "P": P, "view_off": np.array(view_off, np.int32), "point_id": np.concatenate(pid).astype(np.int32),
            "obs_x": np.concatenate(ox).astype(np.float64), "obs_y": np.concatenate(oy).astype(np.float64),
            "trans": trans, "intr": TOOLBA_INTR.copy(), "points": X, "p_gt": np.concatenate([q_gt, tg]), "p0": p0}
