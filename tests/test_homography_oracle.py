"""CPU: pins the oracle's homography LM restatement (oracle/cvx_oracle.c: orc_homography_*, following
homography.cpp:159-236 on levmar's dlevmar_der) against
  * tests/golden/homography_lm.json -- the REFERENCE's own levmar 2.5 run on the functor (generator:
    tests/golden/make_golden_homography.py), bit for bit, and
  * the reference levmar itself (oracle/_ref/liblevmar_ref.so) on fresh random problems when that library is
    present (it is built from /root/reference in the build container and travels to the GPU box)."""
import numpy as np
import pytest

from conftest import load_golden
from test_helpers import homography_problem

GOLD = load_golden("homography_lm.json")


def _arrays(g):
    return [np.array(g[k], dtype=np.float32) for k in ("x1", "y1", "x2", "y2")]


@pytest.mark.parametrize("name", sorted(GOLD))
def test_restatement_reproduces_reference_levmar_run(orc, name):
    g = GOLD[name]
    x1, y1, x2, y2 = _arrays(g)
    ret, h, info = orc.homography_lm(x1, y1, x2, y2, g["H0"], itmax=g["itmax"], opts=g["opts"])
    assert ret == g["ret"]
    assert h.tolist() == g["h"]                   # bit-identical parameters
    if ret >= 0:
        assert info.tolist() == g["info"]         # |e|^2 before/after, |J^T e|_inf, |Dp|^2, mu/max, counts, reason


def test_restatement_vs_reference_library_on_random_problems(orc):
    if orc.levmar_ref() is None:
        pytest.skip("oracle/_ref/liblevmar_ref.so not built (reference absent)")
    rng = np.random.default_rng(77)
    for _ in range(60):
        N = int(rng.integers(4, 200))
        x1, y1, x2, y2, H0 = homography_problem(rng, N)
        a = orc.homography_lm(x1, y1, x2, y2, H0, engine="restatement")
        b = orc.homography_lm(x1, y1, x2, y2, H0, engine="reference")
        assert a[0] == b[0] and a[1].tolist() == b[1].tolist() and a[2].tolist() == b[2].tolist()


def test_analytic_jacobian_matches_central_differences(orc):
    import ctypes as C
    rng = np.random.default_rng(3)
    x1, y1, x2, y2, H0 = homography_problem(rng, 30)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    dp = lambda a: a.ctypes.data_as(orc.c_double_p)
    h = np.ascontiguousarray(H0[:8])
    J = np.zeros((60, 8))
    orc.lib().orc_homography_jacobian(30, fp(x1), fp(y1), dp(h), dp(J))

    def err(hh):
        e = np.zeros(60)
        hh = np.ascontiguousarray(hh)
        orc.lib().orc_homography_errors(30, fp(x1), fp(y1), fp(x2), fp(y2), dp(hh), dp(e))
        return e
    for k in range(8):
        step = 1e-6 * max(1.0, abs(h[k]))
        hp, hm = h.copy(), h.copy()
        hp[k] += step; hm[k] -= step
        fd = (err(hp) - err(hm)) / (2 * step)
        assert np.max(np.abs(fd - J[:, k])) <= 1e-6 * max(1.0, np.max(np.abs(J[:, k])))


def test_refinement_reaches_the_least_squares_minimum(orc):
    """size-independent property: at the returned h the gradient J^T e vanishes relative to its start"""
    rng = np.random.default_rng(9)
    x1, y1, x2, y2, H0 = homography_problem(rng, 108)
    ret, h, info = orc.homography_lm(x1, y1, x2, y2, H0)
    assert ret > 0 and info[1] < info[0] and info[6] in (1, 2, 6)
    ret2, h2, info2 = orc.homography_lm(x1, y1, x2, y2, np.append(h, 1.0))      # restart: nothing left to gain
    assert abs(info2[1] - info[1]) <= 1e-9 * info[1]


# ---- HomographyEstimator::solve (normalised DLT) and ::residuals, homography.cpp:17-110 -----------------------------
# PARITY UNPINNED against the reference (it holds no vectors for them, and the SVD is Eigen's); pinned here against
# independent statements of the same arithmetic: numpy float32 for normalize_points, LAPACK's SVD for the null vector.

def _normalize_np(x, y):
    """normalize_points (homography.cpp:17-42) in numpy scalars: float32 sums in point order"""
    f = np.float32
    cx = f(0); cy = f(0)
    for a, b in zip(x, y):
        cx = f(cx + a); cy = f(cy + b)
    n = f(len(x))
    cx = f(cx / n); cy = f(cy / n)
    nrm = f(0)
    for a, b in zip(x, y):
        dx = f(a - cx); dy = f(b - cy)
        nrm = f(nrm + f(f(dx * dx) + f(dy * dy)))
    nrm = f(nrm / n)
    scale = np.sqrt(2.0) / np.float64(np.sqrt(nrm))           # np.sqrt of a float32 is the float square root
    nx = ((x - cx).astype(np.float64) * scale).astype(np.float32)
    ny = ((y - cy).astype(np.float64) * scale).astype(np.float32)
    T = np.array([[scale, 0, -scale * np.float64(cx)], [0, scale, -scale * np.float64(cy)], [0, 0, 1]])
    return nx, ny, T


def _dlt_np(x1, y1, x2, y2):
    a, b, T1 = _normalize_np(x1, y1)
    c, d, T2 = _normalize_np(x2, y2)
    u1, v1, u2, v2 = (t.astype(np.float64) for t in (a, b, c, d))
    n = len(x1)
    A = np.zeros((2 * n, 9))
    A[0::2, 0] = u1; A[0::2, 1] = v1; A[0::2, 2] = 1; A[0::2, 6] = -u2 * u1; A[0::2, 7] = -u2 * v1; A[0::2, 8] = -u2
    A[1::2, 3] = -u1; A[1::2, 4] = -v1; A[1::2, 5] = -1; A[1::2, 6] = v2 * u1; A[1::2, 7] = v2 * v1; A[1::2, 8] = v2
    v = np.linalg.svd(A)[2][-1]
    H = np.linalg.inv(T2) @ v.reshape(3, 3) @ T1
    return A, T1, T2, (H / H[2, 2]).ravel()


@pytest.mark.parametrize("N", [4, 5, 9, 54, 108, 333, 1024])
def test_dlt_matrix_is_the_reference_arithmetic_and_the_null_vector_is_lapacks(orc, N):
    rng = np.random.default_rng(100 + N)
    x1, y1, x2, y2, _ = homography_problem(rng, N)
    A, T1, T2 = orc.homography_dlt_matrix(x1, y1, x2, y2)
    An, T1n, T2n, Hn = _dlt_np(x1, y1, x2, y2)
    assert A.tolist() == An.tolist() and T1.tolist() == T1n.tolist() and T2.tolist() == T2n.tolist()   # bit for bit
    ret, H = orc.homography_dlt(x1, y1, x2, y2)
    assert 0 < ret < 60                                      # the sweeps converge
    assert np.max(np.abs(H - Hn)) <= 1e-12 * np.max(np.abs(Hn))
    assert H[8] == 1.0


def test_dlt_recovers_an_exact_homography_and_its_residuals_vanish(orc):
    rng = np.random.default_rng(12)
    Ht = np.array([[1.1, 0.05, 30.0], [-0.04, 0.95, -12.0], [1e-4, -2e-4, 1.0]])
    x1 = rng.uniform(0, 1280, 40).astype(np.float32); y1 = rng.uniform(0, 960, 40).astype(np.float32)
    w = Ht[2, 0] * x1 + Ht[2, 1] * y1 + 1
    x2 = ((Ht[0, 0] * x1 + Ht[0, 1] * y1 + Ht[0, 2]) / w).astype(np.float32)
    y2 = ((Ht[1, 0] * x1 + Ht[1, 1] * y1 + Ht[1, 2]) / w).astype(np.float32)
    ret, H = orc.homography_dlt(x1, y1, x2, y2)
    assert np.max(np.abs(H - Ht.ravel()) / np.maximum(np.abs(Ht.ravel()), 1e-3)) < 1e-3      # float32 points: ~1e-7 relative on pixels
    res = orc.homography_residuals(x1, y1, x2, y2, H)
    assert res.shape == (40,) and np.all(res < 1e-3)
    # the residual is the symmetric transfer error: against a direct numpy statement
    Hm = H.reshape(3, 3); Hi = np.linalg.inv(Hm)
    p1 = np.stack([x1, y1, np.ones(40)]).astype(np.float64); p2 = np.stack([x2, y2, np.ones(40)]).astype(np.float64)
    f = Hm @ p1; g = Hi @ p2
    ref = np.hypot(f[0] / f[2] - p2[0], f[1] / f[2] - p2[1]) + np.hypot(g[0] / g[2] - p1[0], g[1] / g[2] - p1[1])
    assert np.max(np.abs(res - ref)) <= 1e-9


def test_dlt_on_the_five_pairs_the_reference_test_lists(orc):
    """modules/calib/test/test_homography.cpp:73,77 (commented out there): the only point data the reference holds"""
    p1 = np.array([[1544, 592], [2236, 512], [1552, 2140], [2144, 1608], [2416, 1332]], np.float32)
    p2 = np.array([[292, 656], [992, 596], [360, 2332], [964, 1700], [1212, 1412]], np.float32)
    ret, H = orc.homography_dlt(p1[:, 0], p1[:, 1], p2[:, 0], p2[:, 1])
    _, _, _, Hn = _dlt_np(p1[:, 0], p1[:, 1], p2[:, 0], p2[:, 1])
    assert ret > 0 and np.max(np.abs(H - Hn)) <= 1e-12 * np.max(np.abs(Hn))
    res = orc.homography_residuals(p1[:, 0], p1[:, 1], p2[:, 0], p2[:, 1], H)
    assert np.all(res < 10.0)            # the reference's own inlier threshold for these images (test_homography.cpp:95)


def test_dlt_with_fewer_than_four_pairs_is_refused(orc):
    x = np.array([1, 2, 3], np.float32)
    ret, H = orc.homography_dlt(x, x * 2, x + 1, x - 1)
    assert ret == -1 and not H.any()


# ---- HomographyEstimator::solveRansac (homography.cpp:112-156 on ransac.hpp:28-78): parity unpinned, see cvx_oracle.h ----
def _py_sample(N, draw):
    """rng.hpp:62-75 word for word on a Python list"""
    vidx, max_idx, subset = list(range(N)), N - 1, []
    for k in range(4):
        index = draw[k]
        vidx[index], vidx[max_idx] = vidx[max_idx], vidx[index]
        subset.append(vidx[max_idx])
        max_idx -= 1
    return subset


def ransac_problem(rng, N, outliers, noise=0.3):
    x1, y1, x2, y2, _ = homography_problem(rng, N, noise=noise)
    bad = rng.permutation(N)[:outliers]
    x2 = x2.copy(); y2 = y2.copy()
    x2[bad] += rng.uniform(20, 200, outliers).astype(np.float32) * rng.choice([-1, 1], outliers).astype(np.float32)
    y2[bad] += rng.uniform(20, 200, outliers).astype(np.float32)
    return x1, y1, x2, y2


def draw_subsets(orc, rng, N, trials):
    return np.array([orc.rng_sample4(N, [int(rng.integers(0, N - k)) for k in range(4)]) for _ in range(trials)], dtype=np.int32)


def _py_ransac(orc, x1, y1, x2, y2, subsets, p):
    """ransac.hpp:28-78 in Python on the oracle's solve() and residuals(), which other tests pin"""
    n = len(x1)
    log_probability, one_over = np.log(1.0 - p.outlier_probability), 1.0 / n
    trial, max_trials, min_residual, best = 0, p.max_num_trials, np.finfo(np.float64).max, None
    while trial < min(p.max_num_trials, max_trials):
        s = subsets[trial]
        _, H = orc.homography_dlt(x1[s], y1[s], x2[s], y2[s])
        with np.errstate(all="ignore"):
            res = orc.homography_residuals(x1, y1, x2, y2, H)
            inl = np.flatnonzero(res < p.max_error)
            total = 0.0
            for i in inl:
                total += res[i]
            total = total / len(inl) if len(inl) else float("nan")
        if len(inl) > p.min_inliers and total < min_residual:
            min_residual, best = total, inl
            pno = 1.0 - (len(inl) * one_over) ** 4
            pno = min(max(pno, np.finfo(np.float64).eps), 1.0 - np.finfo(np.float64).eps)
            max_trials = int(log_probability / np.log(pno))
        trial += 1
    if best is None:
        return False, None, np.zeros(0, int), trial
    _, H = orc.homography_dlt(x1[best], y1[best], x2[best], y2[best])
    return True, H, best, trial


def test_sample_is_the_reference_swap_sequence(orc):
    rng = np.random.default_rng(5)
    for N in (4, 5, 9, 108, 1000):
        for _ in range(50):
            draw = [int(rng.integers(0, N - k)) for k in range(4)]
            s = orc.rng_sample4(N, draw)
            assert s == _py_sample(N, draw) and len(set(s)) == 4 and all(0 <= v < N for v in s)


@pytest.mark.parametrize("case", ["outliers", "tight_threshold", "all_inliers", "few_trials", "nothing_qualifies"])
def test_ransac_loop_follows_the_reference_loop(orc, case):
    rng = np.random.default_rng({"outliers": 1, "tight_threshold": 2, "all_inliers": 3, "few_trials": 4, "nothing_qualifies": 5}[case])
    kw = dict(outliers=dict(max_error=2.0), tight_threshold=dict(max_error=0.6, outlier_probability=0.5, max_num_trials=60),
              all_inliers=dict(max_error=50.0), few_trials=dict(max_error=1.5, max_num_trials=3, min_inliers=20),
              nothing_qualifies=dict(max_error=1e-6))[case]
    p = orc.ransac_params(**kw)
    stopped_early = 0
    for rep in range(6):
        N = int(rng.integers(12, 109))
        x1, y1, x2, y2 = ransac_problem(rng, N, outliers=0 if case == "all_inliers" else N // 4)
        subsets = draw_subsets(orc, rng, N, p.max_num_trials)
        ok, H, inl, trials = orc.homography_ransac(x1, y1, x2, y2, subsets, p)
        ok2, H2, inl2, trials2 = _py_ransac(orc, x1, y1, x2, y2, subsets, p)
        assert (ok, trials) == (ok2, trials2) and inl.tolist() == list(inl2)
        if ok:
            assert H.tolist() == H2.tolist()
            stopped_early += trials < p.max_num_trials
            assert len(inl) > p.min_inliers
        else:
            assert trials == p.max_num_trials and len(inl) == 0
    if case == "nothing_qualifies":
        assert not ok
    if case in ("outliers", "all_inliers"):
        assert stopped_early > 0          # the first acceptable model cuts max_trials (ransac.hpp:62-66)


def test_ransac_needs_four_pairs(orc):
    rng = np.random.default_rng(6)
    x1, y1, x2, y2 = ransac_problem(rng, 12, 0)
    ok, H, inl, trials = orc.homography_ransac(x1[:3], y1[:3], x2[:3], y2[:3], np.zeros((100, 4), np.int32))
    assert not ok and trials == 0 and len(inl) == 0

