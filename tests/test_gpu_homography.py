"""GPU: the batched homography LM kernel (cvx_amd/csrc/homography_kernels.h) through the C ABI against the
oracle.  Bar: BIT-exact -- the kernel evaluates every sum in the reference's order, so parameters, levmar's
info[] and the return value must equal the CPU reference's exactly (tests/golden/homography_lm.json holds runs
of the reference's own levmar; oracle/cvx_oracle.c is pinned to it by tests/test_homography_oracle.py)."""
import numpy as np
import pytest

import cvx_amd
from cvx_amd import capi
from cvx_amd.problem import HomographyBatch
from conftest import load_golden
from test_helpers import homography_batch, homography_problem, rel_err

pytestmark = pytest.mark.gpu


def _oracle_batch(orc, off, x1, y1, x2, y2, H0, itmax=100, which=None):
    out = {}
    for i in (range(len(off) - 1) if which is None else which):
        s = slice(off[i], off[i + 1])
        if off[i + 1] - off[i] < 4:
            out[i] = (-1, np.array(H0[i][:8]), np.zeros(10))
        else:
            out[i] = orc.homography_lm(x1[s], y1[s], x2[s], y2[s], H0[i], itmax=itmax)
    return out


def test_golden_reference_levmar_runs_bit_exact():
    gold = load_golden("homography_lm.json")
    names = sorted(gold)
    for itmax in sorted({gold[n]["itmax"] for n in names}):
        sel = [n for n in names if gold[n]["itmax"] == itmax]
        off = np.concatenate([[0], np.cumsum([len(gold[n]["x1"]) for n in sel])]).astype(np.int32)
        cat = lambda k: np.concatenate([np.array(gold[n][k], dtype=np.float32) for n in sel])
        b = HomographyBatch(off, cat("x1"), cat("y1"), cat("x2"), cat("y2"))
        H, info, ret = b.solve(np.stack([gold[n]["H0"] for n in sel]), max_iter=itmax)
        for i, n in enumerate(sel):
            g = gold[n]
            assert ret[i] == g["ret"], n
            if g["ret"] >= 0:
                assert H[i, :8].tolist() == g["h"], n
                assert H[i, 8] == 1.0
                assert info[i].tolist() == g["info"], n
            else:                                   # fewer than 4 pairs: untouched (lm_core.c:117-120)
                assert H[i].tolist() == g["H0"], n


def test_random_ragged_batch_bit_exact_against_oracle(orc):
    rng = np.random.default_rng(2026)
    sizes = np.concatenate([rng.integers(4, 200, 250), [4, 5, 63, 64, 65, 108, 512, 3, 0, 1]])
    off, x1, y1, x2, y2, H0 = homography_batch(rng, np.maximum(sizes, 1))
    b = HomographyBatch(off, x1, y1, x2, y2)
    H, info, ret = b.solve(H0)
    ref = _oracle_batch(orc, off, x1, y1, x2, y2, H0)
    for i, (r, h, inf) in ref.items():
        assert ret[i] == r, i
        if r >= 0:
            assert H[i, :8].tolist() == h.tolist(), i
            assert info[i].tolist() == inf.tolist(), i
        else:
            assert H[i].tolist() == H0[i].tolist(), i


def test_empty_and_tiny_problems_inside_a_batch(orc):
    rng = np.random.default_rng(5)
    off, x1, y1, x2, y2, H0 = homography_batch(rng, [20, 3, 30])
    off = np.array([0, 20, 20, 23, 53, 53], dtype=np.int32)            # 20, 0, 3, 30, 0 pairs
    H0 = np.stack([H0[0], H0[1], H0[1], H0[2], H0[2]])
    H, info, ret = HomographyBatch(off, x1, y1, x2, y2).solve(H0)
    assert ret[1] == -1 and ret[2] == -1 and ret[4] == -1
    for i in (1, 2, 4):
        assert H[i].tolist() == H0[i].tolist()
    ref = _oracle_batch(orc, off, x1, y1, x2, y2, H0, which=[0, 3])
    for i, (r, h, inf) in ref.items():
        assert ret[i] == r and H[i, :8].tolist() == h.tolist() and info[i].tolist() == inf.tolist()


def test_full_size_batch_properties_and_sampled_parity(orc):
    """BASELINE config 4's view count: 10 000 views x 108 board points.  Properties that hold at any size
    (cost never increases, a restart from the result gains nothing) plus bit-exact parity on a sample."""
    rng = np.random.default_rng(4)
    n = 10000
    base = homography_batch(rng, [108] * 50)
    rep = n // 50
    off = (np.arange(n + 1) * 108).astype(np.int32)
    x1, y1, x2, y2 = (np.tile(a, rep) for a in base[1:5])
    H0 = np.tile(base[5], (rep, 1)) * (1 + 1e-3 * rng.normal(size=(n, 1)))       # distinct starts per view
    H0 /= H0[:, 8:9]
    b = HomographyBatch(off, x1, y1, x2, y2)
    H, info, ret = b.solve(H0)
    assert np.all(ret > 0) and np.all(info[:, 1] <= info[:, 0]) and np.all(np.isin(info[:, 6], (1, 2, 3, 6)))
    H2, info2, _ = b.solve(H)
    assert np.all(np.abs(info2[:, 1] - info[:, 1]) <= 1e-9 * info[:, 1])
    ref = _oracle_batch(orc, off, x1, y1, x2, y2, H0, which=range(0, n, 397))
    for i, (r, h, inf) in ref.items():
        assert ret[i] == r and H[i, :8].tolist() == h.tolist() and info[i].tolist() == inf.tolist()


def test_custom_options_and_iteration_cap(orc):
    rng = np.random.default_rng(8)
    off, x1, y1, x2, y2, H0 = homography_batch(rng, [54, 108, 70])
    opts = (1e-2, 1e-9, 1e-9, 1e-3)
    H, info, ret = HomographyBatch(off, x1, y1, x2, y2).solve(H0, max_iter=7, opts=opts)
    for i in range(3):
        s = slice(off[i], off[i + 1])
        r, h, inf = orc.homography_lm(x1[s], y1[s], x2[s], y2[s], H0[i], itmax=7, opts=opts)
        assert ret[i] == r and H[i, :8].tolist() == h.tolist() and info[i].tolist() == inf.tolist()


# ---- HomographyEstimator::solve (normalised DLT) and ::residuals on the same handle (homography.cpp:17-110) ---------
# Bar: floating point.  normalize_points and the constraint matrix are the reference's arithmetic in the reference's
# order on both sides (the oracle's are pinned bit for bit to a numpy-float32 statement, tests/test_homography_oracle.py);
# the null vector comes from two different, backward-stable SVDs (oracle: one-sided Jacobi on the 2N x 9 matrix, pinned
# to LAPACK at 1e-12; device: Householder QR, then one-sided Jacobi on the triangle), so H agrees to the conditioning of
# the smallest singular vector: DLT_TOL relative to max |H| of the problem, stated here.
DLT_TOL = 1e-13                # measured 2e-15 over the ragged batch below
DLT_TOL_MINIMAL = 1e-11        # four pairs: eight equations, an exact null space, no averaging of the rounding


def _general_position_problem(rng, N):
    """N >= 4 pairs that include the board's four corners: a few random grid points are often collinear, and a
    degenerate set has a null space of more than one dimension (any vector of it is as good as another)"""
    x1, y1, x2, y2, H0 = homography_problem(rng, 108)
    corner = ((x1 == 0) | (x1 == 275)) & ((y1 == 0) | (y1 == 200))
    idx = np.concatenate([np.flatnonzero(corner), np.flatnonzero(~corner)[:N - 4]])
    return x1[idx], y1[idx], x2[idx], y2[idx], H0


def _batch_of(parts):
    off = np.concatenate([[0], np.cumsum([len(p[0]) for p in parts])]).astype(np.int32)
    cat = lambda k: np.concatenate([p[k] for p in parts])
    return off, cat(0), cat(1), cat(2), cat(3), np.stack([p[4] for p in parts])


def _dlt_oracle(orc, off, x1, y1, x2, y2, which=None):
    out = {}
    for i in (range(len(off) - 1) if which is None else which):
        s = slice(off[i], off[i + 1])
        out[i] = orc.homography_dlt(x1[s], y1[s], x2[s], y2[s])
    return out


def test_dlt_ragged_batch_against_oracle(orc):
    rng = np.random.default_rng(31)
    parts = [homography_problem(rng, int(n)) for n in np.concatenate([rng.integers(12, 200, 120), [63, 64, 65, 108, 512, 1000, 1024]])]
    parts += [_general_position_problem(rng, n) for n in (4, 4, 5, 6, 8, 9)]
    parts += [homography_problem(rng, n) for n in (3, 1, 2)]                # too few pairs: left alone
    off, x1, y1, x2, y2, H0 = _batch_of(parts)
    b = HomographyBatch(off, x1, y1, x2, y2)
    b.set(H0)
    H = b.dlt()
    worst = worst4 = 0.0
    for i, (ret, Ho) in _dlt_oracle(orc, off, x1, y1, x2, y2).items():
        n = off[i + 1] - off[i]
        if n < 4:
            assert ret == -1 and H[i].tolist() == H0[i].tolist()          # keeps its current estimate
            continue
        err = np.max(np.abs(H[i] - Ho)) / np.max(np.abs(Ho))
        worst = max(worst, err if n > 4 else 0.0)
        worst4 = max(worst4, err if n == 4 else 0.0)
        assert err <= (DLT_TOL if n > 4 else DLT_TOL_MINIMAL), (i, n, err)
        assert H[i, 8] == 1.0
    print("worst relative DLT difference over the batch: %.2e (n > 4), %.2e (n = 4)" % (worst, worst4))
    # residuals of the DLT estimates, against the oracle's on the same H
    res = b.residuals()
    for i in range(len(off) - 1):
        s = slice(off[i], off[i + 1])
        if off[i + 1] - off[i] >= 4:
            ro = orc.homography_residuals(x1[s], y1[s], x2[s], y2[s], H[i])
            assert np.max(np.abs(res[s] - ro)) <= 1e-11 * max(1.0, np.max(np.abs(ro)))


def test_dlt_then_refinement_on_the_device_equals_the_reference_flow(orc):
    """solve -> solveLM without the estimates leaving the device; the refinement from the device's DLT estimate is the
    oracle's bit for bit (the LM is order-exact), and ends where the refinement from the oracle's DLT estimate ends"""
    rng = np.random.default_rng(32)
    off, x1, y1, x2, y2, _ = _batch_of([homography_problem(rng, n) for n in (108, 54, 70, 12, 200)] +
                                       [_general_position_problem(rng, n) for n in (4, 5)])
    b = HomographyBatch(off, x1, y1, x2, y2)
    Hd = b.dlt()
    b.run()
    H, info, ret = b.get()
    for i in range(len(off) - 1):
        s = slice(off[i], off[i + 1])
        r, h, inf = orc.homography_lm(x1[s], y1[s], x2[s], y2[s], Hd[i])
        assert ret[i] == r and H[i, :8].tolist() == h.tolist() and info[i].tolist() == inf.tolist()
        _, Ho = orc.homography_dlt(x1[s], y1[s], x2[s], y2[s])
        r2, h2, inf2 = orc.homography_lm(x1[s], y1[s], x2[s], y2[s], Ho)
        assert np.max(np.abs(h2 - h)) <= 1e-7 * np.max(np.abs(h))          # two LM runs to levmar's own tolerances
        assert abs(inf2[1] - inf[1]) <= 1e-9 * max(inf[1], 1e-12)
        assert inf[1] <= inf[0]                                            # the refinement never loses to the DLT


def test_residuals_with_given_estimates(orc):
    rng = np.random.default_rng(33)
    off, x1, y1, x2, y2, H0 = homography_batch(rng, [30, 4, 77])
    res = HomographyBatch(off, x1, y1, x2, y2).residuals(H0)
    for i in range(3):
        s = slice(off[i], off[i + 1])
        ro = orc.homography_residuals(x1[s], y1[s], x2[s], y2[s], H0[i])
        assert np.max(np.abs(res[s] - ro)) <= 1e-11 * max(1.0, np.max(np.abs(ro)))


def test_dlt_full_size_batch_properties_and_sampled_parity(orc):
    """10 000 views x 108 pairs: every estimate is finite and normalised, its transfer residuals are at the noise level
    of the synthetic pairs, a sample equals the oracle"""
    rng = np.random.default_rng(34)
    n = 10000
    base = homography_batch(rng, [108] * 50)
    rep = n // 50
    off = (np.arange(n + 1) * 108).astype(np.int32)
    x1, y1, x2, y2 = (np.tile(a, rep) for a in base[1:5])
    x2 = (x2 + rng.normal(0, 0.05, x2.shape)).astype(np.float32)          # distinct problems
    b = HomographyBatch(off, x1, y1, x2, y2)
    H = b.dlt()
    assert np.all(np.isfinite(H)) and np.all(H[:, 8] == 1.0)
    res = b.residuals().reshape(n, 108)
    assert np.all(res.mean(axis=1) < 5.0)
    for i, (ret, Ho) in _dlt_oracle(orc, off, x1, y1, x2, y2, which=range(0, n, 397)).items():
        assert np.max(np.abs(H[i] - Ho)) <= DLT_TOL * np.max(np.abs(Ho)), i


# ---- HomographyEstimator::solveRansac for a batch (cvx_homography_ransac) ----
from test_homography_oracle import draw_subsets, ransac_problem      # noqa: E402


def _ransac_batch(orc, rng, sizes, trials, outlier_share=4):
    parts, subs = [], []
    for n in sizes:
        n = int(n)
        if n >= 4:
            parts.append(ransac_problem(rng, n, n // outlier_share if outlier_share else 0))
            subs.append(draw_subsets(orc, rng, n, trials))
        else:
            x1, y1, x2, y2, _ = homography_problem(rng, max(n, 1))
            parts.append((x1[:n], y1[:n], x2[:n], y2[:n]))
            subs.append(np.zeros((trials, 4), np.int32))
    off = np.concatenate([[0], np.cumsum([len(p[0]) for p in parts])]).astype(np.int32)
    cat = lambda k: np.concatenate([p[k] for p in parts])
    return off, cat(0), cat(1), cat(2), cat(3), np.stack(subs)


def _check_against_oracle(orc, off, x1, y1, x2, y2, subsets, params, out, which=None, H_tol=1e-10):
    for i in (range(len(off) - 1) if which is None else which):
        s = slice(off[i], off[i + 1])
        ok, H, inl, trials = orc.homography_ransac(x1[s], y1[s], x2[s], y2[s], subsets[i], orc.ransac_params(**params))
        assert out["trials_run"][i] == trials, i
        assert out["n_inliers"][i] == len(inl), i
        assert np.flatnonzero(out["mask"][s]).tolist() == inl.tolist(), i
        if ok:
            # the model of the inliers: the DLT's own bar (DLT_TOL; the null vector comes from different rotations)
            assert rel_err(out["H"][i], H) < H_tol, (i, rel_err(out["H"][i], H))


@pytest.mark.parametrize("params", [dict(max_error=2.0), dict(max_error=0.8, outlier_probability=0.6, max_num_trials=40, min_inliers=8),
                                    dict(max_error=1e-7, max_num_trials=12)])
def test_ransac_batch_with_given_subsets_follows_the_oracle(orc, params):
    rng = np.random.default_rng(41)
    T = params.get("max_num_trials", 100)
    sizes = np.concatenate([rng.integers(10, 200, 40), [4, 5, 7, 64, 65, 108, 640, 3, 0, 1024]])
    off, x1, y1, x2, y2, subsets = _ransac_batch(orc, rng, sizes, T)
    b = HomographyBatch(off, x1, y1, x2, y2)
    H_before = np.tile(np.arange(9.0), (b.n, 1))
    b.set(H_before)
    out = b.ransac(subsets=subsets, params=params)
    _check_against_oracle(orc, off, x1, y1, x2, y2, subsets, params, out)
    none = out["n_inliers"] == 0
    assert none[-3] and none[-2]                          # fewer than four pairs: estimate() cannot run
    assert np.array_equal(out["H"][none], H_before[none])   # no model: the estimate stays what it was
    if params["max_error"] < 1e-6:
        assert none.all() and (out["trials_run"][sizes >= 4] == T).all()
    else:
        assert (out["n_inliers"][sizes >= 12] > 5).all()


def test_ransac_subsets_drawn_on_the_device(orc):
    rng = np.random.default_rng(42)
    sizes = np.concatenate([rng.integers(8, 150, 30), [4, 5, 108]])
    off, x1, y1, x2, y2, _ = _ransac_batch(orc, rng, sizes, 1)
    b = HomographyBatch(off, x1, y1, x2, y2)
    params = dict(max_error=2.0, max_num_trials=50)
    out = b.ransac(seed=2026, params=params, want_subsets=True)
    sub = out["subsets"]
    for i, n in enumerate(sizes):
        assert ((sub[i] >= 0) & (sub[i] < n)).all()
        assert all(len(set(row.tolist())) == 4 for row in sub[i])
    # RNG::sample's property: uniform over the pairs (a loose check: every pair of the largest problem is drawn sometime)
    again = b.ransac(seed=2026, params=params, want_subsets=True)
    other = b.ransac(seed=2027, params=params, want_subsets=True)
    assert np.array_equal(again["subsets"], sub) and not np.array_equal(other["subsets"], sub)
    assert np.array_equal(again["H"], out["H"])
    _check_against_oracle(orc, off, x1, y1, x2, y2, sub, params, out)


def test_ransac_rejects_bad_subsets(orc):
    rng = np.random.default_rng(43)
    off, x1, y1, x2, y2, subsets = _ransac_batch(orc, rng, [20, 30], 10)
    b = HomographyBatch(off, x1, y1, x2, y2)
    for bad in ((1, 3, 2, 30), (0, 0, 0, -1), (1, 2, 0, 2)):        # (problem, trial, slot, value): out of range, negative, repeated
        s = subsets.copy()
        s[bad[0], bad[1], bad[2]] = s[bad[0], bad[1], 1] if bad == (1, 2, 0, 2) else bad[3]
        with pytest.raises(capi.CvxError):
            b.ransac(subsets=s, params=dict(max_num_trials=10))
    with pytest.raises(capi.CvxError):
        b.ransac(seed=1, params=dict(max_num_trials=0))
    out = b.ransac(subsets=subsets, params=dict(max_num_trials=10, max_error=2.0))     # and the handle still works
    assert (out["n_inliers"] > 5).all()


def test_ransac_then_lm_over_the_inliers_is_the_reference_flow(orc):
    """solveRansac -> inliers -> solveLM(pts1, pts2, inliers, H) (test_homography.cpp:101-110): the inlier batch that
    cvx_homography_ransac leaves runs the LM bit for bit as the oracle runs it on the gathered inliers from the same start"""
    rng = np.random.default_rng(44)
    sizes = rng.integers(20, 150, 12)
    params = dict(max_error=2.0)
    off, x1, y1, x2, y2, subsets = _ransac_batch(orc, rng, sizes, 100)
    b = HomographyBatch(off, x1, y1, x2, y2)
    out = b.ransac(subsets=subsets, params=params, keep_inliers=True)
    nb = out["inliers"]
    assert nb.off.tolist() == np.concatenate([[0], np.cumsum(out["n_inliers"])]).tolist()
    res_in = nb.residuals()                             # of the refit, over the inliers only
    nb.run(100)
    H, info, ret = nb.get()
    for i in range(b.n):
        s = slice(off[i], off[i + 1])
        inl = np.flatnonzero(out["mask"][s])
        r, h, inf = orc.homography_lm(x1[s][inl], y1[s][inl], x2[s][inl], y2[s][inl], out["H"][i], itmax=100)
        assert ret[i] == r and H[i, :8].tolist() == h.tolist() and info[i].tolist() == inf.tolist(), i
        ref = orc.homography_residuals(x1[s][inl], y1[s][inl], x2[s][inl], y2[s][inl], out["H"][i])
        assert np.allclose(res_in[nb.off[i]:nb.off[i + 1]], ref, rtol=1e-12, atol=0)
    nb.close(); b.close()


def test_ransac_full_size_batch_properties_and_sampled_parity(orc):
    """10 000 views x 108 pairs x 100 trials = 1 M hypotheses in one call"""
    rng = np.random.default_rng(45)
    n, N, T = 10000, 108, 100
    parts = [ransac_problem(rng, N, 20) for _ in range(64)]
    pick = rng.integers(0, 64, n)
    off = (np.arange(n + 1) * N).astype(np.int32)
    cat = lambda k: np.concatenate([parts[j][k] for j in pick])
    x1, y1, x2, y2 = cat(0), cat(1), cat(2), cat(3)
    b = HomographyBatch(off, x1, y1, x2, y2)
    params = dict(max_error=2.0)
    out = b.ransac(seed=7, params=params, want_subsets=True)
    assert (out["n_inliers"] > 5).all() and (out["n_inliers"] <= 88).all()
    assert np.array_equal(out["mask"].reshape(n, N).sum(1), out["n_inliers"])
    assert (out["trials_run"] >= 1).all() and (out["trials_run"] <= T).all()
    res = b.residuals(out["H"]).reshape(n, N)
    # (the first acceptable model cuts max_trials to a handful, ransac.hpp:62-66: many problems stop on a mediocre model)
    good = out["n_inliers"] >= 80                       # the planted outliers are out, and the refit is at the noise level
    assert good.mean() > 0.2
    assert np.median(res[good][out["mask"].reshape(n, N)[good] == 1]) < 1.0
    _check_against_oracle(orc, off, x1, y1, x2, y2, out["subsets"], params, out, which=rng.integers(0, n, 24))

