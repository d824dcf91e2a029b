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
