"""GPU: tool-level bundle adjustment (cvx_amd/csrc/toolba_kernels.h, toolba.hip) through the C ABI against the
oracle restatement of hand_eye_registration.cpp:48-238.  Tolerances as for the main path (north_star): blocks of
the normal equations and costs 1e-12 relative, final cost / pose / points 1e-9 relative.  The oracle for this
row is parity-unpinned against the reference (no Ceres here); see tests/test_toolba_oracle.py."""
import numpy as np
import pytest

from cvx_amd import capi
from cvx_amd.problem import ToolBA
from test_helpers import rel_err, toolba_problem

pytestmark = pytest.mark.gpu


def _pair(orc, d, huber=1.0):
    args = (d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], d["trans"], d["intr"], d["P"])
    return ToolBA(*args, huber=huber), orc.ToolBA(*args, huber=huber)


@pytest.mark.parametrize("V,ragged,huber,outliers", [(5, 0, 1.0, 0.05), (40, 3, 1.0, 0.1), (40, 0, 0.0, 0.0), (300, 5, 2.5, 0.05)])
def test_cost_and_normal_equation_blocks(orc, V, ragged, huber, outliers):
    rng = np.random.default_rng(V)
    d = toolba_problem(rng, V, ragged=ragged, outliers=outliers)
    g, o = _pair(orc, d, huber)
    p = d["p0"] * 1.7                      # un-normalised quaternion: the functor normalises, Plus does not
    p[4:] = d["p0"][4:]
    c = g.cost(p, d["points"])
    assert abs(c - o.cost(p, d["points"])) <= 1e-12 * c
    Hpp, gp, W, U, gx, c2 = g.normal_eq(p, d["points"])
    H, gr, c3 = o.normal_eq(p, d["points"])
    assert abs(c2 - c3) <= 1e-12 * c3
    P = d["P"]
    assert rel_err(Hpp, H[:6, :6]) <= 1e-12 and rel_err(gp, gr[:6]) <= 1e-11
    for q in range(P):
        s = slice(6 + 3 * q, 9 + 3 * q)
        scale = max(np.max(np.abs(H[:6, s])), 1e-300)
        assert np.max(np.abs(W[q] - H[:6, s])) <= 1e-12 * max(scale, np.max(np.abs(H[:6, 6:])) * 1e-3)
        assert rel_err(U[q], H[s, s]) <= 1e-12
        assert np.max(np.abs(gx[q] - gr[s])) <= 1e-11 * np.max(np.abs(gr[6:]))
    # the point blocks do not couple with each other
    off = H[6:, 6:].copy()
    for q in range(P):
        off[3 * q:3 * q + 3, 3 * q:3 * q + 3] = 0
    assert np.all(off == 0)


@pytest.mark.parametrize("V,ragged,outliers", [(30, 0, 0.03), (60, 4, 0.08)])
def test_solve_follows_the_oracle(orc, V, ragged, outliers):
    rng = np.random.default_rng(100 + V)
    d = toolba_problem(rng, V, ragged=ragged, outliers=outliers)
    g, o = _pair(orc, d)
    po, xo, so = o.solve(d["p0"], d["points"], orc.options("toolba"))
    pg, xg, sg = g.solve(d["p0"], d["points"], capi.options("toolba"))
    assert abs(sg.initial_cost - so.initial_cost) <= 1e-12 * so.initial_cost
    # use_inner_iterations (hand_eye_registration.cpp:205): the same candidates go through a pass on either side
    assert capi.options("toolba").inner_iterations == 1 and orc.options("toolba").inner_iterations == 1
    assert sg.num_inner_iteration_steps == so.num_inner_iteration_steps >= 1
    # tolerances of 1e-20 make the last iterations a matter of rounding: compare the common prefix of the traces
    n = min(sg.trace_len, so.trace_len, 12)
    tg, to = np.array(sg.trace_cost[:n]), np.array(so.trace_cost[:n])
    assert np.max(np.abs(tg - to) / to) <= 1e-9
    assert abs(sg.final_cost - so.final_cost) <= 1e-9 * so.final_cost
    assert np.max(np.abs(pg - po)) <= 1e-8 and np.max(np.abs(xg - xo)) <= 1e-8
    assert sg.final_cost < 0.3 * sg.initial_cost


@pytest.mark.parametrize("V,ragged,outliers", [(30, 0, 0.03), (60, 4, 0.08)])
def test_inner_pass_matches_the_oracle(orc, V, ragged, outliers):
    """One coordinate-descent pass over the blocks t, q and every point (Ceres' automatic ordering on this graph),
    each block's own LM to its own termination, twice in a row; and the solve without the passes ends at the same
    cost."""
    rng = np.random.default_rng(200 + V)
    d = toolba_problem(rng, V, ragged=ragged, outliers=outliers)
    g, o = _pair(orc, d)
    c0 = g.cost(d["p0"], d["points"])
    po, xo, co = o.inner_pass(d["p0"], d["points"])
    pg, xg, cg = g.inner_pass(d["p0"], d["points"])
    assert cg < c0 and abs(cg - co) <= 1e-10 * co
    assert np.max(np.abs(pg - po)) <= 1e-10 and np.max(np.abs(xg - xo)) <= 1e-10 * np.max(np.abs(xo))
    assert abs(g.cost(pg, xg) - cg) <= 1e-13 * cg
    po2, xo2, co2 = o.inner_pass(po, xo)
    pg2, xg2, cg2 = g.inner_pass(pg, xg)
    assert cg2 <= cg and abs(cg2 - co2) <= 1e-10 * co2
    assert np.max(np.abs(pg2 - po2)) <= 1e-9 and np.max(np.abs(xg2 - xo2)) <= 1e-9 * np.max(np.abs(xo2))
    op = capi.options("toolba"); op.max_iterations = 40
    _, _, s1 = g.solve(d["p0"], d["points"], op)
    op.inner_iterations = 0
    _, _, s0 = g.solve(d["p0"], d["points"], op)
    assert s1.num_inner_iteration_steps >= 1 and s0.num_inner_iteration_steps == 0
    assert abs(s1.final_cost - s0.final_cost) <= 1e-8 * s0.final_cost


def test_inner_pass_with_a_single_point_in_use(orc):
    """With one point all three degrees of the Hessian graph tie and the naming order decides: the point first,
    then t, then q."""
    rng = np.random.default_rng(14)
    d = toolba_problem(rng, 12)
    keep = d["point_id"] == 20
    view_of = np.repeat(np.arange(12), np.diff(d["view_off"]))
    vo = np.concatenate([[0], np.cumsum(np.bincount(view_of[keep], minlength=12))]).astype(np.int32)
    d2 = dict(d, view_off=vo, point_id=d["point_id"][keep], obs_x=d["obs_x"][keep], obs_y=d["obs_y"][keep])
    g, o = _pair(orc, d2)
    po, xo, co = o.inner_pass(d["p0"], d["points"])
    pg, xg, cg = g.inner_pass(d["p0"], d["points"])
    assert abs(cg - co) <= 1e-9 * max(co, 1e-12) + 1e-18
    assert np.max(np.abs(pg - po)) <= 1e-8 and np.max(np.abs(xg - xo)) <= 1e-8 * np.max(np.abs(xo))
    moved = np.any(xg != d["points"], axis=1)
    assert moved[20] and not np.any(np.delete(moved, 20))


def test_repeated_runs_are_bitwise_identical(orc):
    rng = np.random.default_rng(11)
    d = toolba_problem(rng, 500, ragged=6)
    g, _ = _pair(orc, d)
    a = g.normal_eq(d["p0"], d["points"]); b = g.normal_eq(d["p0"], d["points"])
    for x, y in zip(a[:5], b[:5]):
        assert np.array_equal(x, y)


def test_full_size_cost_equals_sum_over_shards(orc):
    """10 000 views x 54 points: the device cost of the whole problem equals the sum of the oracle's costs over
    view shards (size-independent property), and one solve decreases it."""
    rng = np.random.default_rng(12)
    d = toolba_problem(rng, 10000)
    g, _ = _pair(orc, d)
    c = g.cost(d["p0"], d["points"])
    tot = 0.0
    for a in range(0, 10000, 2500):
        vo = d["view_off"][a:a + 2501] - d["view_off"][a]
        s = slice(d["view_off"][a], d["view_off"][a + 2500])
        o = orc.ToolBA(vo, d["point_id"][s], d["obs_x"][s], d["obs_y"][s], d["trans"][a:a + 2500], d["intr"], d["P"])
        tot += o.cost(d["p0"], d["points"])
    assert abs(c - tot) <= 1e-11 * tot
    op = capi.options("toolba"); op.max_iterations = 6
    p, x, s = g.solve(d["p0"], d["points"], op)
    assert s.final_cost < 0.2 * s.initial_cost
    assert np.max(np.abs(p[4:] - d["p_gt"][4:])) < 1e-3


def test_points_without_observations_and_empty_views(orc):
    rng = np.random.default_rng(13)
    d = toolba_problem(rng, 20)
    keep = (d["point_id"] != 7) & (d["point_id"] != 50)              # two board points never seen
    counts = np.diff(d["view_off"])
    view_of = np.repeat(np.arange(20), counts)
    keep &= view_of != 4                                              # one view without markers
    vo = np.concatenate([[0], np.cumsum(np.bincount(view_of[keep], minlength=20))]).astype(np.int32)
    d2 = dict(d, view_off=vo, point_id=d["point_id"][keep], obs_x=d["obs_x"][keep], obs_y=d["obs_y"][keep])
    g, o = _pair(orc, d2)
    Hpp, gp, W, U, gx, c = g.normal_eq(d["p0"], d["points"])
    assert np.all(U[7] == 0) and np.all(W[50] == 0) and np.all(gx[7] == 0)
    pg, xg, sg = g.solve(d["p0"], d["points"], capi.options("toolba"))
    po, xo, so = o.solve(d["p0"], d["points"], orc.options("toolba"))
    assert abs(sg.final_cost - so.final_cost) <= 1e-9 * so.final_cost
    assert np.array_equal(xg[7], d["points"][7]) and np.array_equal(xg[50], d["points"][50])     # never moved
