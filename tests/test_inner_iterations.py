"""Inner iterations (Solver::Options::use_inner_iterations = true, calibration.cpp:424), CPU side: the automatic
ordering the library computes on the host against the oracle's independent restatement of it (explicit Hessian graph
there, observation lists here) and against what Ceres' rule must give on shapes where that is obvious, and the
oracle's coordinate-descent pass itself.  Ceres is un-vendored and the reference holds no vectors for this: parity
unpinned, both sides restate the published algorithm."""
import numpy as np
import pytest

import cvx_amd
from cvx_amd import capi
from cvx_amd.problem import inner_ordering

SHAPES = [
    # (cameras, views per camera, board w, h, seed, ragged, fix_intrinsics, fix_points)
    (1, 20, 9, 6, 20260001, 0, False, False),      # BASELINE config 1
    (2, 7, 12, 9, 11, 4, False, False),
    (3, 5, 20, 15, 12, 3, False, False),
    (2, 60, 3, 3, 31, 6, False, False),            # many near-empty views: sets that mix poses and points
    (1, 200, 3, 3, 5, 0, False, False),            # more views than points: the poses have the lowest degree
    (2, 9, 9, 6, 14, 2, True, False),
    (2, 9, 9, 6, 14, 2, False, True),
    (1, 6, 9, 6, 13, 0, True, True),
    (4, 3, 28, 16, 23, 5, False, False),           # CVX_MAX_POINTS board points
    (64, 1, 4, 3, 24, 0, False, False),            # CVX_MAX_CAMERAS cameras of one view each: cameras and poses tie in degree
]


def problem(shape):
    C_, VP, bw, bh, seed, ragged, fi, fp = shape
    return cvx_amd.synth_calib(C_, VP, bw, bh, seed, ragged=ragged), fi, fp


@pytest.mark.parametrize("shape", SHAPES)
def test_ordering_matches_the_oracle_and_is_a_valid_coordinate_descent_ordering(orc, shape):
    d, fi, fp = problem(shape)
    C_, P, V = d["C"], d["P"], d["V"]
    g, n = inner_ordering(d["cam_of_view"], d["view_off"], d["point_id"], C_, P, fi, fp)
    o = orc.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], C_, P, fix_intrinsics=fi, fix_points=fp)
    go, no = o.inner_ordering()
    assert n == no and np.array_equal(g, go)
    view_of = np.repeat(np.arange(V), np.diff(d["view_off"]))
    cam = d["cam_of_view"][view_of]
    gc, gp, gv = g[cam], g[C_ + d["point_id"]], g[C_ + P + view_of]
    # independence: no residual block touches two blocks of one set
    for a, b in ((gc, gp), (gc, gv), (gp, gv)):
        assert not np.any((a == b) & (a >= 0))
    # coverage: exactly the free blocks that occur in a residual are ordered, the sets are 0 .. n-1
    seen = np.zeros(C_ + P + V, bool)
    if not fi:
        seen[cam] = True
    if not fp:
        seen[C_ + d["point_id"]] = True
    seen[C_ + P + view_of] = True
    assert np.array_equal(g >= 0, seen)
    assert set(g[g >= 0]) == set(range(n))


def test_ordering_on_the_shapes_where_ceres_rule_is_obvious():
    # config 1 (20 views, 54 points, every point in every view): degrees are point 21 < pose 55 < camera 74, so the
    # greedy rounds take the points, then the poses (degree 1 once the points are gone), then the camera; reversed:
    d = cvx_amd.synth_calib(1, 20, 9, 6, 20260001)
    g, n = inner_ordering(d["cam_of_view"], d["view_off"], d["point_id"], 1, 54)
    assert n == 3 and g[0] == 0 and np.all(g[1:55] == 2) and np.all(g[55:] == 1)
    # 200 views of 9 points: pose 10 < point 201 < camera 209: poses, then points, then the camera; reversed:
    d = cvx_amd.synth_calib(1, 200, 3, 3, 5)
    g, n = inner_ordering(d["cam_of_view"], d["view_off"], d["point_id"], 1, 9)
    assert n == 3 and g[0] == 0 and np.all(g[1:10] == 1) and np.all(g[10:] == 2)
    # BASELINE config 4 (10 000 views of 108 points, 8 cameras): pose 109 < point 10 008 < camera 1 358 + ...: the
    # poses go first, then the points (degree 8 once the poses are gone, the cameras have 108), then the cameras;
    # reversed: cameras, points, poses
    d4 = cvx_amd.synth_calib(8, 1250, 12, 9, 20260004)
    g4, n4 = inner_ordering(d4["cam_of_view"], d4["view_off"], d4["point_id"], 8, 108)
    assert n4 == 3 and np.all(g4[:8] == 0) and np.all(g4[8:116] == 1) and np.all(g4[116:] == 2)
    # nothing shared is free: every pose stands alone, one set
    g, n = inner_ordering(d["cam_of_view"], d["view_off"], d["point_id"], 1, 9, True, True)
    assert n == 1 and g[0] == -1 and np.all(g[1:10] == -1) and np.all(g[10:] == 0)
    # a view without markers has no residual block: its pose is in no set
    vo = d["view_off"].copy(); vo[-1] = vo[-2]
    g, n = inner_ordering(d["cam_of_view"], vo, d["point_id"][:vo[-1]], 1, 9)
    assert g[-1] == -1 and np.all(g[10:-1] == 2)


def test_ordering_matches_the_oracle_on_random_shapes(orc):
    """120 small random problems (1-4 cameras, ragged views, fixed blocks): the library's ordering (stored degrees, lazy
    lists, early-out greedy pass) equals the oracle's explicit-graph one; some of them have sets that mix kinds."""
    rng = np.random.default_rng(0)
    mixed = 0
    for _ in range(120):
        C_, VP, bw, bh = int(rng.integers(1, 5)), int(rng.integers(1, 30)), int(rng.integers(2, 7)), int(rng.integers(2, 6))
        rag = int(rng.choice([0, 2, 3, 6]))
        fi, fp = bool(rng.uniform() < 0.15), bool(rng.uniform() < 0.3)
        d = cvx_amd.synth_calib(C_, VP, bw, bh, int(rng.integers(1, 10 ** 6)), ragged=rag)
        P = bw * bh
        g, n = inner_ordering(d["cam_of_view"], d["view_off"], d["point_id"], C_, P, fi, fp)
        o = orc.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], C_, P, fix_intrinsics=fi, fix_points=fp)
        go, no = o.inner_ordering()
        assert n == no and np.array_equal(g, go)
        kind = np.where(np.arange(len(g)) < C_, 0, np.where(np.arange(len(g)) < C_ + P, 1, 2))
        mixed += any(len(set(kind[g == s])) > 1 for s in range(n))
    assert mixed >= 1


def test_ordering_rejects_malformed_input():
    d = cvx_amd.synth_calib(1, 4, 3, 3, 5)
    ok = (d["cam_of_view"], d["view_off"], d["point_id"], 1, 9)
    inner_ordering(*ok)
    bad_off = d["view_off"].copy(); bad_off[2] = bad_off[1] - 1
    bad_pid = d["point_id"].copy(); bad_pid[3] = 9
    bad_cam = d["cam_of_view"].copy(); bad_cam[1] = 1
    for args in ((d["cam_of_view"], bad_off, d["point_id"], 1, 9), (d["cam_of_view"], d["view_off"], bad_pid, 1, 9),
                 (bad_cam, d["view_off"], d["point_id"], 1, 9)):
        with pytest.raises(capi.CvxError) as e:
            inner_ordering(*args)
        assert e.value.status == 1          # CVX_ERR_INVALID_ARG


def test_refine_options_ask_for_inner_iterations():
    o = capi.options("refine")                     # calibration.cpp:424
    assert o.inner_iterations == 1 and o.inner_iteration_tolerance == 1e-3
    assert capi.options("toolba").inner_iterations == 1       # hand_eye_registration.cpp:205
    assert capi.options("handeye").inner_iterations == 0      # handeye.cpp:315 leaves it commented out
    assert capi.options("levmar").inner_iterations == 0


def test_oracle_pass_is_block_coordinate_descent_in_the_order_of_the_sets(orc):
    """An independent statement of what a pass is: set after set, every block of the set moved to the minimum of the
    cost over that block alone (scipy's MINPACK LM to convergence, on the oracle's residuals).  The oracle's blocks
    stop at Ceres' function tolerance of 1e-6, so the two agree to that order, not to rounding -- which is enough to
    tell a wrong order of the sets, a block left out, or a block minimised over the wrong residuals."""
    from scipy.optimize import least_squares
    d = cvx_amd.synth_calib(2, 5, 4, 3, 41, ragged=5)
    C_, P, V = d["C"], d["P"], d["V"]
    o = orc.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], C_, P)
    group, n = o.inner_ordering()
    x = [d["intr_init"].copy(), d["poses_init"].copy(), d["points_init"].copy()]
    free_intr = [0, 1, 2, 3, 4, 5, 7, 8]                    # k3 is held (calibration.cpp:410-415)

    def residuals():
        return o.eval(*x)[0].ravel()
    for s in range(n):
        for b in np.nonzero(group == s)[0]:
            if b < C_:
                arr, idx = x[0][b], free_intr
            elif b < C_ + P:
                arr, idx = x[2][b - C_], [0, 1, 2]
            else:
                arr, idx = x[1][b - C_ - P], list(range(6))

            def f(z, arr=arr, idx=idx):
                arr[idx] = z
                return residuals()
            z = least_squares(f, arr[idx].copy(), method="lm", xtol=1e-14, ftol=1e-14, gtol=1e-14).x
            arr[idx] = z
    exact = o.cost(*x)
    io, po, xo, co = o.inner_pass(d["intr_init"], d["poses_init"], d["points_init"])
    c0 = o.cost(d["intr_init"], d["poses_init"], d["points_init"])
    assert exact < 0.1 * c0
    assert exact <= co * (1 + 1e-9) and co - exact <= 2e-3 * exact     # the oracle stops a little short of each minimum
    assert np.max(np.abs(io - x[0]) / np.maximum(np.abs(x[0]), 1e-2)) < 5e-2
    assert np.max(np.abs(xo - x[2])) < 5e-2 * 25.0 and np.max(np.abs(po - x[1])) < 5e-2 * np.max(np.abs(x[1]))


def test_oracle_pass_descends_and_the_solve_ends_where_the_plain_one_does(orc):
    d = cvx_amd.synth_calib(2, 7, 12, 9, 11, ragged=4)
    o = orc.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], 2, 108)
    x = (d["intr_init"], d["poses_init"], d["points_init"])
    c0 = o.cost(*x)
    i1, p1, x1, c1 = o.inner_pass(*x)
    i2, p2, x2, c2 = o.inner_pass(i1, p1, x1)
    assert c1 < 1e-2 * c0 and c2 <= c1                 # coordinate descent from a perturbed start gains a lot, then little
    assert abs(o.cost(i1, p1, x1) - c1) <= 1e-14 * c1
    # every block of the set minimised last is left at a minimum of its own problem: another LM on it moves nothing
    g, n = o.inner_ordering()
    on = orc.options("refine"); assert on.inner_iterations == 1
    off = orc.options("refine"); off.inner_iterations = 0
    sn, sf = o.solve(*x, on)[3], o.solve(*x, off)[3]
    assert sn.num_inner_iteration_steps >= 1 and sf.num_inner_iteration_steps == 0
    assert sn.termination == sf.termination == 3
    assert abs(sn.final_cost - sf.final_cost) <= 1e-9 * sf.final_cost
    assert sn.trace()["cost"][0] < sf.trace()["cost"][0]      # the first candidate is better for its pass
