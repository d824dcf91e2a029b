"""GPU tests of the inner iterations (use_inner_iterations = true, calibration.cpp:424) through the C ABI: the
batched per-set block LM of inner_kernels.h against the oracle's block-by-block coordinate descent, and the
trust-region loop that uses the passes (lm_driver.cpp) against the oracle's.  Parity unpinned (Ceres is
un-vendored, the reference holds no vectors): both sides restate Ceres' published algorithm, independently."""
import ctypes as C

import numpy as np
import pytest

import cvx_amd
from cvx_amd import capi
from test_helpers import assert_params_close, param_block_errs, rel_err

pytestmark = pytest.mark.gpu

# one pass, block by block (measured on the box: see DESIGN.md section 2)
PASS_TOL = {"focal_pp": 1e-10, "distortion": 1e-10, "rotation": 1e-10, "translation": 1e-10, "points": 1e-10}

CASES = {
    "config1": dict(spec=(1, 20, 9, 6, 20260001), ragged=0, fix={}),
    "two_cameras_ragged": dict(spec=(2, 7, 12, 9, 11), ragged=4, fix={}),
    "three_cameras_free_k3_big_board": dict(spec=(3, 5, 20, 15, 12), ragged=3, fix={"fix_k3": False}),
    "near_empty_views_mixed_sets": dict(spec=(2, 60, 3, 3, 31), ragged=6, fix={}),
    "more_views_than_points": dict(spec=(1, 200, 3, 3, 5), ragged=0, fix={}),
    "fixed_intrinsics": dict(spec=(1, 6, 9, 6, 13), ragged=0, fix={"fix_intrinsics": True}),
    "fixed_points": dict(spec=(2, 9, 9, 6, 14), ragged=2, fix={"fix_points": True}),
    "poses_only": dict(spec=(2, 4, 9, 6, 14), ragged=2, fix={"fix_intrinsics": True, "fix_points": True}),
    # the limits of the build: CVX_MAX_POINTS board points; CVX_MAX_CAMERAS cameras (one view each: cameras and poses
    # tie in degree, the naming order decides)
    "max_points": dict(spec=(4, 3, 28, 16, 23), ragged=5, fix={}),
    "max_cameras": dict(spec=(64, 1, 4, 3, 24), ragged=0, fix={"fix_points": True}),
}


def make(case, orc):
    c = CASES[case]
    C_, VP, bw, bh, seed = c["spec"]
    d = cvx_amd.synth_calib(C_, VP, bw, bh, seed, ragged=c["ragged"])
    args = (d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], C_, bw * bh)
    return d, cvx_amd.Calib(*args, **c["fix"]), orc.Calib(*args, **c["fix"])


@pytest.mark.parametrize("case", list(CASES))
def test_one_pass_matches_the_oracle_block_by_block(orc, case):
    """Two passes in a row from the perturbed start (the second starts where the blocks' trust regions, Jacobi scales
    and iteration counts have to be set up afresh).  Every block runs its own LM to its own termination, so this
    covers accept / reject / tolerance decisions of some hundreds of small problems per case."""
    d, g, o = make(case, orc)
    x = (d["intr_init"], d["poses_init"], d["points_init"])
    g.set_params(*x)
    c0 = g.cost()
    io, po, xo, co = o.inner_pass(*x)
    cg = g.inner_pass()
    ig, pg, xg = g.get_params()
    assert cg < c0
    assert abs(cg - co) <= 1e-11 * co
    assert_params_close((ig, pg, xg), (io, po, xo), PASS_TOL, case)
    assert abs(g.cost() - cg) <= 1e-13 * cg                     # the rotation tables follow the improved poses
    io2, po2, xo2, co2 = o.inner_pass(io, po, xo)
    cg2 = g.inner_pass()
    ig, pg, xg = g.get_params()
    assert cg2 <= cg and abs(cg2 - co2) <= 1e-11 * co2
    assert_params_close((ig, pg, xg), (io2, po2, xo2), PASS_TOL, case)
    g.close()


@pytest.mark.parametrize("case", ["config1", "two_cameras_ragged", "three_cameras_free_k3_big_board", "fixed_intrinsics",
                                  "fixed_points", "poses_only"])
def test_refine_with_inner_iterations_follows_the_oracle(orc, case):
    d, g, o = make(case, orc)
    x = (d["intr_init"], d["poses_init"], d["points_init"])
    g.set_params(*x)
    og, oo = capi.options("refine"), orc.options("refine")
    assert og.inner_iterations == 1 and oo.inner_iterations == 1      # calibration.cpp:424
    s = g.solve(og)
    io, po, xo, so = o.solve(*x, oo)
    ig, pg, xg = g.get_params()
    assert (s.iterations, s.termination, s.num_inner_iteration_steps) == (so.iterations, so.termination, so.num_inner_iteration_steps)
    assert s.num_inner_iteration_steps >= 1
    tg, to = s.trace()["cost"], so.trace()["cost"]
    assert len(tg) == len(to) and np.array_equal(s.trace()["accepted"], so.trace()["accepted"])
    n = s.num_inner_iteration_steps
    assert np.all(np.abs(tg[:n] - to[:n]) <= 1e-11 * to[:n])       # candidates that went through a pass
    assert np.all(np.abs(tg - to) <= 1e-9 * to)
    assert abs(s.final_cost - so.final_cost) <= 1e-11 * so.final_cost
    assert_params_close((ig, pg, xg), (io, po, xo), None, case)      # every block to north_star's 1e-9
    # without the passes the run ends at the same cost (the minimum is the same, the path is not)
    g.set_params(*x)
    og.inner_iterations = 0
    s0 = g.solve(og)
    assert s0.num_inner_iteration_steps == 0
    assert abs(s0.final_cost - s.final_cost) <= 1e-8 * s.final_cost
    g.close()


def test_passes_on_random_shapes(orc):
    """Thirty small random problems (1-4 cameras, ragged views down to near-empty ones, fixed blocks, free k3): the
    sets differ from problem to problem (some mix poses and points) and every block runs its own LM."""
    rng = np.random.default_rng(7)
    worst = 0.0
    for _ in range(30):
        C_, VP, bw, bh = int(rng.integers(1, 5)), int(rng.integers(3, 25)), int(rng.integers(3, 8)), int(rng.integers(3, 6))
        rag = int(rng.choice([0, 2, 3, 6]))
        fix = {"fix_intrinsics": bool(rng.uniform() < 0.15), "fix_points": bool(rng.uniform() < 0.3), "fix_k3": bool(rng.uniform() < 0.7)}
        d = cvx_amd.synth_calib(C_, VP, bw, bh, int(rng.integers(1, 10 ** 6)), ragged=rag)
        args = (d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], C_, bw * bh)
        g, o = cvx_amd.Calib(*args, **fix), orc.Calib(*args, **fix)
        x = (d["intr_init"], d["poses_init"], d["points_init"])
        g.set_params(*x)
        io, po, xo, co = o.inner_pass(*x)
        cg = g.inner_pass()
        ig, pg, xg = g.get_params()
        g.close()
        assert abs(cg - co) <= 1e-9 * co
        e = param_block_errs((ig, pg, xg), (io, po, xo))
        worst = max(worst, max(v for k, v in e.items() if k != "dist_cols"))
    assert worst < 1e-8


def test_blocks_without_residuals_stay_where_they_are(orc):
    """A view without markers and a board point nobody sees are in no residual block: Ceres never learns of them,
    the ordering leaves them out (-1) and a pass must not touch them; a handle without views has nothing to do."""
    d = cvx_amd.synth_calib(2, 6, 9, 6, 21)
    view_of = np.repeat(np.arange(d["V"]), np.diff(d["view_off"]))
    keep = (d["point_id"] != 7) & (view_of != 3)
    vo = np.concatenate([[0], np.cumsum(np.bincount(view_of[keep], minlength=d["V"]))]).astype(np.int32)
    args = (d["cam_of_view"], vo, d["point_id"][keep], d["obs_x"][keep], d["obs_y"][keep], 2, 54)
    g, o = cvx_amd.Calib(*args), orc.Calib(*args)
    grp, n = o.inner_ordering()
    assert grp[2 + 7] == -1 and grp[2 + 54 + 3] == -1 and np.sum(grp < 0) == 2
    x = (d["intr_init"], d["poses_init"], d["points_init"])
    g.set_params(*x)
    io, po, xo, co = o.inner_pass(*x)
    cg = g.inner_pass()
    ig, pg, xg = g.get_params()
    assert abs(cg - co) <= 1e-11 * co
    assert_params_close((ig, pg, xg), (io, po, xo), PASS_TOL, "unseen blocks")
    assert np.array_equal(xg[7], d["points_init"][7]) and np.array_equal(pg[3], d["poses_init"][3])
    g.set_params(*x)
    s = g.solve(capi.options("refine"))
    so = o.solve(*x, orc.options("refine"))[3]
    assert (s.iterations, s.termination, s.num_inner_iteration_steps) == (so.iterations, so.termination, so.num_inner_iteration_steps)
    tg, to = s.trace()["cost"], so.trace()["cost"]
    assert np.all(np.abs(tg - to) <= 1e-9 * to)
    assert abs(s.final_cost - so.final_cost) <= 1e-10 * so.final_cost
    ig, pg, xg = g.get_params()
    assert np.array_equal(xg[7], d["points_init"][7])      # the outer iteration moves an unseen point nowhere either
    g.close()
    # no views at all: nothing to minimise over, the pass returns a zero cost
    e = cvx_amd.Calib(np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0), np.zeros(0), 1, 54)
    e.set_params(d["intr_init"][:1], np.zeros((0, 6)), d["points_init"])
    assert e.inner_pass() == 0.0
    e.close()


def test_config4_size_passes(orc):
    """BASELINE config 4 (8 cameras x 1 250 views x 108 corners = 1.08 M observations): the pass at full size through
    properties -- it descends, a second one gains little, each pose block ends where the oracle's LM on that block
    alone ends (sampled views), and refine with the passes ends at the cost refine without them ends at."""
    d = cvx_amd.synth_calib(8, 1250, 12, 9, 20260004)
    assert d["N"] == 1080000
    g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], 8, 108)
    x = (d["intr_init"], d["poses_init"], d["points_init"])
    g.set_params(*x)
    c0 = g.cost()
    c1 = g.inner_pass()
    i1, p1, x1 = g.get_params()
    c2 = g.inner_pass()
    assert c1 < 1e-2 * c0 and c2 <= c1 and (c1 - c2) < 0.5 * c1
    # the sets at this shape: cameras, then points, then poses (the poses have the lowest degree).  The poses are
    # minimised last, against the improved cameras and points: the oracle's pass on a slice of views with those
    # held (fix_intrinsics, fix_points: one set, the poses) must end at the same poses
    for v0 in (0, 5 * 1250 + 600, d["V"] - 25):
        v1 = v0 + 25
        o0, o1 = int(d["view_off"][v0]), int(d["view_off"][v1])
        o = orc.Calib(d["cam_of_view"][v0:v1], d["view_off"][v0:v1 + 1] - o0, d["point_id"][o0:o1], d["obs_x"][o0:o1],
                      d["obs_y"][o0:o1], 8, 108, fix_intrinsics=True, fix_points=True)
        _, po, _, _ = o.inner_pass(i1, d["poses_init"][v0:v1], x1)
        assert rel_err(p1[v0:v1], po) < 1e-10
    on = capi.options("refine")
    g.set_params(*x)
    s = g.solve(on)
    on.inner_iterations = 0
    g.set_params(*x)
    s0 = g.solve(on)
    assert s.num_inner_iteration_steps >= 1 and s.termination == s0.termination == 3
    assert abs(s.final_cost - s0.final_cost) <= 1e-9 * s0.final_cost
    assert s.trace()["cost"][0] < s0.trace()["cost"][0]
    rms = capi.lib().cvx_calib_refine_return(s.final_cost, d["N"])
    assert abs(rms - 0.5 * np.sqrt(2) * 0.1) < 0.02 * 0.0707
    g.close()


def test_first_use_prepared_ahead_gives_the_same_solve():
    """cvx_calib_desc.prepare_inner_iterations / cvx_calib_prepare_inner_iterations: the host half of the first use
    on a host thread changes when it is made, not what it is -- summary, trace and parameters bit for bit; and a handle
    can be destroyed while that thread is still at work"""
    d = cvx_amd.synth_calib(2, 150, 12, 9, 77, ragged=3)
    args = (d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], 2, 108)
    runs = []
    for mode in ("on use", "flag", "call"):
        g = cvx_amd.Calib(*args, prepare_inner=(mode == "flag"))
        if mode == "call":
            g.prepare_inner_iterations()
            g.prepare_inner_iterations()                # a second call is a no-op
        g.set_params(d["intr_init"], d["poses_init"], d["points_init"])
        s = g.solve(capi.options("refine"))
        runs.append((s.iterations, s.termination, s.num_inner_iteration_steps, s.final_cost, s.trace()["cost"].tolist(),
                     [a.tolist() for a in g.get_params()]))
        g.close()
    assert runs[0][2] >= 1                              # the passes did run
    assert runs[1] == runs[0] and runs[2] == runs[0]
    for _ in range(3):                                  # destroyed at once: the destructor waits for the thread
        cvx_amd.Calib(*args, prepare_inner=True).close()
    big = cvx_amd.synth_calib(8, 1250, 12, 9, 20260004)
    cvx_amd.Calib(big["cam_of_view"], big["view_off"], big["point_id"], big["obs_x"], big["obs_y"], 8, 108, prepare_inner=True).close()
