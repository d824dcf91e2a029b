"""GPU tests of the BASELINE configurations the other files do not reach -- config 3 (stereo rig, 2 x 500 views of a
12x9 board) as a full solve against the oracle, config 5 (100 000 views, 10.8 M observations, one GPU) through
size-independent properties -- plus the gauge-fixed reference-levmar trace and the file -> GPU -> file flow of
SURVEY.md 8(f1).  All through the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest

import cvx_amd
from cvx_amd import capi, formats
from conftest import GOLDEN, load_golden
from test_helpers import assert_params_close, rel_err

pytestmark = pytest.mark.gpu


def test_config3_stereo_rig_full_solve_follows_oracle(orc):
    """BASELINE config 3: 2 cameras x 500 views x 108 corners = 108 000 observations, 340 shared unknowns."""
    d = cvx_amd.synth_calib(2, 500, 12, 9, 20260003)
    assert d["N"] == 108000 and d["V"] == 1000
    args = (d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], 2, 108)
    g, o = cvx_amd.Calib(*args), orc.Calib(*args)
    assert g.ns == 2 * 8 + 3 * 108
    x = (d["intr_init"], d["poses_init"], d["points_init"])
    g.set_params(*x)
    # residuals and Jacobian blocks of every observation at identical parameters: 1e-12
    cost, r, Jc, Jp, Jx = g.evaluate()
    ro, Jco, Jpo, Jxo = o.eval(*x)
    assert rel_err(r, ro) < 1e-12 and rel_err(Jc, Jco) < 1e-12 and rel_err(Jp, Jpo) < 1e-12 and rel_err(Jx, Jxo) < 1e-12
    s = g.solve(capi.options("refine"))
    io, po, xo, so = o.solve(*x, orc.options("refine"))
    ig, pg, xg = g.get_params()
    assert abs(s.initial_cost - so.initial_cost) <= 1e-13 * so.initial_cost
    assert (s.iterations, s.termination) == (so.iterations, so.termination)
    tg, to = s.trace()["cost"], so.trace()["cost"]
    assert len(tg) == len(to)
    assert np.all(np.abs(tg[:5] - to[:5]) <= 1e-12 * to[:5])        # per-iteration costs, north_star's 1e-12
    assert np.all(np.abs(tg - to) <= 1e-10 * to)                    # all of them
    assert np.array_equal(s.trace()["accepted"], so.trace()["accepted"])
    assert abs(s.final_cost - so.final_cost) <= 1e-12 * so.final_cost
    rg = capi.lib().cvx_calib_refine_return(s.final_cost, d["N"])
    assert abs(rg - orc.lib().orc_calib_refine_return(so.final_cost, d["N"])) <= 1e-12 * rg
    # block by block; 1e-9 asked, ten times the measured values held (scripts/parity_blocks.py: focal lengths 4.6e-15,
    # distortion 5.9e-15, rotations 7.9e-13, translations 2.3e-13, points 6.6e-13)
    assert_params_close((ig, pg, xg), (io, po, xo), {"focal_pp": 5e-14, "distortion": 6e-14, "rotation": 8e-12,
                                                     "translation": 3e-12, "points": 7e-12}, "config 3")
    # and the answer is the rig that made the data
    assert np.allclose(ig[:, :4], d["intr_gt"][:, :4], rtol=3e-3)
    assert abs(rg - 0.5 * np.sqrt(2) * 0.1) < 0.03 * 0.0707
    g.close()


def test_config5_full_size_properties(orc):
    """BASELINE config 5 on ONE GPU: 8 cameras x 12 500 views = 100 000 pose blocks, 10.8 M observations (Z alone is
    1.6 GB).  The oracle would need minutes, so: size-independent properties plus direct oracle comparisons on slices."""
    d = cvx_amd.synth_calib(8, 12500, 12, 9, 20260005)
    assert d["N"] == 10800000 and d["V"] == 100000
    g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], 8, 108)
    g.set_params(d["intr_init"], d["poses_init"], d["points_init"])
    # (1) residuals of all 10.8 M observations: the cost kernel equals 1/2 sum r^2 summed on the host in long double
    N = d["N"]
    r = np.zeros((N, 2))
    c = C.c_double()
    capi.check(capi.lib().cvx_calib_evaluate(g.h, C.byref(c), capi.pd(r), None, None, None), "evaluate")
    ref = 0.5 * float(np.sum(r.astype(np.longdouble) ** 2))
    assert abs(g.cost() - ref) <= 1e-13 * ref and abs(c.value - ref) <= 1e-13 * ref
    # (2) slices against the oracle: residuals of the first and last 40 views and of 40 views in the middle of
    #     camera 3, where a handle of their own gives the Jacobian blocks too
    for v0 in (0, 3 * 12500 + 6000, d["V"] - 40):
        v1 = v0 + 40
        o0, o1 = int(d["view_off"][v0]), int(d["view_off"][v1])
        sl = (d["cam_of_view"][v0:v1], d["view_off"][v0:v1 + 1] - o0, d["point_id"][o0:o1], d["obs_x"][o0:o1],
              d["obs_y"][o0:o1], 8, 108)
        o = orc.Calib(*sl)
        ro, Jco, Jpo, Jxo = o.eval(d["intr_init"], d["poses_init"][v0:v1], d["points_init"])
        assert rel_err(r[o0:o1], ro) < 1e-12
        gs = cvx_amd.Calib(*sl)
        gs.set_params(d["intr_init"], d["poses_init"][v0:v1], d["points_init"])
        _, rs, Jc, Jp, Jx = gs.evaluate()
        assert np.array_equal(rs, r[o0:o1])         # the same observation gives the same bits in either handle
        assert rel_err(Jc, Jco) < 1e-12 and rel_err(Jp, Jpo) < 1e-12 and rel_err(Jx, Jxo) < 1e-12
        gs.close()
    del r
    # (3) additivity over view shards (how an 8-GPU run cuts the problem): sum of the shards' costs == the whole
    first = cvx_amd.partition_views(d["view_off"], 8)
    assert first[0] == 0 and first[8] == d["V"] and np.all(np.diff(first) == 12500)
    parts = 0.0
    for rank in range(8):
        sh = cvx_amd.shard_calib(d, first, rank)
        gsh = cvx_amd.Calib(sh["cam_of_view"], sh["view_off"], sh["point_id"], sh["obs_x"], sh["obs_y"], 8, 108)
        gsh.set_params(d["intr_init"], sh["poses_init"], d["points_init"])
        part = gsh.cost()
        gsh.close()
        o0, o1 = int(d["view_off"][first[rank]]), int(d["view_off"][first[rank + 1]])
        assert o1 - o0 == 1350000
        parts += part
    assert abs(parts - ref) <= 1e-13 * ref
    # (4) one damped linear step: the candidate cost the fused kernel reports equals the cost kernel's value at the
    #     candidate point, and the model predicts a decrease
    g.init_scaling(True)
    st = g.linear_step(1e-4, 0.0, 1e-6, 1e32)
    assert abs(st["cost"] - ref) <= 1e-12 * ref
    assert st["model_change"] > 0 and st["cand_cost"] < st["cost"]
    # (5) the solve: monotone accepted costs, refine()'s return value settles at the noise floor
    #     1/2 sqrt(2 sigma^2) = 0.0707 px for sigma = 0.1 px (calibration.cpp:463), the intrinsics come back
    s = g.solve(capi.options("refine"))
    tr = s.trace()
    acc = tr["cost"][tr["accepted"] > 0]
    assert s.termination in (1, 2, 3) and s.num_successful >= 5
    assert np.all(np.diff(acc) <= 1e-9 * acc[:-1])
    rms = capi.lib().cvx_calib_refine_return(s.final_cost, N)
    assert abs(rms - 0.5 * np.sqrt(2) * 0.1) < 0.02 * 0.0707
    assert abs(g.cost() - s.final_cost) <= 1e-12 * s.final_cost       # the point handed back has the reported cost
    intr = g.get_params()[0]
    assert np.allclose(intr[:, :4], d["intr_gt"][:, :4], rtol=2e-3)
    # (6) idempotence: solving again from the optimum stops at once
    s2 = g.solve(capi.options("refine"))
    assert s2.iterations <= 2 and abs(s2.final_cost - s.final_cost) <= 1e-9 * s.final_cost
    g.close()


def test_golden_levmar_trace_with_the_gauge_fixed(orc):
    """tests/golden/levmar_calib_fixed_points_trace.json is a run of the REFERENCE's levmar (dense J^T J, LU) on
    BASELINE config 1 with the board points held fixed.  Without the gauge freedom the minimiser is unique and the
    HIP path (Schur complement + Cholesky, Nielsen damping) must follow that run: every candidate cost while the run
    still moves to 1e-10, and the end point block by block.

    Where a levmar run with machine-epsilon tolerances ends is decided by rounding noise: the last twelve candidate
    costs of the reference run itself wander in the 14th digit (9e-13 of 19.655...), dF > 0 is a coin toss there, and
    the run stops when enough rejections in a row have shrunk the step.  A cost difference dc cannot tell two points
    apart that are closer than sqrt(2 dc (S^-1)_ii) in parameter i (S: the reduced normal matrix at the end point) --
    2.7e-5 px in fx, 1.2e-7 in k2, 1.9e-9 in p2.  Measured on the box (scripts/parity_blocks.py):
    the HIP run ends 1e-3 .. 3e-3 of that radius away from the reference's end point in EVERY intrinsic (the
    same fraction for all eight: a displacement along the flat direction, not an error in one block), which is
    2.5e-11 (focal lengths, principal point), 2.6e-9 (distortion block; 3e-10 absolute in k2), 6.7e-11 (rotations),
    3.1e-11 (translations) block-relative.  Held at ten times the measured values."""
    gold = load_golden("levmar_calib_fixed_points_trace.json")
    d = cvx_amd.synth_calib(**gold["spec"])
    args = (d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], d["C"], d["P"])
    g = cvx_amd.Calib(*args, fix_points=True)
    g.set_params(d["intr_init"], d["poses_init"], d["points_init"])
    s = g.solve(capi.options("levmar"))
    ref = 0.5 * np.array(gold["evals"])
    assert abs(s.initial_cost - ref[0]) <= 1e-13 * ref[0]
    tr = s.trace()["cost"]
    # candidate costs, one per linear solve, for as long as the run still moves
    live = np.nonzero(np.abs(ref[1:] - 0.5 * gold["info"][1]) > 1e-7 * 0.5 * gold["info"][1])[0]
    m = min(int(live[-1]) + 1 if len(live) else 0, len(tr))
    assert m >= 8
    assert np.all(np.abs(tr[:m] - ref[1:m + 1]) <= 1e-10 * ref[1:m + 1])
    assert s.termination == int(gold["info"][6])
    assert abs(s.final_cost - 0.5 * gold["info"][1]) <= 1e-12 * s.final_cost
    intr, poses, pts = g.get_params()
    g.close()
    exp = np.array(gold["p"])
    ie = d["intr_init"].copy()
    ie[0, [0, 1, 2, 3, 4, 5, 7, 8]] = exp[:8]
    pe = exp[8:].reshape(-1, 6)
    assert_params_close((intr, poses, pts), (ie, pe, d["points_init"]),
                        {"focal_pp": 3e-10, "distortion": 3e-8, "rotation": 7e-10, "translation": 4e-10, "points": 0.0},
                        "reference levmar, gauge fixed")
    # the distance to the reference's end point in units of what the cost can resolve there
    o = orc.Calib(*args, fix_points=True)
    S, _, _ = o.reduced_system(ie, pe, d["points_init"], np.zeros(6 * d["V"]), np.zeros(o.ns))
    tail = np.array(gold["evals"][-12:])
    dc = 0.5 * (tail.max() - tail.min())
    radius = np.sqrt(2.0 * dc * np.diag(np.linalg.inv(S)))
    frac = np.abs(intr[0, [0, 1, 2, 3, 4, 5, 7, 8]] - exp[:8]) / radius
    assert np.all(frac <= 0.03), frac                     # measured: 1e-3 .. 3e-3


def test_files_to_gpu_to_files(orc, tmp_path):
    """SURVEY.md 8(f1) end to end: a calib.data file (calibration.cpp:22-79) -> cvx_calib_data_load -> cvx_calib_create
    -> cvx_calib_solve -> PinholeCamera YAML (camera.hpp:138-180) written and read back."""
    # (a) the hand-made fixture (ragged: an empty view, a two-marker view) reaches the kernels unchanged
    f = formats.load_calib_data(os.path.join(GOLDEN, "formats", "calib_small.data"))
    V = len(f["view_off"]) - 1
    args = (np.zeros(V, np.int32), f["view_off"], f["point_id"], f["obs_x"], f["obs_y"], 1, len(f["coords"]))
    g, o = cvx_amd.Calib(*args), orc.Calib(*args)
    intr = np.array([[525.5, 524.25, 319.5, 239.5, 0.1, -0.25, 0.05, 0.001, -0.002]])
    poses = np.tile(np.array([0.1, -0.2, 0.05, -10.0, -12.0, 400.0]), (V, 1)) + 0.01 * np.arange(V)[:, None]
    g.set_params(intr, poses, f["coords"])
    cost, r, Jc, Jp, Jx = g.evaluate()
    ro, Jco, Jpo, Jxo = o.eval(intr, poses, f["coords"])
    assert rel_err(r, ro) < 1e-12 and rel_err(Jp, Jpo) < 1e-12 and rel_err(Jc, Jco) < 1e-12
    g.close()
    # (b) a data set of config 1's shape written in the reference's text format (floats at the stream's default
    #     six significant digits), read back, refined, and the camera written as the reference's YAML
    d = cvx_amd.synth_calib(1, 20, 9, 6, 20260001)
    path = str(tmp_path / "calib.data")
    names = ["img/view_%03d.png" % i for i in range(d["V"])]
    formats.save_calib_data(path, 1280, 960, d["points_gt"], names, d["view_off"], d["point_id"], d["obs_x"], d["obs_y"])
    f = formats.load_calib_data(path)
    assert (f["width"], f["height"]) == (1280, 960) and f["image_paths"] == names
    assert np.array_equal(f["view_off"], d["view_off"]) and np.array_equal(f["point_id"], d["point_id"])
    assert np.max(np.abs(f["obs_x"] - d["obs_x"])) < 5e-3            # six significant digits of a pixel coordinate
    args = (d["cam_of_view"], f["view_off"], f["point_id"], f["obs_x"], f["obs_y"], 1, len(f["coords"]))
    g, o = cvx_amd.Calib(*args), orc.Calib(*args)
    g.set_params(d["intr_init"], d["poses_init"], f["coords"])
    s = g.solve(capi.options("refine"))
    io, po, xo, so = o.solve(d["intr_init"], d["poses_init"], f["coords"], orc.options("refine"))
    ig, pg, xg = g.get_params()
    g.close()
    assert (s.iterations, s.termination) == (so.iterations, so.termination)
    assert abs(s.final_cost - so.final_cost) <= 1e-12 * so.final_cost
    assert_params_close((ig, pg, xg), (io, po, xo), {"focal_pp": 2e-13, "distortion": 2e-12, "translation": 6e-11}, "calib.data")
    K = ig[0, :4]
    dist = ig[0, [4, 5, 7, 8, 6]]                                    # a1 order -> OpenCV's k1 k2 p1 p2 k3 (calibration.cpp:452-456)
    ypath = str(tmp_path / "camera.yaml")
    formats.write_camera_yaml(ypath, f["width"], f["height"], K, dist)
    w, h, K2, dist2 = formats.read_camera_yaml(ypath)
    assert (w, h) == (1280, 960)
    assert np.array_equal(K2, K) and np.array_equal(dist2, dist)     # the emitter writes round-trip digits
    assert np.allclose(K2, d["intr_gt"][0, :4], rtol=1e-2)
