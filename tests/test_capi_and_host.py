"""CPU: the C-ABI library loads and exports every symbol include/cvx_amd.h declares, host-side logic
(conversions, pair construction, partitioning, synthetic workloads, option presets), and loud failure
without a device.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import cvx_amd
from cvx_amd import capi
from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cvx_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cvx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = C.CDLL(capi.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(L, n), "libcvx_amd.so does not export %s" % n
    # and the ctypes table covers exactly the header
    assert sorted(capi.SIGNATURES) == names


def test_product_does_not_touch_the_oracle():
    # the product path may not include, link or load anything under oracle/
    import subprocess
    for root, _, files in os.walk(os.path.join(ROOT, "cvx_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "oracle" not in txt.lower() or f == "capi.py" and False, os.path.join(root, f)
    out = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_status_strings_and_option_presets():
    L = capi.lib()
    assert L.cvx_status_string(0) == b"ok"
    assert b"device" in L.cvx_status_string(capi.ERR_NO_DEVICE)
    o = capi.options("refine")      # calibration.cpp:420-428
    assert (o.strategy, o.max_iterations, o.nonmonotonic) == (0, 100, 1)
    assert o.function_tolerance == 1e-10 and o.gradient_tolerance == 1e-10 and o.parameter_tolerance == 1e-8
    assert o.initial_radius == 1e4 and o.min_lm_diagonal == 1e-6 and o.max_lm_diagonal == 1e32 and o.jacobi_scaling == 1
    o = capi.options("handeye")     # handeye.cpp:310-319
    assert (o.strategy, o.max_iterations, o.nonmonotonic) == (0, 100, 0) and o.function_tolerance == 1e-6
    o = capi.options("levmar")      # lm.hpp:43-50,61
    assert o.strategy == 1 and o.tau == 1e-3 and o.eps1 == o.eps2 == o.eps3 == np.finfo(float).eps


def test_options_match_the_oracle_presets(orc):
    for kind in ("refine", "handeye", "levmar"):
        a, b = capi.options(kind), orc.options(kind)
        for name, _ in orc.LMOptions._fields_:
            assert getattr(a, name) == getattr(b, name), (kind, name)


def test_no_device_is_a_loud_error():
    if capi.lib().cvx_device_count() > 0:
        pytest.skip("a device is present")
    d = cvx_amd.synth_calib(1, 2, 3, 3, 1)
    with pytest.raises(capi.CvxError) as e:
        cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], 1, 9)
    assert e.value.status == capi.ERR_NO_DEVICE
    with pytest.raises(capi.CvxError):
        cvx_amd.HandEye(np.ones((2, 4)), np.ones((2, 3)), np.ones((2, 4)), np.ones((2, 3)))


def test_rotation_conversions_round_trip_and_match_oracle(orc):
    L = capi.lib()
    rng = np.random.default_rng(0)
    for k in range(200):
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        ang = [0.0, 1e-9, 1e-4, 3.1][k] if k < 4 else rng.uniform(0, np.pi)
        w = ax * ang
        R = np.zeros(9); Ro = np.zeros(9); w2 = np.zeros(3); wo = np.zeros(3)
        L.cvx_angle_axis_to_rotation(capi.pd(w), capi.pd(R))
        orc.lib().orc_angle_axis_to_rotation(capi.pd(w), capi.pd(Ro))
        assert np.array_equal(R, Ro)
        Rm = R.reshape(3, 3).T   # column-major storage
        assert np.allclose(Rm @ Rm.T, np.eye(3), atol=1e-14 if ang > 1e-6 else 1e-8)
        L.cvx_rotation_to_angle_axis(capi.pd(R), capi.pd(w2))
        orc.lib().orc_rotation_to_angle_axis(capi.pd(R), capi.pd(wo))
        assert np.allclose(w2, wo, atol=1e-15)
        if ang > 1e-6:
            assert np.allclose(w2, w, atol=1e-9)
        q = np.zeros(4)
        L.cvx_rotation_to_quaternion(capi.pd(R), capi.pd(q))
        assert abs(np.linalg.norm(q) - 1) < 1e-12
        T = np.zeros(16); p = np.concatenate([q, [1.0, 2.0, 3.0]])
        L.cvx_quat_trans_to_affine(capi.pd(p), capi.pd(T))
        assert np.allclose(T.reshape(4, 4).T[:3, :3], Rm, atol=1e-12)
        assert np.allclose(T.reshape(4, 4).T[:3, 3], [1, 2, 3])


def test_affine_rotation_is_the_polar_factor():
    """cvx_affine_to_quat_trans restates Quaterniond(Affine3d::rotation()) (handeye.cpp:287-290): rotation() is the
    orthogonal polar factor of the linear part (Eigen's computeRotationScaling: U diag(1/x, 1, 1) V^T, x = det(U V^T)).
    Rigid inputs -- exact rotations and products of them, as the motion pairs are -- keep the bits of the plain
    conversion; a noisy 4x4 (a pose_N.txt written with six digits) gives the quaternion of its nearest rotation."""
    from cvx_amd.capi import lib, pd
    L = lib()
    rng = np.random.default_rng(3)

    def rot(w):
        R = np.zeros(9)
        L.cvx_angle_axis_to_rotation(pd(np.asarray(w, float)), pd(R))
        return R.reshape(3, 3).T                               # column-major -> numpy

    def quat_of_affine(M):
        T = np.ascontiguousarray(M.T).reshape(-1)              # column-major
        q, t = np.zeros(4), np.zeros(3)
        L.cvx_affine_to_quat_trans(pd(T), pd(q), pd(t))
        return q, t

    def quat_of_rotation(R):
        q = np.zeros(4)
        L.cvx_rotation_to_quaternion(pd(np.ascontiguousarray(R.T).reshape(-1)), pd(q))
        return q

    def eigen_rotation(A):                                     # Transform::computeRotationScaling, Eigen 3.3
        U, s, Vt = np.linalg.svd(A)
        x = np.linalg.det(U @ Vt)
        U = U.copy(); U[:, 0] /= x
        return U @ Vt

    for k in range(50):
        Ra, Rb = rot(rng.normal(size=3)), rot(rng.normal(size=3) * 2.0)
        for R in (Ra, Ra @ Rb.T, Ra @ Rb @ Ra):                # exact, and products as A = G_j G_i^-1 are
            M = np.eye(4); M[:3, :3] = R; M[:3, 3] = rng.normal(size=3)
            q, t = quat_of_affine(M)
            assert np.array_equal(q, quat_of_rotation(R)) and np.array_equal(t, M[:3, 3])
        # noisy linear part: rounded to six digits, and a grossly non-rigid one (scale + shear)
        for A in (np.round(Ra, 6), Ra @ (np.eye(3) + 0.05 * rng.normal(size=(3, 3))), 1.7 * Rb):
            M = np.eye(4); M[:3, :3] = A; M[:3, 3] = rng.normal(size=3)
            q, t = quat_of_affine(M)
            assert abs(np.linalg.norm(q) - 1.0) < 1e-14        # an orthonormal R gives a unit quaternion
            Rq = np.zeros(16)
            L.cvx_quat_trans_to_affine(pd(np.concatenate([q, t])), pd(Rq))
            Rq = Rq.reshape(4, 4).T[:3, :3]
            assert np.max(np.abs(Rq.T @ Rq - np.eye(3))) < 1e-14 and np.linalg.det(Rq) > 0
            assert np.max(np.abs(Rq - eigen_rotation(A))) < 1e-13
            assert np.max(np.abs(quat_of_rotation(A) - q)) > 1e-9       # the plain conversion would differ
    # a reflection: Eigen 3.3 flips U's first column
    A = rot([0.3, -0.2, 0.5]) @ np.diag([1.0, 1.0, -1.0]) @ (np.eye(3) + 0.01 * rng.normal(size=(3, 3)))
    M = np.eye(4); M[:3, :3] = A
    q, _ = quat_of_affine(M)
    Rq = np.zeros(16)
    L.cvx_quat_trans_to_affine(pd(np.concatenate([q, np.zeros(3)])), pd(Rq))
    assert np.max(np.abs(Rq.reshape(4, 4).T[:3, :3] - eigen_rotation(A))) < 1e-13


def test_synthetic_calibration_workload_is_deterministic_and_float32_representable():
    a = cvx_amd.synth_calib(2, 5, 12, 9, 20260003)
    b = cvx_amd.synth_calib(2, 5, 12, 9, 20260003)
    for k in ("obs_x", "obs_y", "poses_init", "intr_init", "point_id", "view_off"):
        assert np.array_equal(a[k], b[k])
    assert a["N"] == 2 * 5 * 108 and a["view_off"][-1] == a["N"]
    # observations and board points come from cv::Point2f / Point3f (pattern.hpp:18-24, 40)
    assert np.array_equal(a["obs_x"], a["obs_x"].astype(np.float32).astype(np.float64))
    assert np.array_equal(a["points_gt"], a["points_gt"].astype(np.float32).astype(np.float64))
    # OCVCalibrationPattern layout, pattern.hpp:53-55
    assert np.allclose(a["points_gt"][13], [25.0, 25.0, 0.0])
    r = cvx_amd.synth_calib(2, 5, 12, 9, 20260003, ragged=3)
    assert 8 * 10 <= r["N"] < a["N"]
    n_per_view = np.diff(r["view_off"])
    assert n_per_view.min() >= 8 and len(set(n_per_view.tolist())) > 1
    assert np.array_equal(r["poses_gt"], a["poses_gt"])


def test_motion_pairs_obey_ax_xb_without_noise():
    for moving in (False, True):
        G, T, xg, x0 = cvx_amd.synth_handeye(30, 5, rot_noise=0.0, trans_noise=0.0, moving=moving)
        A, B = cvx_amd.handeye_pairs(G, T, moving=moving)
        if moving:
            assert len(A) == 29                      # consecutive pairs of the sorted stations, handeye.cpp:742-746
        else:
            assert 0 < len(A) <= 30 * 29 // 2        # all i<j beyond 10 degrees, handeye.cpp:681-693
        X = np.zeros(16)
        capi.lib().cvx_quat_trans_to_affine(capi.pd(xg), capi.pd(X))
        X = X.reshape(4, 4).T
        for a, b in zip(A[:50], B[:50]):
            a = a.reshape(4, 4).T; b = b.reshape(4, 4).T
            assert np.allclose(a @ X, X @ b, atol=1e-9)


def test_partition_views_balances_observations():
    d = cvx_amd.synth_calib(3, 40, 12, 9, 7, ragged=3)
    for nr in (1, 2, 3, 8):
        first = cvx_amd.partition_views(d["view_off"], nr)
        assert first[0] == 0 and first[-1] == d["V"] and np.all(np.diff(first) >= 0)
        counts = [d["view_off"][first[r + 1]] - d["view_off"][first[r]] for r in range(nr)]
        assert sum(counts) == d["N"]
        assert max(counts) - min(counts) <= 2 * 108
    # more ranks than views: empty shards are legal
    first = cvx_amd.partition_views(np.array([0, 5, 9], dtype=np.int32), 4)
    assert first[0] == 0 and first[-1] == 2 and np.all(np.diff(first) >= 0)


def test_refine_return_value():
    # sqrt(total_cost / (2 N)), calibration.cpp:463
    assert capi.lib().cvx_calib_refine_return(8.0, 4) == 1.0


def test_homography_batch_argument_checks():
    """host-side validation happens before any device work (include/cvx_amd.h, CVX_HOMOGRAPHY_MAX_POINTS)"""
    from cvx_amd.problem import HomographyBatch
    z = np.zeros(1025, np.float32)
    with pytest.raises(capi.CvxError) as e:
        HomographyBatch([0, 1025], z, z, z, z)
    assert e.value.status == capi.ERR_UNSUPPORTED
    with pytest.raises(capi.CvxError) as e:
        HomographyBatch([1, 5], z, z, z, z)
    assert e.value.status == capi.ERR_INVALID_ARG
    with pytest.raises(capi.CvxError) as e:
        HomographyBatch([0, 5, 3], z, z, z, z)
    assert e.value.status == capi.ERR_INVALID_ARG


def test_lm_driver_hands_back_the_lowest_cost_point():
    """The trust-region loop with non-monotonic steps (calibration.cpp:421) may accept an uphill step; what the solve
    returns is the lowest-cost point it visited (documented behaviour of ceres use_nonmonotonic_steps).  Scripted
    evaluator, host only: tests/cpp/lm_driver_test.cpp."""
    import json
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "lm_driver_test")
    assert os.path.exists(exe), "run __graft_entry__.build() (make -C cvx_amd/csrc)"
    rows = {r["name"]: r for r in map(json.loads, subprocess.check_output([exe], timeout=60).decode().splitlines())}
    assert all(r["status"] == 0 and r["iterations"] == 4 for r in rows.values())
    m = rows["monotone"]
    assert (m["final_cost"], m["at"], m["saved"], m["restored"], m["successful"]) == (60.0, 3, 0, 0, 4)
    u = rows["uphill_then_end"]      # 50, 40, 45 (uphill, accepted), 44: back to the 40
    assert (u["final_cost"], u["at"], u["saved"], u["restored"], u["successful"]) == (40.0, 1, 1, 1, 4)
    r = rows["monotonic_rejects"]    # the same costs with monotonic steps: 45 and 44 are never accepted
    assert (r["final_cost"], r["at"], r["saved"], r["restored"], r["successful"]) == (40.0, 1, 0, 0, 2)
    b = rows["uphill_then_better"]   # a later, lower minimum makes the saved copy stale: not restored
    assert (b["final_cost"], b["at"], b["restored"]) == (30.0, 3, 0)
    assert all(r["inner_calls"] == 0 for k, r in rows.items() if k != "inner_iterations")
    i = rows["inner_iterations"]     # a pass that ends below the current cost makes the step; passes stop when they stop paying
    assert (i["final_cost"], i["at"], i["successful"], i["inner_calls"], i["inner_steps"]) == (93.0, 3, 3, 3, 3)


def test_schur_task_list_covers_every_block_and_every_stage_row_once():
    """k4_build_tasks (calib_kernels.h): for every width k_schur_syrk4 takes, each 16x16 block of the upper triangle
    belongs to exactly one task, and in every workgroup the sixteen rows of a stage are copied into LDS exactly once,
    never by a DIAG wave (which has no registers to spare for it).  Host only: tests/cpp/k4_tasks_test.cpp."""
    import json
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "k4_tasks_test")
    assert os.path.exists(exe), "run __graft_entry__.build() (make -C cvx_amd/csrc)"
    p = subprocess.run([exe], capture_output=True, timeout=60)
    rows = [json.loads(line) for line in p.stdout.decode().splitlines()]
    assert p.returncode == 0, rows
    assert [r["nb"] for r in rows] == list(range(1, 26))
    assert all(r["blocks_wrong"] == 0 and r["copy_wrong"] == 0 and r["diag_copies"] == 0 for r in rows)
    assert {r["nb"]: r["parts"] for r in rows}[21] == 2      # config 4: 15 tasks on 2 x 8 waves


def test_public_header_is_plain_c(tmp_path):
    """the boundary is a C ABI: include/cvx_amd.h compiles as C99 with -pedantic, no C++ needed to bind it"""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "cvx_amd.h"\nint main(void) { cvx_ransac_params p; cvx_ransac_default_params(&p); '
                   'cvx_lm_options o; cvx_lm_options_refine(&o); return p.max_num_trials + o.max_iterations > 0 ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                        "-o", str(tmp_path / "t.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_ransac_defaults_are_the_reference_parameters():
    p = capi.RansacParams()
    capi.lib().cvx_ransac_default_params(C.byref(p))
    assert (p.max_error, p.outlier_probability, p.max_num_trials, p.min_inliers) == (0.1, 0.1, 100, 5)     # ransac.hpp:14-22

