"""CPU: the oracle's restatement of the tool-level bundle adjustment (hand_eye_registration.cpp:48-238 =
camera_ba.cpp:47-240).  PARITY UNPINNED against the reference: the tools need Ceres, Eigen and OpenCV, none of
which is in the image, and the reference holds no vectors for them.  What can be checked is internal
consistency: analytic Jacobians vs forward-mode duals over the literal functor (what Ceres' autodiff computes),
the robustified gradient vs finite differences of 1/2 sum rho, Huber's published values, and convergence to
the generating pose in the presence of outliers."""
import numpy as np

from test_helpers import toolba_problem


def test_analytic_jacobian_matches_dual_numbers(orc):
    rng = np.random.default_rng(1)
    d = toolba_problem(rng, 6)
    worst = 0.0
    for i in range(0, len(d["point_id"]), 7):
        v = np.searchsorted(d["view_off"], i, side="right") - 1
        q = d["p0"][:4] * rng.uniform(0.5, 2.0)                   # the functor normalises q itself
        args = (d["intr"], d["trans"][v], q, d["p0"][4:], d["points"][d["point_id"][i]], d["obs_x"][i], d["obs_y"][i])
        a = orc.toolba_functor(*args); b = orc.toolba_functor(*args, autodiff=True)
        for x, y in zip(a, b):
            worst = max(worst, float(np.max(np.abs(x - y)) / max(1.0, np.max(np.abs(y)))))
    assert worst < 1e-12


def test_huber_values(orc):
    import ctypes as C
    rho = np.zeros(3)
    p = rho.ctypes.data_as(orc.c_double_p)
    orc.lib().orc_huber(1.0, 0.25, p); assert rho.tolist() == [0.25, 1.0, 0.0]
    orc.lib().orc_huber(1.0, 4.0, p); assert rho.tolist() == [3.0, 0.5, -0.0625]       # 2 a sqrt(s) - a^2, a/sqrt(s)
    orc.lib().orc_huber(0.0, 4.0, p); assert rho.tolist() == [4.0, 1.0, 0.0]           # no loss function


def test_gradient_is_the_derivative_of_the_robust_cost(orc):
    rng = np.random.default_rng(2)
    d = toolba_problem(rng, 12, outliers=0.1)
    o = orc.ToolBA(d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], d["trans"], d["intr"], d["P"])
    H, g, c = o.normal_eq(d["p0"], d["points"])
    assert abs(c - o.cost(d["p0"], d["points"])) <= 1e-12 * c
    assert np.allclose(H, H.T)
    for k in range(3):                                             # translation
        e = np.zeros(7); e[4 + k] = 1e-6
        fd = (o.cost(d["p0"] + e, d["points"]) - o.cost(d["p0"] - e, d["points"])) / 2e-6
        assert abs(fd - g[3 + k]) <= 1e-6 * abs(g[3 + k]) + 1e-3
    for pid, k in ((0, 0), (17, 1), (53, 2)):                      # board points
        X = d["points"].copy(); X[pid, k] += 1e-6; cp = o.cost(d["p0"], X)
        X[pid, k] -= 2e-6; cm = o.cost(d["p0"], X)
        assert abs((cp - cm) / 2e-6 - g[6 + 3 * pid + k]) <= 1e-5 * abs(g[6 + 3 * pid + k]) + 1e-3
    # rotation: tangent direction through QuaternionParameterization::Plus
    import ctypes as C
    for k in range(3):
        dl = np.zeros(3); dl[k] = 1e-6; qp = np.zeros(4); qm = np.zeros(4)
        dp = lambda a: a.ctypes.data_as(orc.c_double_p)
        q0 = np.ascontiguousarray(d["p0"][:4])
        orc.lib().orc_quaternion_plus(dp(q0), dp(dl), dp(qp)); orc.lib().orc_quaternion_plus(dp(q0), dp(-dl), dp(qm))
        fd = (o.cost(np.concatenate([qp, d["p0"][4:]]), d["points"]) - o.cost(np.concatenate([qm, d["p0"][4:]]), d["points"])) / 2e-6
        assert abs(fd - g[k]) <= 1e-6 * abs(g[k]) + 1e-3


def test_solve_recovers_the_pose_despite_outliers(orc):
    rng = np.random.default_rng(3)
    d = toolba_problem(rng, 40, outliers=0.05)
    o = orc.ToolBA(d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], d["trans"], d["intr"], d["P"])
    p, pts, s = o.solve(d["p0"], d["points"], orc.options("toolba"))
    assert s.final_cost < 0.2 * s.initial_cost and s.iterations <= 100
    q = p[:4] / np.linalg.norm(p[:4])
    assert abs(abs(q @ d["p_gt"][:4]) - 1.0) < 1e-6 and np.max(np.abs(p[4:] - d["p_gt"][4:])) < 2e-4
    # without the loss function the outliers drag the estimate further away
    o2 = orc.ToolBA(d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], d["trans"], d["intr"], d["P"], huber=0.0)
    p2, _, _ = o2.solve(d["p0"], d["points"], orc.options("toolba"))
    assert np.linalg.norm(p2[4:] - d["p_gt"][4:]) > np.linalg.norm(p[4:] - d["p_gt"][4:])
