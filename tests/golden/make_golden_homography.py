"""Generates tests/golden/homography_lm.json: the reference's own levmar 2.5 (dlevmar_der, compiled from
/root/reference into oracle/_ref/liblevmar_ref.so by oracle/Makefile) run on the homography functor
(homography.cpp:166-207, restated in oracle/cvx_oracle.c because the reference file needs Eigen + OpenCV) with
the options LMSolver::minimizeDer passes (lm.hpp:61).  Run in the build container only:
    python tests/golden/make_golden_homography.py
Inputs are float32 values (cv::Point2f) stored exactly as Python floats; outputs are doubles (repr round-trips)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import oracle as orc  # noqa: E402
from test_helpers import homography_problem  # noqa: E402


def main():
    rng = np.random.default_rng(20260004)
    cases = []
    spec = [("minimal_4_points", 4, {}), ("small_path_20", 20, {}), ("small_path_63", 63, {}),
            ("large_path_64", 64, {}), ("board_9x6", 54, {}), ("board_12x9", 108, {}),
            ("ragged_77", 77, {}), ("noise_free_108", 108, {"noise": 0.0}),
            ("far_start_108", 108, {"perturb": 8e-2}), ("many_points_300", 300, {}),
            ("too_few_3", 3, {})]
    for name, N, kw in spec:
        x1, y1, x2, y2, H0 = homography_problem(rng, N, **kw)
        cases.append((name, x1, y1, x2, y2, H0, 100))
    # a start whose projective row makes w <= eps for some points (homography.cpp:177 guard), few iterations
    x1, y1, x2, y2, H0 = homography_problem(rng, 40)
    H0 = H0.copy(); H0[6] = -1.0 / 100.0; H0[7] = 0.0        # w = 1 - x/100 <= 0 for x >= 100
    cases.append(("guarded_w", x1, y1, x2, y2, H0, 100))
    x1, y1, x2, y2, H0 = homography_problem(rng, 108)
    cases.append(("iteration_cap_3", x1, y1, x2, y2, H0, 3))
    out = {}
    for name, x1, y1, x2, y2, H0, itmax in cases:
        if len(x1) >= 4:
            ret, h, info = orc.homography_lm(x1, y1, x2, y2, H0, itmax=itmax, engine="reference")
        else:
            ret, h, info = -1, np.array(H0[:8]), np.zeros(10)
        out[name] = {"x1": [float(v) for v in x1], "y1": [float(v) for v in y1], "x2": [float(v) for v in x2],
                     "y2": [float(v) for v in y2], "H0": [float(v) for v in H0], "itmax": itmax,
                     "opts": list(orc.LEVMAR_OPTS), "ret": int(ret), "h": [float(v) for v in h],
                     "info": [float(v) for v in info]}
        print("%-18s N=%3d ret=%3d stop=%d |e|^2 %.6g -> %.6g" % (name, len(x1), ret, info[6], info[0], info[1]))
    with open(os.path.join(HERE, "homography_lm.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
