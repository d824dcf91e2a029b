"""What one drop-in refine() call costs end to end at config 4 (development aid): handle creation (allocations, upload
of the observations), parameter upload, the solve with the reference's options (first use of the inner iterations
included), download, destruction."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvx_amd  # noqa: E402
from cvx_amd import capi  # noqa: E402

SPEC = {"1": (1, 20, 9, 6, 20260001), "3": (2, 500, 12, 9, 20260003), "4": (8, 1250, 12, 9, 20260004)}[os.environ.get("CVX_CONFIG", "4")]
d = cvx_amd.synth_calib(*SPEC)
x = (d["intr_init"], d["poses_init"], d["points_init"])
import numpy as np  # noqa: E402
oxf, oyf = d["obs_x"].astype(np.float32), d["obs_y"].astype(np.float32)     # what the reference holds (cv::Point2f)
assert np.array_equal(oxf.astype(np.float64), d["obs_x"])
for rep in range(6):
    ahead = rep >= 3        # the adaptor's way: the inner iterations' host preparation starts inside create
    t0 = time.perf_counter()
    ox, oy = (oxf, oyf) if ahead else (d["obs_x"], d["obs_y"])
    g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], ox, oy, SPEC[0], SPEC[2] * SPEC[3], prepare_inner=ahead)
    t1 = time.perf_counter()
    g.set_params(*x)
    t2 = time.perf_counter()
    s = g.solve(capi.options("refine"))
    t3 = time.perf_counter()
    g.get_params()
    t4 = time.perf_counter()
    g.close()
    t5 = time.perf_counter()
    print(("ahead  " if ahead else "on use ") + "create %.2f | set %.2f | solve %.2f (%d iterations) | get %.2f | destroy %.2f | total %.2f ms" % (
        1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), s.iterations, 1e3 * (t4 - t3), 1e3 * (t5 - t4), 1e3 * (t5 - t0)))
