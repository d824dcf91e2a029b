"""Distribution of cvx_calib_create's duration with and without the host-thread preparation (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cvx_amd
SPEC = {"1": (1, 20, 9, 6, 20260001), "3": (2, 500, 12, 9, 20260003), "4": (8, 1250, 12, 9, 20260004)}[os.environ.get("CVX_CONFIG", "1")]
d = cvx_amd.synth_calib(*SPEC)
for ahead in (False, True, False, True):
    ts = []
    for _ in range(30):
        t = time.perf_counter()
        g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], SPEC[0], SPEC[2] * SPEC[3], prepare_inner=ahead)
        ts.append(1e3 * (time.perf_counter() - t))
        g.close()
    ts = np.array(ts)
    print("ahead" if ahead else "plain", "median %.2f  p90 %.2f  max %.2f ms" % (np.median(ts), np.percentile(ts, 90), ts.max()), np.round(np.sort(ts)[-4:], 2))
