"""First-use cost of the inner iterations at config 4 (ordering, per-point lists, uploads): handle creation, the first
pass, a later pass (development aid)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvx_amd  # noqa: E402

d = cvx_amd.synth_calib(8, 1250, 12, 9, 20260004)
x = (d["intr_init"], d["poses_init"], d["points_init"])
for rep in range(3):
    t0 = time.perf_counter()
    g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], 8, 108)
    t1 = time.perf_counter()
    g.set_params(*x)
    t2 = time.perf_counter()
    g.inner_pass()
    t3 = time.perf_counter()
    g.set_params(*x)
    t4 = time.perf_counter()
    g.inner_pass()
    t5 = time.perf_counter()
    print("create %.2f ms | first pass %.2f ms | later pass %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t3 - t2), 1e3 * (t5 - t4)))
    g.close()
