"""One inner-iteration pass (cvx_calib_inner_pass) at BASELINE config 4 from the perturbed start, repeated: wall time
per pass; run under rocprofv3 --kernel-trace --stats for the per-kernel split (profiles/r02f_inner_*)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvx_amd
d = cvx_amd.synth_calib(8, 1250, 12, 9, 20260004)
g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], 8, 108)
x = (d["intr_init"], d["poses_init"], d["points_init"])
g.set_params(*x); g.inner_pass()
ts = []
for _ in range(10):
    g.set_params(*x)
    t = time.perf_counter(); c = g.inner_pass(); ts.append(time.perf_counter() - t)
print("pass from the start: %.3f ms (min %.3f), cost %.6g" % (1e3 * sum(ts) / len(ts), 1e3 * min(ts), c))
t = time.perf_counter(); c2 = g.inner_pass(); print("second pass: %.3f ms, cost %.6g" % (1e3 * (time.perf_counter() - t), c2))
g.close()
