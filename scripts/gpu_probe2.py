import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, cvx_amd
from cvx_amd import capi
from oracle import oracle as orc
d = cvx_amd.synth_calib(2, 4, 9, 6, 14, ragged=2)
args = (d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], 2, 54)
fix = {"fix_intrinsics": True, "fix_points": True}
g = cvx_amd.Calib(*args, **fix); o = orc.Calib(*args, **fix)
x = (d["intr_init"], d["poses_init"], d["points_init"])
g.set_params(*x)
s = g.solve(capi.options("levmar")); io, po, xo, so = o.solve(*x, orc.options("levmar"))
print(np.diff(d["view_off"]))
tg, to = s.trace(), so.trace()
for i in range(min(s.trace_len, so.trace_len)):
    print(i, "%.15g %.15g %.2e acc %d %d mu %.3e %.3e" % (tg["cost"][i], to["cost"][i], abs(tg["cost"][i]-to["cost"][i])/to["cost"][i], tg["accepted"][i], to["accepted"][i], tg["damping"][i], to["damping"][i]))
print(s.iterations, so.iterations, s.termination, so.termination, s.final_cost, so.final_cost)
