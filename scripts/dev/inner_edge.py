import sys
import numpy as np
sys.path.insert(0, ".")
import cvx_amd
from cvx_amd import capi
from oracle import oracle as orc
d = cvx_amd.synth_calib(2, 6, 9, 6, 21)
view_of = np.repeat(np.arange(d["V"]), np.diff(d["view_off"]))
for name, keep in (("no point 7", d["point_id"] != 7), ("no view 3", view_of != 3), ("both", (d["point_id"] != 7) & (view_of != 3))):
    vo = np.concatenate([[0], np.cumsum(np.bincount(view_of[keep], minlength=d["V"]))]).astype(np.int32)
    args = (d["cam_of_view"], vo, d["point_id"][keep], d["obs_x"][keep], d["obs_y"][keep], 2, 54)
    g, o = cvx_amd.Calib(*args), orc.Calib(*args)
    x = (d["intr_init"], d["poses_init"], d["points_init"])
    for inner in (0, 1):
        g.set_params(*x)
        og = capi.options("refine"); og.inner_iterations = inner; og.max_iterations = 3
        oo = orc.options("refine"); oo.inner_iterations = inner; oo.max_iterations = 3
        s = g.solve(og); so = o.solve(*x, oo)[3]
        print(name, "inner", inner, s.trace()["cost"], so.trace()["cost"])
    # the candidate of the first step, then a pass on it
    g.set_params(*x); g.init_scaling(True)
    st = g.linear_step(1e-4, 0.0, 1e-6, 1e32)
    print("  cand", st["cand_cost"])
    g.close()
