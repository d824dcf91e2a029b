"""A few LM iterations of config 4 (development aid: run under rocprofv3 --kernel-trace --stats to time k_normal_eq
of the library CVX_AMD_LIB points at)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvx_amd  # noqa: E402
from cvx_amd import capi  # noqa: E402

d = cvx_amd.synth_calib(8, 1250, 12, 9, 20260004)
g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], 8, 108)
o = capi.options("refine")
o.inner_iterations = 0      # the LM iteration itself
o.function_tolerance = o.gradient_tolerance = o.parameter_tolerance = -1.0
o.max_iterations = 6
for rep in range(3):
    g.set_params(d["intr_init"], d["poses_init"], d["points_init"])
    try:
        s = g.solve(o)
        print("solve:", s.iterations, s.final_cost)
    except Exception as e:  # ablation builds produce nonsense on purpose
        print("solve raised", e)
