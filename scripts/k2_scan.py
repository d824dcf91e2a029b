"""k_normal_eq at 1, 2, 5 and 10 views per workgroup with the same number of workgroups (development aid: run under
rocprofv3 --kernel-trace; the kernel's duration against the views per workgroup separates its fixed part from the
time a view takes on its critical path)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvx_amd  # noqa: E402
from cvx_amd import capi  # noqa: E402

for vpc in (128, 256, 640, 1250):
    d = cvx_amd.synth_calib(8, vpc, 12, 9, 20260004)
    g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], 8, 108)
    o = capi.options("refine")
    o.inner_iterations = 0      # the LM iteration itself
    o.function_tolerance = o.gradient_tolerance = o.parameter_tolerance = -1.0
    o.max_iterations = 4
    g.set_params(d["intr_init"], d["poses_init"], d["points_init"])
    s = g.solve(o)
    print("views per camera", vpc, "iterations", s.iterations, flush=True)
