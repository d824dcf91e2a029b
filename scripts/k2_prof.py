"""Per-phase cycle counts of k_normal_eq (library built with -DK2_PROF; development aid)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvx_amd  # noqa: E402
from cvx_amd import capi  # noqa: E402

d = cvx_amd.synth_calib(8, 1250, 12, 9, 20260004)
g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], 8, 108)
o = capi.options("refine")
o.inner_iterations = 0      # the LM iteration itself
o.function_tolerance = o.gradient_tolerance = o.parameter_tolerance = -1.0
o.max_iterations = 3
g.set_params(d["intr_init"], d["poses_init"], d["points_init"])
g.solve(o)
g.time_schur(1)      # the last k_normal_eq launch is then the damped form (MODE 0), not the gradient-only pass
L = capi.lib()
L.cvx_calib_debug_group_scalars.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
L.cvx_calib_debug_group_scalars.restype = C.c_int
buf = np.zeros(4 * 4096)
n = L.cvx_calib_debug_group_scalars(g.h, buf.ctypes.data_as(C.POINTER(C.c_double)))
ph = buf[: 4 * n].reshape(n, 4)[:, 3]
names = ["loads + jacobian", "point accumulators", "stage + gram mfma", "barriers + tile sum", "damp + factor", "camera columns + records", "point columns of Z"]
views = d["V"] / n
tot = 0
for k in range(7):
    v = ph[k::8].mean() / views
    tot += v
    print("%-22s %8.0f cycles per view" % (names[k], v))
print("total %.0f cycles per view (%.1f views per group, %d groups)" % (tot, views, n))
