"""Wall-clock stamps of k_normal_eq's workgroups (library built with -DK2_WALL; development aid): when they start,
leave the prologue, leave the view loop and end, in microseconds from the first start."""
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
g.time_schur(1)      # the last k_normal_eq launch is then the damped form (MODE 0)
L = capi.lib()
L.cvx_calib_debug_group_scalars.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
L.cvx_calib_debug_group_scalars.restype = C.c_int
buf = np.zeros(4 * 4096)
n = L.cvx_calib_debug_group_scalars(g.h, buf.ctypes.data_as(C.POINTER(C.c_double)))
t = buf[: 4 * n].reshape(n, 4)[:, 3] * 0.01      # 100 MHz -> us
t0 = t[0::4].min()
for k, name in enumerate(["start", "prologue done", "view loop done", "end"]):
    x = t[k::4] - t0
    print("%-15s min %7.2f  mean %7.2f  max %7.2f us" % (name, x.min(), x.mean(), x.max()))
x = t[2::4] - t0
gi = np.arange(n)[2::4]
print("loop-done percentiles 5/25/50/75/95:", np.percentile(x, [5, 25, 50, 75, 95]).round(1))
for m in (8, 32, 256):
    print("mean by group index mod %d:" % m, [round(float(x[(gi % m) == r].mean()), 1) if ((gi % m) == r).any() else None for r in range(0, m, max(1, m // 16))])
print("first 40 by group:", [(int(a), round(float(b), 1)) for a, b in zip(gi[:40], x[:40])])
