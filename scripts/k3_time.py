"""Times k_schur_syrk alone at BASELINE config 4 (development aid).  Usage: python scripts/k3_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvx_amd  # noqa: E402
from cvx_amd import capi  # noqa: E402

d = cvx_amd.synth_calib(8, 1250, 12, 9, 20260004)
g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], 8, 108)
g.set_params(d["intr_init"], d["poses_init"], d["points_init"])
o = capi.options("refine"); o.max_iterations = 2
o.inner_iterations = 0      # the LM iteration itself
g.solve(o)
ts = [g.time_schur(20) for _ in range(3)]
print("k_schur_syrk ms per launch:", ["%.4f" % t for t in ts], " cost after 2 its %.6g" % g.cost())
