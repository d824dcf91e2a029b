"""Scratch: phase times of the LM iteration at BASELINE config 4 scale."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cvx_amd
from cvx_amd import capi

C, VP, bw, bh = int(sys.argv[1]), int(sys.argv[2]), 12, 9
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
t = time.time(); d = cvx_amd.synth_calib(C, VP, bw, bh, 20260004); print("synth %.2fs N=%d" % (time.time() - t, d['N']), flush=True)
t = time.time(); g = cvx_amd.Calib(d['cam_of_view'], d['view_off'], d['point_id'], d['obs_x'], d['obs_y'], C, bw * bh); print("create %.2fs" % (time.time() - t), g.geometry(), flush=True)
o = capi.options("refine"); o.max_iterations = K
o.inner_iterations = 0      # the LM iteration itself
o.function_tolerance = -1; o.gradient_tolerance = -1; o.parameter_tolerance = -1
for rep in range(3):
    g.set_params(d['intr_init'], d['poses_init'], d['points_init'])
    g.enable_timing(rep == 2)
    t = time.time(); s = g.solve(o); dt = time.time() - t
    print("solve: it=%d term=%d cost %.6g -> %.6g  %.3f ms/iter (%.2f Mobs-iter/s)" % (s.iterations, s.termination, s.initial_cost, s.final_cost, 1e3 * dt / max(s.iterations, 1), d['N'] * s.iterations / dt / 1e6), flush=True)
print({k: "%.3f ms x%d" % (v[0] / max(v[1], 1), v[1]) for k, v in g.phase_times().items()})
t = time.time()
for i in range(20): p = g.evaluate_device()
g.cost(); dt = (time.time() - t) / 20
print("eval_device: %.3f ms -> %.2f Gobs/s, %.1f GB/s algorithmic" % (dt * 1e3, d['N'] / dt / 1e9, 328 * d['N'] / dt / 1e9))
