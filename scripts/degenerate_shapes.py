"""Degenerate shapes through cvx_calib_create / solve / destroy (development aid): no views, views without markers,
float observations of length zero, the preparation thread on an empty problem."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cvx_amd
from cvx_amd import capi

i32 = lambda a: np.array(a, dtype=np.int32)
for name, cam, off, pid, ox, P in (
        ("no views", i32([]), i32([0]), i32([]), np.zeros(0), 4),
        ("one empty view", i32([0]), i32([0, 0]), i32([]), np.zeros(0), 4),
        ("empty view among others", i32([0, 0, 0]), i32([0, 4, 4, 8]), i32([0, 1, 2, 3, 0, 1, 2, 3]), np.linspace(100, 200, 8), 4),
        ("float, no observations", i32([0]), i32([0, 0]), i32([]), np.zeros(0, np.float32), 4)):
    for ahead in (False, True):
        try:
            g = cvx_amd.Calib(cam, off, pid, ox, ox.copy(), 1, P, prepare_inner=ahead)
            V = len(cam)
            intr = np.array([[1000.0, 1000, 640, 480, 0, 0, 0, 0, 0]])
            poses = np.tile(np.array([0.01, 0.02, 0.03, 0.0, 0.0, 500.0]), (V, 1)).reshape(V, 6)
            pts = np.array([[0.0, 0, 0], [25, 0, 0], [0, 25, 0], [25, 25, 0]])
            g.set_params(intr, poses, pts)
            c = g.cost()
            try:
                s = g.solve(capi.options("refine"))
                res = "solve: %d iterations, termination %d" % (s.iterations, s.termination)
            except capi.CvxError as e:
                res = "solve refused: %s" % e
            g.close()
            print("%-26s ahead=%d cost %.4g | %s" % (name, ahead, c, res))
        except capi.CvxError as e:
            print("%-26s ahead=%d create refused: %s" % (name, ahead, e))
