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

# ---- the other handles ----
from cvx_amd.problem import HomographyBatch
for n in (0, 1, 2):
    try:
        q = (np.tile([1.0, 0, 0, 0], (n, 1)), np.zeros((n, 3)), np.tile([1.0, 0, 0, 0], (n, 1)), np.zeros((n, 3)))
        he = cvx_amd.HandEye(*q)
        p, s = he.solve(np.array([1.0, 0, 0, 0, 0, 0, 0]), capi.options("handeye"))
        print("hand-eye, %d pairs: %d iterations, termination %d, cost %.3g" % (n, s.iterations, s.termination, s.final_cost))
        he.close()
    except capi.CvxError as e:
        print("hand-eye, %d pairs: refused: %s" % (n, e))
try:
    b = HomographyBatch(i32([0, 0, 0]), np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32))
    H = b.dlt(); b.run(); Hr, info, ret = b.get(); r = b.residuals()
    print("homography, two empty problems: dlt", H.tolist(), "ret", ret.tolist(), "residuals", r.shape)
    b.close()
except capi.CvxError as e:
    print("homography, empty problems: refused: %s" % e)
