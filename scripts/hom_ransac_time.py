"""GPU: time of the batched RANSAC (cvx_homography_ransac) on BASELINE config 4's view count, 10 000 problems x 108 pairs
(20 of them pushed away) x 100 trials; run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import sys, time
import numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if os.environ.get('WITH_TORCH'):
    import torch; torch.cuda.init()
import cvx_amd
from cvx_amd.problem import HomographyBatch
from test_homography_oracle import ransac_problem

rng = np.random.default_rng(4)
n, N = 10000, 108
parts = [ransac_problem(rng, N, 20) for _ in range(50)]
off = (np.arange(n + 1) * N).astype(np.int32)
x1, y1, x2, y2 = (np.tile(np.concatenate([p[k] for p in parts]), n // 50) for k in range(4))
b = HomographyBatch(off, x1, y1, x2, y2)
params = dict(max_error=2.0)
b.ransac(seed=1, params=params)
for _ in range(4):
    t = time.perf_counter(); b.ransac(seed=1, params=params); print("  one call %.3f ms" % (1e3 * (time.perf_counter() - t)))
for trials in (100, 25):
    params["max_num_trials"] = trials
    t = time.perf_counter()
    for _ in range(3):
        out = b.ransac(seed=1, params=params)
    print("%3d trials: %.3f ms per batch, mean inliers %.1f, the loop ran %.1f trials on average" % (
        trials, 1e3 * (time.perf_counter() - t) / 3, out["n_inliers"].mean(), out["trials_run"].mean()))
