"""GPU: time of the batched DLT (cvx_homography_dlt) and of the residual pass on BASELINE config 4's view count,
10 000 problems x 108 pairs; run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import sys, time
import numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cvx_amd
from cvx_amd.problem import HomographyBatch
from test_helpers import homography_batch

rng = np.random.default_rng(4)
n = 10000
base = homography_batch(rng, [108] * 50)
off = (np.arange(n + 1) * 108).astype(np.int32)
x1, y1, x2, y2 = (np.tile(a, n // 50) for a in base[1:5])
x2 = (x2 + rng.normal(0, 0.05, x2.shape)).astype(np.float32)
b = HomographyBatch(off, x1, y1, x2, y2)
b.dlt(fetch=False)
for what, fn in (("dlt, estimates left on the device", lambda: b.dlt(fetch=False)), ("dlt + download", lambda: b.dlt()),
                 ("residuals + download", lambda: b.residuals()), ("dlt + lm on the device", lambda: (b.dlt(fetch=False), b.run()))):
    t = time.perf_counter()
    for _ in range(5):
        fn()
    print("%-36s %.3f ms per batch" % (what, 1e3 * (time.perf_counter() - t) / 5))
