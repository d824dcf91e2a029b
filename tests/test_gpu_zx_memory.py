"""GPU: handles give back what they take.  Every kind of handle is made, used and destroyed many times; after
cvx_release_cached_memory() the device's free memory is what it was (hipMemGetInfo of the HIP runtime the library itself
brought into the process).  A leak of one buffer per life shows at once: the smallest allocation here is 2 MB of address
space in the runtime's books, the allowance below is a fraction of one life's footprint."""
import ctypes as C
import gc

import numpy as np
import pytest

import cvx_amd
from cvx_amd import capi
from cvx_amd.problem import HomographyBatch
from test_helpers import homography_batch

pytestmark = pytest.mark.gpu


def _free_bytes():
    capi.lib()
    paths = {l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64.so" in l}
    assert len(paths) == 1, paths
    hip = C.CDLL(paths.pop())
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipSetDevice(0) == 0 and hip.hipDeviceSynchronize() == 0
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def _lives(orc_unused=None):
    d = cvx_amd.synth_calib(2, 40, 12, 9, 77)
    oxf, oyf = d["obs_x"].astype(np.float32), d["obs_y"].astype(np.float32)
    rng = np.random.default_rng(3)
    off, x1, y1, x2, y2, H0 = homography_batch(rng, [108] * 40)
    qa = rng.normal(size=(30, 4)); qa /= np.linalg.norm(qa, axis=1, keepdims=True)
    qb = rng.normal(size=(30, 4)); qb /= np.linalg.norm(qb, axis=1, keepdims=True)

    def one(i):
        g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], oxf if i % 2 else d["obs_x"], oyf if i % 2 else d["obs_y"], 2, 108,
                          prepare_inner=bool(i % 2))
        g.set_params(d["intr_init"], d["poses_init"], d["points_init"])
        g.solve(capi.options("refine"))
        g.close()
        b = HomographyBatch(off, x1, y1, x2, y2)
        b.dlt(fetch=False); b.run(); b.residuals()
        out = b.ransac(seed=i, params=dict(max_error=2.0, max_num_trials=20), keep_inliers=True)
        out["inliers"].run(); out["inliers"].close()
        b.close()
        he = cvx_amd.HandEye(qa, rng.normal(size=(30, 3)), qb, rng.normal(size=(30, 3)))
        he.solve(np.array([1.0, 0, 0, 0, 0, 0, 0]), capi.options("handeye"))
        he.close()
    return one


def test_device_memory_comes_back():
    L = capi.lib()
    one = _lives()
    one(0); one(1)                       # whatever the runtime sets up on first use (code objects, queues, pools)
    gc.collect()
    assert L.cvx_release_cached_memory() == 0
    before = _free_bytes()
    for i in range(12):
        one(i)
    gc.collect()
    held = before - _free_bytes()        # the kept block and stream of the last handle
    assert L.cvx_release_cached_memory() == 0
    after = _free_bytes()
    assert before - after <= (4 << 20), (before - after, held)
    assert held >= 0
