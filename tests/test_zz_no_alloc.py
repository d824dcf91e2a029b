"""GPU test of cvx_handeye_normal_eq_static, the entry point that allocates nothing (include/cvx_amd.h): its output
against cvx_handeye_normal_eq's on the same motion pairs, bit for bit, on two calls in a row, the second with a
smaller input (the scratch array keeps the first call's partial sums).  The caller's device buffers come straight
from the HIP runtime the library itself is linked against (ctypes on libamdhip64)."""
import ctypes as C

import numpy as np
import pytest

import cvx_amd
from cvx_amd import capi

pytestmark = pytest.mark.gpu


def _pairs(n, seed):
    rng = np.random.default_rng(seed)
    qa = rng.normal(size=(n, 4)); qa /= np.linalg.norm(qa, axis=1, keepdims=True)
    qb = rng.normal(size=(n, 4)); qb /= np.linalg.norm(qb, axis=1, keepdims=True)
    return qa, rng.normal(size=(n, 3)), qb, rng.normal(size=(n, 3))


_HIP = None


def _hip():
    global _HIP
    if _HIP is None:
        capi.lib()
        # the HIP runtime libcvx_amd.so itself brought into the process (whatever its version), not a soname of ours
        paths = {l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64.so" in l}
        assert len(paths) == 1, paths
        _HIP = C.CDLL(paths.pop())
        _HIP.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _HIP.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _HIP.hipFree.argtypes = [C.c_void_p]
        _HIP.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        _HIP.hipStreamDestroy.argtypes = [C.c_void_p]
        _HIP.hipStreamSynchronize.argtypes = [C.c_void_p]
        _HIP.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
        _HIP.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        _HIP.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        _HIP.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
        _HIP.hipGraphExecDestroy.argtypes = [C.c_void_p]
        _HIP.hipGraphDestroy.argtypes = [C.c_void_p]
    return _HIP


class _Dev:
    """a device buffer holding a copy of a host array"""

    def __init__(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        self.n = a.size
        self.p = C.c_void_p()
        assert _hip().hipSetDevice(0) == 0
        assert _hip().hipMalloc(C.byref(self.p), max(8 * a.size, 8)) == 0
        if a.size:
            assert _hip().hipMemcpy(self.p, a.ctypes.data_as(C.c_void_p), 8 * a.size, 1) == 0

    def host(self):
        out = np.empty(self.n)
        assert _hip().hipDeviceSynchronize() == 0
        assert _hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), self.p, 8 * self.n, 2) == 0
        return out

    def free(self):
        _hip().hipFree(self.p)


def _static(n, qa, ta, qb, tb, p7):
    bufs = [_Dev(a) for a in (qa, ta, qb, tb, p7)]
    out = _Dev(np.full(43, np.nan))
    st = capi.lib().cvx_handeye_normal_eq_static(n, bufs[0].p, bufs[1].p, bufs[2].p, bufs[3].p, bufs[4].p, out.p, 0, None)
    capi.check(st, "cvx_handeye_normal_eq_static")
    o = out.host()
    for b in bufs + [out]:
        b.free()
    return o[:36].reshape(6, 6), o[36:42], o[42]


def test_static_entry_point_matches_the_handle_one_twice_in_a_row():
    p7 = np.array([0.9, 0.1, -0.3, 0.2, 0.05, -0.03, 0.12])
    for n, seed in ((19548, 1), (777, 2), (0, 3), (1, 4)):     # second call smaller than the first; then the edges
        qa, ta, qb, tb = _pairs(n, seed)
        g = cvx_amd.HandEye(qa, ta, qb, tb)
        H, gr, c = g.normal_eq(p7)
        g.close()
        Hs, gs, cs = _static(n, qa, ta, qb, tb, p7)
        assert np.array_equal(Hs, H) and np.array_equal(gs, gr) and cs == c, n


def test_static_entry_point_rejects_sizes_beyond_its_scratch():
    z = _Dev(np.zeros(64))
    st = capi.lib().cvx_handeye_normal_eq_static(262145, z.p, z.p, z.p, z.p, z.p, z.p, 0, None)
    z.free()
    assert st == capi.ERR_UNSUPPORTED
    assert b"CVX_HANDEYE_STATIC_MAX_PAIRS" in capi.lib().cvx_last_error()
    # the largest size it takes
    n = 262144
    qa, ta, qb, tb = _pairs(n, 9)
    p7 = np.array([1.0, 0, 0, 0, 0.1, 0.2, 0.3])
    g = cvx_amd.HandEye(qa, ta, qb, tb)
    H, gr, c = g.normal_eq(p7)
    g.close()
    Hs, gs, cs = _static(n, qa, ta, qb, tb, p7)
    assert np.array_equal(Hs, H) and np.array_equal(gs, gr) and cs == c


def test_static_entry_point_under_stream_capture():
    """What the entry point is for: after the first (address look-up) call on the device, a call between
    hipStreamBeginCapture and hipStreamEndCapture on a non-default stream, in global capture mode -- where hipMalloc,
    hipFree and every synchronising call fail and invalidate the capture -- succeeds, records its two kernels, and the
    replayed graph gives the handle path's bits, twice (the graph re-uses the scratch array)."""
    hip = _hip()
    p7 = np.array([0.7, -0.2, 0.4, 0.1, 0.3, -0.1, 0.02])
    n = 5000
    qa, ta, qb, tb = _pairs(n, 21)
    g = cvx_amd.HandEye(qa, ta, qb, tb)
    H, gr, c = g.normal_eq(p7)
    g.close()
    _static(64, *_pairs(64, 22), p7)                          # the look-up call, outside any capture
    bufs = [_Dev(a) for a in (qa, ta, qb, tb, p7)]
    out = _Dev(np.full(43, np.nan))
    stream, graph, gexec = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert hip.hipStreamCreate(C.byref(stream)) == 0
    assert hip.hipStreamBeginCapture(stream, 0) == 0          # hipStreamCaptureModeGlobal
    st = capi.lib().cvx_handeye_normal_eq_static(n, bufs[0].p, bufs[1].p, bufs[2].p, bufs[3].p, bufs[4].p, out.p, 0, stream)
    end = hip.hipStreamEndCapture(stream, C.byref(graph))
    assert st == capi.OK and end == 0 and graph.value        # an allocation inside would have broken the capture
    assert np.all(np.isnan(out.host()))                      # captured, not run
    assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
    for _ in range(2):
        assert hip.hipGraphLaunch(gexec, stream) == 0 and hip.hipStreamSynchronize(stream) == 0
        o = out.host()
        assert np.array_equal(o[:36].reshape(6, 6), H) and np.array_equal(o[36:42], gr) and o[42] == c
        assert hip.hipMemcpy(out.p, np.full(43, np.nan).ctypes.data_as(C.c_void_p), 8 * 43, 1) == 0
    hip.hipGraphExecDestroy(gexec); hip.hipGraphDestroy(graph); hip.hipStreamDestroy(stream)
    for b in bufs + [out]:
        b.free()
