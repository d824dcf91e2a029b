"""GPU: host cost of one call of bench.py's exchange hook (torch.distributed on "nccl" = RCCL, a world of one rank) and
of the library's built-in hook (cvx_rccl_allreduce) on the same buffer: what an N > 1 iteration pays twice."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
import bench
from cvx_amd import capi
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
hook = bench.make_hook(torch, dist, capi)
L = capi.lib()
st = torch.cuda.Stream()
for n in (4, 76_000):
    with torch.cuda.stream(st):
        t = torch.ones(n, dtype=torch.float64, device="cuda")
        p = C.cast(t.data_ptr(), C.POINTER(C.c_double))
        for _ in range(20):
            hook(None, p, n, 1, C.c_void_p(st.cuda_stream))
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            hook(None, p, n, 1, C.c_void_p(st.cuda_stream))
        t1 = time.perf_counter()
        st.synchronize()
        t2 = time.perf_counter()
    print("torch hook   %6d doubles: %.1f us per call to enqueue, %.1f us per call with the device's share" % (n, 1e6 * (t1 - t0) / 200, 1e6 * (t2 - t0) / 200))
ident = C.create_string_buffer(256)
capi.check(L.cvx_rccl_unique_id(ident), "id")
comm = C.c_void_p()
capi.check(L.cvx_rccl_create(0, 1, ident, 0, C.byref(comm)), "create")
for n in (4, 76_000):
    t = torch.ones(n, dtype=torch.float64, device="cuda")
    p = C.cast(t.data_ptr(), capi.dp)
    for _ in range(20):
        L.cvx_rccl_allreduce(comm, p, n, 1, C.c_void_p(st.cuda_stream))
    st.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        L.cvx_rccl_allreduce(comm, p, n, 1, C.c_void_p(st.cuda_stream))
    t1 = time.perf_counter()
    st.synchronize()
    t2 = time.perf_counter()
    print("built-in hook %6d doubles: %.1f us per call to enqueue, %.1f us per call with the device's share" % (n, 1e6 * (t1 - t0) / 200, 1e6 * (t2 - t0) / 200))
L.cvx_rccl_destroy(comm)
dist.destroy_process_group()
