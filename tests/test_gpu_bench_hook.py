"""GPU: the exchange hook bench.py installs for --gpus N (torch.distributed, backend nccl = RCCL) wraps the
library's raw device pointer without copying and reduces it in place.  One rank here (the driver owns the
multi-GPU runs); the multi-rank logic of the hook protocol is covered on CPU by test_multirank_gloo.py."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_hook_reduces_device_and_host_buffers_in_place():
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    from cvx_amd import capi
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        hook = bench.make_hook(torch, dist, capi)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            t = torch.arange(79003, dtype=torch.float64, device="cuda") * 0.5
            ref = t.clone()
            rc = hook(None, C.cast(t.data_ptr(), capi.dp), t.numel(), 1, C.c_void_p(stream.cuda_stream))
            torch.cuda.synchronize()
            assert rc == 0 and torch.equal(t, ref)
            # the wrapped tensor aliases the buffer: an in-place change through the hook's view is seen
            v = torch.as_tensor(bench._DevBuf(t.data_ptr(), t.numel()), device="cuda")
            v += 1.0
            torch.cuda.synchronize()
            assert torch.equal(t, ref + 1.0)
            # the hook orders the collective on the stream it is GIVEN, not on torch's current one: data written on a
            # side stream behind a long-running kernel is what gets reduced, and work queued on that stream
            # afterwards sees the result, while the current stream is a different one
            side = torch.cuda.Stream()
            w = torch.ones(4096, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                torch.cuda._sleep(20_000_000)
                w.fill_(3.0)
            assert torch.cuda.current_stream().cuda_stream != side.cuda_stream
            rc = hook(None, C.cast(w.data_ptr(), capi.dp), w.numel(), 1, C.c_void_p(side.cuda_stream))
            with torch.cuda.stream(side):
                w2 = w * 2.0
            torch.cuda.synchronize()
            assert rc == 0 and bool((w == 3.0).all()) and bool((w2 == 6.0).all())
            h = np.arange(16, dtype=np.float64)
            rc = hook(None, capi.pd(h), h.size, 0, None)
            assert rc == 0 and np.array_equal(h, np.arange(16, dtype=np.float64))
    finally:
        dist.destroy_process_group()
