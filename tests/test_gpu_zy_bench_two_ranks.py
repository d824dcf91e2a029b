"""GPU: the whole `bench.py --gpus 2` flow, two ranks sharing cuda:0 (CVX_BENCH_REHEARSAL=1: gloo instead of RCCL,
the device-buffer hook summed through the host).  Checks the contract of the N > 1 run -- one JSON line from rank 0,
strong scaling by default (one config-4 problem over both ranks) and weak scaling on request, all ranks' work
counted -- not its speed.  (Collected late on purpose: it starts two more processes.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(scaling):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, CVX_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--no-cpu-baseline", "--scaling", scaling]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                      # rank 0 alone reports
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == scaling
    assert d["unit"] == "observations/s" and d["value"] > 0 and d["higher_is_better"] is True
    assert "cpu_baseline" not in d              # rank 0 at N = 1 only
    return d


def test_two_rank_bench_line_strong_scaling():
    """the default: ONE config-4 problem (1 080 000 observations) cut by view over the two ranks"""
    d = _run("strong")
    assert d["config"]["observations_total"] == 1080000
    assert d["config"]["observations_rank0"] == 540000 and d["config"]["views_rank0"] == 5000
    # value = the whole problem's observations x iterations / the slowest rank's time
    assert abs(d["value"] - 1080000 * 4 / (d["ms_per_step"] * 4 * 1e-3)) <= 1e-6 * d["value"]
    # the sharded solve walks the same path as the single-GPU one: same accepted steps, same cost to ~1e-10
    assert d["config"]["accepted_steps_in_last_solve"] >= 1
    # north_star's config 5 as a strong-scaling block of the same line: the 100 000-view problem over both ranks
    c5 = d["config5_strong"]
    assert c5["observations"] == 10800000 and c5["pose_blocks"] == 100000 and c5["views_this_rank"] == 50000
    assert c5["steps"] == 20 and c5["n_gpus"] == 2 and c5["scaling"] == "strong" and c5["shared_unknowns"] == 388
    assert abs(c5["value"] - 10800000 * 20 / c5["seconds"]) <= 1e-6 * c5["value"]
    assert c5["cost"][1] < 1e-3 * c5["cost"][0]                 # the sharded iterations do converge
    assert "configs" not in d                                   # the one-GPU blocks stay out of the N > 1 line
    # and the weak reading beside it: a config-4-sized shard per rank, one problem of twice the views
    w = d["config4_weak"]
    assert w["observations"] == 2 * 1080000 and w["pose_blocks"] == 20000 and w["views_this_rank"] == 10000
    assert w["n_gpus"] == 2 and w["scaling"] == "weak" and w["shared_unknowns"] == 388 and w["steps"] == 20
    assert abs(w["value"] - 2 * 1080000 * 20 / w["seconds"]) <= 1e-6 * w["value"]
    assert w["cost"][1] < 1e-3 * w["cost"][0]


def test_two_rank_bench_line_weak_scaling():
    d = _run("weak")
    assert d["config"]["observations_total"] == 2 * 1080000 and d["config"]["observations_rank0"] == 1080000
    assert abs(d["value"] - 2 * 1080000 * 4 / (d["ms_per_step"] * 4 * 1e-3)) <= 1e-6 * d["value"]
    assert d["config5_strong"]["observations"] == 10800000      # config 5 is cut over the ranks in either mode
    assert "config4_weak" not in d                              # it is the headline here


def test_bench_exchange_hook_on_rccl_with_one_rank():
    """the hook bench.py hands the library for N > 1 (a device buffer wrapped as a torch tensor, reduced by
    torch.distributed on the stream the library names) on the REAL backend -- "nccl" = RCCL -- as far as one GPU allows:
    a world of one rank.  What it pins: RCCL initialises in this process, the external stream and the foreign pointer are
    accepted, the sum lands in place (x1 with one rank), work enqueued behind it on that stream sees it."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    code = r"""
import ctypes as C, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import bench
from cvx_amd import capi
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
hook = bench.make_hook(torch, dist, capi)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    t = torch.arange(1000, dtype=torch.float64, device="cuda") * 0.5
    rc = hook(None, C.cast(t.data_ptr(), C.POINTER(C.c_double)), t.numel(), 1, C.c_void_p(st.cuda_stream))
    u = t * 2.0                      # behind the collective on the same stream
st.synchronize()
assert rc == 0
assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float64) * 0.5) and torch.equal(u.cpu(), torch.arange(1000, dtype=torch.float64))
host = np.arange(16, dtype=np.float64)
rc = hook(None, host.ctypes.data_as(C.POINTER(C.c_double)), host.size, 0, None)
assert rc == 0 and host.tolist() == list(range(16))
dist.destroy_process_group()
print("hook ok")
""" % ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=200)
    assert p.returncode == 0 and "hook ok" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])

