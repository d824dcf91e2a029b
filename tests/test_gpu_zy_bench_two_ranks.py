"""GPU: the whole `bench.py --gpus 2` flow, two ranks sharing cuda:0 (CVX_BENCH_REHEARSAL=1: gloo instead of RCCL,
the device-buffer hook summed through the host).  Checks the contract of the N > 1 run -- one JSON line from rank 0,
weak scaling, both ranks' work counted -- not its speed.  (Collected late on purpose: it starts two more processes.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_rank_bench_line():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, CVX_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=100)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                      # rank 0 alone reports
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["unit"] == "observations/s" and d["value"] > 0 and d["higher_is_better"] is True
    # value = the observations of BOTH ranks x iterations / the slowest rank's time
    per_rank = d["config"]["observations_per_gpu"]
    assert abs(d["value"] - 2 * per_rank * 4 / (d["ms_per_step"] * 4 * 1e-3)) <= 1e-6 * d["value"]
    assert "cpu_baseline" not in d              # rank 0 at N = 1 only
