"""CPU, world_size 2 over gloo: the view-sharded exchange step of SURVEY.md section 8(e).

Each rank holds a contiguous range of views (cvx_partition_views); the per-rank Schur pieces
[S | gred | cost] summed by ONE all-reduce must equal the unsharded system, and every rank must then
compute the same step.  The per-rank evaluator here is the oracle (the HIP evaluator needs a GPU: the same
exchange through the C-ABI hook is exercised on the card by tests/test_gpu_multirank.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cvx_amd
    from oracle import oracle as orc
    d = cvx_amd.synth_calib(2, 6, 5, 4, 31, ragged=3)
    first = cvx_amd.partition_views(d["view_off"], world)
    sh = cvx_amd.shard_calib(d, first, rank)
    o = orc.Calib(sh["cam_of_view"], sh["view_off"], sh["point_id"], sh["obs_x"], sh["obs_y"], d["C"], d["P"])
    ns = o.ns
    Ds = np.full(ns, 0.5)
    Dp = np.full(6 * sh["V"], 0.25)
    S, g, cost = o.reduced_system(d["intr_init"], sh["poses_init"], d["points_init"], Dp, np.zeros(ns))
    buf = torch.from_numpy(np.concatenate([S.reshape(-1), g, [cost]]))
    dist.all_reduce(buf)                       # the one exchange step per linear solve
    buf = buf.numpy()
    S = buf[:ns * ns].reshape(ns, ns) + np.diag(Ds)
    g = buf[ns * ns:ns * ns + ns]
    delta = np.linalg.solve(S, -g)             # every rank factorises the same system redundantly
    gathered = [torch.zeros(ns, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(delta))
    if rank == 0:
        full = orc.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], d["C"], d["P"])
        Sf, gf, cf = full.reduced_system(d["intr_init"], d["poses_init"], d["points_init"], np.full(6 * d["V"], 0.25), Ds)
        out["S"] = float(np.max(np.abs(S - Sf)) / np.max(np.abs(Sf)))
        out["g"] = float(np.max(np.abs(g - gf)) / np.max(np.abs(gf)))
        out["cost"] = abs(buf[-1] - cf) / cf
        out["same_step"] = bool(all(torch.equal(gathered[0], t) for t in gathered))
        out["views"] = [int(first[r + 1] - first[r]) for r in range(world)]
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_normal_equations_sum_to_the_full_system():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out["views"][0] > 0 and out["views"][1] > 0
    assert out["S"] < 1e-13 and out["g"] < 1e-12 and out["cost"] < 1e-14
    assert out["same_step"]   # identical all-reduce result on every rank => bitwise identical steps
