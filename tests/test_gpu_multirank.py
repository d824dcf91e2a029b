"""GPU, world_size 2 (both ranks on cuda:0, exchange over gloo through the C-ABI all-reduce hook):
the view-sharded LM solve reproduces the single-rank solve."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cvx_amd
    from cvx_amd import capi
    d = cvx_amd.synth_calib(2, 9, 12, 9, 77, ragged=5)
    first = cvx_amd.partition_views(d["view_off"], world)
    sh = cvx_amd.shard_calib(d, first, rank)
    g = cvx_amd.Calib(sh["cam_of_view"], sh["view_off"], sh["point_id"], sh["obs_x"], sh["obs_y"], d["C"], d["P"])

    def hook(ctx, buf, count, on_device, stream):
        a = np.ctypeslib.as_array(buf, shape=(count,))
        t = torch.from_numpy(a)
        dist.all_reduce(t)
        return 0
    cb = capi.ALLREDUCE_FN(hook)
    g.set_exchange(cb, rank, world, device_buffers=False)
    g.set_params(d["intr_init"], sh["poses_init"], d["points_init"])
    s = g.solve(capi.options("refine"))
    intr, poses, pts = g.get_params()
    res = {"it": s.iterations, "term": s.termination, "cost": s.final_cost, "trace": s.trace()["cost"].tolist(),
           "intr": intr.tolist(), "pts": pts.tolist(), "poses": poses.tolist(), "v0": int(first[rank])}
    # hand-eye shards by pair
    G, T, xg, x0 = cvx_amd.synth_handeye(40, 9)
    A, B = cvx_amd.handeye_pairs(G, T)
    qa, ta, qb, tb = cvx_amd.pairs_to_quat_trans(A, B)
    lo, hi = (len(A) * rank) // world, (len(A) * (rank + 1)) // world
    he = cvx_amd.HandEye(qa[lo:hi], ta[lo:hi], qb[lo:hi], tb[lo:hi])
    he.set_exchange(cb, rank, world, device_buffers=False)
    p, hs = he.solve(x0)
    res["he_p"] = p.tolist(); res["he_cost"] = hs.final_cost
    out[rank] = res
    g.close(); he.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_solve_equals_single_rank():
    import cvx_amd
    from cvx_amd import capi
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    d = cvx_amd.synth_calib(2, 9, 12, 9, 77, ragged=5)
    g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], d["C"], d["P"])
    g.set_params(d["intr_init"], d["poses_init"], d["points_init"])
    s = g.solve(capi.options("refine"))
    intr, poses, pts = g.get_params()
    r0, r1 = out[0], out[1]
    # both ranks hold bitwise identical shared parameters and costs (identical all-reduce results)
    assert r0["intr"] == r1["intr"] and r0["pts"] == r1["pts"] and r0["trace"] == r1["trace"]
    assert (r0["it"], r0["term"]) == (s.iterations, s.termination)
    tr = np.array(r0["trace"])
    assert np.all(np.abs(tr - s.trace()["cost"]) <= 1e-10 * tr)
    assert abs(r0["cost"] - s.final_cost) <= 1e-11 * s.final_cost
    assert np.max(np.abs(np.array(r0["intr"]) - intr) / np.maximum(np.abs(intr), 1e-3)) < 1e-9
    got_poses = np.concatenate([np.array(r0["poses"]), np.array(r1["poses"])])
    assert np.max(np.abs(got_poses - poses)) < 1e-8 * np.max(np.abs(poses))
    # hand-eye
    G, T, xg, x0 = cvx_amd.synth_handeye(40, 9)
    A, B = cvx_amd.handeye_pairs(G, T)
    he = cvx_amd.HandEye(*cvx_amd.pairs_to_quat_trans(A, B))
    p, hs = he.solve(x0)
    assert r0["he_p"] == r1["he_p"]
    assert np.max(np.abs(np.array(r0["he_p"]) - p)) < 1e-12 and abs(r0["he_cost"] - hs.final_cost) <= 1e-12 * hs.final_cost
    g.close(); he.close()


def test_builtin_rccl_hook_single_rank():
    """The dlopen'ed RCCL path: a 1-rank communicator all-reduces a device buffer in place (identity)."""
    from cvx_amd import capi
    L = capi.lib()
    ident = C.create_string_buffer(capi.RCCL_ID_BYTES)
    capi.check(L.cvx_rccl_unique_id(ident), "unique id")
    comm = C.c_void_p()
    capi.check(L.cvx_rccl_create(0, 1, ident, 0, C.byref(comm)), "rccl create")
    host = np.arange(1000, dtype=np.float64)
    capi.check(L.cvx_rccl_allreduce(comm, capi.pd(host), host.size, 0, None), "host all-reduce")
    assert np.array_equal(host, np.arange(1000, dtype=np.float64))
    t = torch.arange(4096, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    capi.check(L.cvx_rccl_allreduce(comm, C.cast(t.data_ptr(), capi.dp), t.numel(), 1,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), "device all-reduce")
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(4096, dtype=torch.float64))
    L.cvx_rccl_destroy(comm)
