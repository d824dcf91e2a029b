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


def sharded_options(capi):
    """refine's options without the inner iterations, on the sharded and on the single-rank side of the comparisons
    of the LM iteration itself (the inner iterations on shards have a comparison of their own below)"""
    o = capi.options("refine")
    o.inner_iterations = 0
    return o


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
    s = g.solve(sharded_options(capi))
    intr, poses, pts = g.get_params()
    res = {"it": s.iterations, "term": s.termination, "cost": s.final_cost, "trace": s.trace()["cost"].tolist(),
           "intr": intr.tolist(), "pts": pts.tolist(), "poses": poses.tolist(), "v0": int(first[rank])}
    # the same solve over the device-buffer protocol, the path a real RCCL run takes on the C side (packed lower
    # triangle exchanged in place on the device, k_damp_prepare reading it, the 4-double all-reduce of the candidate
    # scalars, k_publish_scalars + the host poll), with bench.py's own hook: under gloo it wraps the device pointer
    # on the stream the library hands over and takes the sum on host copies
    import bench
    g2 = cvx_amd.Calib(sh["cam_of_view"], sh["view_off"], sh["point_id"], sh["obs_x"], sh["obs_y"], d["C"], d["P"])
    cb2 = bench.make_hook(torch, dist, capi)
    g2.set_exchange(cb2, rank, world, device_buffers=True)
    g2.set_params(d["intr_init"], sh["poses_init"], d["points_init"])
    s2 = g2.solve(sharded_options(capi))
    intr2, poses2, pts2 = g2.get_params()
    res["dev"] = {"it": s2.iterations, "term": s2.termination, "cost": s2.final_cost, "trace": s2.trace()["cost"].tolist(),
                  "intr": intr2.tolist(), "pts": pts2.tolist(), "poses": poses2.tolist()}
    g2.close()
    # inner iterations (calibration.cpp:424) on shards: the ordering from the whole problem's structure (two
    # exchanges at the first pass), the shared blocks' sums all-reduced every round; over both buffer protocols
    for dev in (False, True):
        g3 = cvx_amd.Calib(sh["cam_of_view"], sh["view_off"], sh["point_id"], sh["obs_x"], sh["obs_y"], d["C"], d["P"])
        g3.set_exchange(cb2 if dev else cb, rank, world, device_buffers=dev)
        g3.set_params(d["intr_init"], sh["poses_init"], d["points_init"])
        o3 = capi.options("refine")
        assert o3.inner_iterations == 1
        s3 = g3.solve(o3)
        intr3, poses3, pts3 = g3.get_params()
        res["inner_dev" if dev else "inner"] = {
            "it": s3.iterations, "term": s3.termination, "cost": s3.final_cost, "trace": s3.trace()["cost"].tolist(),
            "passes": s3.num_inner_iteration_steps, "intr": intr3.tolist(), "pts": pts3.tolist(), "poses": poses3.tolist()}
        g3.close()
    # hand-eye shards by pair
    G, T, xg, x0 = cvx_amd.synth_handeye(40, 9)
    A, B = cvx_amd.handeye_pairs(G, T)
    qa, ta, qb, tb = cvx_amd.pairs_to_quat_trans(A, B)
    lo, hi = (len(A) * rank) // world, (len(A) * (rank + 1)) // world
    he = cvx_amd.HandEye(qa[lo:hi], ta[lo:hi], qb[lo:hi], tb[lo:hi])
    he.set_exchange(cb, rank, world, device_buffers=False)
    p, hs = he.solve(x0)
    res["he_p"] = p.tolist(); res["he_cost"] = hs.final_cost
    # tool-level bundle adjustment shards by view: one all-reduce of the P x 55 block sums per linearisation
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_helpers import toolba_problem
    from cvx_amd.problem import ToolBA
    tb = toolba_problem(np.random.default_rng(5), 41, ragged=4, outliers=0.05)
    v0, v1 = (41 * rank) // world, (41 * (rank + 1)) // world
    o0, o1 = int(tb["view_off"][v0]), int(tb["view_off"][v1])
    for dev in (False, True):
        t = ToolBA(tb["view_off"][v0:v1 + 1] - o0, tb["point_id"][o0:o1], tb["obs_x"][o0:o1], tb["obs_y"][o0:o1],
                   tb["trans"][v0:v1], tb["intr"], tb["P"])
        t.set_exchange(cb2 if dev else cb, rank, world, device_buffers=dev)
        pt, xt, stb = t.solve(tb["p0"], tb["points"], capi.options("toolba"))
        res["tba_dev" if dev else "tba"] = {"p": pt.tolist(), "x": xt.tolist(), "cost": stb.final_cost, "it": stb.iterations,
                                              "trace": list(stb.trace_cost[:stb.trace_len])}
        t.close()
    out[rank] = res
    g.close(); he.close()
    dist.barrier()
    dist.destroy_process_group()


def _blocks_close(ra, rb, ref, what):
    """shared blocks of rank 0 + the two ranks' pose ranges against a single-rank (intr, poses, pts), block by block"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_helpers import assert_params_close
    got = (np.array(ra["intr"]), np.concatenate([np.array(ra["poses"]), np.array(rb["poses"])]), np.array(ra["pts"]))
    # a two-rank run sums the reduced system in another order than one rank: every block to north_star's 1e-9
    return assert_params_close(got, ref, None, what)


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
    s = g.solve(sharded_options(capi))
    intr, poses, pts = g.get_params()
    r0, r1 = out[0], out[1]
    # both ranks hold bitwise identical shared parameters and costs (identical all-reduce results)
    assert r0["intr"] == r1["intr"] and r0["pts"] == r1["pts"] and r0["trace"] == r1["trace"]
    assert (r0["it"], r0["term"]) == (s.iterations, s.termination)
    tr = np.array(r0["trace"])
    assert np.all(np.abs(tr - s.trace()["cost"]) <= 1e-10 * tr)
    assert abs(r0["cost"] - s.final_cost) <= 1e-11 * s.final_cost
    _blocks_close(r0, r1, (intr, poses, pts), "host buffers")
    # device-buffer protocol: the same iterations, termination, trace and parameters; identical shared blocks on both ranks
    d0, d1 = r0["dev"], r1["dev"]
    assert d0["intr"] == d1["intr"] and d0["pts"] == d1["pts"] and d0["trace"] == d1["trace"]
    assert (d0["it"], d0["term"]) == (s.iterations, s.termination)
    trd = np.array(d0["trace"])
    assert np.all(np.abs(trd - s.trace()["cost"]) <= 1e-10 * trd)
    assert np.all(np.abs(trd - tr) <= 1e-12 * tr)          # and it agrees with the host-buffer protocol
    assert abs(d0["cost"] - s.final_cost) <= 1e-11 * s.final_cost
    _blocks_close(d0, d1, (intr, poses, pts), "device buffers")
    # with the inner iterations: the sharded runs follow the single-rank run pass for pass
    g.set_params(d["intr_init"], d["poses_init"], d["points_init"])
    si = g.solve(capi.options("refine"))
    intr_i, poses_i, pts_i = g.get_params()
    assert si.num_inner_iteration_steps >= 1
    for key in ("inner", "inner_dev"):
        a0, a1 = r0[key], r1[key]
        assert a0["intr"] == a1["intr"] and a0["pts"] == a1["pts"] and a0["trace"] == a1["trace"]
        assert (a0["it"], a0["term"], a0["passes"]) == (si.iterations, si.termination, si.num_inner_iteration_steps)
        tri = np.array(a0["trace"])
        assert np.all(np.abs(tri - si.trace()["cost"]) <= 1e-10 * tri)
        assert abs(a0["cost"] - si.final_cost) <= 1e-11 * si.final_cost
        _blocks_close(a0, a1, (intr_i, poses_i, pts_i), key)
    # hand-eye
    G, T, xg, x0 = cvx_amd.synth_handeye(40, 9)
    A, B = cvx_amd.handeye_pairs(G, T)
    he = cvx_amd.HandEye(*cvx_amd.pairs_to_quat_trans(A, B))
    p, hs = he.solve(x0)
    assert r0["he_p"] == r1["he_p"]
    assert np.max(np.abs(np.array(r0["he_p"]) - p)) < 1e-12 and abs(r0["he_cost"] - hs.final_cost) <= 1e-12 * hs.final_cost
    g.close(); he.close()
    # tool-level bundle adjustment, host-buffer and device-buffer protocol
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_helpers import toolba_problem
    from cvx_amd.problem import ToolBA
    tb = toolba_problem(np.random.default_rng(5), 41, ragged=4, outliers=0.05)
    t = ToolBA(tb["view_off"], tb["point_id"], tb["obs_x"], tb["obs_y"], tb["trans"], tb["intr"], tb["P"])
    pt, xt, stb = t.solve(tb["p0"], tb["points"], capi.options("toolba"))
    t.close()
    for key in ("tba", "tba_dev"):
        a, b = r0[key], r1[key]
        assert a["p"] == b["p"] and a["x"] == b["x"] and a["trace"] == b["trace"]      # both ranks: the same bits
        n = min(len(a["trace"]), stb.trace_len, 12)
        tr1 = np.array(stb.trace_cost[:n])
        assert np.max(np.abs(np.array(a["trace"][:n]) - tr1) / tr1) <= 1e-9
        assert abs(a["cost"] - stb.final_cost) <= 1e-9 * stb.final_cost
        assert np.max(np.abs(np.array(a["p"]) - pt)) <= 1e-8 and np.max(np.abs(np.array(a["x"]) - xt)) <= 1e-8


def _one_camera_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cvx_amd
    from cvx_amd import capi
    import bench
    d = cvx_amd.synth_calib(world, 6, 12, 9, 78)                # dense: the cut by observations falls between the cameras
    first = cvx_amd.partition_views(d["view_off"], world)
    sh = cvx_amd.shard_calib(d, first, rank)
    assert set(sh["cam_of_view"].tolist()) == {rank}            # this rank sees ONE camera of `world`
    g = cvx_amd.Calib(sh["cam_of_view"], sh["view_off"], sh["point_id"], sh["obs_x"], sh["obs_y"], d["C"], d["P"])
    g.set_exchange(bench.make_hook(torch, dist, capi), rank, world, device_buffers=True)
    g.set_params(d["intr_init"], sh["poses_init"], d["points_init"])
    s = g.solve(capi.options("refine"))
    intr, poses, pts = g.get_params()
    out[rank] = {"it": s.iterations, "term": s.termination, "cost": s.final_cost, "trace": s.trace()["cost"].tolist(),
                 "passes": s.num_inner_iteration_steps, "intr": intr.tolist(), "pts": pts.tolist(), "poses": poses.tolist()}
    g.close()
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_one_camera_each_equal_single_rank():
    """the shape an 8-GPU run of config 4 has (8 cameras x 1250 views over 8 ranks: every rank holds the views of exactly one
    camera, the other cameras' blocks have no local residuals), at world 4 on this one GPU: the reference's options with
    their inner iterations, device-buffer protocol, against the single-rank solve"""
    import cvx_amd
    from cvx_amd import capi
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_helpers import assert_params_close
    world = 4
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_one_camera_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    d = cvx_amd.synth_calib(world, 6, 12, 9, 78)
    g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], d["C"], d["P"])
    g.set_params(d["intr_init"], d["poses_init"], d["points_init"])
    s = g.solve(capi.options("refine"))
    ref = g.get_params()
    g.close()
    r = [out[k] for k in range(world)]
    for k in range(1, world):
        assert r[k]["intr"] == r[0]["intr"] and r[k]["pts"] == r[0]["pts"] and r[k]["trace"] == r[0]["trace"]
    assert (r[0]["it"], r[0]["term"], r[0]["passes"]) == (s.iterations, s.termination, s.num_inner_iteration_steps)
    tr = np.array(r[0]["trace"])
    assert np.all(np.abs(tr - s.trace()["cost"]) <= 1e-10 * tr)
    assert abs(r[0]["cost"] - s.final_cost) <= 1e-11 * s.final_cost
    got = (np.array(r[0]["intr"]), np.concatenate([np.array(x["poses"]) for x in r]), np.array(r[0]["pts"]))
    assert_params_close(got, ref, None, "four ranks, one camera each")


def _rccl_worker(rank, world, port, out):
    """one process per GPU: the built-in RCCL hook (cvx_rccl_*) under a view-sharded solve"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)     # only carries the 128-byte RCCL id
    import cvx_amd
    from cvx_amd import capi
    L = capi.lib()
    ident = C.create_string_buffer(capi.RCCL_ID_BYTES)
    if rank == 0:
        capi.check(L.cvx_rccl_unique_id(ident), "unique id")
    t = torch.frombuffer(bytearray(ident.raw), dtype=torch.uint8).clone()
    dist.broadcast(t, 0)
    ident = C.create_string_buffer(capi.RCCL_ID_BYTES)
    ident.raw = bytes(t.tolist())
    comm = C.c_void_p()
    capi.check(L.cvx_rccl_create(rank, world, ident, rank, C.byref(comm)), "rccl create")
    d = cvx_amd.synth_calib(2, 9, 12, 9, 77, ragged=5)
    first = cvx_amd.partition_views(d["view_off"], world)
    sh = cvx_amd.shard_calib(d, first, rank)
    g = cvx_amd.Calib(sh["cam_of_view"], sh["view_off"], sh["point_id"], sh["obs_x"], sh["obs_y"], d["C"], d["P"], device=rank)
    g.set_exchange(C.cast(L.cvx_rccl_allreduce, C.c_void_p).value, rank, world, device_buffers=True, ctx=comm)
    g.set_params(d["intr_init"], sh["poses_init"], d["points_init"])
    s = g.solve(sharded_options(capi))
    intr, poses, pts = g.get_params()
    out[rank] = {"it": s.iterations, "term": s.termination, "cost": s.final_cost, "trace": s.trace()["cost"].tolist(),
                 "intr": intr.tolist(), "pts": pts.tolist(), "poses": poses.tolist()}
    g.close()
    L.cvx_rccl_destroy(comm)
    dist.barrier()
    dist.destroy_process_group()


def test_two_gpus_rccl_solve_equals_single_rank():
    """world 2 over RCCL, one GPU per rank; skipped on a one-GPU box (the driver's 8-GPU node runs it)"""
    import cvx_amd
    from cvx_amd import capi
    if capi.lib().cvx_device_count() < 2:
        pytest.skip("needs two GPUs")
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rccl_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    d = cvx_amd.synth_calib(2, 9, 12, 9, 77, ragged=5)
    g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], d["C"], d["P"])
    g.set_params(d["intr_init"], d["poses_init"], d["points_init"])
    s = g.solve(sharded_options(capi))
    intr, poses, pts = g.get_params()
    g.close()
    r0, r1 = out[0], out[1]
    assert r0["intr"] == r1["intr"] and r0["pts"] == r1["pts"] and r0["trace"] == r1["trace"]
    assert (r0["it"], r0["term"]) == (s.iterations, s.termination)
    tr = np.array(r0["trace"])
    assert np.all(np.abs(tr - s.trace()["cost"]) <= 1e-10 * tr)
    _blocks_close(r0, r1, (intr, poses, pts), "rccl")


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
