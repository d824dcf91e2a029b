#!/usr/bin/env python3
"""bench.py -- the cvx_calibration LM inner loop on MI355X, measured as BASELINE.json asks.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE Levenberg-Marquardt iteration of CameraCalibration::refine
(/root/reference/modules/calib/src/calibration.cpp:354-464) over one rank's shard:
linearise (residual + Jacobian of every observation, normal-equation blocks, Schur
contraction), exchange + solve the damped reduced system, back-substitute the poses and
evaluate the candidate cost.  Workload: BASELINE config 4 -- 8 cameras x 1 250 views of a
12x9 board = 10 000 pose blocks, 1 080 000 observations.

--scaling strong (default; what north_star's config 4 says: "~1M corner observations, sharded by view
across 8 GPUs"): ONE config-4 problem, its views cut into N contiguous ranges balanced by observation
count (cvx_partition_views), the shared block replicated, one all-reduce of the reduced system and one
4-double all-reduce per iteration.  --scaling weak: every rank holds its own config-4-sized shard
(N = 8: 80 000 pose blocks, 8.64 M observations -- config 5's regime).  At N = 1 the two are the same run.

Rank 0 prints one JSON line.  `value` = observations processed by LM iterations per second,
whole job (all ranks' observations x K iterations / max-over-ranks time); inputs are
resident in HBM before the timed region.  `roofline` is for the kernel that dominates the
iteration (per-launch hipEvent durations on the handle's stream); `eval` carries the
materialised residual/Jacobian evaluation (HBM-bound, 328 B/observation); `cpu_baseline`
is the oracle (CPU restatement, 1 thread, as the reference's Ceres runs) on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

CAMERAS, VIEWS_PER_CAMERA, BOARD_W, BOARD_H = 8, 1250, 12, 9      # BASELINE config 4
SEED = 20260004
ITERS_PER_SOLVE = 10          # every timed solve restarts from the perturbed initial guess
HBM_PEAK_GBS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
F64_PEAK_TFLOPS = 78.6        # MI355X FP64 vector = FP64 matrix peak (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
EVAL_BYTES_PER_OBS = 328      # 24 B read + 16 B residual + 288 B Jacobian


class _DevBuf:
    """a raw device pointer as something torch.as_tensor can wrap without copying"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def make_hook(torch, dist, capi):
    """cvx_allreduce_fn on torch.distributed (backend nccl = RCCL over xGMI).  A device buffer is reduced in place on
    the stream the library hands over (cvx_amd.h: `stream` is the hipStream_t the data is ordered on): the
    collective is issued with that stream current, so RCCL's own stream waits for the kernels before it and the
    library's next kernels wait for the result, whatever torch's current stream is at the time of the call."""
    calls = [0]
    gloo = dist.get_backend() == "gloo"

    def hook(ctx, buf, count, on_device, stream):
        calls[0] += 1
        if os.environ.get("CVX_BENCH_TRACE_HOOK") == "1":
            sys.stderr.write("[rank %s] exchange %d: %d doubles\n" % (os.environ.get("RANK"), calls[0], count)); sys.stderr.flush()
        try:
            if on_device:
                ext = torch.cuda.ExternalStream(stream) if stream else torch.cuda.default_stream()
                with torch.cuda.stream(ext):
                    t = torch.as_tensor(_DevBuf(C.cast(buf, C.c_void_p).value, count), device="cuda")
                    if gloo:    # one-GPU rehearsal: the sum is taken on host copies
                        th = t.cpu()
                        dist.all_reduce(th)
                        t.copy_(th)
                    else:
                        dist.all_reduce(t)
            else:
                a = np.ctypeslib.as_array(buf, shape=(count,))
                if gloo:
                    dist.all_reduce(torch.from_numpy(a))
                else:
                    t = torch.from_numpy(a).cuda()
                    dist.all_reduce(t)
                    a[:] = t.cpu().numpy()
            return 0
        except Exception as e:      # the C side turns this into CVX_ERR_EXCHANGE
            sys.stderr.write("exchange hook failed: %r\n" % (e,))
            return 1
    return capi.ALLREDUCE_FN(hook)


def cpu_baseline(views_per_camera, iters):
    """The oracle's LM iterations on a bounded sample of the same workload, one thread."""
    import cvx_amd
    from oracle import oracle as orc
    d = cvx_amd.synth_calib(CAMERAS, views_per_camera, BOARD_W, BOARD_H, SEED)
    o = orc.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], CAMERAS, BOARD_W * BOARD_H)
    op = orc.options("refine")
    op.inner_iterations = 0      # the same work as the GPU's timed steps
    op.max_iterations = iters
    op.function_tolerance = op.gradient_tolerance = op.parameter_tolerance = -1.0
    t = time.perf_counter()
    s = o.solve(d["intr_init"], d["poses_init"], d["points_init"], op)[3]
    dt = time.perf_counter() - t
    # the same sample on all the cores this job may use (16 per GPU on the box, where
    # OMP_NUM_THREADS says so; os.cpu_count() there is the whole machine)
    cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16"))))
    orc.set_threads(cores)
    try:
        o.solve(d["intr_init"], d["poses_init"], d["points_init"], op)       # thread start-up, page faults
        t = time.perf_counter()
        sa = o.solve(d["intr_init"], d["poses_init"], d["points_init"], op)[3]
        dta = time.perf_counter() - t
    finally:
        orc.set_threads(1)
    return {"value": d["N"] * s.iterations / dt, "unit": "observations/s", "cores": 1, "kind": "port",
            "sample": "config 4 cut to %d of %d views per camera (%d observations), %d LM iterations, %.1f s" % (
                views_per_camera, VIEWS_PER_CAMERA, d["N"], s.iterations, dt),
            "lm_iterations_per_s_at_full_size": (d["N"] * s.iterations / dt) / (CAMERAS * VIEWS_PER_CAMERA * BOARD_W * BOARD_H),
            "all_cores": {"value": d["N"] * sa.iterations / dta, "unit": "observations/s", "cores": cores,
                          "note": "same sample, OpenMP over views with per-thread accumulators, %.1f s" % dta}}


def homography_bench(cvx_amd, torch, device, with_cpu, n=CAMERAS * VIEWS_PER_CAMERA, P=BOARD_W * BOARD_H):
    """HomographyEstimator::solveLM for every view of config 4 in one launch (board -> image point pairs with
    0.3 px noise, starts 1 % off), beside the reference's levmar on a sample of the same problems."""
    from cvx_amd.problem import HomographyBatch
    rng = np.random.default_rng(SEED)
    bx, by = np.meshgrid(np.arange(BOARD_W) * 25.0, np.arange(BOARD_H) * 25.0)
    x1 = np.tile(bx.ravel(), n).astype(np.float32); y1 = np.tile(by.ravel(), n).astype(np.float32)
    H = np.array([3.5, 0.3, 400.0, -0.2, 3.6, 300.0, 2e-4, -3e-4, 1.0]) + rng.normal(size=(n, 9)) * np.array(
        [0.1, 0.1, 20, 0.1, 0.1, 20, 1e-4, 1e-4, 0])
    Hr = np.repeat(H, P, axis=0)
    w = Hr[:, 6] * x1 + Hr[:, 7] * y1 + 1.0
    x2 = ((Hr[:, 0] * x1 + Hr[:, 1] * y1 + Hr[:, 2]) / w + rng.normal(0, 0.3, n * P)).astype(np.float32)
    y2 = ((Hr[:, 3] * x1 + Hr[:, 4] * y1 + Hr[:, 5]) / w + rng.normal(0, 0.3, n * P)).astype(np.float32)
    H0 = H * (1 + rng.normal(0, 1e-2, (n, 9)))
    H0 /= H0[:, 8:9]
    off = (np.arange(n + 1) * P).astype(np.int32)
    b = HomographyBatch(off, x1, y1, x2, y2, device=device)
    b.set(H0); b.run()
    reps = 5
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        b.set(H0)
        b.run()
    dt = (time.perf_counter() - t) / reps
    Hf, info, ret = b.get()
    # the step before it in the reference's flow, HomographyEstimator::solve (the normalised DLT, homography.cpp:46-93),
    # and both in a row with the estimates staying on the device
    b.dlt(fetch=False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        b.dlt(fetch=False)
    dt_dlt = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    for _ in range(reps):
        b.dlt(fetch=False)
        b.run()
    dt_both = (time.perf_counter() - t) / reps
    Hd = b.dlt()
    res = b.residuals()
    b.close()
    out = {"problems": n, "point_pairs_per_problem": P, "ms_per_batch": 1e3 * dt, "problems_per_s": n / dt,
           "lm_iterations_total": int(info[:, 5].sum()), "residual_evaluations_total": int(info[:, 7].sum()),
           "converged": int(np.sum(ret > 0)), "kernel": "k_homography_lm (includes the 0.7 MB upload of the starts)",
           "dlt": {"ms_per_batch": 1e3 * dt_dlt, "problems_per_s": n / dt_dlt, "ms_per_batch_dlt_then_lm": 1e3 * dt_both,
                   "mean_transfer_residual_px": float(res.mean()),
                   "kernels": "k_homography_dlt_reduce + k_homography_dlt_finish, estimates left on the device"}}
    # and the step before that, HomographyEstimator::solveRansac (homography.cpp:151-156): the same views with 20 of the 108
    # pairs pushed 20-200 px away, RANSAC::Parameters' defaults but max_error = 2 px (the default 0.1 px is below the noise),
    # subsets drawn on the device; then solveLM over each view's inliers
    bad = rng.permutation(P)[:20]
    x2o = x2.reshape(n, P).copy(); y2o = y2.reshape(n, P).copy()
    x2o[:, bad] += rng.uniform(20, 200, (n, 20)).astype(np.float32); y2o[:, bad] -= rng.uniform(20, 200, (n, 20)).astype(np.float32)
    x2o = x2o.ravel(); y2o = y2o.ravel()
    br = HomographyBatch(off, x1, y1, x2o, y2o, device=device)
    rparams = dict(max_error=2.0)
    br.ransac(seed=SEED, params=rparams)
    t = time.perf_counter()
    for _ in range(3):
        rout = br.ransac(seed=SEED, params=rparams)
    dt_r = (time.perf_counter() - t) / 3
    t = time.perf_counter()
    rk = br.ransac(seed=SEED, params=rparams, keep_inliers=True)
    rk["inliers"].run()
    Hl, _, retl = rk["inliers"].get()
    dt_rl = time.perf_counter() - t
    rk["inliers"].close()
    out["ransac"] = {"ms_per_batch": 1e3 * dt_r, "problems_per_s": n / dt_r, "hypotheses_per_batch": n * 100,
                     "hypotheses_per_s": n * 100 / dt_r, "ms_ransac_then_lm_over_inliers": 1e3 * dt_rl,
                     "mean_inliers": float(rout["n_inliers"].mean()), "mean_trials_the_loop_ran": float(rout["trials_run"].mean()),
                     "planted_outliers_among_inliers": int(rout["mask"].reshape(n, P)[:, bad].sum()),
                     "lm_converged": int(np.sum(retl > 0)),
                     "note": "all 100 trials of every view fitted (DLT on four pairs) and scored at once, the reference's loop "
                             "walked over the scores; H, inlier counts, the inlier mask and trial counts downloaded"}
    if with_cpu:
        from oracle import oracle as orc
        m = 100
        ru = br.ransac(seed=SEED, params=rparams, want_subsets=True)
        t = time.perf_counter()
        same = 0
        for i in range(m):
            sl = slice(off[i], off[i + 1])
            ok, Ho, inl, tr = orc.homography_ransac(x1[sl], y1[sl], x2o[sl], y2o[sl], ru["subsets"][i], orc.ransac_params(**rparams))
            same += int(np.flatnonzero(ru["mask"][sl]).tolist() == inl.tolist() and tr == ru["trials_run"][i])
        dtc = time.perf_counter() - t
        out["ransac"]["cpu_baseline"] = {"value": m / dtc, "unit": "problems/s", "cores": 1, "kind": "port",
                                         "sample": "first %d of the %d problems on the subsets the device drew (the loop stops after %.1f trials on average), %.2f s" % (m, n, float(ru["trials_run"][:m].mean()), dtc),
                                         "same_inliers_and_trial_count_as_gpu": "%d of %d" % (same, m)}
    br.close()
    if with_cpu:
        from oracle import oracle as orc
        kind = "reference" if orc.levmar_ref() is not None else "port"
        m = 400
        t = time.perf_counter()
        same = 0
        for i in range(m):
            s = slice(off[i], off[i + 1])
            r, h, inf = orc.homography_lm(x1[s], y1[s], x2[s], y2[s], H0[i], engine="reference" if kind == "reference" else "restatement")
            same += int(h.tolist() == Hf[i, :8].tolist())
        dtc = time.perf_counter() - t
        out["cpu_baseline"] = {"value": m / dtc, "unit": "problems/s", "cores": 1, "kind": kind,
                               "sample": "first %d of the %d problems through levmar dlevmar_der, %.2f s" % (m, n, dtc),
                               "bit_identical_to_gpu": "%d of %d" % (same, m)}
        t = time.perf_counter()
        worst = 0.0
        for i in range(m):
            s = slice(off[i], off[i + 1])
            _, Ho = orc.homography_dlt(x1[s], y1[s], x2[s], y2[s])
            worst = max(worst, float(np.max(np.abs(Ho - Hd[i])) / np.max(np.abs(Ho))))
        dtc = time.perf_counter() - t
        out["dlt"]["cpu_baseline"] = {"value": m / dtc, "unit": "problems/s", "cores": 1, "kind": "port",
                                      "sample": "first %d of the %d problems, %.2f s" % (m, n, dtc),
                                      "worst_relative_difference_to_gpu": worst}
    return out


def toolba_bench(cvx_amd, torch, device, with_cpu, V=CAMERAS * VIEWS_PER_CAMERA, bw=BOARD_W, bh=BOARD_H):
    """bundleAdjustment of tools/hand_eye_registration.cpp at config 4's size: one camera
    pose, V views with known pre-transforms, bw x bh free board points, Huber loss; time per LM iteration."""
    from cvx_amd import capi
    from cvx_amd.problem import ToolBA
    rng = np.random.default_rng(SEED + 3)
    P = bw * bh
    gx, gy = np.meshgrid(np.arange(bw) * 0.025, np.arange(bh) * 0.025)
    X = np.stack([gx.ravel(), gy.ravel(), np.zeros(P)], 1).astype(np.float32).astype(np.float64)

    def rot(axis, ang):
        axis = axis / np.linalg.norm(axis, axis=-1, keepdims=True)
        K = np.zeros(axis.shape[:-1] + (3, 3))
        K[..., 0, 1], K[..., 0, 2], K[..., 1, 0] = -axis[..., 2], axis[..., 1], axis[..., 2]
        K[..., 1, 2], K[..., 2, 0], K[..., 2, 1] = -axis[..., 0], -axis[..., 1], axis[..., 0]
        ang = np.asarray(ang)[..., None, None]
        return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    Rg = rot(np.array([1.0, 2.0, 3.0]), 0.4); tg = np.array([0.05, -0.03, 0.12])
    Rb = rot(rng.normal(size=(V, 3)), rng.uniform(0, 0.6, V))
    tb = np.stack([rng.uniform(-0.15, 0.15, V) - gx.mean(), rng.uniform(-0.1, 0.1, V) - gy.mean(), rng.uniform(0.5, 0.9, V)], 1)
    trans = np.concatenate([Rg.T @ Rb, (Rg.T @ (tb - tg)[..., None])], axis=2).reshape(V, 12)     # base <- board
    xc = np.einsum("vij,pj->vpi", Rb, X) + tb[:, None, :]
    fx, fy, cx, cy, k1, k2, k3, p1, p2 = intr = np.array([1000.0, 1005.0, 640.0, 480.0, -0.12, 0.05, 0.01, 1e-3, -1.5e-3])
    xn, yn = xc[..., 0] / xc[..., 2], xc[..., 1] / xc[..., 2]
    r2 = xn * xn + yn * yn
    rc = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    bad = rng.uniform(size=(V, P)) < 0.03
    u = fx * (xn * rc + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)) + cx + rng.normal(0, 0.3, (V, P)) + bad * rng.normal(0, 20, (V, P))
    w = fy * (yn * rc + 2 * p2 * xn * yn + p1 * (r2 + 2 * yn * yn)) + cy + rng.normal(0, 0.3, (V, P)) + bad * rng.normal(0, 20, (V, P))
    view_off = (np.arange(V + 1) * P).astype(np.int32); pid = np.tile(np.arange(P, dtype=np.int32), V)
    ox = u.astype(np.float32).astype(np.float64).ravel(); oy = w.astype(np.float32).astype(np.float64).ravel()
    qw = np.sqrt(1 + np.trace(Rg)) / 2
    q = np.array([qw, (Rg[2, 1] - Rg[1, 2]) / (4 * qw), (Rg[0, 2] - Rg[2, 0]) / (4 * qw), (Rg[1, 0] - Rg[0, 1]) / (4 * qw)])
    p0 = np.concatenate([q + rng.normal(0, 5e-3, 4), tg + rng.normal(0, 0.01, 3)])
    g = ToolBA(view_off, pid, ox, oy, trans, intr, P, device=device)
    op = capi.options("toolba")
    op.max_iterations = 10
    op.inner_iterations = 0      # the LM iteration itself, as for refine above
    g.solve(p0, X, op)
    torch.cuda.synchronize()
    reps = 5
    t = time.perf_counter()
    for _ in range(reps):
        pp, xx, sm = g.solve(p0, X, op)
    dt = (time.perf_counter() - t) / (reps * max(sm.iterations, 1))
    g.close()
    out = {"views": V, "points": P, "observations": V * P, "ms_per_lm_iteration": 1e3 * dt,
           "observations_per_s": V * P / dt, "iterations_per_solve": sm.iterations,
           "cost": [sm.initial_cost, sm.final_cost], "translation_error_m": float(np.max(np.abs(pp[4:] - tg)))}
    if with_cpu:
        from oracle import oracle as orc
        Vs = V
        o = orc.ToolBA(view_off[:Vs + 1], pid[:Vs * P], ox[:Vs * P], oy[:Vs * P], trans[:Vs], intr, P)
        oo = orc.options("toolba"); oo.max_iterations = 5; oo.inner_iterations = 0
        t = time.perf_counter()
        so = o.solve(p0, X, oo)[2]
        dtc = time.perf_counter() - t
        out["cpu_baseline"] = {"value": Vs * P * so.iterations / dtc, "unit": "observations/s", "cores": 1, "kind": "port",
                               "sample": "first %d views (%d observations), %d LM iterations, dense normal equations, %.1f s" % (
                                   Vs, Vs * P, so.iterations, dtc)}
    return out


def lm_iteration_ms(cvx_amd, capi, torch, spec, device, steps=20, hook=None, rank=0, world=1, fence=None, host_exchange=False,
                    weak=False):
    """ms per LM iteration (the same step as the headline: no inner-iteration passes, every iteration asked is run) of the
    synthetic calibration problem `spec` = (cameras, views per camera, board w, board h, seed); with world > 1 the
    problem's views are cut over the ranks (strong scaling) and the exchange goes through `hook`.  weak: every rank
    holds a shard of `spec`'s size (its own poses and observations; cameras and board shared), one problem of world x
    that many views."""
    C_, VP, bw, bh, seed = spec
    full = cvx_amd.synth_calib(C_, VP, bw, bh, seed)
    intr0, pts0 = full["intr_init"], full["points_init"]       # the generator is deterministic: the same on every rank
    if weak and world > 1:
        d = cvx_amd.synth_calib(C_, VP, bw, bh, seed + 1000 * rank) if rank else full
        n_total, v_total = world * full["N"], world * full["V"]
    elif world > 1:
        n_total, v_total = full["N"], full["V"]
        first = cvx_amd.partition_views(full["view_off"], world)
        d = cvx_amd.shard_calib(full, first, rank)
    else:
        n_total, v_total = full["N"], full["V"]
        d = full
    del full
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], C_, bw * bh,
                          device=device, stream=stream.cuda_stream)
        if world > 1:
            g.set_exchange(hook, rank, world, device_buffers=not host_exchange)
        o = capi.options("refine")
        o.inner_iterations = 0
        o.function_tolerance = o.gradient_tolerance = o.parameter_tolerance = -1.0
        o.max_iterations = ITERS_PER_SOLVE

        def run(n):
            for _ in range(max(1, n // ITERS_PER_SOLVE)):
                g.set_params(intr0, d["poses_init"], pts0)
                s = g.solve(o)
                if s.iterations != ITERS_PER_SOLVE:
                    raise RuntimeError("solve stopped after %d iterations (termination %d)" % (s.iterations, s.termination))
            return s
        run(ITERS_PER_SOLVE)
        if fence:
            fence()
        else:
            torch.cuda.synchronize()
        t = time.perf_counter()
        s = run(steps)
        if fence:
            fence()
        else:
            torch.cuda.synchronize()
        dt = time.perf_counter() - t
        ns = g.ns
        g.close()
    n_steps = max(1, steps // ITERS_PER_SOLVE) * ITERS_PER_SOLVE
    return {"observations": n_total, "pose_blocks": v_total, "shared_unknowns": ns, "steps": n_steps,
            "views_this_rank": d["V"], "seconds": dt, "ms_per_lm_iteration": 1e3 * dt / n_steps,
            "cost": [s.initial_cost, s.final_cost]}


def handeye_cpu_baseline(q, x0, iters):
    """solveNonLinear (handeye.cpp:272-333) of the oracle on the same motion pairs, one thread"""
    from oracle import oracle as orc
    ho = orc.HandEye(*q)
    op = orc.options("handeye")
    op.function_tolerance = op.gradient_tolerance = op.parameter_tolerance = -1.0
    op.max_iterations = iters
    t = time.perf_counter()
    p, s = ho.solve(x0, op)
    dt = time.perf_counter() - t
    n = max(s.iterations, 1)
    return {"value": 1e3 * dt / n, "unit": "ms per LM iteration", "cores": 1, "kind": "port",
            "sample": "the same %d motion pairs, %d LM iterations, %.3f s" % (len(q[0]), n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs 1 / 3 / 5 block (profiling runs: keeps every kernel's launches at config 4's size)")
    ap.add_argument("--cpu-views", type=int, default=1250, help="views per camera of the CPU sample")
    ap.add_argument("--cpu-iters", type=int, default=10)
    ap.add_argument("--host-exchange", action="store_true", help="all-reduce host copies instead of device buffers")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong: one config-4 problem sharded by view over the ranks; weak: a config-4-sized shard per rank")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
        a.gpus = world

    import torch
    import torch.distributed as dist
    import cvx_amd
    from cvx_amd import capi

    if capi.lib().cvx_device_count() < 1:
        sys.exit("bench.py needs a HIP device: the product has no CPU path")
    # rehearsal on a one-GPU box (not a measurement): CVX_BENCH_REHEARSAL=1 puts every rank on cuda:0 and exchanges
    # over gloo, so that the N > 1 code path -- sharding, device-buffer hook, packed exchange -- runs end to end
    rehearsal = world > 1 and os.environ.get("CVX_BENCH_REHEARSAL") == "1"

    if rehearsal:       # kill -USR1 <pid> prints every thread's Python stack (where a stalled rehearsal sits)
        import faulthandler
        import signal
        faulthandler.register(signal.SIGUSR1, all_threads=True)

    def trace(msg):
        if rehearsal:
            sys.stderr.write("[rank %d] %s\n" % (rank, msg)); sys.stderr.flush()
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # ---- this rank's shard of the global problem, the shared block replicated ----
    gloo = world > 1 and dist.get_backend() == "gloo"

    def on_wire(t):     # gloo (rehearsal): CPU tensors only; nccl: device tensors
        return t if gloo else t.cuda()
    if a.scaling == "strong" or world == 1:
        # one config-4 problem (the same on every rank: the generator is deterministic), this rank's range of views
        full_problem = cvx_amd.synth_calib(CAMERAS, VIEWS_PER_CAMERA, BOARD_W, BOARD_H, SEED)
        first = cvx_amd.partition_views(full_problem["view_off"], world)
        d = cvx_amd.shard_calib(full_problem, first, rank) if world > 1 else full_problem
        n_obs_total = full_problem["N"]
        del full_problem
    else:
        d = cvx_amd.synth_calib(CAMERAS, VIEWS_PER_CAMERA, BOARD_W, BOARD_H, SEED + 1000 * rank)
        n_obs_total = world * d["N"]
    intr0, pts0 = d["intr_init"].copy(), d["points_init"].copy()
    if world > 1:       # every rank must start from the same shared parameters
        t = on_wire(torch.from_numpy(np.concatenate([intr0.ravel(), pts0.ravel()])))
        dist.broadcast(t, 0)
        flat = t.cpu().numpy()
        intr0 = flat[:intr0.size].reshape(intr0.shape).copy()
        pts0 = flat[intr0.size:].reshape(pts0.shape).copy()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        g = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], CAMERAS,
                          BOARD_W * BOARD_H, device=local_rank, stream=stream.cuda_stream)
        hook = None
        if world > 1:
            hook = make_hook(torch, dist, capi)
            g.set_exchange(hook, rank, world, device_buffers=not a.host_exchange)
        opt = capi.options("refine")
        opt.function_tolerance = opt.gradient_tolerance = opt.parameter_tolerance = -1.0   # run every iteration asked
        # The timed step is the LM iteration itself (linearise, Schur, solve, candidate).  Ceres' inner iterations
        # (calibration.cpp:424) run on the candidates of the first iterations of a solve only -- they switch
        # themselves off once a pass gains less than 1e-3 -- so they are not part of the steady-state step; what they
        # cost is measured below (`inner_iterations` in the line: one pass, and the whole refine with and without).
        opt.inner_iterations = 0

        def run(n_steps):
            done = 0
            while done < n_steps:
                k = min(ITERS_PER_SOLVE, n_steps - done)
                opt.max_iterations = k
                g.set_params(intr0, d["poses_init"], pts0)
                s = g.solve(opt)
                if s.iterations != k:
                    raise RuntimeError("solve stopped after %d of %d iterations (termination %d)" % (s.iterations, k, s.termination))
                done += k
            return s

        def fence():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        trace("handle ready")
        run(a.warmup)
        trace("warm-up done")
        fence()
        t0 = time.perf_counter()
        last = run(a.steps)
        fence()
        dt = time.perf_counter() - t0
        trace("timed region done")
        # a second, untimed pass with per-phase hipEvents on the handle's stream (the event records cost a few
        # microseconds of stream bubbles each, so they stay out of the timed region)
        g.enable_timing(True)
        run(a.steps)
        torch.cuda.synchronize()
        phases = g.phase_times()
        g.enable_timing(False)
        geo = g.geometry()
        # the Schur kernel alone, 20 launches back to back between two hipEvents on the handle's stream: the
        # duration rocprofv3 reports for it (an event pair around a single launch inside a running iteration
        # reads 15-20 us more)
        trace("phase pass done")
        schur_ms = g.time_schur(20)
        trace("schur timing done")

        # ---- one whole refine() with the reference's own options (calibration.cpp:420-428), start to convergence ----
        fo = capi.options("refine")
        if world > 1:
            fo.inner_iterations = 0      # the scaling runs measure the LM iteration; the passes on shards are covered by tests/test_gpu_multirank.py
        g.set_params(intr0, d["poses_init"], pts0)
        g.solve(fo)                      # first use builds the inner-iteration ordering and lists (host work)
        g.set_params(intr0, d["poses_init"], pts0)
        fence()
        tf = time.perf_counter()
        fs = g.solve(fo)
        torch.cuda.synchronize()
        trace("full refine done")
        full = {"iterations": fs.iterations, "termination": fs.termination, "ms_total": 1e3 * (time.perf_counter() - tf),
                "initial_cost": fs.initial_cost, "final_cost": fs.final_cost,
                "inner_iterations": bool(fo.inner_iterations), "inner_iteration_steps": fs.num_inner_iteration_steps}
        inner = None
        if world == 1:
            fo.inner_iterations = 0
            g.set_params(intr0, d["poses_init"], pts0)
            fence()
            tf = time.perf_counter()
            fs0 = g.solve(fo)
            torch.cuda.synchronize()
            ms0 = 1e3 * (time.perf_counter() - tf)
            g.set_params(intr0, d["poses_init"], pts0)
            fence()
            tf = time.perf_counter()
            c_pass = g.inner_pass()
            ms_pass = 1e3 * (time.perf_counter() - tf)
            inner = {"in_timed_steps": False, "one_pass_from_start_ms": ms_pass, "cost_before": fs.initial_cost,
                     "cost_after_pass": c_pass, "refine_ms_with": full["ms_total"], "refine_iterations_with": fs.iterations,
                     "refine_ms_without": ms0, "refine_iterations_without": fs0.iterations,
                     "final_cost_with": fs.final_cost, "final_cost_without": fs0.final_cost,
                     # the whole refine with the reference's options, set-up and passes included, in the line's unit
                     "observation_iterations_per_s_whole_refine_with": d["N"] * fs.iterations / (full["ms_total"] * 1e-3),
                     "observation_iterations_per_s_whole_refine_without": d["N"] * fs0.iterations / (ms0 * 1e-3)}

        # ---- what one drop-in refine() costs end to end: a handle's whole life around the solve above ----
        end_to_end = None
        if world == 1:
            oxf, oyf = d["obs_x"].astype(np.float32), d["obs_y"].astype(np.float32)     # cv::Point2f, as the adaptor hands them over
            for _ in range(2):
                tl = [time.perf_counter()]
                g2 = cvx_amd.Calib(d["cam_of_view"], d["view_off"], d["point_id"], oxf, oyf, CAMERAS,
                                   BOARD_W * BOARD_H, device=local_rank, prepare_inner=True)
                tl.append(time.perf_counter())
                g2.set_params(intr0, d["poses_init"], pts0)
                fs2 = g2.solve(capi.options("refine"))
                tl.append(time.perf_counter())
                g2.get_params()
                g2.close()
                tl.append(time.perf_counter())
            end_to_end = {"create_ms": 1e3 * (tl[1] - tl[0]), "set_and_solve_ms": 1e3 * (tl[2] - tl[1]),
                          "get_and_destroy_ms": 1e3 * (tl[3] - tl[2]), "total_ms": 1e3 * (tl[3] - tl[0]),
                          "iterations": fs2.iterations, "inner_iteration_steps": fs2.num_inner_iteration_steps,
                          "note": "cvx_calib_create (13 MB uploaded from pageable memory: float observations, int point ids) -> set_params -> "
                                  "cvx_calib_solve with the reference's options, first use of the inner iterations included "
                                  "(its host half prepared on a host thread from inside create) -> get_params -> destroy"}

        # ---- evaluation mode: residuals + Jacobian blocks materialised in HBM (k_eval), hipEvent-timed ----
        g.set_params(intr0, d["poses_init"], pts0)
        g.evaluate_device()
        torch.cuda.synchronize()
        n_eval = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n_eval):
            g.evaluate_device()
        e1.record(stream)
        torch.cuda.synchronize()
        eval_ms = e0.elapsed_time(e1) / n_eval
        g.close()

    # ---- the other calibration configurations of BASELINE.json, the same step (ms per LM iteration) ----
    configs = config5_strong = config4_weak = None
    if world == 1 and not a.no_configs:
        configs = {}
        for name, spec in (("config1", (1, 20, 9, 6, 20260001)), ("config3", (2, 500, 12, 9, 20260003)),
                           ("config5", (8, 12500, 12, 9, 20260005))):
            r = lm_iteration_ms(cvx_amd, capi, torch, spec, local_rank, steps=20)
            r["observation_iterations_per_s"] = r["observations"] * r["steps"] / r["seconds"]
            configs[name] = r
    elif world > 1:
        # north_star's config 5 ("100k views x 108 corners, 8xMI355X"): ONE 10.8 M-observation problem cut by view over
        # the ranks -- the configuration where the per-rank kernels dominate the replicated reduced solve
        trace("config 5, strong")
        r = lm_iteration_ms(cvx_amd, capi, torch, (8, 12500, 12, 9, 20260005), local_rank, steps=20, hook=hook, rank=rank,
                            world=world, fence=fence, host_exchange=a.host_exchange)
        tt = on_wire(torch.tensor([r["seconds"]], dtype=torch.float64))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        r["seconds"] = float(tt.item())
        r["ms_per_lm_iteration"] = 1e3 * r["seconds"] / r["steps"]
        r["value"] = r["observations"] * r["steps"] / r["seconds"]
        r.update(unit="observations/s", n_gpus=world, scaling="strong",
                 workload="BASELINE config 5: 8 cameras x 12500 views x 12x9 board, ONE problem sharded by view over %d GPUs" % world)
        config5_strong = r
        trace("config 5 done")
        # and the other way to read "1M observations at 1/2/4/8 GPUs": a config-4-sized shard PER GPU (the rig's cameras
        # and board shared, every rank its own 10 000 pose blocks), unless that is what the headline already measures
        if a.scaling == "strong":
            r = lm_iteration_ms(cvx_amd, capi, torch, (CAMERAS, VIEWS_PER_CAMERA, BOARD_W, BOARD_H, SEED), local_rank, steps=20,
                                hook=hook, rank=rank, world=world, fence=fence, host_exchange=a.host_exchange, weak=True)
            tt = on_wire(torch.tensor([r["seconds"]], dtype=torch.float64))
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            r["seconds"] = float(tt.item())
            r["ms_per_lm_iteration"] = 1e3 * r["seconds"] / r["steps"]
            r["value"] = r["observations"] * r["steps"] / r["seconds"]
            r.update(unit="observations/s", n_gpus=world, scaling="weak",
                     workload="a BASELINE-config-4-sized shard per GPU: 8 cameras, %d x 10000 pose blocks, ONE problem" % world)
            config4_weak = r
            trace("config 4 weak done")

    # ---- BASELINE config 2: hand-eye AX = XB refinement, 200 poses (launch-latency-bound by construction:
    #      SURVEY.md 8d asks for time per LM iteration only) ----
    handeye = None
    if rank == 0:
        handeye = {}
        for name, moving in (("fixed_camera_all_pairs", False), ("moving_camera_consecutive_pairs", True)):
            G, T, xgt, x0 = cvx_amd.synth_handeye(200, 20260002, moving=moving)
            A, B = cvx_amd.handeye_pairs(G, T, moving=moving)
            q = cvx_amd.pairs_to_quat_trans(A, B)
            he = cvx_amd.HandEye(*q, device=local_rank)
            ho = capi.options("handeye")
            ho.function_tolerance = ho.gradient_tolerance = ho.parameter_tolerance = -1.0
            ho.max_iterations = 20
            he.solve(x0, ho)
            torch.cuda.synchronize()
            th = time.perf_counter()
            reps = 10
            for _ in range(reps):
                p, hs = he.solve(x0, ho)
            torch.cuda.synchronize()
            handeye[name] = {"pairs": int(len(A)), "ms_per_lm_iteration": 1e3 * (time.perf_counter() - th) / (reps * max(hs.iterations, 1)),
                             "iterations_per_solve": hs.iterations, "final_cost": hs.final_cost}
            # and the solve as solveNonLinear itself runs it (handeye.cpp:310-319: Ceres' default tolerances)
            hd = capi.options("handeye")
            he.solve(x0, hd)
            torch.cuda.synchronize()
            th = time.perf_counter()
            for _ in range(reps):
                p, hs = he.solve(x0, hd)
            handeye[name]["solve_with_the_reference_options"] = {"ms_per_solve": 1e3 * (time.perf_counter() - th) / reps,
                                                                 "iterations": hs.iterations, "accepted": hs.num_successful,
                                                                 "termination": hs.termination}
            he.close()
            # a handle's whole life, as the adaptor's solveNonLinear runs it: create (upload of the pairs), solve, destroy
            for _ in range(2):
                th = time.perf_counter()
                he2 = cvx_amd.HandEye(*q, device=local_rank)
                he2.solve(x0, hd)
                he2.close()
                life = 1e3 * (time.perf_counter() - th)
            handeye[name]["solve_with_the_reference_options"]["ms_create_solve_destroy"] = life
            if world == 1 and not a.no_cpu_baseline:
                handeye[name]["cpu_baseline"] = handeye_cpu_baseline(q, x0, 20 if moving else 5)

    # ---- batched homography refinement, one LM problem per view of config 4 ----
    homography = toolba = None
    if rank == 0:
        homography = homography_bench(cvx_amd, torch, local_rank, with_cpu=(world == 1 and not a.no_cpu_baseline))
        toolba = toolba_bench(cvx_amd, torch, local_rank, with_cpu=(world == 1 and not a.no_cpu_baseline))

    if world > 1:
        tt = on_wire(torch.tensor([dt], dtype=torch.float64))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        N = d["N"]
        V = d["V"]
        ns = geo["ns"]
        # algorithmic work of one launch of each phase kernel (DESIGN.md "Kernels")
        work = {
            # Z^T Z over the upper triangle of the (ns + 1) columns (reduced matrix + gradient column)
            "schur": ("mfma", 2.0 * 6 * V * (ns + 1) * (ns + 2) / 2, "TFLOP/s", F64_PEAK_TFLOPS, 1e12),
            # residual + Jacobian + 16x16 Gram + point blocks: SURVEY.md 8(d) 1 100 flop / observation
            "normal_eq": ("mfma", 1100.0 * N, "TFLOP/s", F64_PEAK_TFLOPS, 1e12),
            # slab reduction + assembly: read every partial tile once, write S once
            "assemble": ("hbm", 8.0 * (geo["n_chunks"] * geo["tile_pairs"] * 256 + ns * ns), "GB/s", HBM_PEAK_GBS, 1e9),
            "backsubst": ("hbm", 8.0 * 6 * V * geo["ZC"], "GB/s", HBM_PEAK_GBS, 1e9),
            "cost": ("hbm", 24.0 * N, "GB/s", HBM_PEAK_GBS, 1e9),
            # damped reduced solve: Cholesky ns^3 / 3 flop (latency-bound chain of tile columns)
            "solve": ("mfma", ns ** 3 / 3.0, "TFLOP/s", F64_PEAK_TFLOPS, 1e12),
        }
        per_launch = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in phases.items()}
        if schur_ms > 0:
            per_launch["schur"] = schur_ms
        per_iter = {k: phases[k][0] / a.steps for k in phases}
        dom = max(per_iter, key=lambda k: per_iter[k])
        bound, amount, unit, peak, scale = work[dom]
        achieved = amount / (per_launch[dom] * 1e-3) / scale if per_launch[dom] > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and world == 1:
            try:
                traffic = json.load(open(tpath)).get(dom, {}).get("hbm_bytes")
            except Exception:
                traffic = None
        roof = {"kernel": {"schur": "k_schur_syrk4", "normal_eq": "k_normal_eq", "assemble": "k_reduce_all+k_assemble",
                           "backsubst": "k_backsubst_cost_step", "cost": "k_final_scalars",
                           "solve": "k_chol_step"}[dom],
                "bound": bound, "achieved": achieved, "peak": peak, "unit": unit,
                "frac": achieved / peak, "traffic": traffic,
                "avg_launch_ms": per_launch[dom], "algorithmic_per_launch": amount,
                "timing": "20 back-to-back launches between two hipEvents" if dom == "schur" and schur_ms > 0
                          else "hipEvent pair around each launch of the phase"}
        if dom == "schur" and per_launch[dom] > 0:
            # three ways to count the contraction's flops: `frac` uses SURVEY.md 8(d)'s figure,
            # the upper triangle of all ns + 1 columns; a view's Z row is zero outside its own camera's columns, so
            # the non-zero part is smaller; the kernel issues whole 16x16 blocks over the columns of ONE camera
            t_s = per_launch[dom] * 1e-3
            cols = 3 * BOARD_W * BOARD_H + 8 + 1      # free points + the view's own camera (k3 fixed) + gradient
            nonzero = 2.0 * 6 * V * cols * (cols + 1) / 2
            issued = 512.0 * geo["tile_pairs"] * 6 * V
            roof.update({
                # the headline fraction counts NON-ZERO work only: a view's Z row is zero outside its own camera's
                # columns, so the blocks between different cameras' columns are structural zeros the kernel never issues
                "achieved": nonzero / t_s / 1e12, "frac": nonzero / t_s / 1e12 / peak, "algorithmic_per_launch": nonzero,
                "algorithmic_dense_triangle_per_launch": amount, "achieved_dense_triangle": amount / t_s / 1e12,
                "frac_dense_triangle": amount / t_s / 1e12 / peak,
                "issued_per_launch": issued, "frac_issued": issued / t_s / 1e12 / peak,
                "note": "frac: upper triangle of the %d columns a view's Z row is non-zero in (own camera + points + "
                        "gradient); frac_dense_triangle: SURVEY.md 8(d)'s figure, the upper triangle of all %d columns, "
                        "which counts structural zeros; frac_issued: what the MFMA pipe executes, %d blocks of 16x16 per "
                        "4 rows" % (cols, ns + 1, geo["tile_pairs"])})
        # the iteration as a whole against the fp64 peak: SURVEY.md 8(d) flops per step (1 100 per observation on the
        # vector ALU + the symmetric half of the Schur contraction, 6 ns^2 per view) / wall time per step
        step_flops = 1100.0 * N + 6.0 * ns * ns * V
        step_s = dt / a.steps
        out = {
            "metric": "LM iterations/sec + residual/Jacobian evals/sec at 1M obs (observations processed by LM iterations per second)",
            "value": n_obs_total * a.steps / dt,
            "unit": "observations/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps,
            "lm_iterations_per_s": a.steps / dt,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("BASELINE config 4: 8 cameras x 1250 views x 12x9 board (10000 pose blocks, 1080000 "
                                    "observations, 388 shared unknowns)%s, CameraCalibration::refine LM iteration, "
                                    "k3 fixed, board points free") % (
                                        "" if world == 1 else (", ONE problem sharded by view over %d GPUs" % world)
                                        if a.scaling == "strong" else ", one such shard PER GPU"),
                       "observations_total": n_obs_total,
                       "observations_rank0": N, "views_rank0": V, "shared_unknowns": ns,
                       "parallelism": "views sharded over %d GPU(s), one all-reduce of the reduced system per iteration" % world,
                       "iterations_per_solve": ITERS_PER_SOLVE, "accepted_steps_in_last_solve": last.num_successful,
                       "inner_iterations": "not in the timed steps (Ceres runs them on a solve's first candidates only, until a pass gains < 1e-3); their cost is in `inner_iterations`",
                       "final_cost_after_%d_iterations" % ITERS_PER_SOLVE: last.final_cost},
            "roofline": roof,
            "iteration": {"flops_per_step_rank0": step_flops, "achieved": step_flops / step_s / 1e12,
                          "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": step_flops / step_s / 1e12 / F64_PEAK_TFLOPS,
                          "note": "SURVEY.md 8(d) flops of one LM iteration on rank 0's shard / ms_per_step"},
            "phases_ms_per_iteration": per_iter,
            "device_ms_per_iteration": sum(per_iter.values()),
            "eval": {"kernel": "k_eval", "ms": eval_ms, "observations_per_s": N / (eval_ms * 1e-3),
                     "roofline": {"bound": "hbm", "achieved": EVAL_BYTES_PER_OBS * N / (eval_ms * 1e-3) / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": EVAL_BYTES_PER_OBS * N / (eval_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}},
        }
        # every phase of the iteration: time per launch and per iteration (hipEvent pairs of the phase pass), its
        # algorithmic work and the fraction of the roofline that bounds it
        kern = {}
        names = {"schur": "k_schur_syrk4", "normal_eq": "k_normal_eq", "assemble": "k_reduce_all + k_assemble",
                 "backsubst": "k_backsubst_cost_step", "cost": "k_final_scalars", "solve": "k_chol_step x %d" % geo.get("tile_columns", 0)}
        for k, (bnd, amt, un, pk, sc) in work.items():
            if k == "schur":
                amt = roof.get("algorithmic_per_launch", amt) if dom == "schur" else amt
            ms_it = per_iter.get(k, 0.0)
            ms_l = per_launch.get(k, 0.0) if k == "schur" else ms_it
            kern[k] = {"kernel": names[k], "us_per_iteration": 1e3 * ms_it, "share_of_device_time": ms_it / max(sum(per_iter.values()), 1e-30),
                       "bound": bnd, "algorithmic": amt, "achieved": (amt / (ms_l * 1e-3) / sc) if ms_l > 0 else 0.0, "unit": un,
                       "peak": pk, "frac": (amt / (ms_l * 1e-3) / sc / pk) if ms_l > 0 else 0.0}
        out["kernels"] = kern
        out["host_and_gaps_us_per_step"] = 1e3 * (1e3 * dt / a.steps - sum(per_iter.values()))
        if configs is not None:
            out["configs"] = configs
        if config5_strong is not None:
            out["config5_strong"] = config5_strong
        if config4_weak is not None:
            out["config4_weak"] = config4_weak
        out["full_refine_to_convergence"] = full
        out["inner_iterations"] = inner
        out["handeye"] = handeye
        out["homography"] = homography
        if end_to_end is not None:
            out["refine_end_to_end"] = end_to_end
        out["tool_bundle_adjustment"] = toolba
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_views, a.cpu_iters)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
