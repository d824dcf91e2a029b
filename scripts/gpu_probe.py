"""Scratch GPU probe: prints HIP-vs-oracle differences (development aid, not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cvx_amd
from cvx_amd import capi
from oracle import oracle as orc

def rel(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))

def probe(C, VP, bw, bh, seed, ragged=0, **fix):
    d = cvx_amd.synth_calib(C, VP, bw, bh, seed, ragged=ragged)
    P = bw * bh
    print("== calib C=%d V=%d P=%d N=%d ragged=%d %s" % (C, d['V'], P, d['N'], ragged, fix), flush=True)
    g = cvx_amd.Calib(d['cam_of_view'], d['view_off'], d['point_id'], d['obs_x'], d['obs_y'], C, P, **fix)
    o = orc.Calib(d['cam_of_view'], d['view_off'], d['point_id'], d['obs_x'], d['obs_y'], C, P, **fix)
    g.set_params(d['intr_init'], d['poses_init'], d['points_init'])
    print("geometry", g.geometry(), flush=True)
    c, r, Jc, Jp, Jx = g.evaluate()
    ro, Jco, Jpo, Jxo = o.eval(d['intr_init'], d['poses_init'], d['points_init'])
    co = o.cost(d['intr_init'], d['poses_init'], d['points_init'])
    print("eval rel: cost %.2e r %.2e Jc %.2e Jp %.2e Jx %.2e" % (abs(c - co) / co, rel(r, ro), rel(Jc, Jco), rel(Jp, Jpo), rel(Jx, Jxo)), flush=True)
    print("cost kernel rel %.2e" % (abs(g.cost() - co) / co), flush=True)
    # linear step with Ceres-style damping
    g.init_scaling(True)
    Vv, gv, U, gs, W, cost = o.normal_blocks(d['intr_init'], d['poses_init'], d['points_init'], with_W=False)
    radius = 1e4
    dU = np.diag(U); s2s = 1.0 / (1.0 + np.sqrt(dU)) ** 2
    Ds = np.clip(s2s * dU, 1e-6, 1e32) / (s2s * radius)
    dV = np.einsum('vii->vi', Vv).reshape(-1); s2p = 1.0 / (1.0 + np.sqrt(dV)) ** 2
    Dp = np.clip(s2p * dV, 1e-6, 1e32) / (s2p * radius)
    S, gred, _ = o.reduced_system(d['intr_init'], d['poses_init'], d['points_init'], Dp, Ds)
    st = g.linear_step(1.0 / radius, 0.0, 1e-6, 1e32)
    print("linear: S %.2e gred %.2e cost %.2e" % (rel(st['S'], S), rel(st['gred'], gred), abs(st['cost'] - cost) / cost), flush=True)
    if g.ns:
        ds = np.linalg.solve(S, -gred)
        print("  delta_shared rel %.2e  cond(S) %.2e" % (rel(st['delta_shared'], ds), np.linalg.cond(S)), flush=True)
    for kind in ("refine", "levmar"):
        g.set_params(d['intr_init'], d['poses_init'], d['points_init'])
        t = time.time(); s = g.solve(capi.options(kind)); tg = time.time() - t
        t = time.time(); io, po, xo, so = o.solve(d['intr_init'], d['poses_init'], d['points_init'], orc.options(kind)); to = time.time() - t
        ig, pg, xg = g.get_params()
        n = min(s.trace_len, so.trace_len)
        tc, toc = s.trace()['cost'][:n], so.trace()['cost'][:n]
        print(" %s: gpu it=%d term=%d cost=%.12g (%.3fs) | orc it=%d term=%d cost=%.12g (%.3fs)" % (kind, s.iterations, s.termination, s.final_cost, tg, so.iterations, so.termination, so.final_cost, to), flush=True)
        print("   trace rel diffs:", " ".join("%.1e" % (abs(a - b) / b) for a, b in zip(tc, toc)), flush=True)
        print("   params rel: intr %.2e poses %.2e points %.2e" % (rel(ig, io), rel(pg, po), rel(xg, xo)), flush=True)
    g.close()

def probe_he(n, seed, moving):
    G, T, xg, x0 = cvx_amd.synth_handeye(n, seed, moving=moving)
    A, B = cvx_amd.handeye_pairs(G, T, moving=moving)
    qa, ta, qb, tb = cvx_amd.pairs_to_quat_trans(A, B)
    print("== handeye n=%d pairs=%d moving=%d" % (n, len(A), moving), flush=True)
    g = cvx_amd.HandEye(qa, ta, qb, tb); o = orc.HandEye(qa, ta, qb, tb)
    print(" cost rel %.2e" % (abs(g.cost(x0) - o.cost(x0)) / o.cost(x0)))
    H, gg, c = g.normal_eq(x0); Ho, go, co = o.normal_eq(x0)
    print(" H %.2e g %.2e cost %.2e" % (rel(H, Ho), rel(gg, go), abs(c - co) / co))
    r, Jq, Jt = g.evaluate(x0)
    ro = np.zeros_like(r); Jqo = np.zeros_like(Jq); Jto = np.zeros_like(Jt)
    for i in range(len(A)):
        ro[i], Jqo[i], Jto[i] = orc.handeye(x0[:4], x0[4:], qa[i], ta[i], qb[i], tb[i])
    print(" eval r %.2e Jq %.2e Jt %.2e" % (rel(r, ro), rel(Jq, Jqo), rel(Jt, Jto)))
    for kind in ("handeye", "levmar"):
        p, s = g.solve(x0, capi.options(kind)); po, so = o.solve(x0, orc.options(kind))
        n_ = min(s.trace_len, so.trace_len)
        print(" %s: gpu it=%d term=%d cost=%.12g | orc it=%d term=%d cost=%.12g | p rel %.2e" % (kind, s.iterations, s.termination, s.final_cost, so.iterations, so.termination, so.final_cost, rel(p, po)))
        print("   trace rel:", " ".join("%.1e" % (abs(a - b) / b) for a, b in zip(s.trace()['cost'][:n_], so.trace()['cost'][:n_])))
    g.close()

if __name__ == "__main__":
    print("devices", capi.lib().cvx_device_count(), flush=True)
    probe(1, 20, 9, 6, 20260001)
    probe(1, 20, 9, 6, 20260001, fix_points=True)
    probe(2, 7, 12, 9, 11, ragged=4)
    probe(3, 5, 20, 15, 12, ragged=3, fix_k3=False)
    probe(1, 6, 9, 6, 13, fix_intrinsics=True)
    probe_he(200, 20260002, False)
    probe_he(200, 20260002, True)
