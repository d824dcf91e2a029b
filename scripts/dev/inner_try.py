"""development aid: inner-iteration pass and solve on the GPU against the oracle"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import cvx_amd
from cvx_amd import capi
from cvx_amd.problem import inner_ordering
from oracle import oracle as orc

def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))

for spec, rag, fix in (((1, 20, 9, 6, 20260001), 0, {}), ((2, 7, 12, 9, 11), 4, {}), ((3, 5, 20, 15, 12), 3, {"fix_k3": False}),
                       ((2, 30, 4, 3, 5), 2, {}), ((1, 6, 9, 6, 13), 0, {"fix_intrinsics": True})):
    C_, VP, bw, bh, seed = spec
    d = cvx_amd.synth_calib(C_, VP, bw, bh, seed, ragged=rag)
    args = (d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], C_, bw * bh)
    g, o = cvx_amd.Calib(*args, **fix), orc.Calib(*args, **fix)
    go, no = o.inner_ordering()
    gg, ng = inner_ordering(d["cam_of_view"], d["view_off"], d["point_id"], C_, bw * bh,
                            fix.get("fix_intrinsics", False), fix.get("fix_points", False))
    print(spec, "ordering equal", no == ng and np.array_equal(go, gg), "sets", ng)
    x = (d["intr_init"], d["poses_init"], d["points_init"])
    g.set_params(*x)
    t = time.time(); cg = g.inner_pass(); tg = time.time() - t
    t = time.time(); cg2 = g.inner_pass(); tg2 = time.time() - t
    io, po, Xo, co = o.inner_pass(*x)
    io2, po2, Xo2, co2 = o.inner_pass(io, po, Xo)
    ig, pg, Xg = g.get_params()
    print("  pass: cost", cg2, co2, "rel", abs(cg2 - co2) / co2, "intr", rel(ig, io2), "poses", rel(pg, po2), "points", rel(Xg, Xo2),
          "time %.1f ms / %.1f ms" % (tg * 1e3, tg2 * 1e3))
    for inner in (0, 1):
        g.set_params(*x)
        og = capi.options("refine"); og.inner_iterations = inner
        oo = orc.options("refine"); oo.inner_iterations = inner
        sg = g.solve(og)
        i2, p2, X2, so = o.solve(*x, oo)
        ig, pg, Xg = g.get_params()
        n = min(sg.trace_len, so.trace_len)
        tc = rel(np.array(sg.trace_cost[:n]), np.array(so.trace_cost[:n]))
        print("  solve inner=%d: it %d/%d term %d/%d inner steps %d/%d final %.12g/%.12g trace %.2e intr %.2e poses %.2e pts %.2e  %.1f ms" % (
            inner, sg.iterations, so.iterations, sg.termination, so.termination, sg.num_inner_iteration_steps,
            so.num_inner_iteration_steps, sg.final_cost, so.final_cost, tc, rel(ig, i2), rel(pg, p2), rel(Xg, X2), sg.seconds_total * 1e3))
    g.close()
