"""Per-block parameter parity table: GPU vs oracle after a whole refine() on the test shapes,
config 1 and config 3 (with and without inner iterations), and GPU / oracle vs the REFERENCE levmar run with the gauge
fixed (tests/golden/levmar_calib_fixed_points_trace.json).  Prints one JSON line per case.
Usage on the GPU box:  python scripts/parity_blocks.py [tag]   (CVX_AMD_LIB picks a development build)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cvx_amd                                   # noqa: E402
from cvx_amd import capi                         # noqa: E402
from oracle import oracle as orc                 # noqa: E402
from test_helpers import param_block_errs        # noqa: E402

CASES = {
    "config1": dict(spec=(1, 20, 9, 6, 20260001), ragged=0, fix={}),
    "config1_fixed_points": dict(spec=(1, 20, 9, 6, 20260001), ragged=0, fix={"fix_points": True}),
    "two_cameras_ragged": dict(spec=(2, 7, 12, 9, 11), ragged=4, fix={}),
    "three_cameras_free_k3_big_board": dict(spec=(3, 5, 20, 15, 12), ragged=3, fix={"fix_k3": False}),
    "fixed_intrinsics": dict(spec=(1, 6, 9, 6, 13), ragged=0, fix={"fix_intrinsics": True}),
    "all_shared_fixed": dict(spec=(2, 4, 9, 6, 14), ragged=2, fix={"fix_intrinsics": True, "fix_points": True}),
    "config3": dict(spec=(2, 500, 12, 9, 20260003), ragged=0, fix={}),
}


def flat(e):
    out = {k: float("%.3g" % v) for k, v in e.items() if k != "dist_cols"}
    out.update({k: float("%.3g" % v) for k, v in e["dist_cols"].items()})
    return out


def run(tag):
    rows = []
    for name, c in CASES.items():
        for inner in (0, 1):
            C_, VP, bw, bh, seed = c["spec"]
            d = cvx_amd.synth_calib(C_, VP, bw, bh, seed, ragged=c["ragged"])
            args = (d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], C_, bw * bh)
            g, o = cvx_amd.Calib(*args, **c["fix"]), orc.Calib(*args, **c["fix"])
            x = (d["intr_init"], d["poses_init"], d["points_init"])
            g.set_params(*x)
            og, oo = capi.options("refine"), orc.options("refine")
            og.inner_iterations = inner
            oo.inner_iterations = inner
            s = g.solve(og)
            io, po, xo, so = o.solve(*x, oo)
            e = param_block_errs(g.get_params(), (io, po, xo))
            rows.append({"case": name, "inner": inner, "vs": "oracle", "iters": [s.iterations, so.iterations],
                         "cost_rel": abs(s.final_cost - so.final_cost) / so.final_cost, **flat(e)})
            g.close()
    # the reference-output pin
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "levmar_calib_fixed_points_trace.json")))
    d = cvx_amd.synth_calib(**gold["spec"])
    args = (d["cam_of_view"], d["view_off"], d["point_id"], d["obs_x"], d["obs_y"], d["C"], d["P"])
    x = (d["intr_init"], d["poses_init"], d["points_init"])
    exp = np.array(gold["p"])
    ie = d["intr_init"].copy()
    ie[0, [0, 1, 2, 3, 4, 5, 7, 8]] = exp[:8]
    ref = (ie, exp[8:].reshape(-1, 6), d["points_init"])
    g = cvx_amd.Calib(*args, fix_points=True)
    g.set_params(*x)
    s = g.solve(capi.options("levmar"))
    got = g.get_params()
    tr = s.trace()["cost"]
    refc = 0.5 * np.array(gold["evals"])
    n = min(len(tr), len(refc) - 1)
    rows.append({"case": "levmar_pin", "vs": "reference levmar", "who": "gpu", "iters": [s.iterations, int(gold["info"][5])],
                 "trace_rel_max": float(np.max(np.abs(tr[:n] - refc[1:n + 1]) / refc[1:n + 1])),
                 "abs_dist": float(np.max(np.abs(got[0][0, [4, 5, 7, 8]] - exp[4:8]))), **flat(param_block_errs(got, ref))})
    g.close()
    o = orc.Calib(*args, fix_points=True)
    io, po, xo, so = o.solve(*x, orc.options("levmar"))
    rows.append({"case": "levmar_pin", "vs": "reference levmar", "who": "oracle", "iters": [so.iterations, int(gold["info"][5])],
                 "abs_dist": float(np.max(np.abs(io[0, [4, 5, 7, 8]] - exp[4:8]))), **flat(param_block_errs((io, po, xo), ref))})
    rows.append({"case": "levmar_pin", "vs": "oracle", "who": "gpu", **flat(param_block_errs(got, (io, po, xo)))})
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_blocks_%s.jsonl" % tag), "w") as f:
        for r in rows:
            line = json.dumps(r)
            print(line)
            f.write(line + "\n")


if __name__ == "__main__":
    run(sys.argv[1] if len(sys.argv) > 1 else "x")
