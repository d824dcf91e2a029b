"""Shared helpers of the tests (not a test module)."""
import importlib.util
import os

import numpy as np

_g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py")


def _load_generator():
    spec = importlib.util.spec_from_file_location("make_golden", _g)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


_gen = _load_generator()
# name -> (f, jac) of the classic problems, shared with the fixture generator
KAT_FUNCS = {k: (v[0], v[1]) for k, v in _gen.PROBLEMS.items()}


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300)) if a.size else 0.0


def calib_from_fixture(g):
    """functor_mp.json 'calib' block -> arrays of a calibration problem (8 views x 4 points, 2 cameras)."""
    V, P = len(g["poses"]), len(g["points"])
    obs = sorted(g["obs"], key=lambda o: (o["view"], o["point"]))
    view_off = np.arange(V + 1, dtype=np.int32) * P
    point_id = np.array([o["point"] for o in obs], dtype=np.int32)
    ox = np.array([o["ox"] for o in obs]); oy = np.array([o["oy"] for o in obs])
    r = np.array([[float(s) for s in o["r"]] for o in obs])
    J = np.array([[[float(s) for s in row] for row in o["J"]] for o in obs])
    return {"C": len(g["cameras"]), "V": V, "P": P, "cam_of_view": np.array(g["cam_of_view"], dtype=np.int32),
            "view_off": view_off, "point_id": point_id, "obs_x": ox, "obs_y": oy,
            "intr": np.array(g["cameras"]), "poses": np.array(g["poses"]), "points": np.array(g["points"]),
            "r": r, "Jc": J[:, :, :9], "Jp": J[:, :, 9:15], "Jx": J[:, :, 15:]}
