"""cvx_amd: MI355X-native drop-in for the nonlinear-least-squares inner loop of angeltop/cvx's
modules/calib (camera-calibration bundle adjustment and hand-eye AX = XB refinement).

The product is cvx_amd/libcvx_amd.so (HIP for gfx950 + host C++, C ABI in include/cvx_amd.h) and
the header-only C++ adaptor in cvx_amd/include/.  The Python modules here only bind that C ABI.
"""
from . import capi  # noqa: F401
from .problem import (Calib, HandEye, handeye_linear, handeye_pairs, pairs_to_quat_trans, partition_views, shard_calib,  # noqa: F401
                      synth_calib, synth_handeye)
