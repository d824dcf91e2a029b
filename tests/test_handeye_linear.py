"""CPU: the closed-form hand-eye initialisers behind cvx_handeye_linear (handeye.cpp:335-599; SURVEY.md 8f
row 2) against the independent numpy restatement in oracle/handeye_init.py and against ground truth.
Host-only code: no GPU needed.  Parity with the reference itself is unpinned (it holds no vectors for these
functions and Eigen is absent); tolerances are those of an SVD-based closed form in fp64."""
import numpy as np
import pytest

import cvx_amd
from oracle import handeye_init as hi


def mats(flat):
    return [np.asarray(m).reshape(4, 4).T for m in flat]      # column-major flat -> [row, col]


@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("moving", [False, True])
def test_exact_motions_recover_ground_truth(method, moving):
    G, T, xgt, x0 = cvx_amd.synth_handeye(40, 5, rot_noise=0.0, trans_noise=0.0, moving=moving)
    A, B = cvx_amd.handeye_pairs(G, T, moving=moving)
    X = cvx_amd.handeye_linear(A, B, method).reshape(4, 4).T
    Xgt = np.zeros(16)
    cvx_amd.capi.lib().cvx_quat_trans_to_affine(cvx_amd.capi.pd(xgt.copy()), cvx_amd.capi.pd(Xgt))
    Xgt = Xgt.reshape(4, 4).T
    assert np.max(np.abs(X[:3, :3] - Xgt[:3, :3])) < 1e-9
    assert np.max(np.abs(X[:3, 3] - Xgt[:3, 3])) < 1e-9
    assert np.allclose(X[3], [0, 0, 0, 1])


@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("moving,n", [(False, 30), (True, 200)])
def test_noisy_motions_match_numpy_restatement(method, moving, n):
    G, T, xgt, x0 = cvx_amd.synth_handeye(n, 20260002, moving=moving)       # BASELINE config 2 noise levels
    A, B = cvx_amd.handeye_pairs(G, T, moving=moving)
    X = cvx_amd.handeye_linear(A, B, method).reshape(4, 4).T
    Xo = hi.METHODS[method](mats(A), mats(B))
    # two SVD implementations (one-sided Jacobi vs LAPACK) of the same closed form
    assert np.max(np.abs(X[:3, :3] - Xo[:3, :3])) < 1e-9
    assert np.max(np.abs(X[:3, 3] - Xo[:3, 3])) < 1e-9 * max(1.0, np.max(np.abs(Xo[:3, 3])))
    # and the estimate is close to the truth at these noise levels (1e-3 rad, 1e-4)
    Xgt = np.zeros(16)
    cvx_amd.capi.lib().cvx_quat_trans_to_affine(cvx_amd.capi.pd(xgt.copy()), cvx_amd.capi.pd(Xgt))
    assert np.max(np.abs(X - Xgt.reshape(4, 4).T)) < 5e-2


def test_invalid_arguments():
    from cvx_amd import capi
    L = capi.lib()
    X = np.zeros(16)
    A = np.tile(np.eye(4).reshape(-1), (2, 1))
    assert L.cvx_handeye_linear(3, 2, capi.pd(A), capi.pd(A), capi.pd(X)) == capi.ERR_INVALID_ARG
    assert L.cvx_handeye_linear(1, 2, capi.pd(A), capi.pd(A), capi.pd(X)) == capi.ERR_INVALID_ARG   # Tsai: n > 2
    assert L.cvx_handeye_linear(0, 0, capi.pd(A), capi.pd(A), capi.pd(X)) == capi.ERR_INVALID_ARG
