"""numpy restatement of the reference's closed-form hand-eye initialisers -- TEST INFRASTRUCTURE ONLY
(imported by tests/ alone; nothing under cvx_amd/ may import it).

Follows /root/reference/modules/calib/src/handeye.cpp:
  getDualQuaternion :335-356, solveHandEyeLinearDualQuaternion :366-473, solveHandEyeLinearTsai :475-533,
  solveHandEyeLinearHD :537-599.
Eigen (un-vendored by the reference) is replaced by its published algorithms: Quaterniond(Matrix3d) is
Shoemake's branchy conversion, AngleAxisd(Quaterniond) is 2 atan2(|v|, |w|), JacobiSVD -> numpy.linalg.svd,
JacobiSVD::solve -> numpy.linalg.lstsq.  PARITY UNPINNED: the reference holds no test vectors for these
functions; the tests compare the product's C++ against this independent restatement and
against the ground truth of noise-free synthetic motions.

Matrices are 4x4 numpy arrays (row, col)."""
import numpy as np


def quat_of(R):
    tr = np.trace(R)
    q = np.zeros(4)
    if tr > 0:
        t = np.sqrt(tr + 1.0)
        q[0] = 0.5 * t
        t = 0.5 / t
        q[1] = (R[2, 1] - R[1, 2]) * t
        q[2] = (R[0, 2] - R[2, 0]) * t
        q[3] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i + 1] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[k, j] - R[j, k]) * t
        q[j + 1] = (R[j, i] + R[i, j]) * t
        q[k + 1] = (R[k, i] + R[i, k]) * t
    return q


def angle_axis_of(R):
    q = quat_of(R)
    n = np.linalg.norm(q[1:])
    if n == 0:
        return 0.0, np.array([1.0, 0.0, 0.0])
    angle = 2.0 * np.arctan2(n, abs(q[0]))
    if q[0] < 0:
        n = -n
    return angle, q[1:] / n


def crossprod(a):
    return np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])


def quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def quat_mul(a, b):
    w = a[0] * b[0] - a[1:] @ b[1:]
    v = a[0] * b[1:] + b[0] * a[1:] + np.cross(a[1:], b[1:])
    return np.concatenate([[w], v])


def dual_quaternion(T):
    theta, a = angle_axis_of(T[:3, :3])
    hc, hs = np.cos(theta / 2), np.sin(theta / 2)
    as_ = hs * a
    t = T[:3, 3]
    q = np.concatenate([[hc], as_])
    qp = np.concatenate([[-t @ as_ / 2.0], (np.cross(t, as_) + hc * t) / 2.0])
    return q, qp


def compose(R, t):
    X = np.eye(4)
    X[:3, :3] = R
    X[:3, 3] = t
    return X


def linear_dual_quaternion(A, B):
    n = len(A)
    T = np.zeros((6 * n, 8))
    for i in range(n):
        qa, qpa = dual_quaternion(A[i])
        qb, qpb = dual_quaternion(B[i])
        s1, s2 = qa[1:] - qb[1:], crossprod(qa[1:] + qb[1:])
        t1, t2 = qpa[1:] - qpb[1:], crossprod(qpa[1:] + qpb[1:])
        T[6 * i:6 * i + 3, 0] = s1
        T[6 * i:6 * i + 3, 1:4] = s2
        T[6 * i + 3:6 * i + 6, 0] = t1
        T[6 * i + 3:6 * i + 6, 1:4] = t2
        T[6 * i + 3:6 * i + 6, 4] = s1
        T[6 * i + 3:6 * i + 6, 5:8] = s2
    V = np.linalg.svd(T)[2].T
    v7, v8 = V[:, 6], V[:, 7]
    u1, v1, u2, v2 = v7[:4], v7[4:], v8[:4], v8[4:]
    a, b, c = u1 @ v1, u1 @ v2 + u2 @ v1, u2 @ v2
    det = np.sqrt(b * b - 4 * a * c)
    s1, s2 = (-b + det) / 2 / a, (-b - det) / 2 / a
    a_, b_, c_ = u1 @ u1, u1 @ u2, u2 @ u2
    val1, val2 = s1 * s1 * a_ + 2 * s1 * b_ + c_, s2 * s2 * a_ + 2 * s2 * b_ + c_
    s, val = (s1, val1) if val1 > val2 else (s2, val2)
    l2 = np.sqrt(1 / val)
    sol = s * l2 * v7 + l2 * v8
    q, qp = sol[:4], sol[4:]
    conj = np.concatenate([[q[0]], -q[1:]])
    return compose(quat_to_R(q), 2 * quat_mul(qp, conj)[1:])


def _translation(A, B, R):
    MA = np.vstack([a[:3, :3] - np.eye(3) for a in A])
    MB = np.concatenate([R @ b[:3, 3] - a[:3, 3] for a, b in zip(A, B)])
    return np.linalg.lstsq(MA, MB, rcond=None)[0]


def linear_tsai(A, B):
    M, rhs = [], []
    for a, b in zip(A, B):
        tg, ng = angle_axis_of(a[:3, :3])
        tc, nc = angle_axis_of(b[:3, :3])
        Pg, Pc = 2 * np.sin(tg / 2) * ng, 2 * np.sin(tc / 2) * nc
        M.append(crossprod(Pg + Pc))
        rhs.append(Pc - Pg)
    Pp = np.linalg.lstsq(np.vstack(M), np.concatenate(rhs), rcond=None)[0]
    P = 2 * Pp / np.sqrt(1 + Pp @ Pp)
    R = (1 - P @ P / 2) * np.eye(3) + 0.5 * (np.outer(P, P) + np.sqrt(4 - P @ P) * crossprod(P))
    return compose(R, _translation(A, B, R))


def linear_horaud(A, B):
    M = []
    for a, b in zip(A, B):
        qa, qb = quat_of(a[:3, :3]), quat_of(b[:3, :3])
        Q = np.array([[qa[0], -qa[1], -qa[2], -qa[3]], [qa[1], qa[0], -qa[3], qa[2]],
                      [qa[2], qa[3], qa[0], -qa[1]], [qa[3], -qa[2], qa[1], qa[0]]])
        W = np.array([[qb[0], -qb[1], -qb[2], -qb[3]], [qb[1], qb[0], qb[3], -qb[2]],
                      [qb[2], -qb[3], qb[0], qb[1]], [qb[3], qb[2], -qb[1], qb[0]]])
        M.append(Q - W)
    q = np.linalg.svd(np.vstack(M))[2].T[:, 3]
    R = quat_to_R(q)
    return compose(R, _translation(A, B, R))


METHODS = {0: linear_dual_quaternion, 1: linear_tsai, 2: linear_horaud}
