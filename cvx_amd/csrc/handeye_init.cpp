// Closed-form hand-eye initial estimates from motion pairs A_i X = X B_i (host only, no device work):
//   method 0  dual quaternions (Daniilidis)   /root/reference/modules/calib/src/handeye.cpp:335-473
//   method 1  Tsai                            handeye.cpp:475-533
//   method 2  Horaud and Dornaika             handeye.cpp:537-599
// The reference calls Eigen::JacobiSVD (un-vendored); the singular vectors and least-squares solutions it
// asks for are computed here with a one-sided (Hestenes) Jacobi SVD of the tall matrix, which like Eigen's
// works on the matrix itself and not on its normal equations.  Every result used downstream is invariant to
// the sign / ordering freedom of singular vectors.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

// One-sided Jacobi SVD of an m x k matrix (row-major, k <= 8): on return the columns of W = A V are mutually
// orthogonal; sv[j] = |W_j| descending, V (k x k, row-major) holds the right singular vectors as columns.
void cvx::jacobi_svd(std::vector<double> &W, int m, int k, double *V, double *sv) {
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) V[i * k + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < k - 1; ++p)
            for (int q = p + 1; q < k; ++q) {
                double app = 0, aqq = 0, apq = 0;
                for (int r = 0; r < m; ++r) {
                    const double a = W[(size_t)r * k + p], b = W[(size_t)r * k + q];
                    app += a * a; aqq += b * b; apq += a * b;
                }
                if (apq == 0.0 || std::fabs(apq) <= 1e-300) continue;
                const double rel = std::fabs(apq) / std::sqrt(std::max(app * aqq, 1e-300));
                off = std::max(off, rel);
                if (rel < 1e-16) continue;
                const double zeta = (aqq - app) / (2.0 * apq);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < m; ++r) {
                    const double a = W[(size_t)r * k + p], b = W[(size_t)r * k + q];
                    W[(size_t)r * k + p] = c * a - s * b;
                    W[(size_t)r * k + q] = s * a + c * b;
                }
                for (int r = 0; r < k; ++r) {
                    const double a = V[r * k + p], b = V[r * k + q];
                    V[r * k + p] = c * a - s * b;
                    V[r * k + q] = s * a + c * b;
                }
            }
        if (off < 1e-15) break;
    }
    int order[8];
    double nrm[8];
    for (int j = 0; j < k; ++j) {
        double s = 0;
        for (int r = 0; r < m; ++r) s += W[(size_t)r * k + j] * W[(size_t)r * k + j];
        nrm[j] = std::sqrt(s);
        order[j] = j;
    }
    std::sort(order, order + k, [&](int a, int b) { return nrm[a] > nrm[b]; });
    double Vs[64];
    std::vector<double> Ws(W.size());
    for (int j = 0; j < k; ++j) {
        sv[j] = nrm[order[j]];
        for (int r = 0; r < k; ++r) Vs[r * k + j] = V[r * k + order[j]];
        for (int r = 0; r < m; ++r) Ws[(size_t)r * k + j] = W[(size_t)r * k + order[j]];
    }
    std::memcpy(V, Vs, sizeof(double) * k * k);
    W.swap(Ws);
}

namespace {
using cvx::jacobi_svd;

// JacobiSVD(A, ComputeThinU | ComputeThinV).solve(b): x = V S^-1 U^T b (singular values below
// eps * max(m, k) * s_max are dropped, Eigen's default threshold)
void svd_solve(std::vector<double> A, int m, int k, const std::vector<double> &b, double *x) {
    double V[64], sv[8];
    jacobi_svd(A, m, k, V, sv);     // A now holds U S (columns)
    const double thresh = 2.220446049250313e-16 * std::max(m, k) * sv[0];
    double y[8];
    for (int j = 0; j < k; ++j) {
        double s = 0;
        for (int r = 0; r < m; ++r) s += A[(size_t)r * k + j] * b[r];
        y[j] = (sv[j] > thresh) ? s / (sv[j] * sv[j]) : 0.0;     // (U_j . b) / s_j with U_j = W_j / s_j
    }
    for (int i = 0; i < k; ++i) {
        double s = 0;
        for (int j = 0; j < k; ++j) s += V[i * k + j] * y[j];
        x[i] = s;
    }
}

struct Motion {
    double R[9];    // row-major rotation (linear part)
    double q[4];    // Eigen::Quaterniond(R): w x y z
    double t[3];
    double angle, axis[3];   // Eigen::AngleAxisd(R)
};

Motion motion_of(const double *T) {
    Motion m;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m.R[i * 3 + j] = T[j * 4 + i];
    cvx_affine_to_quat_trans(T, m.q, m.t);
    // Eigen AngleAxis <- Quaternion
    double n = std::sqrt(m.q[1] * m.q[1] + m.q[2] * m.q[2] + m.q[3] * m.q[3]);
    if (n != 0.0) {
        m.angle = 2.0 * std::atan2(n, std::fabs(m.q[0]));
        if (m.q[0] < 0) n = -n;
        for (int i = 0; i < 3; ++i) m.axis[i] = m.q[1 + i] / n;
    } else {
        m.angle = 0.0; m.axis[0] = 1.0; m.axis[1] = 0.0; m.axis[2] = 0.0;
    }
    return m;
}

void cross(const double a[3], const double b[3], double c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}

// rows r0..r0+2, columns c0..c0+2 of M (row stride k) = [a]x
void put_cross(std::vector<double> &M, int k, int r0, int c0, const double a[3]) {
    const double v[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[(size_t)(r0 + i) * k + c0 + j] = v[i * 3 + j];
}

// handeye.cpp:335-356: dual quaternion (q, q') of a rigid motion
void dual_quaternion(const Motion &m, double q[4], double qp[4]) {
    const double hc = std::cos(m.angle / 2.0), hs = std::sin(m.angle / 2.0);
    const double as[3] = {hs * m.axis[0], hs * m.axis[1], hs * m.axis[2]};
    q[0] = hc; q[1] = as[0]; q[2] = as[1]; q[3] = as[2];
    qp[0] = -(m.t[0] * as[0] + m.t[1] * as[1] + m.t[2] * as[2]) / 2.0;
    double c[3];
    cross(m.t, as, c);
    for (int i = 0; i < 3; ++i) qp[1 + i] = (c[i] + hc * m.t[i]) / 2.0;
}

void quat_to_rot_rowmajor(const double q[4], double R[9]) {      // Eigen toRotationMatrix, no normalisation
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    const double r[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx,
                         txz - twy, tyz + twx, 1 - (txx + tyy)};
    std::memcpy(R, r, sizeof r);
}

void compose(const double R[9], const double t[3], double X[16]) {   // Translation3d(t) * R, column-major 4x4
    std::memset(X, 0, 16 * sizeof(double));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) X[j * 4 + i] = R[i * 3 + j];
    X[12] = t[0]; X[13] = t[1]; X[14] = t[2]; X[15] = 1.0;
}

// (R_A - I) T = R B.t - A.t in the least-squares sense (handeye.cpp:518-527, 582-592)
void solve_translation(const std::vector<Motion> &A, const std::vector<Motion> &B, const double R[9], double T[3]) {
    const int n = (int)A.size();
    std::vector<double> M((size_t)3 * n * 3), rhs((size_t)3 * n);
    for (int i = 0; i < n; ++i)
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) M[(size_t)(3 * i + r) * 3 + c] = A[i].R[r * 3 + c] - (r == c ? 1.0 : 0.0);
            rhs[3 * i + r] = R[r * 3] * B[i].t[0] + R[r * 3 + 1] * B[i].t[1] + R[r * 3 + 2] * B[i].t[2] - A[i].t[r];
        }
    svd_solve(M, 3 * n, 3, rhs, T);
}

int linear_dual_quaternion(const std::vector<Motion> &A, const std::vector<Motion> &B, double X[16]) {
    const int n = (int)A.size();
    std::vector<double> T((size_t)6 * n * 8, 0.0);
    for (int i = 0; i < n; ++i) {
        double qa[4], qpa[4], qb[4], qpb[4];
        dual_quaternion(A[i], qa, qpa);
        dual_quaternion(B[i], qb, qpb);
        double s1[3], s2[3], t1[3], t2[3];
        for (int c = 0; c < 3; ++c) {
            s1[c] = qa[1 + c] - qb[1 + c]; s2[c] = qa[1 + c] + qb[1 + c];
            t1[c] = qpa[1 + c] - qpb[1 + c]; t2[c] = qpa[1 + c] + qpb[1 + c];
        }
        for (int r = 0; r < 3; ++r) {
            T[(size_t)(6 * i + r) * 8 + 0] = s1[r];
            T[(size_t)(6 * i + 3 + r) * 8 + 0] = t1[r];
            T[(size_t)(6 * i + 3 + r) * 8 + 4] = s1[r];
        }
        put_cross(T, 8, 6 * i, 1, s2);
        put_cross(T, 8, 6 * i + 3, 1, t2);
        put_cross(T, 8, 6 * i + 3, 5, s2);
    }
    double V[64], sv[8];
    jacobi_svd(T, 6 * n, 8, V, sv);
    // the two right singular vectors of the smallest singular values span the solutions (q, q'); the unit-norm and
    // orthogonality conditions q.q = 1, q.q' = 0 pick the combination mu (ns_a + ratio ns_b): a quadratic in ratio
    double ns_a[8], ns_b[8];
    for (int r = 0; r < 8; ++r) { ns_a[r] = V[r * 8 + 6]; ns_b[r] = V[r * 8 + 7]; }
    auto dot4 = [](const double *x, const double *y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3]; };
    const double *real_a = ns_a, *dual_a = ns_a + 4, *real_b = ns_b, *dual_b = ns_b + 4;
    const double qa2 = dot4(real_a, dual_a), qa1 = dot4(real_a, dual_b) + dot4(real_b, dual_a), qa0 = dot4(real_b, dual_b);
    const double disc = std::sqrt(qa1 * qa1 - 4 * qa2 * qa0);
    const double root[2] = {(-qa1 + disc) / 2 / qa2, (-qa1 - disc) / 2 / qa2};
    // |real part|^2 of (root ns_a + ns_b): the root with the larger value is the better conditioned one (handeye.cpp:441-452)
    const double naa = dot4(real_a, real_a), nab = dot4(real_a, real_b), nbb = dot4(real_b, real_b);
    const double norm2[2] = {root[0] * root[0] * naa + 2 * root[0] * nab + nbb, root[1] * root[1] * naa + 2 * root[1] * nab + nbb};
    const int pick = norm2[0] > norm2[1] ? 0 : 1;
    const double w_b = std::sqrt(1 / norm2[pick]), w_a = root[pick] * w_b;
    double sol[8];
    for (int r = 0; r < 8; ++r) sol[r] = w_a * ns_a[r] + w_b * ns_b[r];
    if (!std::isfinite(sol[0])) return CVX_ERR_NUMERIC;
    const double *q = sol, *qp = sol + 4;
    double R[9];
    quat_to_rot_rowmajor(q, R);
    // trans = 2 (qp * conj(q)).vec
    const double cw = q[0], cx = -q[1], cy = -q[2], cz = -q[3];
    const double t[3] = {2 * (qp[0] * cx + qp[1] * cw + qp[2] * cz - qp[3] * cy),
                         2 * (qp[0] * cy - qp[1] * cz + qp[2] * cw + qp[3] * cx),
                         2 * (qp[0] * cz + qp[1] * cy - qp[2] * cx + qp[3] * cw)};
    compose(R, t, X);
    return CVX_OK;
}

int linear_tsai(const std::vector<Motion> &A, const std::vector<Motion> &B, double X[16]) {
    const int n = (int)A.size();
    if (n <= 2) return CVX_ERR_INVALID_ARG;     // assert(n>2), handeye.cpp:481
    std::vector<double> M((size_t)3 * n * 3, 0.0), rhs((size_t)3 * n);
    for (int i = 0; i < n; ++i) {
        double Pg[3], Pc[3], sum[3];
        for (int c = 0; c < 3; ++c) {
            Pg[c] = 2 * std::sin(A[i].angle / 2) * A[i].axis[c];
            Pc[c] = 2 * std::sin(B[i].angle / 2) * B[i].axis[c];
            sum[c] = Pg[c] + Pc[c];
            rhs[3 * i + c] = Pc[c] - Pg[c];
        }
        put_cross(M, 3, 3 * i, 0, sum);
    }
    double Pp[3];
    svd_solve(M, 3 * n, 3, rhs, Pp);
    const double n2 = Pp[0] * Pp[0] + Pp[1] * Pp[1] + Pp[2] * Pp[2];
    double P[3];
    for (int c = 0; c < 3; ++c) P[c] = 2 * Pp[c] / std::sqrt(1 + n2);
    const double p2 = P[0] * P[0] + P[1] * P[1] + P[2] * P[2];
    const double k = std::sqrt(4 - p2);
    const double cr[9] = {0, -P[2], P[1], P[2], 0, -P[0], -P[1], P[0], 0};
    double R[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[i * 3 + j] = (1 - p2 / 2) * (i == j ? 1.0 : 0.0) + 0.5 * (P[i] * P[j] + k * cr[i * 3 + j]);
    double T[3];
    solve_translation(A, B, R, T);
    compose(R, T, X);
    return std::isfinite(R[0]) && std::isfinite(T[0]) ? CVX_OK : CVX_ERR_NUMERIC;
}

int linear_horaud(const std::vector<Motion> &A, const std::vector<Motion> &B, double X[16]) {
    const int n = (int)A.size();
    std::vector<double> M((size_t)4 * n * 4);
    for (int i = 0; i < n; ++i) {
        const double *a = A[i].q, *b = B[i].q;
        const double Q[16] = {a[0], -a[1], -a[2], -a[3], a[1], a[0], -a[3], a[2], a[2], a[3], a[0], -a[1], a[3], -a[2], a[1], a[0]};
        const double W[16] = {b[0], -b[1], -b[2], -b[3], b[1], b[0], b[3], -b[2], b[2], -b[3], b[0], b[1], b[3], b[2], -b[1], b[0]};
        for (int e = 0; e < 16; ++e) M[(size_t)(4 * i + e / 4) * 4 + e % 4] = Q[e] - W[e];
    }
    double V[64], sv[8];
    jacobi_svd(M, 4 * n, 4, V, sv);
    const double q[4] = {V[0 * 4 + 3], V[1 * 4 + 3], V[2 * 4 + 3], V[3 * 4 + 3]};
    double R[9], T[3];
    quat_to_rot_rowmajor(q, R);
    solve_translation(A, B, R, T);
    compose(R, T, X);
    return std::isfinite(T[0]) ? CVX_OK : CVX_ERR_NUMERIC;
}

}  // namespace

extern "C" int cvx_handeye_linear(int method, int n, const double *A, const double *B, double X[16]) {
    if (!A || !B || !X || n < 1 || method < 0 || method > 2) return CVX_ERR_INVALID_ARG;
    std::vector<Motion> a(n), b(n);
    for (int i = 0; i < n; ++i) { a[i] = motion_of(A + 16 * (size_t)i); b[i] = motion_of(B + 16 * (size_t)i); }
    switch (method) {
        case 1: return linear_tsai(a, b, X);
        case 2: return linear_horaud(a, b, X);
        default: return linear_dual_quaternion(a, b, X);
    }
}
