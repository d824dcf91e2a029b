// Host-only pieces of libcvx_amd.so: error text, the small dense solve of the reduced system,
// the conversions the C++ adaptor needs, motion-pair construction, view partitioning and the
// synthetic workloads of SURVEY.md section 8(d).  No device code here.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "common.h"

namespace cvx {

static thread_local std::string g_last_error;
void set_last_error(const std::string &s) { g_last_error = s; }

bool handle_cache_on() {
    static const bool on = [] { const char *e = std::getenv("CVX_ARENA_CACHE"); return !(e && e[0] == '0'); }();
    return on;
}
namespace {
struct StreamCache {
    std::mutex mu;
    hipStream_t s[16] = {};
} g_streams;
}  // namespace
hipStream_t stream_take(int dev) {
    if (!handle_cache_on() || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lock(g_streams.mu);
    hipStream_t s = g_streams.s[dev];
    g_streams.s[dev] = nullptr;
    return s;
}
void stream_give(int dev, hipStream_t s) {
    if (!s) return;
    if (handle_cache_on() && dev >= 0 && dev < 16) {
        std::lock_guard<std::mutex> lock(g_streams.mu);
        if (!g_streams.s[dev]) { g_streams.s[dev] = s; return; }
    }
    (void)hipStreamDestroy(s);
}
void stream_cache_release() {
    for (int dev = 0; dev < 16; ++dev) {
        hipStream_t s = nullptr;
        {
            std::lock_guard<std::mutex> lock(g_streams.mu);
            std::swap(s, g_streams.s[dev]);
        }
        if (s) { (void)hipSetDevice(dev); (void)hipStreamDestroy(s); }
    }
}

// ---------------------------------------------------------------------------------------
// SPD solve.  Left-looking blocked Cholesky on the lower triangle (row-major): every entry
// is one long dot product over contiguous row prefixes, four columns at a time.
// ---------------------------------------------------------------------------------------
static inline void dot4(const double *a, const double *b0, const double *b1, const double *b2, const double *b3,
                        int n, double out[4]) {
    typedef double v4 __attribute__((vector_size(32)));
    v4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
    int k = 0;
    for (; k + 4 <= n; k += 4) {
        v4 x;
        std::memcpy(&x, a + k, 32);
        v4 y0, y1, y2, y3;
        std::memcpy(&y0, b0 + k, 32); std::memcpy(&y1, b1 + k, 32);
        std::memcpy(&y2, b2 + k, 32); std::memcpy(&y3, b3 + k, 32);
        s0 += x * y0; s1 += x * y1; s2 += x * y2; s3 += x * y3;
    }
    double r0 = (s0[0] + s0[1]) + (s0[2] + s0[3]), r1 = (s1[0] + s1[1]) + (s1[2] + s1[3]);
    double r2 = (s2[0] + s2[1]) + (s2[2] + s2[3]), r3 = (s3[0] + s3[1]) + (s3[2] + s3[3]);
    for (; k < n; ++k) { r0 += a[k] * b0[k]; r1 += a[k] * b1[k]; r2 += a[k] * b2[k]; r3 += a[k] * b3[k]; }
    out[0] = r0; out[1] = r1; out[2] = r2; out[3] = r3;
}

bool host_cholesky_solve(int n, double *A, double *b) {
    for (int j0 = 0; j0 < n; j0 += 4) {
        const int jb = std::min(4, n - j0);
        // rows j0.. : subtract the contribution of columns < j0 from the jb columns of this block
        for (int i = j0; i < n; ++i) {
            double d[4];
            const double *ai = A + (size_t)i * n;
            const double *b0 = A + (size_t)j0 * n;
            const double *b1 = A + (size_t)std::min(j0 + 1, n - 1) * n;
            const double *b2 = A + (size_t)std::min(j0 + 2, n - 1) * n;
            const double *b3 = A + (size_t)std::min(j0 + 3, n - 1) * n;
            dot4(ai, b0, b1, b2, b3, j0, d);
            for (int q = 0; q < jb; ++q) A[(size_t)i * n + j0 + q] -= d[q];
        }
        // factor the jb x jb diagonal block and solve the panel below it
        for (int q = 0; q < jb; ++q) {
            const int j = j0 + q;
            double s = A[(size_t)j * n + j];
            for (int k = j0; k < j; ++k) s -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
            if (!(s > 0.0) || !std::isfinite(s)) return false;
            const double l = std::sqrt(s), inv = 1.0 / l;
            A[(size_t)j * n + j] = l;
            for (int i = j + 1; i < n; ++i) {
                double t = A[(size_t)i * n + j];
                for (int k = j0; k < j; ++k) t -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
                A[(size_t)i * n + j] = t * inv;
            }
        }
    }
    for (int i = 0; i < n; ++i) {
        double t = b[i];
        const double *ai = A + (size_t)i * n;
        for (int k = 0; k < i; ++k) t -= ai[k] * b[k];
        b[i] = t / ai[i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double t = b[i];
        for (int k = i + 1; k < n; ++k) t -= A[(size_t)k * n + i] * b[k];
        b[i] = t / A[(size_t)i * n + i];
    }
    return true;
}

}  // namespace cvx

extern "C" const char *cvx_last_error(void) { return cvx::g_last_error.c_str(); }

extern "C" const char *cvx_status_string(int s) {
    switch (s) {
        case CVX_OK: return "ok";
        case CVX_ERR_INVALID_ARG: return "invalid argument";
        case CVX_ERR_NO_DEVICE: return "no usable HIP device";
        case CVX_ERR_HIP: return "HIP runtime error";
        case CVX_ERR_UNSUPPORTED: return "size beyond a kernel limit of this build";
        case CVX_ERR_DUPLICATE_MARKER: return "a view lists the same board point twice";
        case CVX_ERR_EXCHANGE: return "all-reduce hook failed";
        case CVX_ERR_NUMERIC: return "non-finite cost or singular damped system";
        default: return "unknown status";
    }
}

extern "C" int cvx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------------------------------
// rotation conversions (column-major 3x3, as Eigen's &R(0,0))
// ---------------------------------------------------------------------------------------
#define RR(i, j) R[(j) * 3 + (i)]

extern "C" void cvx_rotation_to_quaternion(const double R[9], double q[4]) {
    // Eigen::Quaternion(Matrix3) / ceres::RotationMatrixToQuaternion (Shoemake)
    const double trace = RR(0, 0) + RR(1, 1) + RR(2, 2);
    if (trace > 0.0) {
        double t = std::sqrt(trace + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (RR(2, 1) - RR(1, 2)) * t;
        q[2] = (RR(0, 2) - RR(2, 0)) * t;
        q[3] = (RR(1, 0) - RR(0, 1)) * t;
    } else {
        int i = 0;
        if (RR(1, 1) > RR(0, 0)) i = 1;
        if (RR(2, 2) > RR(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double t = std::sqrt(RR(i, i) - RR(j, j) - RR(k, k) + 1.0);
        q[i + 1] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (RR(k, j) - RR(j, k)) * t;
        q[j + 1] = (RR(j, i) + RR(i, j)) * t;
        q[k + 1] = (RR(k, i) + RR(i, k)) * t;
    }
}

extern "C" void cvx_rotation_to_angle_axis(const double R[9], double w[3]) {
    // ceres::RotationMatrixToAngleAxis = RotationMatrixToQuaternion + QuaternionToAngleAxis
    double q[4];
    cvx_rotation_to_quaternion(R, q);
    const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (s2 > 0.0) {
        const double st = std::sqrt(s2), ct = q[0];
        const double two_theta = 2.0 * ((ct < 0.0) ? std::atan2(-st, -ct) : std::atan2(st, ct));
        const double k = two_theta / st;
        w[0] = q[1] * k; w[1] = q[2] * k; w[2] = q[3] * k;
    } else {
        w[0] = q[1] * 2.0; w[1] = q[2] * 2.0; w[2] = q[3] * 2.0;
    }
}

extern "C" void cvx_angle_axis_to_rotation(const double w[3], double R[9]) {
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (th2 > 2.220446049250313e-16) {
        const double th = std::sqrt(th2);
        const double wx = w[0] / th, wy = w[1] / th, wz = w[2] / th;
        const double c = std::cos(th), s = std::sin(th);
        RR(0, 0) = c + wx * wx * (1.0 - c);
        RR(1, 0) = wz * s + wx * wy * (1.0 - c);
        RR(2, 0) = -wy * s + wx * wz * (1.0 - c);
        RR(0, 1) = wx * wy * (1.0 - c) - wz * s;
        RR(1, 1) = c + wy * wy * (1.0 - c);
        RR(2, 1) = wx * s + wy * wz * (1.0 - c);
        RR(0, 2) = wy * s + wx * wz * (1.0 - c);
        RR(1, 2) = -wx * s + wy * wz * (1.0 - c);
        RR(2, 2) = c + wz * wz * (1.0 - c);
    } else {
        RR(0, 0) = 1; RR(1, 0) = w[2]; RR(2, 0) = -w[1];
        RR(0, 1) = -w[2]; RR(1, 1) = 1; RR(2, 1) = w[0];
        RR(0, 2) = w[1]; RR(1, 2) = -w[0]; RR(2, 2) = 1;
    }
}
#undef RR

// Eigen's Transform<double,3,Affine>::rotation() (what handeye.cpp:287-290 converts to a quaternion): the orthogonal
// factor of the polar decomposition of the linear part L = U S V^T, rotation = U diag(1/x, 1, 1) V^T with
// x = det(U V^T) = +-1 dividing U's FIRST column (Transform::computeRotationScaling of Eigen 3.2 / 3.3, the
// versions the reference's Ceres <= 2.1 API dates it to; Eigen 3.4 flips the last column instead -- the two agree
// whenever det L > 0, which every rigid or noisy-rigid pose has).  [external: Eigen is not in /root/reference]
// R, out: column-major 3x3.  A linear part that is orthonormal to rounding (|L^T L - I| <= 64 eps) is its own polar
// factor to rounding and is returned as it is, bit for bit; Eigen's SVD would return it perturbed in the last
// bits, by its own rounding, which no restatement reproduces.
static void polar_rotation(const double L[9], double out[9]) {
    double dev = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += L[i * 3 + k] * L[j * 3 + k];       // columns i and j
            dev = std::max(dev, std::fabs(s - (i == j ? 1.0 : 0.0)));
        }
    if (!(dev > 64 * 2.220446049250313e-16)) {
        std::memcpy(out, L, 9 * sizeof(double));
        return;
    }
    std::vector<double> W(9);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) W[i * 3 + j] = L[j * 3 + i];                // row-major copy
    double V[9], sv[3];
    cvx::jacobi_svd(W, 3, 3, V, sv);                                            // W = U S (columns), sv descending
    double U[9];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) U[i * 3 + j] = sv[j] > 0.0 ? W[i * 3 + j] / sv[j] : 0.0;
    if (!(sv[2] > 1e-300 * sv[0])) {
        // rank-deficient linear part: complete U to an orthonormal basis (Eigen's full U is one too)
        if (!(sv[1] > 1e-300 * sv[0])) { std::memcpy(out, L, 9 * sizeof(double)); return; }
        U[0 * 3 + 2] = U[1 * 3 + 0] * U[2 * 3 + 1] - U[2 * 3 + 0] * U[1 * 3 + 1];
        U[1 * 3 + 2] = U[2 * 3 + 0] * U[0 * 3 + 1] - U[0 * 3 + 0] * U[2 * 3 + 1];
        U[2 * 3 + 2] = U[0 * 3 + 0] * U[1 * 3 + 1] - U[1 * 3 + 0] * U[0 * 3 + 1];
    }
    double Q[9];                                                                // U V^T, row-major
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Q[i * 3 + j] = U[i * 3] * V[j * 3] + U[i * 3 + 1] * V[j * 3 + 1] + U[i * 3 + 2] * V[j * 3 + 2];
    const double x = Q[0] * (Q[4] * Q[8] - Q[5] * Q[7]) - Q[1] * (Q[3] * Q[8] - Q[5] * Q[6]) + Q[2] * (Q[3] * Q[7] - Q[4] * Q[6]);
    if (x < 0.0)                                                                // m.col(0) /= x
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Q[i * 3 + j] -= 2.0 * U[i * 3] * V[j * 3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) out[j * 3 + i] = Q[i * 3 + j];
}

extern "C" void cvx_affine_to_quat_trans(const double T[16], double q[4], double t[3]) {
    // Quaterniond(A.rotation()), A.translation()  (handeye.cpp:287-290)
    double L[9], R[9];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) L[j * 3 + i] = T[j * 4 + i];
    polar_rotation(L, R);
    cvx_rotation_to_quaternion(R, q);
    t[0] = T[12]; t[1] = T[13]; t[2] = T[14];
}

extern "C" void cvx_quat_trans_to_affine(const double p[7], double T[16]) {
    const double w = p[0], x = p[1], y = p[2], z = p[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx,
                         txz - twy, tyz + twx, 1 - (txx + tyy)};   // row-major
    std::memset(T, 0, 16 * sizeof(double));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[j * 4 + i] = R[i * 3 + j];
    T[12] = p[4]; T[13] = p[5]; T[14] = p[6]; T[15] = 1.0;
}

// ---------------------------------------------------------------------------------------
// 4x4 column-major affine helpers
// ---------------------------------------------------------------------------------------
namespace {
struct Aff {
    double m[16];
};
Aff aff_mul(const Aff &a, const Aff &b) {
    Aff r;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += a.m[k * 4 + i] * b.m[j * 4 + k];
            r.m[j * 4 + i] = s;
        }
    return r;
}
Aff aff_inv(const Aff &a) {
    // Eigen::Affine3d::inverse(): general inverse of the linear part
    const double *m = a.m;
    const double a00 = m[0], a10 = m[1], a20 = m[2], a01 = m[4], a11 = m[5], a21 = m[6], a02 = m[8], a12 = m[9],
                 a22 = m[10];
    const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
    const double det = a00 * c00 + a01 * c01 + a02 * c02, id = 1.0 / det;
    double inv[9];   // row-major inverse
    inv[0] = c00 * id; inv[1] = (a02 * a21 - a01 * a22) * id; inv[2] = (a01 * a12 - a02 * a11) * id;
    inv[3] = c01 * id; inv[4] = (a00 * a22 - a02 * a20) * id; inv[5] = (a02 * a10 - a00 * a12) * id;
    inv[6] = c02 * id; inv[7] = (a01 * a20 - a00 * a21) * id; inv[8] = (a00 * a11 - a01 * a10) * id;
    Aff r;
    std::memset(r.m, 0, sizeof r.m);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[j * 4 + i] = inv[i * 3 + j];
    for (int i = 0; i < 3; ++i) r.m[12 + i] = -(inv[i * 3] * m[12] + inv[i * 3 + 1] * m[13] + inv[i * 3 + 2] * m[14]);
    r.m[15] = 1.0;
    return r;
}
double angular_distance(const double qa[4], const double qb[4]) {
    // Eigen::QuaternionBase::angularDistance: d = a * conj(b); 2 atan2(|d.vec|, |d.w|)
    const double w = qa[0] * qb[0] + qa[1] * qb[1] + qa[2] * qb[2] + qa[3] * qb[3];
    const double x = -qa[0] * qb[1] + qa[1] * qb[0] - qa[2] * qb[3] + qa[3] * qb[2];
    const double y = -qa[0] * qb[2] + qa[1] * qb[3] + qa[2] * qb[0] - qa[3] * qb[1];
    const double z = -qa[0] * qb[3] - qa[1] * qb[2] + qa[2] * qb[1] + qa[3] * qb[0];
    return 2.0 * std::atan2(std::sqrt(x * x + y * y + z * z), std::fabs(w));
}
void quat_of(const double *T, double q[4]) {
    double t[3];
    cvx_affine_to_quat_trans(T, q, t);
}
}  // namespace

extern "C" int cvx_handeye_pairs_fixed(int n, const double *G, const double *Tt, double *A, double *B, int cap,
                                       int *n_pairs) {
    if (n < 0 || !G || !Tt || !n_pairs) return CVX_ERR_INVALID_ARG;
    std::vector<double> q((size_t)4 * n);
    for (int i = 0; i < n; ++i) quat_of(G + 16 * (size_t)i, &q[4 * (size_t)i]);
    int cnt = 0;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            if (angular_distance(&q[4 * (size_t)i], &q[4 * (size_t)j]) > M_PI * 10 / 180.0) {
                if (A && B && cnt < cap) {
                    Aff gi, gj, ti, tj;
                    std::memcpy(gi.m, G + 16 * (size_t)i, sizeof gi.m); std::memcpy(gj.m, G + 16 * (size_t)j, sizeof gj.m);
                    std::memcpy(ti.m, Tt + 16 * (size_t)i, sizeof ti.m); std::memcpy(tj.m, Tt + 16 * (size_t)j, sizeof tj.m);
                    const Aff a = aff_mul(gj, aff_inv(gi)), b = aff_mul(tj, aff_inv(ti));
                    std::memcpy(A + 16 * (size_t)cnt, a.m, sizeof a.m);
                    std::memcpy(B + 16 * (size_t)cnt, b.m, sizeof b.m);
                }
                ++cnt;
            }
        }
    *n_pairs = cnt;
    return (A && B && cnt > cap) ? CVX_ERR_INVALID_ARG : CVX_OK;
}

extern "C" int cvx_handeye_pairs_moving(int n, const double *G, const double *Tt, double *A, double *B, int cap,
                                        int *n_pairs) {
    if (n < 0 || !G || !Tt || !n_pairs) return CVX_ERR_INVALID_ARG;
    // sortStationMovements(target_to_sensor_, gripper_to_base_) -- note the swapped arguments at
    // handeye.cpp:740: the greedy max-angle ordering runs on the TARGET rotations.
    std::vector<double> q((size_t)4 * n);
    for (int i = 0; i < n; ++i) quat_of(Tt + 16 * (size_t)i, &q[4 * (size_t)i]);
    std::vector<int> order;
    if (n > 0) {
        std::vector<char> checked(n, 0);
        int i = 0;
        for (;;) {
            checked[i] = 1;
            order.push_back(i);
            double best = 0.0;
            int bestj = -1;
            for (int j = 0; j < n; ++j)
                if (!checked[j]) {
                    const double a = angular_distance(&q[4 * (size_t)i], &q[4 * (size_t)j]);
                    if (a > best) { best = a; bestj = j; }
                }
            if (bestj < 0) break;
            i = bestj;
        }
    }
    const int m = (int)order.size();
    const int cnt = m > 0 ? m - 1 : 0;
    *n_pairs = cnt;
    if (!A || !B) return CVX_OK;
    if (cnt > cap) return CVX_ERR_INVALID_ARG;
    for (int k = 0; k + 1 < m; ++k) {
        Aff g0, g1, t0, t1;
        std::memcpy(g0.m, G + 16 * (size_t)order[k], sizeof g0.m); std::memcpy(g1.m, G + 16 * (size_t)order[k + 1], sizeof g1.m);
        std::memcpy(t0.m, Tt + 16 * (size_t)order[k], sizeof t0.m); std::memcpy(t1.m, Tt + 16 * (size_t)order[k + 1], sizeof t1.m);
        const Aff a = aff_mul(aff_inv(g1), g0), b = aff_mul(t1, aff_inv(t0));
        std::memcpy(A + 16 * (size_t)k, a.m, sizeof a.m);
        std::memcpy(B + 16 * (size_t)k, b.m, sizeof b.m);
    }
    return CVX_OK;
}

// ---------------------------------------------------------------------------------------
// Ordering of the inner iterations: what CoordinateDescentMinimizer::CreateOrdering makes of the Hessian graph of
// refine's problem when Solver::Options::inner_iteration_ordering is left empty (calibration.cpp:420-428).
// Vertices: the free parameter blocks that occur in a residual block; edges: blocks that share one.  The graph is
// tripartite (camera -- pose -- point -- camera) and is never built: a vertex's neighbours are enumerated from the
// observation lists (a pose's own observations, a point's observations by view, a camera's views and the points it
// sees), so a round costs O(observations).
// ---------------------------------------------------------------------------------------
extern "C" int cvx_calib_inner_ordering(int C, int V, int P, const int *cam_of_view, const int *view_off,
                                        const int *point_id, int fix_intrinsics, int fix_points, int *group, int *n_sets) {
    if (C < 1 || V < 0 || P < 1 || (V > 0 && !cam_of_view) || !view_off || !group || !n_sets) return CVX_ERR_INVALID_ARG;
    const int NB = C + P + V;
    if (view_off[0] != 0) return CVX_ERR_INVALID_ARG;
    for (int v = 0; v < V; ++v)
        if (view_off[v + 1] < view_off[v]) return CVX_ERR_INVALID_ARG;
    const int64_t N = view_off[V];
    if (N > 0 && !point_id) return CVX_ERR_INVALID_ARG;
    // One pass over the observations: how often each point is seen, which camera sees which point, the views of a
    // camera, and the rank of every block = the order in which refine's loop first names it (intrinsics, pose, point
    // per marker).  The lists of a point's views are made only if a point ever has to name its poses (below).
    std::vector<int64_t> pt_off((size_t)P + 1, 0);
    std::vector<char> sees((size_t)C * P, 0);
    std::vector<std::vector<int>> cam_views(C);
    std::vector<int> rank(NB, -1);
    std::vector<char> present(NB, 0);
    int nr = 0, left = 0;
    for (int v = 0; v < V; ++v) {
        const int c = cam_of_view[v];
        if (c < 0 || c >= C) return CVX_ERR_INVALID_ARG;
        if (view_off[v + 1] == view_off[v]) continue;
        cam_views[c].push_back(v);
        if (!fix_intrinsics && rank[c] < 0) { rank[c] = nr++; present[c] = 1; ++left; }
        rank[C + P + v] = nr++; present[C + P + v] = 1; ++left;
        char *sc = sees.data() + (size_t)c * P;
        for (int o = view_off[v]; o < view_off[v + 1]; ++o) {
            const int p = point_id[o];
            if (p < 0 || p >= P) return CVX_ERR_INVALID_ARG;
            pt_off[p + 1]++;
            sc[p] = 1;
            if (!fix_points && rank[C + p] < 0) { rank[C + p] = nr++; present[C + p] = 1; ++left; }
        }
    }
    for (int p = 0; p < P; ++p) pt_off[p + 1] += pt_off[p];
    std::vector<int> pt_view;
    auto need_pt_view = [&]() {
        if (!pt_view.empty() || N == 0) return;
        pt_view.resize((size_t)N);
        std::vector<int64_t> pos(pt_off.begin(), pt_off.end() - 1);
        for (int v = 0; v < V; ++v)
            for (int o = view_off[v]; o < view_off[v + 1]; ++o) pt_view[pos[point_id[o]]++] = v;
    };
    // f(neighbour) for the neighbours of kind k (0 cameras, 1 points, 2 poses) of vertex id
    auto for_neighbours = [&](int id, int k, auto &&f) {
        if (id < C) {
            if (k == 2) for (int v : cam_views[id]) f(C + P + v);
            if (k == 1) for (int p = 0; p < P; ++p) if (sees[(size_t)id * P + p]) f(C + p);
        } else if (id < C + P) {
            const int p = id - C;
            if (k == 2) { need_pt_view(); for (int64_t j = pt_off[p]; j < pt_off[p + 1]; ++j) f(C + P + pt_view[j]); }
            if (k == 0) for (int c = 0; c < C; ++c) if (sees[(size_t)c * P + p]) f(c);
        } else {
            const int v = id - C - P;
            if (k == 0) f(cam_of_view[v]);
            if (k == 1) for (int o = view_off[v]; o < view_off[v + 1]; ++o) f(C + point_id[o]);
        }
    };
    std::vector<int> round(NB, -1);
    struct Vert { int deg, rank, id; };
    std::vector<Vert> q;
    std::vector<char> color(NB);
    int rounds = 0;
    // A vertex's degree is the number of its neighbours still there.  Whole kinds of blocks come and go together in
    // the usual shapes, so the count over one kind of neighbour is taken from the lists only while that kind is partly
    // gone: all there -> the full count, made once; none there -> zero.
    int there[3] = {0, 0, 0}, at_start[3];          // cameras, points, poses
    auto kind = [&](int id) { return id < C ? 0 : id < C + P ? 1 : 2; };
    for (int i = 0; i < NB; ++i) there[kind(i)] += present[i];
    for (int k = 0; k < 3; ++k) at_start[k] = there[k];
    std::vector<int> full_a(NB, 0), full_b(NB, 0);  // camera: views, points; point: poses, cameras; pose: camera, points
    for (int c = 0; c < C; ++c) {
        full_a[c] = (int)cam_views[c].size();
        for (int p = 0; p < P; ++p) full_b[c] += sees[(size_t)c * P + p] && present[C + p];
    }
    for (int p = 0; p < P; ++p) {
        full_a[C + p] = (int)(pt_off[p + 1] - pt_off[p]);
        for (int c = 0; c < C; ++c) full_b[C + p] += sees[(size_t)c * P + p] && present[c];
    }
    for (int v = 0; v < V; ++v) {
        full_a[C + P + v] = present[cam_of_view[v]];
        full_b[C + P + v] = fix_points ? 0 : view_off[v + 1] - view_off[v];
    }
    auto count_kind = [&](int id, int k, int full) {        // neighbours of `id` of kind k that are still there
        if (there[k] == 0) return 0;
        if (there[k] == at_start[k]) return full;
        int d = 0;
        for_neighbours(id, k, [&](int n) { d += present[n]; });
        return d;
    };
    while (left > 0) {
        q.clear();
        for (int i = 0; i < NB; ++i) {
            if (!present[i]) continue;
            int d;
            if (i < C) d = count_kind(i, 2, full_a[i]) + count_kind(i, 1, full_b[i]);
            else if (i < C + P) d = count_kind(i, 2, full_a[i]) + count_kind(i, 0, full_b[i]);
            else d = count_kind(i, 0, full_a[i]) + count_kind(i, 1, full_b[i]);
            q.push_back({d, rank[i], i});
            color[i] = 0;
        }
        std::sort(q.begin(), q.end(), [](const Vert &a, const Vert &b) { return a.deg != b.deg ? a.deg < b.deg : a.rank < b.rank; });
        // greedy pass; white[k] = vertices of kind k that can still be taken or greyed: once a kind has none left
        // (after the first few poses of a dense problem every point and camera is grey) its lists are not walked
        int white[3] = {there[0], there[1], there[2]};
        for (const Vert &x : q) {
            if (color[x.id]) continue;
            color[x.id] = 2; round[x.id] = rounds; --white[kind(x.id)];
            for (int k = 0; k < 3; ++k)
                if (k != kind(x.id) && white[k] > 0)
                    for_neighbours(x.id, k, [&](int n) { if (present[n] && !color[n]) { color[n] = 1; --white[k]; } });
        }
        for (const Vert &x : q) if (color[x.id] == 2) { present[x.id] = 0; --left; --there[kind(x.id)]; }
        ++rounds;
    }
    for (int i = 0; i < NB; ++i) group[i] = round[i] < 0 ? -1 : rounds - 1 - round[i];      // CreateOrdering reverses
    *n_sets = rounds;
    return CVX_OK;
}

extern "C" int cvx_partition_views(int n_views, const int *view_off, int nranks, int *first_view) {
    if (n_views < 0 || !view_off || nranks < 1 || !first_view) return CVX_ERR_INVALID_ARG;
    const long long total = view_off[n_views];
    first_view[0] = 0;
    int v = 0;
    for (int r = 1; r < nranks; ++r) {
        const long long target = total * r / nranks;
        while (v < n_views && view_off[v] < target) ++v;
        // keep ranges non-decreasing and leave at least the remaining ranks room when possible
        first_view[r] = std::max(first_view[r - 1], std::min(v, n_views));
    }
    first_view[nranks] = n_views;
    return CVX_OK;
}

// ---------------------------------------------------------------------------------------
// synthetic workloads, SURVEY.md section 8(d)
// ---------------------------------------------------------------------------------------
namespace {
struct Rng {
    std::mt19937_64 g;   // the engine cvx's RNG wraps (modules/util/include/cvx/util/math/rng.hpp:90)
    explicit Rng(uint64_t seed) : g(seed) {}
    double uni() { return (double)(g() >> 11) * (1.0 / 9007199254740992.0); }   // [0,1)
    double uni(double a, double b) { return a + (b - a) * uni(); }
    double normal() {
        double u1 = 1.0 - uni(), u2 = uni();
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
    }
};

void rodrigues_rowmajor(const double w[3], double R[9]) {
    double Rc[9];
    cvx_angle_axis_to_rotation(w, Rc);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i * 3 + j] = Rc[j * 3 + i];
}

void project(const double in[9], const double pose[6], const double X[3], double &u, double &v) {
    double R[9];
    rodrigues_rowmajor(pose, R);
    const double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + pose[3];
    const double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + pose[4];
    const double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + pose[5];
    const double xn = x / z, yn = y / z;
    const double r2 = xn * xn + yn * yn, r4 = r2 * r2, r6 = r4 * r2;
    const double rc = 1 + in[4] * r2 + in[5] * r4 + in[6] * r6;
    const double xd = xn * rc + 2 * in[7] * xn * yn + in[8] * (r2 + 2 * xn * xn);
    const double yd = yn * rc + 2 * in[8] * xn * yn + in[7] * (r2 + 2 * yn * yn);
    u = in[0] * xd + in[2];
    v = in[1] * yd + in[3];
}
}  // namespace

extern "C" int64_t cvx_synth_calib_count(const cvx_synth_calib_spec *s) {
    if (!s) return -1;
    if (!s->ragged) return (int64_t)s->n_cameras * s->views_per_camera * s->board_w * s->board_h;
    // ragged: replay the visibility draws
    const int P = s->board_w * s->board_h, V = s->n_cameras * s->views_per_camera;
    Rng vis(s->seed ^ 0x9E3779B97F4A7C15ULL);
    int64_t n = 0;
    for (int v = 0; v < V; ++v) {
        int kept = 0;
        for (int p = 0; p < P; ++p) {
            const bool drop = vis.uni() * s->ragged < 1.0;
            if (!drop || P - p <= 8 - kept) { ++kept; }
        }
        n += kept;
    }
    return n;
}

extern "C" int cvx_synth_calib(const cvx_synth_calib_spec *s, int *cam_of_view, int *view_off, int *point_id,
                               double *obs_x, double *obs_y, double *intr_gt, double *poses_gt, double *points_gt,
                               double *intr_init, double *poses_init, double *points_init) {
    if (!s || !cam_of_view || !view_off || !point_id || !obs_x || !obs_y) return CVX_ERR_INVALID_ARG;
    const int C = s->n_cameras, VP = s->views_per_camera, P = s->board_w * s->board_h, V = C * VP;
    if (C < 1 || VP < 1 || P < 1) return CVX_ERR_INVALID_ARG;
    Rng rng(s->seed);
    Rng vis(s->seed ^ 0x9E3779B97F4A7C15ULL);
    std::vector<double> pts((size_t)3 * P), in((size_t)9 * C), poses((size_t)6 * V);
    // OCVCalibrationPattern layout (pattern.hpp:53-55), float32 storage
    for (int i = 0, k = 0; i < s->board_h; ++i)
        for (int j = 0; j < s->board_w; ++j, ++k) {
            pts[3 * k] = (double)(float)(j * s->tile);
            pts[3 * k + 1] = (double)(float)(i * s->tile);
            pts[3 * k + 2] = 0.0;
        }
    const double bcx = 0.5 * (s->board_w - 1) * s->tile, bcy = 0.5 * (s->board_h - 1) * s->tile;
    for (int c = 0; c < C; ++c) {
        double *q = &in[9 * (size_t)c];
        q[0] = 1000.0 + 10.0 * c; q[1] = 1005.0 + 10.0 * c; q[2] = 640.0; q[3] = 480.0;
        q[4] = -0.12; q[5] = 0.05; q[6] = 0.0; q[7] = 1e-3; q[8] = -1.5e-3;
    }
    int64_t o = 0;
    for (int v = 0; v < V; ++v) {
        const int c = v / VP;
        cam_of_view[v] = c;
        view_off[v] = (int)o;
        double ax[3] = {rng.normal(), rng.normal(), rng.normal()};
        const double nrm = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]) + 1e-300;
        const double ang = rng.uni(0.0, 0.6);
        double *ps = &poses[6 * (size_t)v];
        for (int k = 0; k < 3; ++k) ps[k] = ax[k] / nrm * ang;
        ps[3] = rng.uni(-150.0, 150.0) - bcx;
        ps[4] = rng.uni(-100.0, 100.0) - bcy;
        ps[5] = rng.uni(500.0, 900.0);
        int kept = 0;
        for (int p = 0; p < P; ++p) {
            const double n0 = rng.normal(), n1 = rng.normal();   // drawn for every marker: dense and ragged share poses
            if (s->ragged) {
                const bool drop = vis.uni() * s->ragged < 1.0;
                if (drop && !(P - p <= 8 - kept)) continue;
            }
            ++kept;
            double u, w;
            project(&in[9 * (size_t)c], ps, &pts[3 * (size_t)p], u, w);
            obs_x[o] = (double)(float)(u + s->pixel_noise * n0);
            obs_y[o] = (double)(float)(w + s->pixel_noise * n1);
            point_id[o] = p;
            ++o;
        }
    }
    view_off[V] = (int)o;
    if (intr_gt) std::memcpy(intr_gt, in.data(), sizeof(double) * in.size());
    if (poses_gt) std::memcpy(poses_gt, poses.data(), sizeof(double) * poses.size());
    if (points_gt) std::memcpy(points_gt, pts.data(), sizeof(double) * pts.size());
    Rng pr(s->seed + 0x5DEECE66DULL);
    if (intr_init)
        for (size_t i = 0; i < in.size(); ++i) intr_init[i] = in[i] * (1.0 + 1e-2 * pr.normal());
    if (poses_init)
        for (int v = 0; v < V; ++v)
            for (int k = 0; k < 6; ++k)
                poses_init[6 * (size_t)v + k] = poses[6 * (size_t)v + k] + (k < 3 ? 1e-2 : 1.0) * pr.normal();
    if (points_init) std::memcpy(points_init, pts.data(), sizeof(double) * pts.size());
    return CVX_OK;
}

extern "C" int cvx_synth_handeye(int n, uint64_t seed, double rot_noise, double trans_noise, int moving,
                                 double *G, double *Tt, double x_gt[7], double x_init[7]) {
    if (n < 2 || !G || !Tt) return CVX_ERR_INVALID_ARG;
    Rng rng(seed);
    auto make = [](const double w[3], const double t[3]) {
        Aff a;
        std::memset(a.m, 0, sizeof a.m);
        double R[9];
        cvx_angle_axis_to_rotation(w, R);
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) a.m[j * 4 + i] = R[j * 3 + i];
        a.m[12] = t[0]; a.m[13] = t[1]; a.m[14] = t[2]; a.m[15] = 1.0;
        return a;
    };
    auto rand_axis = [&](double ang, double w[3]) {
        double ax[3] = {rng.normal(), rng.normal(), rng.normal()};
        const double nrm = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]) + 1e-300;
        for (int k = 0; k < 3; ++k) w[k] = ax[k] / nrm * ang;
    };
    const double s14 = std::sqrt(14.0);
    const double wx[3] = {0.4 * 1 / s14, 0.4 * 2 / s14, 0.4 * 3 / s14}, tx[3] = {0.05, -0.03, 0.12};
    const Aff X = make(wx, tx), Xi = aff_inv(X);
    const double wy[3] = {0.2, -0.1, 0.3}, ty[3] = {0.1, 0.2, 1.0};
    const Aff Y = make(wy, ty);
    for (int i = 0; i < n; ++i) {
        double w[3], t[3] = {rng.uni(-0.5, 0.5), rng.uni(-0.5, 0.5), rng.uni(-0.5, 0.5)};
        rand_axis(rng.uni(0.0, M_PI / 2), w);
        const Aff g = make(w, t);
        // fixed camera:  T_i = X^-1 G_i Y  => (T_j T_i^-1) = X^-1 (G_j G_i^-1) X
        // moving camera: T_i = X^-1 G_i^-1 Y => (T_{i+1} T_i^-1) = X^-1 (G_{i+1}^-1 G_i) X
        Aff tt = aff_mul(aff_mul(Xi, moving ? aff_inv(g) : g), Y);
        double wn[3], tn[3] = {trans_noise * rng.normal(), trans_noise * rng.normal(), trans_noise * rng.normal()};
        rand_axis(rot_noise * rng.normal(), wn);
        tt = aff_mul(tt, make(wn, tn));
        std::memcpy(G + 16 * (size_t)i, g.m, sizeof g.m);
        std::memcpy(Tt + 16 * (size_t)i, tt.m, sizeof tt.m);
    }
    if (x_gt) {
        double q[4], t[3];
        cvx_affine_to_quat_trans(X.m, q, t);
        for (int k = 0; k < 4; ++k) x_gt[k] = q[k];
        for (int k = 0; k < 3; ++k) x_gt[4 + k] = t[k];
    }
    if (x_init) {
        double wn[3], tn[3] = {0.01 * rng.normal(), 0.01 * rng.normal(), 0.01 * rng.normal()};
        rand_axis(0.05, wn);
        const Aff X0 = aff_mul(X, make(wn, tn));
        double q[4], t[3];
        cvx_affine_to_quat_trans(X0.m, q, t);
        for (int k = 0; k < 4; ++k) x_init[k] = q[k];
        for (int k = 0; k < 3; ++k) x_init[4 + k] = t[k];
    }
    return CVX_OK;
}
