// HIP kernels (gfx950 / CDNA4, wave64) of the hand-eye AX = XB refinement.
// Included only by handeye.hip.
//
// Functor restated per motion pair (one pair per lane):
//   PoseError2::operator(), /root/reference/modules/calib/src/handeye.cpp:228-261
//     Rx = q.toRotationMatrix() (Eigen formula, q NOT normalised), Ra, Rb from the stored quaternions
//     r[0..8]  = (Ra Rx - Rx Rb) row-major            handeye.cpp:244-249
//     r[9..11] = Ra Tx + Ta - Rx Tb - Tx              handeye.cpp:251-258
// with the Jacobian that ceres::AutoDiffCostFunction<PoseError2,12,4,3> (handeye.cpp:298-301) yields,
// and its product with QuaternionParameterization's 4x3 plus-Jacobian (handeye.cpp:305-308).
//
// Pair constants are stored as component planes: qa[4][n] | ta[3][n] | qb[4][n] | tb[3][n].
#pragma once
#include <hip/hip_runtime.h>

namespace cvxh {

constexpr int HE_ACC = 28;   // 21 (upper triangle of the 6x6 Hessian) + 6 (gradient) + 1 (sum r^2)

// Eigen QuaternionBase::toRotationMatrix, row-major, no normalisation
__device__ __forceinline__ void quat_to_R(double w, double x, double y, double z, double (&R)[9]) {
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// d Rx / d q_k for k = w, x, y, z (row-major)
__device__ __forceinline__ void quat_dR(double w, double x, double y, double z, double (&dR)[4][9]) {
    const double w2 = 2.0 * w, x2 = 2.0 * x, y2 = 2.0 * y, z2 = 2.0 * z;
    const double x4 = 4.0 * x, y4 = 4.0 * y, z4 = 4.0 * z;
    const double dw[9] = {0.0, -z2, y2, z2, 0.0, -x2, -y2, x2, 0.0};
    const double dx[9] = {0.0, y2, z2, y2, -x4, -w2, z2, w2, -x4};
    const double dy[9] = {-y4, x2, w2, x2, 0.0, z2, -w2, z2, -y4};
    const double dz[9] = {-z4, -w2, x2, w2, -z4, y2, x2, y2, 0.0};
#pragma unroll
    for (int i = 0; i < 9; ++i) { dR[0][i] = dw[i]; dR[1][i] = dx[i]; dR[2][i] = dy[i]; dR[3][i] = dz[i]; }
}

struct Pair {
    double Ra[9], Rb[9], ta[3], tb[3];
};

__device__ __forceinline__ Pair load_pair(const double *__restrict__ pairs, long long n, long long i) {
    Pair p;
    const double *c = pairs + i;
    quat_to_R(c[0], c[n], c[2 * n], c[3 * n], p.Ra);
    p.ta[0] = c[4 * n]; p.ta[1] = c[5 * n]; p.ta[2] = c[6 * n];
    quat_to_R(c[7 * n], c[8 * n], c[9 * n], c[10 * n], p.Rb);
    p.tb[0] = c[11 * n]; p.tb[1] = c[12 * n]; p.tb[2] = c[13 * n];
    return p;
}

// where a kernel finds its motion pairs: the handle's component planes, or the caller's own device arrays in the
// layouts cvx_handeye_create takes on the host (qa[n][4], ta[n][3], qb[n][4], tb[n][3])
struct PlaneSrc {
    const double *pairs;
    long long n;
    __device__ __forceinline__ Pair load(long long i) const { return load_pair(pairs, n, i); }
};
struct AosSrc {
    const double *qa, *ta, *qb, *tb;
    __device__ __forceinline__ Pair load(long long i) const {
        Pair p;
        const double *a = qa + 4 * i, *b = qb + 4 * i;
        quat_to_R(a[0], a[1], a[2], a[3], p.Ra);
        p.ta[0] = ta[3 * i]; p.ta[1] = ta[3 * i + 1]; p.ta[2] = ta[3 * i + 2];
        quat_to_R(b[0], b[1], b[2], b[3], p.Rb);
        p.tb[0] = tb[3 * i]; p.tb[1] = tb[3 * i + 1]; p.tb[2] = tb[3 * i + 2];
        return p;
    }
};

__device__ __forceinline__ void he_residual(const Pair &p, const double (&Rx)[9], const double (&t)[3],
                                            double (&r)[12]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double a = p.Ra[i * 3] * Rx[j] + p.Ra[i * 3 + 1] * Rx[3 + j] + p.Ra[i * 3 + 2] * Rx[6 + j];
            const double b = Rx[i * 3] * p.Rb[j] + Rx[i * 3 + 1] * p.Rb[3 + j] + Rx[i * 3 + 2] * p.Rb[6 + j];
            r[i * 3 + j] = a - b;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double rat = p.Ra[i * 3] * t[0] + p.Ra[i * 3 + 1] * t[1] + p.Ra[i * 3 + 2] * t[2];
        const double rxt = Rx[i * 3] * p.tb[0] + Rx[i * 3 + 1] * p.tb[1] + Rx[i * 3 + 2] * p.tb[2];
        r[9 + i] = rat + p.ta[i] - rxt - t[i];
    }
}

// derivative of the 12 residuals along one direction dR of Rx
__device__ __forceinline__ void he_dir(const Pair &p, const double (&dR)[9], double (&col)[12]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double a = p.Ra[i * 3] * dR[j] + p.Ra[i * 3 + 1] * dR[3 + j] + p.Ra[i * 3 + 2] * dR[6 + j];
            const double b = dR[i * 3] * p.Rb[j] + dR[i * 3 + 1] * p.Rb[3 + j] + dR[i * 3 + 2] * p.Rb[6 + j];
            col[i * 3 + j] = a - b;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
        col[9 + i] = -(dR[i * 3] * p.tb[0] + dR[i * 3 + 1] * p.tb[1] + dR[i * 3 + 2] * p.tb[2]);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// -----------------------------------------------------------------------------------------
// cost only: partial[block] = sum r^2 of the block's pairs
// -----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_he_cost(const double *__restrict__ pairs, long long n,
                                                 const double *__restrict__ p7, double *__restrict__ partial) {
    __shared__ double sh[4];
    double Rx[9];
    quat_to_R(p7[0], p7[1], p7[2], p7[3], Rx);
    const double t[3] = {p7[4], p7[5], p7[6]};
    double acc = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const Pair p = load_pair(pairs, n, i);
        double r[12];
        he_residual(p, Rx, t, r);
#pragma unroll
        for (int k = 0; k < 12; ++k) acc += r[k] * r[k];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// -----------------------------------------------------------------------------------------
// K5: local (6-dof) normal equations.  partial[block][28] = upper triangle of J^T J (row-major
// packed), J^T r, sum r^2, with J = [Jq * P | Jt], P the plus-Jacobian at q
//   rows (-x,-y,-z), (w,z,-y), (-z,w,x), (y,-x,w)       (QuaternionParameterization)
// -----------------------------------------------------------------------------------------
template <class Src>
__global__ void __launch_bounds__(256) k_he_normal_eq(const Src src, long long n,
                                                      const double *__restrict__ p7, double *__restrict__ partial) {
    __shared__ double sh[4][HE_ACC];
    const double qw = p7[0], qx = p7[1], qy = p7[2], qz = p7[3];
    double Rx[9], dRg[4][9], dRl[3][9];
    quat_to_R(qw, qx, qy, qz, Rx);
    quat_dR(qw, qx, qy, qz, dRg);
    const double P[4][3] = {{-qx, -qy, -qz}, {qw, qz, -qy}, {-qz, qw, qx}, {qy, -qx, qw}};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int i = 0; i < 9; ++i)
            dRl[a][i] = dRg[0][i] * P[0][a] + dRg[1][i] * P[1][a] + dRg[2][i] * P[2][a] + dRg[3][i] * P[3][a];
    const double t[3] = {p7[4], p7[5], p7[6]};
    double acc[HE_ACC];
#pragma unroll
    for (int k = 0; k < HE_ACC; ++k) acc[k] = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const Pair p = src.load(i);
        double r[12], c0[12], c1[12], c2[12];
        he_residual(p, Rx, t, r);
        he_dir(p, dRl[0], c0);
        he_dir(p, dRl[1], c1);
        he_dir(p, dRl[2], c2);
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            double J[6] = {c0[k], c1[k], c2[k], 0.0, 0.0, 0.0};
            if (k >= 9) {
                const int i3 = k - 9;
#pragma unroll
                for (int j = 0; j < 3; ++j) J[3 + j] = p.Ra[i3 * 3 + j] - (i3 == j ? 1.0 : 0.0);
            }
            int e = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) acc[e++] += J[a] * J[b];
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[21 + a] += J[a] * r[k];
            acc[27] += r[k] * r[k];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < HE_ACC; ++k) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) sh[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < HE_ACC)
        partial[(size_t)blockIdx.x * HE_ACC + threadIdx.x] =
            (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// fixed-order sum of the block partials: out[k] = sum_b partial[b][k], k < width
__global__ void __launch_bounds__(64) k_he_final(const double *__restrict__ partial, int n_blocks, int width,
                                                 double *__restrict__ out) {
    const int k = threadIdx.x;
    if (k >= width) return;
    double s = 0.0;
    for (int b = 0; b < n_blocks; ++b) s += partial[(size_t)b * width + k];
    out[k] = s;
}

// the same sum written out as cvx_handeye_normal_eq hands it to its caller: H[36] (mirrored) | g[6] | cost
__global__ void __launch_bounds__(64) k_he_final_expand(const double *__restrict__ partial, int n_blocks,
                                                        double *__restrict__ out) {
    __shared__ double sh[HE_ACC];
    const int k = threadIdx.x;
    if (k < HE_ACC) {
        double s = 0.0;
        for (int b = 0; b < n_blocks; ++b) s += partial[(size_t)b * HE_ACC + k];
        sh[k] = s;
    }
    __syncthreads();
    if (k < 36) {
        const int a = min(k / 6, k % 6), b = max(k / 6, k % 6);
        out[k] = sh[a * 6 - a * (a - 1) / 2 + (b - a)];
    } else if (k < 42) {
        out[k] = sh[21 + (k - 36)];
    } else if (k == 42) {
        out[k] = 0.5 * sh[27];
    }
}

// -----------------------------------------------------------------------------------------
// evaluation mode: residuals r[n][12] and the GLOBAL Jacobians Jq[n][12][4], Jt[n][12][3]
// (row-major per pair, as the AutoDiffCostFunction returns them)
// -----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_he_eval(const double *__restrict__ pairs, long long n,
                                                 const double *__restrict__ p7, double *__restrict__ r_out,
                                                 double *__restrict__ Jq_out, double *__restrict__ Jt_out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double qw = p7[0], qx = p7[1], qy = p7[2], qz = p7[3];
    double Rx[9], dRg[4][9];
    quat_to_R(qw, qx, qy, qz, Rx);
    quat_dR(qw, qx, qy, qz, dRg);
    const double t[3] = {p7[4], p7[5], p7[6]};
    const Pair p = load_pair(pairs, n, i);
    double r[12];
    he_residual(p, Rx, t, r);
    if (r_out)
#pragma unroll
        for (int k = 0; k < 12; ++k) r_out[i * 12 + k] = r[k];
    if (Jq_out) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            double c[12];
            he_dir(p, dRg[m], c);
#pragma unroll
            for (int k = 0; k < 12; ++k) Jq_out[i * 48 + k * 4 + m] = c[k];
        }
    }
    if (Jt_out) {
#pragma unroll
        for (int k = 0; k < 12; ++k)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                Jt_out[i * 36 + k * 3 + j] = (k >= 9) ? p.Ra[(k - 9) * 3 + j] - ((k - 9) == j ? 1.0 : 0.0) : 0.0;
    }
}

}  // namespace cvxh
