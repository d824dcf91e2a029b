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

#include "../../include/cvx_amd.h"

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

template <class M9, class V3>
__device__ __forceinline__ void he_residual(const Pair &p, const M9 &Rx, const V3 &t, double (&r)[12]) {
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
template <class M9>
__device__ __forceinline__ void he_dir(const Pair &p, const M9 &dR, double (&col)[12]) {
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

// one pair's share of the 28 sums: upper triangle of J^T J (row-major packed), J^T r, sum r^2, with
// J = [Jq * P | Jt] (P folded into the three directions dRl)
// (Rx, t and the three directions are wave-uniform: registers in the multi-workgroup kernel, LDS in k_he_lm)
template <class M9, class D27, class V3>
__device__ __forceinline__ void he_accumulate(const Pair &p, const M9 &Rx, const D27 &dRl, const V3 &t, double (&acc)[28]) {
    // row by row (residual k and its three tangent derivatives, then its 28 products): the same operations on the
    // same operands as he_residual / he_dir, with 4 values live instead of 48
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        double rk, J[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (k < 9) {
            const int i = k / 3, j = k % 3;
            {
                const double a = p.Ra[i * 3] * Rx[j] + p.Ra[i * 3 + 1] * Rx[3 + j] + p.Ra[i * 3 + 2] * Rx[6 + j];
                const double b = Rx[i * 3] * p.Rb[j] + Rx[i * 3 + 1] * p.Rb[3 + j] + Rx[i * 3 + 2] * p.Rb[6 + j];
                rk = a - b;
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const double a = p.Ra[i * 3] * dRl[d][j] + p.Ra[i * 3 + 1] * dRl[d][3 + j] + p.Ra[i * 3 + 2] * dRl[d][6 + j];
                const double b = dRl[d][i * 3] * p.Rb[j] + dRl[d][i * 3 + 1] * p.Rb[3 + j] + dRl[d][i * 3 + 2] * p.Rb[6 + j];
                J[d] = a - b;
            }
        } else {
            const int i = k - 9;
            const double rat = p.Ra[i * 3] * t[0] + p.Ra[i * 3 + 1] * t[1] + p.Ra[i * 3 + 2] * t[2];
            const double rxt = Rx[i * 3] * p.tb[0] + Rx[i * 3 + 1] * p.tb[1] + Rx[i * 3 + 2] * p.tb[2];
            rk = rat + p.ta[i] - rxt - t[i];
#pragma unroll
            for (int d = 0; d < 3; ++d)
                J[d] = -(dRl[d][i * 3] * p.tb[0] + dRl[d][i * 3 + 1] * p.tb[1] + dRl[d][i * 3 + 2] * p.tb[2]);
#pragma unroll
            for (int j = 0; j < 3; ++j) J[3 + j] = p.Ra[i * 3 + j] - (i == j ? 1.0 : 0.0);
        }
        int e = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) acc[e++] += J[a] * J[b];
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[21 + a] += J[a] * rk;
        acc[27] += rk * rk;
    }
}

// the three tangent directions of Rx at q: dRx/dq times QuaternionParameterization's plus-Jacobian
//   rows (-x,-y,-z), (w,z,-y), (-z,w,x), (y,-x,w)
__device__ __forceinline__ void he_local_directions(double qw, double qx, double qy, double qz, double (&dRl)[3][9]) {
    double dRg[4][9];
    quat_dR(qw, qx, qy, qz, dRg);
    const double P[4][3] = {{-qx, -qy, -qz}, {qw, qz, -qy}, {-qz, qw, qx}, {qy, -qx, qw}};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int i = 0; i < 9; ++i)
            dRl[a][i] = dRg[0][i] * P[0][a] + dRg[1][i] * P[1][a] + dRg[2][i] * P[2][a] + dRg[3][i] * P[3][a];
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
    double Rx[9], dRl[3][9];
    quat_to_R(qw, qx, qy, qz, Rx);
    he_local_directions(qw, qx, qy, qz, dRl);
    const double t[3] = {p7[4], p7[5], p7[6]};
    double acc[HE_ACC];
#pragma unroll
    for (int k = 0; k < HE_ACC; ++k) acc[k] = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        he_accumulate(src.load(i), Rx, dRl, t, acc);
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

// sum_b partial[b][k] in the order of the blocks, eight loads in flight per step of the chain (one load per add, each
// waiting for the one before, made this kernel 11 us at the 77 blocks of config 2's 19 548 pairs)
__device__ __forceinline__ double he_ordered_sum(const double *__restrict__ partial, int n_blocks, int width, int k) {
    double s = 0.0;
    int b = 0;
    for (; b + 8 <= n_blocks; b += 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = partial[(size_t)(b + j) * width + k];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; b < n_blocks; ++b) s += partial[(size_t)b * width + k];
    return s;
}

// fixed-order sum of the block partials: out[k] = sum_b partial[b][k], k < width
__global__ void __launch_bounds__(64) k_he_final(const double *__restrict__ partial, int n_blocks, int width,
                                                 double *__restrict__ out) {
    const int k = threadIdx.x;
    if (k >= width) return;
    out[k] = he_ordered_sum(partial, n_blocks, width, k);
}

// the same sum written out as cvx_handeye_normal_eq hands it to its caller: H[36] (mirrored) | g[6] | cost
__global__ void __launch_bounds__(64) k_he_final_expand(const double *__restrict__ partial, int n_blocks,
                                                        double *__restrict__ out) {
    __shared__ double sh[HE_ACC];
    const int k = threadIdx.x;
    if (k < HE_ACC) sh[k] = he_ordered_sum(partial, n_blocks, HE_ACC, k);
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
// The whole solve in ONE launch: solveNonLinear's ceres::Solve (handeye.cpp:310-322) for a problem that one workgroup
// can hold (at most HE_LM_MAX_PAIRS motion pairs: solveMoving's consecutive pairs of up to 513 stations) -- the trust-region loop of lm_driver.cpp's strategy 0 (same tests in the same order on the same
// quantities: Jacobi scaling, LM diagonal clamp, invalid-step handling, parameter / function / gradient tolerances,
// step quality with the non-monotonic reference, radius update, lowest-cost point handed back), restated as scalar
// code of thread 0 between the workgroup's evaluations.  No host round trip per iteration: at 199 motion pairs the
// host-driven form spent 29 us per LM iteration on four launches, two copies and two synchronisations.
//   * one pair per thread, kept in LDS for the whole solve (14 doubles each);
//   * an evaluation at a candidate makes all 28 sums (J^T J, J^T r, sum r^2), not only the cost: an accepted
//     candidate's linearisation is the next iteration's, so a successful iteration costs ONE pass over the pairs;
//   * sums: per-thread chains, wave butterflies, then the waves in ascending order -- fixed, so repeated runs are
//     bitwise identical.
// -----------------------------------------------------------------------------------------
struct HeLmOut {
    double p[7];
    double initial_cost, final_cost;
    int iterations, num_successful, num_linear_solves, termination, trace_len, status;
    double trace[4][CVX_TRACE_MAX];     // candidate cost | accepted | radius | gradient max-norm, as cvx_lm_summary
    long long seq;                      // written last (behind a system-scope fence): the host polls it
};

constexpr int HE_LM_MAX_THREADS = 512;      // two waves per SIMD: the 28 accumulators + a cached pair want ~200 registers
constexpr int HE_LM_MAX_PAIRS = HE_LM_MAX_THREADS;   // one pair per thread; larger problems take the multi-workgroup kernels
                                                     // (measured: one workgroup at 1 738 pairs 25 us per iteration, the host-driven loop 30)

struct HeLmState {
    double x[7], xc[7], xbest[7];
    double cur[HE_ACC], cand[HE_ACC];
    double s2[6];
    double cost, gmax, radius, decrease_factor;
    double ev_min, ev_cur, ev_ref, ev_cand, ev_acc_ref, ev_acc_cand;
    double best_cost, model_change;
    int ev_nonmono, max_nonmono, term, invalid_run, it, need_gcheck, best_is_current;
    int n_success, n_solves, trace_len, action;
};
enum { HE_LM_DONE = 0, HE_LM_EVAL = 1 };

// Ceres QuaternionParameterization::Plus on (q, t) (handeye.cpp:305-308)
__device__ inline void he_plus7(const double *x, const double *d, double *out) {
    const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (n > 0.0) {
        double sv, cv;
        sincos(n, &sv, &cv);                 // one argument reduction for both
        const double sn = sv / n;
        const double q[4] = {cv, sn * d[0], sn * d[1], sn * d[2]};
        out[0] = q[0] * x[0] - q[1] * x[1] - q[2] * x[2] - q[3] * x[3];
        out[1] = q[0] * x[1] + q[1] * x[0] + q[2] * x[3] - q[3] * x[2];
        out[2] = q[0] * x[2] - q[1] * x[3] + q[2] * x[0] + q[3] * x[1];
        out[3] = q[0] * x[3] + q[1] * x[2] - q[2] * x[1] + q[3] * x[0];
    } else {
        for (int i = 0; i < 4; ++i) out[i] = x[i];
    }
    for (int a = 0; a < 3; ++a) out[4 + a] = x[4 + a] + d[3 + a];
}

// max-norm of x - Plus(x, -g): Ceres' projected gradient norm
__device__ inline double he_grad_norm(const double *x, const double *g) {
    double d[6], xo[7], m = 0.0;
    for (int a = 0; a < 6; ++a) d[a] = -g[a];
    he_plus7(x, d, xo);
    for (int i = 0; i < 7; ++i) m = fmax(m, fabs(x[i] - xo[i]));
    return m;
}

// A (6x6, symmetric, row-major; destroyed) x = b in place by Cholesky; false when A is not positive definite
__device__ inline bool he_chol_solve6(double (&A)[36], double (&b)[6]) {
    // Fully unrolled (with run-time indices the arrays live in scratch memory, a round trip per access), and written for
    // a single lane's dependent chain, where every operation costs its full latency: the pivot's reciprocal square root
    // from the hardware estimate and one third-order Newton step (error below the last bit), as K2's chol6_factor takes
    // it, and the substitutions multiply by it -- no division, no sqrt() call on the chain.
    double inv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double s = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= A[j * 6 + k] * A[j * 6 + k];
        if (!(s > 0.0) || !(s < 1.7e308)) return false;
        const double y0 = __builtin_amdgcn_rsq(s);
        const double e = fma(-s * y0, y0, 1.0);
        inv[j] = fma(y0 * e, fma(e, 0.375, 0.5), y0);
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double t = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = t * inv[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double t = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) t -= A[i * 6 + k] * b[k];
        b[i] = t * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double t = b[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) t -= A[k * 6 + i] * b[k];
        b[i] = t * inv[i];
    }
    return true;
}

__device__ inline void he_lm_trace(HeLmState &S, HeLmOut *out, double cost, int acc, double damp, double grad) {
    if (S.trace_len < CVX_TRACE_MAX) {
        out->trace[0][S.trace_len] = cost; out->trace[1][S.trace_len] = acc;
        out->trace[2][S.trace_len] = damp; out->trace[3][S.trace_len] = grad;
        S.trace_len++;
    }
}

// thread 0, after the evaluation at the start point (lm_trust_region's prologue)
__device__ inline void he_lm_begin(HeLmState &S, const cvx_lm_options &o, HeLmOut *out) {
    S.cost = 0.5 * S.cur[27];
    out->initial_cost = S.cost;
    S.term = 0; S.it = 0; S.invalid_run = 0; S.need_gcheck = 0; S.n_success = 0; S.n_solves = 0; S.trace_len = 0;
    out->status = isfinite(S.cost) ? CVX_OK : CVX_ERR_NUMERIC;
    if (out->status) { S.term = -1; return; }
    int e = 0;
    for (int a = 0; a < 6; ++a) {            // diagonal entries of the packed upper triangle
        const double d = S.cur[e];
        if (o.jacobi_scaling) { const double s = 1.0 / (1.0 + sqrt(d)); S.s2[a] = s * s; } else S.s2[a] = 1.0;
        e += 6 - a;
    }
    S.gmax = he_grad_norm(S.x, S.cur + 21);
    S.radius = o.initial_radius; S.decrease_factor = 2.0;
    S.ev_min = S.ev_cur = S.ev_ref = S.ev_cand = S.cost; S.ev_acc_ref = S.ev_acc_cand = 0.0;
    S.ev_nonmono = 0; S.max_nonmono = o.nonmonotonic ? o.max_consecutive_nonmonotonic_steps : 0;
    S.best_cost = S.cost; S.best_is_current = 1;
    if (S.gmax <= o.gradient_tolerance) S.term = 1;
}

// thread 0: everything of an iteration up to the candidate (or the end of the solve)
__device__ inline void he_lm_prepare(HeLmState &S, const cvx_lm_options &o, HeLmOut *out) {
    S.action = HE_LM_DONE;
    while (!S.term) {
        const double *g = S.cur + 21;
        if (S.it >= o.max_iterations) {
            if (S.need_gcheck) {             // Ceres tests the gradient right after the accepted step
                S.gmax = he_grad_norm(S.x, g);
                if (S.gmax <= o.gradient_tolerance) { S.term = 1; break; }
            }
            S.term = 4;
            break;
        }
        S.gmax = he_grad_norm(S.x, g);
        if (S.need_gcheck && S.gmax <= o.gradient_tolerance) { S.term = 1; break; }
        S.need_gcheck = 0;
        ++S.it; ++S.n_solves;
        double A[36], D[6], delta[6];
        {
            int e = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) { A[a * 6 + b] = S.cur[e]; A[b * 6 + a] = S.cur[e]; ++e; }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double c = S.s2[a] * A[a * 6 + a];
            c = fmin(fmax(c, o.min_lm_diagonal), o.max_lm_diagonal);
            D[a] = c * (1.0 / S.radius) / S.s2[a] + 0.0;
            A[a * 6 + a] += D[a];
            delta[a] = -g[a];
        }
        bool ok = he_chol_solve6(A, delta);
#pragma unroll
        for (int a = 0; a < 6; ++a) ok = ok && isfinite(delta[a]);
        double m = 0.0;
        if (ok) {
#pragma unroll
            for (int a = 0; a < 6; ++a) m += delta[a] * (D[a] * delta[a] - g[a]);
            S.model_change = 0.5 * m;
            ok = isfinite(S.model_change) && S.model_change > 0.0;
        }
        if (!ok) {                           // HandleInvalidStep + LevenbergMarquardtStrategy::StepIsInvalid
            he_lm_trace(S, out, NAN, 0, S.radius, S.gmax);
            if (++S.invalid_run >= 5) { S.term = 6; break; }
            S.radius *= 0.5;
            continue;
        }
        S.invalid_run = 0;
        he_plus7(S.x, delta, S.xc);
        S.action = HE_LM_EVAL;
        return;
    }
}

// thread 0, with the candidate's sums in S.cand: tolerances, step quality, accept / reject
__device__ inline void he_lm_decide(HeLmState &S, const cvx_lm_options &o, HeLmOut *out) {
    const double cand = 0.5 * S.cand[27];
    double sn2 = 0.0, xn2 = 0.0;
    for (int i = 0; i < 7; ++i) { sn2 += (S.x[i] - S.xc[i]) * (S.x[i] - S.xc[i]); xn2 += S.x[i] * S.x[i]; }
    const double sn = sqrt(sn2), xn = sqrt(xn2);
    if (sn <= o.parameter_tolerance * (xn + o.parameter_tolerance)) {
        he_lm_trace(S, out, cand, 0, S.radius, S.gmax);
        S.term = 2;
        return;
    }
    if (fabs(S.cost - cand) <= o.function_tolerance * S.cost) {
        he_lm_trace(S, out, cand, 0, S.radius, S.gmax);
        S.term = 3;
        return;
    }
    const double rel = (S.ev_cur - cand) / S.model_change;
    const double hist = (S.ev_ref - cand) / (S.ev_acc_ref + S.model_change);
    const double rho = (S.max_nonmono > 0 && hist > rel) ? hist : rel;
    if (rho > o.min_relative_decrease) {
        he_lm_trace(S, out, cand, 1, S.radius, S.gmax);
        if (cand < S.best_cost) {
            S.best_cost = cand; S.best_is_current = 1;
        } else if (S.best_is_current) {
            for (int i = 0; i < 7; ++i) S.xbest[i] = S.x[i];
            S.best_is_current = 0;
        }
        for (int i = 0; i < 7; ++i) S.x[i] = S.xc[i];
        for (int k = 0; k < HE_ACC; ++k) S.cur[k] = S.cand[k];
        S.cost = cand;
        const double t = 2.0 * rho - 1.0;
        double f = 1.0 - t * t * t;
        if (f < 1.0 / 3.0) f = 1.0 / 3.0;
        S.radius = S.radius / f;
        if (S.radius > o.max_radius) S.radius = o.max_radius;
        S.decrease_factor = 2.0;
        S.ev_cur = cand; S.ev_acc_cand += S.model_change; S.ev_acc_ref += S.model_change;
        if (S.ev_cur < S.ev_min) { S.ev_min = S.ev_cur; S.ev_nonmono = 0; S.ev_cand = S.ev_cur; S.ev_acc_cand = 0.0; }
        else { ++S.ev_nonmono; if (S.ev_cur > S.ev_cand) { S.ev_cand = S.ev_cur; S.ev_acc_cand = 0.0; } }
        if (S.ev_nonmono == S.max_nonmono) { S.ev_ref = S.ev_cand; S.ev_acc_ref = S.ev_acc_cand; }
        ++S.n_success;
        S.need_gcheck = 1;
    } else {
        he_lm_trace(S, out, cand, 0, S.radius, S.gmax);
        S.radius = S.radius / S.decrease_factor;
        S.decrease_factor *= 2.0;
        if (S.radius < o.min_radius) S.term = 5;
    }
}

// all threads: the 28 sums at the point xq (LDS) into dst (LDS)
// sum over the wave by data-parallel-primitive moves (no LDS traffic); valid on lane 63.  A fixed order.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double he_dpp_take(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += he_dpp_take<0xB1, 0xF>(v);        // quad_perm [1,0,3,2]
    v += he_dpp_take<0x4E, 0xF>(v);        // quad_perm [2,3,0,1]
    v += he_dpp_take<0x141, 0xF>(v);       // row_half_mirror
    v += he_dpp_take<0x140, 0xF>(v);       // row_mirror
    v += he_dpp_take<0x142, 0xA>(v);       // row_bcast:15 into rows 1 and 3
    v += he_dpp_take<0x143, 0xC>(v);       // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ void he_lm_eval(long long n, const double (*raw)[HE_LM_MAX_THREADS], const double *xq,
                                           double *dst, double (*red)[HE_ACC]) {
    // every thread makes the wave-uniform operands itself (a hundred independent operations): through thread 0 and LDS
    // they cost a serial section, 39 stores, a barrier and 39 loads per evaluation
    double Rx[9], dRl[3][9];
    quat_to_R(xq[0], xq[1], xq[2], xq[3], Rx);
    he_local_directions(xq[0], xq[1], xq[2], xq[3], dRl);
    const double t[3] = {xq[4], xq[5], xq[6]};
    double acc[HE_ACC];
#pragma unroll
    for (int k = 0; k < HE_ACC; ++k) acc[k] = 0.0;
    if ((long long)threadIdx.x < n) {
        const int i = threadIdx.x;
        Pair p;
        quat_to_R(raw[0][i], raw[1][i], raw[2][i], raw[3][i], p.Ra);
        p.ta[0] = raw[4][i]; p.ta[1] = raw[5][i]; p.ta[2] = raw[6][i];
        quat_to_R(raw[7][i], raw[8][i], raw[9][i], raw[10][i], p.Rb);
        p.tb[0] = raw[11][i]; p.tb[1] = raw[12][i]; p.tb[2] = raw[13][i];
        he_accumulate(p, Rx, dRl, t, acc);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < HE_ACC; ++k) {
        const double s = wave_sum_dpp(acc[k]);
        if (lane == 63) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < HE_ACC) {
        double s = red[0][threadIdx.x];
        for (int w = 1; w < nw; ++w) s += red[w][threadIdx.x];
        dst[threadIdx.x] = s;
    }
    __syncthreads();
}

struct HeLmStart {
    double p[7];
    long long seq;
};

__global__ void __launch_bounds__(HE_LM_MAX_THREADS) k_he_lm(const PlaneSrc src, const HeLmStart start, const cvx_lm_options o,
                                                             HeLmOut *__restrict__ out) {
    __shared__ HeLmState S;
    __shared__ double red[HE_LM_MAX_THREADS / 64][HE_ACC];
    // one pair per thread: its 14 doubles (two quaternions, two translations) stay in LDS for the whole solve
    __shared__ double raw[14][HE_LM_MAX_THREADS];
    if ((long long)threadIdx.x < src.n)
#pragma unroll
        for (int k = 0; k < 14; ++k) raw[k][threadIdx.x] = src.pairs[(long long)k * src.n + threadIdx.x];
    if (threadIdx.x < 7) S.x[threadIdx.x] = start.p[threadIdx.x];
    __syncthreads();
    he_lm_eval(src.n, raw, S.x, S.cur, red);
    if (threadIdx.x == 0) {
        he_lm_begin(S, o, out);
        he_lm_prepare(S, o, out);
    }
    __syncthreads();
    // every wave reads S.action behind a barrier and leaves the loop together; the loop ends: each turn is an
    // iteration counted against max_iterations (invalid steps are bounded inside he_lm_prepare)
#ifdef HE_LM_PROF
    long long pf_eval = 0, pf_scalar = 0;
#endif
    while (S.action == HE_LM_EVAL) {
#ifdef HE_LM_PROF
        const long long c0 = wall_clock64();
#endif
        he_lm_eval(src.n, raw, S.xc, S.cand, red);
#ifdef HE_LM_PROF
        const long long c1 = wall_clock64();
#endif
        if (threadIdx.x == 0) {
            he_lm_decide(S, o, out);
            he_lm_prepare(S, o, out);
        }
        __syncthreads();
#ifdef HE_LM_PROF
        pf_eval += c1 - c0; pf_scalar += wall_clock64() - c1;
#endif
    }
#ifdef HE_LM_PROF
    if (threadIdx.x == 0) { out->trace[3][1000] = (double)pf_eval; out->trace[3][1001] = (double)pf_scalar; }
#endif
    if (threadIdx.x == 0) {
        if (!S.best_is_current && S.term > 0) {
            for (int i = 0; i < 7; ++i) S.x[i] = S.xbest[i];
            S.cost = S.best_cost;
        }
        for (int i = 0; i < 7; ++i) out->p[i] = S.x[i];
        out->final_cost = S.cost;
        out->iterations = S.it; out->num_successful = S.n_success; out->num_linear_solves = S.n_solves;
        out->termination = S.term > 0 ? S.term : 0; out->trace_len = S.trace_len;
        __threadfence_system();
        *(volatile long long *)&out->seq = start.seq;
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
