// Tool-level bundle adjustment on gfx950: the residual blocks of
// bundleAdjustment in /root/reference/modules/calib/tools/hand_eye_registration.cpp:144-238 (= camera_ba.cpp:146-240)
// -- functor OpenCVReprojectionError :48-141, parameter blocks {q[4] with QuaternionParameterization, t[3],
// board point[3]}, constant intrinsics, a fixed rigid pre-transform per view, HuberLoss(1.0) -- reduced on the
// device to the arrowhead normal equations  [H_pp  W; W^T  blockdiag(U_p)]  of the 6 + 3P tangent coordinates.
//
// Layout: observations are stored point-major (all observations of board point 0, then of point 1, ...), so a
// workgroup works on ONE point: its coordinates and the pose are wave-uniform, the observation planes and the
// view ids are read coalesced, and the 55 sums a point needs (21 H_pp, 6 g_p, 18 W_p, 6 U_p, 3 g_x, 1 cost) are
// accumulated in registers per lane, reduced with a fixed-order butterfly and summed over a point's chunks in
// order: results are bitwise reproducible from run to run.
#pragma once
#include <cfloat>

#include "common.h"

namespace cvxt {

constexpr int TBA_ACC = 64;        // 55 sums, padded to the wave width for the transposing reduction
constexpr int TBA_THREADS = 256;
constexpr int TBA_CHUNK = 2048;    // most observations of one point per workgroup (the host picks a multiple of the block)

struct TbaArgs {
    const double *ox, *oy;         // [N] point-major observation planes (cv::Point2f promoted to double)
    const int *view;               // [N] view of each observation
    const double *trans;           // [V][12] row-major 3x4 of gripper_to_base[i].matrix()
    const double *params;          // [7 + 3P]: q.w q.x q.y q.z t | points
    const double *intr;            // [9] fx fy cx cy k1 k2 k3 p1 p2
    double huber;                  // a of ceres::HuberLoss(a); <= 0: no loss
    const int *job_point, *job_start, *job_count;
    double *partial;               // normal equations: [jobs][TBA_ACC]; cost: [jobs]
};

// pose-dependent, observation-independent quantities (wave-uniform)
struct TbaPose {
    double R[9];                   // rotation of the normalised quaternion (ceres::QuaternionRotatePoint)
    double u[4];                   // q / |q|
    double T[12];                  // (I - u u^T) / |q| * PlusJacobian(q): d u / d(local), 4x3
    double t[3];
};

__device__ __forceinline__ void tba_pose(const double *p, TbaPose &o) {
    const double q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    const double sc = 1.0 / sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    const double u0 = q0 * sc, u1 = q1 * sc, u2 = q2 * sc, u3 = q3 * sc;
    o.u[0] = u0; o.u[1] = u1; o.u[2] = u2; o.u[3] = u3;
    const double t2 = u0 * u1, t3 = u0 * u2, t4 = u0 * u3, t5 = -u1 * u1, t6 = u1 * u2, t7 = u1 * u3;
    const double t8 = -u2 * u2, t9 = u2 * u3, t1 = -u3 * u3;
    o.R[0] = 2.0 * (t8 + t1) + 1.0; o.R[1] = 2.0 * (t6 - t4);       o.R[2] = 2.0 * (t3 + t7);
    o.R[3] = 2.0 * (t4 + t6);       o.R[4] = 2.0 * (t5 + t1) + 1.0; o.R[5] = 2.0 * (t9 - t2);
    o.R[6] = 2.0 * (t7 - t3);       o.R[7] = 2.0 * (t2 + t9);       o.R[8] = 2.0 * (t5 + t8) + 1.0;
    // QuaternionParameterization::ComputeJacobian rows (hand_eye_registration.cpp:196-199)
    const double Pq[12] = {-q1, -q2, -q3, q0, q3, -q2, -q3, q0, q1, q2, -q1, q0};
    for (int c = 0; c < 3; ++c) {
        const double up = u0 * Pq[c] + u1 * Pq[3 + c] + u2 * Pq[6 + c] + u3 * Pq[9 + c];
        for (int r = 0; r < 4; ++r) o.T[3 * r + c] = (Pq[3 * r + c] - o.u[r] * up) * sc;
    }
    o.t[0] = p[4]; o.t[1] = p[5]; o.t[2] = p[6];
}

// residual of one observation (hand_eye_registration.cpp:60-140); D = d r / d x_cam (2x3), pb = trans * (X, 1)
__device__ __forceinline__ void tba_project(const double *intr, const TbaPose &P, const double tr[12], const double X[3],
                                            double ox, double oy, double r[2], double D[6], double pb[3], bool jac) {
    for (int i = 0; i < 3; ++i) pb[i] = tr[4 * i] * X[0] + tr[4 * i + 1] * X[1] + tr[4 * i + 2] * X[2] + tr[4 * i + 3];
    double xc[3];
    for (int i = 0; i < 3; ++i) xc[i] = P.R[3 * i] * pb[0] + P.R[3 * i + 1] * pb[1] + P.R[3 * i + 2] * pb[2] + P.t[i];
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    const double k1 = intr[4], k2 = intr[5], k3 = intr[6], p1 = intr[7], p2 = intr[8];
    const double iz = 1.0 / xc[2];
    const double xn = xc[0] * iz, yn = xc[1] * iz;
    const double r2 = xn * xn + yn * yn, r4 = r2 * r2, r6 = r4 * r2;
    const double rc = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
    const double xd = xn * rc + 2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn);
    const double yd = yn * rc + 2.0 * p2 * xn * yn + p1 * (r2 + 2.0 * yn * yn);
    r[0] = fx * xd + cx - ox;
    r[1] = fy * yd + cy - oy;
    if (!jac) return;
    const double rcp = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4;
    const double dxx = rc + 2.0 * xn * xn * rcp + 2.0 * p1 * yn + 6.0 * p2 * xn;
    const double dxy = 2.0 * xn * yn * rcp + 2.0 * p1 * xn + 2.0 * p2 * yn;
    const double dyy = rc + 2.0 * yn * yn * rcp + 2.0 * p2 * xn + 6.0 * p1 * yn;
    const double a11 = fx * dxx, a12 = fx * dxy, a21 = fy * dxy, a22 = fy * dyy;
    D[0] = a11 * iz; D[1] = a12 * iz; D[2] = -(a11 * xn + a12 * yn) * iz;
    D[3] = a21 * iz; D[4] = a22 * iz; D[5] = -(a21 * xn + a22 * yn) * iz;
}

// ceres::HuberLoss::Evaluate at s = |r|^2: rho (value) and rho' (the squared weight ceres::Corrector applies when
// rho'' <= 0, which always holds for Huber)
__device__ __forceinline__ void tba_huber(double a, double s, double &rho0, double &rho1) {
    const double b = a * a;
    if (a > 0.0 && s > b) {
        const double r = sqrt(s);
        rho0 = 2.0 * a * r - b;
        rho1 = fmax(a / r, DBL_MIN);
    } else { rho0 = s; rho1 = 1.0; }
}

// 64 per-lane values in, lane L returns the sum over all lanes of value L (fixed butterfly order).  Template
// recursion keeps every register index a compile-time constant (a runtime `half` sends the array to scratch).
template <int HALF>
__device__ __forceinline__ void tba_reduce_step(double (&v)[TBA_ACC], int lane) {
    const bool upper = (lane & HALF) != 0;
#pragma unroll
    for (int j = 0; j < HALF; ++j) {
        const double send = upper ? v[j] : v[j + HALF];
        const double keep = upper ? v[j + HALF] : v[j];
        v[j] = keep + __shfl_xor(send, HALF, 64);
    }
    if constexpr (HALF > 1) tba_reduce_step<HALF / 2>(v, lane);
}
__device__ __forceinline__ double tba_transpose_reduce(double (&v)[TBA_ACC], int lane) {
    tba_reduce_step<32>(v, lane);
    return v[0];
}

__global__ __launch_bounds__(TBA_THREADS) void k_tba_normal_eq(TbaArgs a) {
    __shared__ double red[TBA_THREADS / 64][TBA_ACC];
    const int job = blockIdx.x, pt = a.job_point[job], start = a.job_start[job], count = a.job_count[job];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    TbaPose P;
    tba_pose(a.params, P);
    const double X[3] = {a.params[7 + 3 * pt], a.params[8 + 3 * pt], a.params[9 + 3 * pt]};
    double intr[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) intr[i] = a.intr[i];
    double acc[TBA_ACC];
#pragma unroll
    for (int i = 0; i < TBA_ACC; ++i) acc[i] = 0.0;

    for (int i = threadIdx.x; i < count; i += TBA_THREADS) {
        const int o = start + i;
        const double *trp = a.trans + (size_t)a.view[o] * 12;
        double tr[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) tr[k] = trp[k];
        double r[2], D[6], pb[3];
        tba_project(intr, P, tr, X, a.ox[o], a.oy[o], r, D, pb, true);
        double rho0, rho1;
        tba_huber(a.huber, r[0] * r[0] + r[1] * r[1], rho0, rho1);
        const double w = sqrt(rho1);
        // d x_cam / d u (3x4) for the unit quaternion, then to the 3 tangent coordinates through T
        const double u0 = P.u[0], u1 = P.u[1], u2 = P.u[2], u3 = P.u[3];
        const double G[12] = {
            2 * (-u3 * pb[1] + u2 * pb[2]), 2 * (u2 * pb[1] + u3 * pb[2]), 2 * (-2 * u2 * pb[0] + u1 * pb[1] + u0 * pb[2]), 2 * (-2 * u3 * pb[0] - u0 * pb[1] + u1 * pb[2]),
            2 * (u3 * pb[0] - u1 * pb[2]), 2 * (u2 * pb[0] - 2 * u1 * pb[1] - u0 * pb[2]), 2 * (u1 * pb[0] + u3 * pb[2]), 2 * (u0 * pb[0] - 2 * u3 * pb[1] + u2 * pb[2]),
            2 * (-u2 * pb[0] + u1 * pb[1]), 2 * (u3 * pb[0] + u0 * pb[1] - 2 * u1 * pb[2]), 2 * (-u0 * pb[0] + u3 * pb[1] - 2 * u2 * pb[2]), 2 * (u1 * pb[0] + u2 * pb[1])};
        double GT[9];       // d x_cam / d local (3x3)
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                GT[3 * rr + c] = G[4 * rr] * P.T[c] + G[4 * rr + 1] * P.T[3 + c] + G[4 * rr + 2] * P.T[6 + c] + G[4 * rr + 3] * P.T[9 + c];
        double RT[9];       // d x_cam / d X = R * trans[:, :3]
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                RT[3 * rr + c] = P.R[3 * rr] * tr[c] + P.R[3 * rr + 1] * tr[4 + c] + P.R[3 * rr + 2] * tr[8 + c];
        double J[2][9];     // [rot local 3 | t 3 | X 3], scaled by sqrt(rho') like ceres::Corrector
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                J[k][c] = w * (D[3 * k] * GT[c] + D[3 * k + 1] * GT[3 + c] + D[3 * k + 2] * GT[6 + c]);
                J[k][3 + c] = w * D[3 * k + c];
                J[k][6 + c] = w * (D[3 * k] * RT[c] + D[3 * k + 1] * RT[3 + c] + D[3 * k + 2] * RT[6 + c]);
            }
        }
        const double rw[2] = {w * r[0], w * r[1]};
        int e = 0;
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int q = p; q < 6; ++q) { acc[e] += J[0][p] * J[0][q] + J[1][p] * J[1][q]; ++e; }       // 0..20  H_pp
#pragma unroll
        for (int p = 0; p < 6; ++p) acc[21 + p] += J[0][p] * rw[0] + J[1][p] * rw[1];                   // 21..26 g_p
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[27 + 3 * p + q] += J[0][p] * J[0][6 + q] + J[1][p] * J[1][6 + q];   // 27..44 W
        e = 45;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p; q < 3; ++q) { acc[e] += J[0][6 + p] * J[0][6 + q] + J[1][6 + p] * J[1][6 + q]; ++e; }   // 45..50 U
#pragma unroll
        for (int p = 0; p < 3; ++p) acc[51 + p] += J[0][6 + p] * rw[0] + J[1][6 + p] * rw[1];           // 51..53 g_x
        acc[54] += rho0;                                                                                 // 54 cost * 2
    }
    const double mine = tba_transpose_reduce(acc, lane);
    red[wave][lane] = mine;
    __syncthreads();
    if (wave == 0) {
        double s = red[0][lane];
#pragma unroll
        for (int wv = 1; wv < TBA_THREADS / 64; ++wv) s += red[wv][lane];
        a.partial[(size_t)job * TBA_ACC + lane] = s;
    }
}

// sums a point's chunks in order: out[point][TBA_ACC]; job_first[P + 1]
__global__ __launch_bounds__(64) void k_tba_reduce(const double *partial, const int *job_first, double *out) {
    const int pt = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int j = job_first[pt]; j < job_first[pt + 1]; ++j) s += partial[(size_t)j * TBA_ACC + lane];
    out[(size_t)pt * TBA_ACC + lane] = s;
}

__device__ __forceinline__ double tba_block_sum(double v, double *red) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = red[0];
    for (int wv = 1; wv < (int)(blockDim.x >> 6); ++wv) s += red[wv];
    __syncthreads();
    return s;
}

// sum of rho(|r|^2) over one chunk
__global__ __launch_bounds__(TBA_THREADS) void k_tba_cost(TbaArgs a) {
    __shared__ double red[TBA_THREADS / 64];
    const int job = blockIdx.x, pt = a.job_point[job], start = a.job_start[job], count = a.job_count[job];
    TbaPose P;
    tba_pose(a.params, P);
    const double X[3] = {a.params[7 + 3 * pt], a.params[8 + 3 * pt], a.params[9 + 3 * pt]};
    double intr[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) intr[i] = a.intr[i];
    double c = 0.0;
    for (int i = threadIdx.x; i < count; i += TBA_THREADS) {
        const int o = start + i;
        const double *trp = a.trans + (size_t)a.view[o] * 12;
        double tr[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) tr[k] = trp[k];
        double r[2], D[6], pb[3], rho0, rho1;
        tba_project(intr, P, tr, X, a.ox[o], a.oy[o], r, D, pb, false);
        tba_huber(a.huber, r[0] * r[0] + r[1] * r[1], rho0, rho1);
        c += rho0;
    }
    const double s = tba_block_sum(c, red);
    if (threadIdx.x == 0) a.partial[job] = s;
}

// one workgroup: out[0] = sum of the chunk costs in a fixed order
// What the host reads after a linearisation (the P x 55 block sums) or a cost evaluation (one double), written into its
// host-mapped copy behind the kernels that made it, then a sequence number the host polls: a copy and a stream
// synchronisation wake the host up 10-20 us after the fact, twice per LM iteration of 95 us.
__global__ __launch_bounds__(256) void k_tba_publish(const double *src, double *dst, long long n, double *flag, double seq) {
    for (long long i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        *flag = seq;
        __threadfence_system();
    }
}

__global__ __launch_bounds__(TBA_THREADS) void k_tba_cost_final(const double *partial, int n, double *out) {
    __shared__ double red[TBA_THREADS / 64];
    double c = 0.0;
    for (int i = threadIdx.x; i < n; i += TBA_THREADS) c += partial[i];
    const double s = tba_block_sum(c, red);
    if (threadIdx.x == 0) out[0] = s;
}

}  // namespace cvxt
