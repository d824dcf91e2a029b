// HIP kernels (gfx950 / CDNA4, wave64) of the camera-calibration LM iteration.
// Included only by calib.hip.
//
// Functor restated per observation (one observation per lane):
//   OpenCVReprojectionError + ApplyRadialDistortionCameraIntrinsics,
//   /root/reference/modules/calib/src/calibration.cpp:231-311
// with the analytic Jacobian that ceres::AutoDiffCostFunction<...,2,9,6,3> (calibration.cpp:391-398)
// evaluates through Jets.  Internal column order of a residual row everywhere in this file:
//   [0..5]  pose block   d/d(omega, t)
//   [6..14] intrinsics   fx fy cx cy k1 k2 k3 p1 p2   (calibration.cpp:219-229)
//   [15]    the residual itself
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "trust_region.h"

namespace cvxk {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// 16-byte accesses with the non-temporal hint, for buffers that are written once and read once (or read for the last
// time): they pass the L2 and the memory-side cache by instead of pushing out what the next kernels come back for
typedef double cvx_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nt_store2(double *p, double a, double b) { __builtin_nontemporal_store(cvx_d2v{a, b}, (cvx_d2v *)p); }
__device__ __forceinline__ double2 nt_load2(const double *p) {
    const cvx_d2v v = __builtin_nontemporal_load((const cvx_d2v *)p);
    return double2{v.x, v.y};
}

constexpr int RT_STRIDE = 40;    // per view: R[9] | dR/dw_j [3][9] | t[3] | pad
constexpr int PT_ACC = 36;       // per point accumulators: Ucx[9][3] | Uxx[6] | gx[3]
constexpr int AROW = 17;         // LDS stride (doubles) of one staged Jacobian row: odd, so that the 64 lanes' row writes spread over all banks

// Accumulations of K2 (point blocks, 6x6 factor, substitutions) use fused multiply-adds.  -DCVX_NO_CONTRACT is the
// development build of the rounding-source experiment (scripts/build_nocontract.sh): the same
// sums as separate multiply and add, which is how a build with -ffp-contract=off rounds them.
#ifdef CVX_NO_CONTRACT
#define CVX_ACC_CONTRACT _Pragma("clang fp contract(off)")
__device__ __forceinline__ double acc_fma(double x, double y, double z) { return x * y + z; }
#else
#define CVX_ACC_CONTRACT _Pragma("clang fp contract(fast)")
__device__ __forceinline__ double acc_fma(double x, double y, double z) { return fma(x, y, z); }
#endif

// -----------------------------------------------------------------------------------------
// K0: per-view rotation matrix and its derivative w.r.t. the global angle-axis vector
// (both branches of ceres::AngleAxisRotatePoint, used at calibration.cpp:286)
// -----------------------------------------------------------------------------------------
// o[40] = R[9] | dR/dw_j [3][9] | t[3] | 0 for the pose (angle-axis, t)
__device__ __forceinline__ void pose_table(const double *__restrict__ pose, double *__restrict__ o) {
    const double w0 = pose[0], w1 = pose[1], w2 = pose[2];
    const double th2 = w0 * w0 + w1 * w1 + w2 * w2;
    double R[9], dR[3][9];
    if (th2 > 2.220446049250313e-16) {
        const double th = sqrt(th2), c = cos(th), s = sin(th), ti = 1.0 / th, omc = 1.0 - c;
        const double a[3] = {w0 * ti, w1 * ti, w2 * ti};
        // [a]x
        const double K[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
        for (int i = 0; i < 3; ++i)
            for (int k = 0; k < 3; ++k)
                R[i * 3 + k] = (i == k ? c : 0.0) + s * K[i * 3 + k] + omc * a[i] * a[k];
        for (int j = 0; j < 3; ++j) {
            // da/dw_j = (e_j - a a_j) / th
            double da[3];
            for (int i = 0; i < 3; ++i) da[i] = ((i == j ? 1.0 : 0.0) - a[i] * a[j]) * ti;
            const double dK[9] = {0, -da[2], da[1], da[2], 0, -da[0], -da[1], da[0], 0};
            for (int i = 0; i < 3; ++i)
                for (int k = 0; k < 3; ++k)
                    dR[j][i * 3 + k] = (i == k ? -s * a[j] : 0.0) + c * a[j] * K[i * 3 + k] + s * dK[i * 3 + k] +
                                       s * a[j] * a[i] * a[k] + omc * (da[i] * a[k] + a[i] * da[k]);
        }
    } else {
        const double Rs[9] = {1, -w2, w1, w2, 1, -w0, -w1, w0, 1};
        for (int i = 0; i < 9; ++i) R[i] = Rs[i];
        const double E0[9] = {0, 0, 0, 0, 0, -1, 0, 1, 0};
        const double E1[9] = {0, 0, 1, 0, 0, 0, -1, 0, 0};
        const double E2[9] = {0, -1, 0, 1, 0, 0, 0, 0, 0};
        for (int i = 0; i < 9; ++i) { dR[0][i] = E0[i]; dR[1][i] = E1[i]; dR[2][i] = E2[i]; }
    }
    for (int i = 0; i < 9; ++i) o[i] = R[i];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 9; ++i) o[9 + 9 * j + i] = dR[j][i];
    o[36] = pose[3];
    o[37] = pose[4];
    o[38] = pose[5];
    o[39] = 0.0;
}

__global__ void k_pose_prepare(const double *__restrict__ poses, double *__restrict__ rt, int V) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    pose_table(poses + 6 * v, rt + (size_t)v * RT_STRIDE);
}

struct Intr {
    double fx, fy, cx, cy, k1, k2, k3, p1, p2;
};

__device__ __forceinline__ Intr load_intr(const double *__restrict__ p) {
    Intr in;
    in.fx = p[0]; in.fy = p[1]; in.cx = p[2]; in.cy = p[3];
    in.k1 = p[4]; in.k2 = p[5]; in.k3 = p[6]; in.p1 = p[7]; in.p2 = p[8];
    return in;
}

// Per-view tables and other data that a whole wave reads at one address, written by an EARLIER kernel: read through
// the constant address space, the compiler then fetches them with scalar loads into SGPRs (one request per wave
// instead of 64 identical lane requests, no VGPRs, and not ordered behind the vector loads in flight).  Plain
// pointers get vector loads here, because the kernels also store through pointers the compiler cannot tell apart.
typedef const __attribute__((address_space(4))) double *uniform_f64;
typedef const __attribute__((address_space(4))) int *uniform_i32;
__device__ __forceinline__ uniform_f64 as_uniform(const double *p) { return (uniform_f64)p; }
__device__ __forceinline__ uniform_i32 as_uniform(const int *p) { return (uniform_i32)p; }

// -----------------------------------------------------------------------------------------
// Speculated steps.  The trust-region loop's decision on a step (accept / reject, next radius, stopping tests) is
// scalar arithmetic on 16 doubles; waiting for the host to make it leaves the GPU idle for a launch round trip per LM
// iteration (10-15 us of a 400 us step at config 4).  So the step's last kernel makes the decision too (the same
// code as the host's: trust_region.h), in a state record in device memory, and the kernels of the NEXT step -- already
// enqueued -- take the point to linearise at and the damping from that record.  The host remains the authority: it
// repeats every decision from the step's 16 doubles, and whenever the loop leaves its common path (a stopping test,
// an uphill step that wants the best point saved, inner iterations, the iteration limit) the record says "inactive"
// and the kernels already enqueued return at once.
// -----------------------------------------------------------------------------------------
struct LmDev {
    double inv_radius, mu, min_diag, max_diag;   // damping of the step in flight
    int cur;                                     // which of the two parameter buffers holds the current point
    int active;                                  // 1: the kernels that read this record run; 0: they return at once
    int last_cand, pad;                          // buffer of the candidate the last step made (-1: it did not run)
    double min_lm_diagonal, max_lm_diagonal;
    cvx::TrState tr;                             // the loop's state BEFORE the decision on the step in flight
    cvx::TrOptions opt;
};
typedef const __attribute__((address_space(4))) LmDev *uniform_lm;
__device__ __forceinline__ uniform_lm as_uniform(const LmDev *p) { return (uniform_lm)p; }

// The 16 doubles R a linear step leaves (k_backsubst_cost_step / k_step_finish / k_final_scalars) as the loop reads them
__host__ __device__ inline cvx::TrStep step_from_record(const double *R, double gtol, double *cost) {
    cvx::TrStep s;
    s.grad_converged = 0; s.ok = 0; s.cand_cost = 0.0; s.model_change = 0.0; s.step_norm2 = 0.0; s.x_norm2 = 0.0;
    *cost = 0.5 * R[11];
    s.grad_max = R[9] > R[13] ? R[9] : R[13];
    if (s.grad_max <= gtol) { s.grad_converged = 1; return s; }
    if (R[12] > 0 || R[8] > 0) return s;
    s.cand_cost = 0.5 * R[0];
    s.model_change = 0.5 * (R[1] + R[4]);
    s.step_norm2 = R[2] + R[5];
    s.x_norm2 = R[3] + R[6];
    // finite: |x| < inf (false for NaN), the same test on host and device
    s.ok = (fabs(s.model_change) < INFINITY) && (fabs(s.step_norm2) < INFINITY);
    return s;
}

// Rotation tables of the candidate a speculated step has just made (poses / rt: buffer 0's).  It runs BEHIND the
// step's decision, which may have renamed the buffers, so the decision leaves the candidate's buffer in `last_cand`
// (-1: the step did not run).
__global__ void k_pose_prepare_cand(const double *__restrict__ poses, double *__restrict__ rt, int V, long long poses_step,
                                    long long rt_step, const LmDev *__restrict__ st) {
    const int which = as_uniform(st)->last_cand;
    if (which < 0) return;
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    pose_table(poses + which * poses_step + 6 * (size_t)v, rt + which * rt_step + (size_t)v * RT_STRIDE);
}

// residual only (calibration.cpp:266-311)
template <class RT>
__device__ __forceinline__ void reproj_residual(const Intr &in, const RT rt, double X0, double X1,
                                                double X2, double ox, double oy, double &r0, double &r1) {
    const double xc = rt[0] * X0 + rt[1] * X1 + rt[2] * X2 + rt[36];
    const double yc = rt[3] * X0 + rt[4] * X1 + rt[5] * X2 + rt[37];
    const double zc = rt[6] * X0 + rt[7] * X1 + rt[8] * X2 + rt[38];
    const double iz = 1.0 / zc;
    const double xn = xc * iz, yn = yc * iz;
    const double r2 = xn * xn + yn * yn, r4 = r2 * r2, r6 = r4 * r2;
    const double rc = 1.0 + in.k1 * r2 + in.k2 * r4 + in.k3 * r6;
    const double xd = xn * rc + 2.0 * in.p1 * xn * yn + in.p2 * (r2 + 2.0 * xn * xn);
    const double yd = yn * rc + 2.0 * in.p2 * xn * yn + in.p1 * (r2 + 2.0 * yn * yn);
    r0 = in.fx * xd + in.cx - ox;
    r1 = in.fy * yd + in.cy - oy;
}

// residual + Jacobian blocks, row-major Jp[2][6], Jc[2][9], Jx[2][3]
template <class RT>
__device__ __forceinline__ void reproj_jacobian(const Intr &in, const RT rt, double X0, double X1,
                                                double X2, double ox, double oy, double &r0, double &r1,
                                                double (&Jp)[12], double (&Jc)[18], double (&Jx)[6]) {
    // (no contraction in here: the functor is evaluated in plain IEEE operations, as the reference's own build does)
    const double xc = rt[0] * X0 + rt[1] * X1 + rt[2] * X2 + rt[36];
    const double yc = rt[3] * X0 + rt[4] * X1 + rt[5] * X2 + rt[37];
    const double zc = rt[6] * X0 + rt[7] * X1 + rt[8] * X2 + rt[38];
    const double iz = 1.0 / zc;
    const double xn = xc * iz, yn = yc * iz;
    const double r2 = xn * xn + yn * yn, r4 = r2 * r2, r6 = r4 * r2;
    const double rc = 1.0 + in.k1 * r2 + in.k2 * r4 + in.k3 * r6;
    const double xy2 = 2.0 * xn * yn;
    const double tx = r2 + 2.0 * xn * xn, ty = r2 + 2.0 * yn * yn;
    const double xd = xn * rc + in.p1 * xy2 + in.p2 * tx;
    const double yd = yn * rc + in.p2 * xy2 + in.p1 * ty;
    r0 = in.fx * xd + in.cx - ox;
    r1 = in.fy * yd + in.cy - oy;
    // intrinsics block
    Jc[0] = xd;  Jc[9] = 0.0;
    Jc[1] = 0.0; Jc[10] = yd;
    Jc[2] = 1.0; Jc[11] = 0.0;
    Jc[3] = 0.0; Jc[12] = 1.0;
    const double fxx = in.fx * xn, fyy = in.fy * yn;
    Jc[4] = fxx * r2; Jc[13] = fyy * r2;
    Jc[5] = fxx * r4; Jc[14] = fyy * r4;
    Jc[6] = fxx * r6; Jc[15] = fyy * r6;
    Jc[7] = in.fx * xy2; Jc[16] = in.fy * ty;
    Jc[8] = in.fx * tx;  Jc[17] = in.fy * xy2;
    // d(xd,yd)/d(xn,yn)
    const double rcp = in.k1 + 2.0 * in.k2 * r2 + 3.0 * in.k3 * r4;
    const double dxx = rc + 2.0 * xn * xn * rcp + 2.0 * in.p1 * yn + 6.0 * in.p2 * xn;
    const double dxy = xy2 * rcp + 2.0 * in.p1 * xn + 2.0 * in.p2 * yn;
    const double dyy = rc + 2.0 * yn * yn * rcp + 2.0 * in.p2 * xn + 6.0 * in.p1 * yn;
    const double a11 = in.fx * dxx, a12 = in.fx * dxy, a21 = in.fy * dxy, a22 = in.fy * dyy;
    // D = d(u,v)/d(x_cam), 2x3
    const double D00 = a11 * iz, D01 = a12 * iz, D02 = -(a11 * xn + a12 * yn) * iz;
    const double D10 = a21 * iz, D11 = a22 * iz, D12 = -(a21 * xn + a22 * yn) * iz;
    // G[i][j] = sum_k dR_j[i][k] X_k
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const RT d = rt + 9 + 9 * j;
        const double g0 = d[0] * X0 + d[1] * X1 + d[2] * X2;
        const double g1 = d[3] * X0 + d[4] * X1 + d[5] * X2;
        const double g2 = d[6] * X0 + d[7] * X1 + d[8] * X2;
        Jp[j] = D00 * g0 + D01 * g1 + D02 * g2;
        Jp[6 + j] = D10 * g0 + D11 * g1 + D12 * g2;
    }
    Jp[3] = D00; Jp[4] = D01; Jp[5] = D02;
    Jp[9] = D10; Jp[10] = D11; Jp[11] = D12;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        Jx[j] = D00 * rt[j] + D01 * rt[3 + j] + D02 * rt[6 + j];
        Jx[3 + j] = D10 * rt[j] + D11 * rt[3 + j] + D12 * rt[6 + j];
    }
}

// -----------------------------------------------------------------------------------------
// K1: evaluation mode -- residuals and Jacobian blocks materialised as component planes
// out layout (doubles): r[2][N] | Jc[18][N] | Jp[12][N] | Jx[6][N]
// -----------------------------------------------------------------------------------------
// One observation per lane (38 outputs = 76 VGPRs: twice the occupancy two observations per lane allow), and
// still nothing but 16-byte stores: lanes 2i and 2i+1 trade halves with a DPP swap -- the even lane writes the
// pair's planes 0..18, the odd lane planes 19..37 (two 512-byte runs per wave instruction).  That needs N even,
// so that every plane starts 16-byte aligned and every observation has its partner; an odd N takes 8-byte stores.
// A workgroup's 256 observations belong to a handful of consecutive views (3 at 108 per view): their rotation
// tables and intrinsics are staged in LDS once (49 doubles per view), so a lane issues ~25 broadcast
// ds_read_b128 instead of 48 global loads and the memory pipe is left to the stores.  Workgroups that span more
// than EVAL_VIEWS views (many near-empty views) read the tables from global memory.
constexpr int EVAL_THREADS = 256;
constexpr int EVAL_VIEWS = 24;
constexpr int EVAL_PITCH = 50;      // 40 table + 9 intrinsics + 1 pad: rows stay 16-byte aligned

__device__ __forceinline__ double dpp_swap_neighbour(double x) {     // lane i <-> lane i ^ 1 (quad_perm [1,0,3,2])
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void eval_one(long long o, long long N, int pid, double ox, double oy,
                                         const double *__restrict__ points, double *__restrict__ out,
                                         const double *t, const double *ip) {
    double val[38];     // plane order: r[2] | Jc[18] | Jp[12] | Jx[6]
    {
        double Jp[12], Jc[18], Jx[6];
        const Intr in = load_intr(ip);
        const double *X = points + 3 * pid;
        reproj_jacobian(in, t, X[0], X[1], X[2], ox, oy, val[0], val[1], Jp, Jc, Jx);
#pragma unroll
        for (int i = 0; i < 18; ++i) val[2 + i] = Jc[i];
#pragma unroll
        for (int i = 0; i < 12; ++i) val[20 + i] = Jp[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) val[32 + i] = Jx[i];
    }
    if (!(N & 1)) {
        const bool odd = o & 1;
        double2 *p = (double2 *)(out + (o & ~1LL) + (odd ? 19 * N : 0));
        const long long h = N >> 1;
#pragma unroll
        for (int s = 0; s < 19; ++s) {
            const double keep = odd ? val[19 + s] : val[s];
            const double give = odd ? val[s] : val[19 + s];
            const double got = dpp_swap_neighbour(give);
            typedef double v2d __attribute__((ext_vector_type(2)));
            const v2d pr = {odd ? got : keep, odd ? keep : got};
            __builtin_nontemporal_store(pr, (v2d *)&p[(long long)s * h]);
        }
    } else {
        double *p = out + o;
#pragma unroll
        for (int i = 0; i < 38; ++i) p[(long long)i * N] = val[i];
    }
}

__global__ void __launch_bounds__(EVAL_THREADS) k_eval(const int *__restrict__ view_id, const int *__restrict__ cam_of_view,
                                                        const int *__restrict__ point_id, const double *__restrict__ obs_x,
                                                        const double *__restrict__ obs_y, const double *__restrict__ intr,
                                                        const double *__restrict__ points, const double *__restrict__ rt,
                                                        long long N, double *__restrict__ out) {
    __shared__ __align__(16) double tab[EVAL_VIEWS * EVAL_PITCH];
    const long long o0 = (long long)blockIdx.x * EVAL_THREADS;      // < N by the grid size
    const long long o1 = (o0 + EVAL_THREADS < N ? o0 + EVAL_THREADS : N) - 1;
    const int v0 = view_id[o0], nv = view_id[o1] - v0 + 1;         // view ids ascend with the observation index
    const bool staged = nv <= EVAL_VIEWS;
    if (staged) {
        for (int e = threadIdx.x; e < nv * EVAL_PITCH; e += EVAL_THREADS) {
            const int vv = e / EVAL_PITCH, k = e - vv * EVAL_PITCH;
            double x = 0.0;
            if (k < RT_STRIDE) x = rt[(size_t)(v0 + vv) * RT_STRIDE + k];
            else if (k < RT_STRIDE + 9) x = intr[9 * cam_of_view[v0 + vv] + (k - RT_STRIDE)];
            tab[e] = x;
        }
    }
    __syncthreads();
    const long long o = o0 + threadIdx.x;
    if (o >= N) return;         // N even: a lane and its partner leave together
    const int v = view_id[o], pid = point_id[o];
    const double ox = obs_x[o], oy = obs_y[o];
    if (staged) {
        const double *t = tab + (v - v0) * EVAL_PITCH;
        eval_one(o, N, pid, ox, oy, points, out, t, t + RT_STRIDE);
    } else {
        eval_one(o, N, pid, ox, oy, points, out, rt + (size_t)v * RT_STRIDE, intr + 9 * cam_of_view[v]);
    }
}

// -----------------------------------------------------------------------------------------
// K4: cost only.  One partial (sum r^2) per block, summed later in fixed order.
// -----------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_256(double v, double *sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    return t;
}

__global__ void __launch_bounds__(256) k_cost(const int *__restrict__ view_id, const int *__restrict__ cam_of_view,
                                              const int *__restrict__ point_id, const double *__restrict__ obs_x,
                                              const double *__restrict__ obs_y, const double *__restrict__ intr,
                                              const double *__restrict__ points, const double *__restrict__ rt,
                                              long long N, double *__restrict__ partial) {
    __shared__ double sh[4];
    double acc = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < N; o += stride) {
        const int v = view_id[o], pid = point_id[o];
        const Intr in = load_intr(intr + 9 * cam_of_view[v]);
        const double *X = points + 3 * pid;
        double r0, r1;
        reproj_residual(in, rt + (size_t)v * RT_STRIDE, X[0], X[1], X[2], obs_x[o], obs_y[o], r0, r1);
        acc += r0 * r0 + r1 * r1;
    }
    const double t = block_sum_256(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// -----------------------------------------------------------------------------------------
// 6x6 SPD block, packed lower triangle idx(i,j)=i(i+1)/2+j: Cholesky factor in place.  On return
// the off-diagonal slots hold L and the diagonal slots 1/L_ii (all the substitutions need).
// -----------------------------------------------------------------------------------------
#define TRI(i, j) ((i) * ((i) + 1) / 2 + (j))

__device__ __forceinline__ bool chol6_factor(double (&A)[21]) {
CVX_ACC_CONTRACT
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double s = A[TRI(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= A[TRI(j, k)] * A[TRI(j, k)];
        ok = ok && (s > 0.0) && (s < 1.7e308);
        // 1 / sqrt(s): hardware estimate and one third-order Newton step (error ~ e^3, below the last bit); the
        // library rsqrt() is six times this dependent chain, and the six pivots are sequential
        const double y0 = __builtin_amdgcn_rsq(s);
        const double e = fma(-s * y0, y0, 1.0);
        const double ri = ok ? fma(y0 * e, fma(e, 0.375, 0.5), y0) : 0.0;
        A[TRI(j, j)] = ri;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double t = A[TRI(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= A[TRI(i, k)] * A[TRI(j, k)];
            A[TRI(i, j)] = t * ri;
        }
    }
    return ok;
}

// z = L^-1 w with the packed factor of chol6_factor
__device__ __forceinline__ void chol6_forward(const double (&Lf)[21], double (&w)[6]) {
CVX_ACC_CONTRACT
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double t = w[i];
#pragma unroll
        for (int k = 0; k < i; ++k) t -= Lf[TRI(i, k)] * w[k];
        w[i] = t * Lf[TRI(i, i)];
    }
}

// -----------------------------------------------------------------------------------------
// K2: fused residual + Jacobian + normal-equation blocks, one workgroup per group of views
// (all of one camera), one observation per lane, views in sequence.
//
// Per view:   G = A^T A for A = [Jp | Jc | r] (rows: 2 per observation) on fp64 MFMA,
//             giving V_v (6x6), W_vc (6x9), g_v, and this view's share of U_cc, g_c, cost;
//             D_v from diag(V_v); L_v = chol(V_v + D_v) (kept per view for the back-substitution);
//             Z_v = L_v^-1 [W_vx | W_vc | g_v] written to HBM (row-major, ZC columns);
// per point:  U_cx, U_xx, g_x accumulated in LDS across the group's views.
// MODE 0: as above.  MODE 1 (initial linearisation): no damping/factor/Z; writes the Jacobi
//         scales s_j^2 = 1/(1+sqrt(d_j))^2 of the pose parameters when s2pose_out != null.
// -----------------------------------------------------------------------------------------
struct K2Args {
    const int *view_off, *cam_of_view, *point_id;
    const double *obs_x, *obs_y, *intr, *points, *rt, *s2pose;
    const int *grp_v0;
    int P, ZC, camcol0, ycol, cam_free, pts_free;
    double inv_radius, mu, min_diag, max_diag;
    double *Z, *Lfac, *Dv, *gv, *s2pose_out, *slab_pt, *slab_cam, *grp_scal;
    // speculated steps (st != null): intr / points / rt are buffer 0's, buffer 1 lies `*_step` doubles behind; the
    // current buffer and the damping come from the record
    const LmDev *st;
    long long intr_step, points_step, rt_step;
};

#ifndef K2_MIN_WAVES
#define K2_MIN_WAVES 2
#endif
template <int T, int MODE, bool ALIGNED>
__global__ void __launch_bounds__(T, K2_MIN_WAVES) k_normal_eq(K2Args a) {
    extern __shared__ double lds[];
    constexpr int W = T / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.st) {
        const uniform_lm st = as_uniform(a.st);
        if (!st->active) return;
        const long long cur = st->cur;
        a.intr += cur * a.intr_step; a.points += cur * a.points_step; a.rt += cur * a.rt_step;
        a.inv_radius = st->inv_radius; a.mu = st->mu; a.min_diag = st->min_diag; a.max_diag = st->max_diag;
    }
#ifdef K2_WALL
    const long long w_start = wall_clock64();
    long long w_pro = 0, w_loop = 0;
#endif
    // ALIGNED: every view of the group lists the same points in the same order, so thread tid meets the
    // same point in every view and its U_cx | U_xx | g_x accumulators stay in registers; otherwise they
    // live in LDS, indexed by point id
    double *accpt = lds;                                            // [P * PT_ACC] (not ALIGNED)
    double *arows = accpt + (ALIGNED ? 0 : a.P * PT_ACC);           // [W][64 * AROW]
    double *tiles = arows + W * 64 * AROW;                          // [W][256]
    double *gsum = tiles + W * 256;                                 // [256]
    if (!ALIGNED)
        for (int e = tid; e < a.P * PT_ACC; e += T) accpt[e] = 0.0;
    double pacc[PT_ACC];
#pragma unroll
    for (int i = 0; i < PT_ACC; ++i) pacc[i] = 0.0;

    const int g = blockIdx.x;
    const int v0 = a.grp_v0[g], v1 = a.grp_v0[g + 1];
    const int cam = a.cam_of_view[v0];
    const Intr in = load_intr(a.intr + 9 * cam);
    f64x4 acc_tot = {0.0, 0.0, 0.0, 0.0};
    double gmax = 0.0, dmax = 0.0, nbad = 0.0;
    double *myrows = arows + wave * 64 * AROW;
    int pid = 0;
    double X0 = 0.0, X1 = 0.0, X2 = 0.0;
    if (ALIGNED && v0 < v1 && tid < a.view_off[v0 + 1] - a.view_off[v0]) {
        pid = a.point_id[a.view_off[v0] + tid];
        X0 = a.points[3 * pid]; X1 = a.points[3 * pid + 1]; X2 = a.points[3 * pid + 2];
    }
    __syncthreads();

#ifdef K2_PROF
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp;
#define K2_STAMP(i) do { const long long tn = clock64(); ph[i] += tn - tp; tp = tn; } while (0)
#else
#define K2_STAMP(i) do {} while (0)
#endif
#ifdef K2_WALL
    w_pro = wall_clock64();
#endif
    int rt_touch = 0;
    for (int v = v0; v < v1; ++v) {
#ifdef K2_PROF
        tp = clock64();
#endif
        const int base = as_uniform(a.view_off)[v];
        const int n = as_uniform(a.view_off)[v + 1] - base;
        const uniform_f64 rt = as_uniform(a.rt + (size_t)v * RT_STRIDE);
        {
            // the NEXT view's table touched now (one scalar load per 64-byte line of its 320 bytes), so that its loads
            // hit the scalar cache a view later: the functor needs the table at once, and the table was written by
            // another kernel, i.e. comes from the Infinity Cache or further (K2 88.7 -> 86.9 us in bench.py)
            const uniform_i32 nx = (uniform_i32)(a.rt + (size_t)(v + 1 < v1 ? v + 1 : v) * RT_STRIDE);
            rt_touch ^= nx[0] ^ nx[16] ^ nx[32] ^ nx[48] ^ nx[64] ^ nx[79];
        }
        const bool active = tid < n;
        double r[2] = {0.0, 0.0}, Jp[12], Jc[18], Jx[6];
        if (active) {
            if (!ALIGNED) {
                pid = a.point_id[base + tid];
                X0 = a.points[3 * pid]; X1 = a.points[3 * pid + 1]; X2 = a.points[3 * pid + 2];
            }
            reproj_jacobian(in, rt, X0, X1, X2, a.obs_x[base + tid], a.obs_y[base + tid], r[0], r[1], Jp, Jc, Jx);
        } else {
#pragma unroll
            for (int i = 0; i < 12; ++i) Jp[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 18; ++i) Jc[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) Jx[i] = 0.0;
        }
        K2_STAMP(0);
        // ---- per-point accumulators (unique point ids inside a view: plain read-modify-write) ----
#ifndef K2_NO_PACC
        if (active && a.pts_free) {
            // pa[i] += Jc[c] Jx[b] + Jc[9 + c] Jx[3 + b] etc., as two fused multiply-adds per accumulator (x row, then y row)
            double *pa = ALIGNED ? pacc : accpt + pid * PT_ACC;
            if (a.cam_free) {
#pragma unroll
                for (int c = 0; c < 9; ++c)
#pragma unroll
                    for (int b = 0; b < 3; ++b) pa[c * 3 + b] = acc_fma(Jc[9 + c], Jx[3 + b], acc_fma(Jc[c], Jx[b], pa[c * 3 + b]));
            }
            pa[27] = acc_fma(Jx[3], Jx[3], acc_fma(Jx[0], Jx[0], pa[27]));
            pa[28] = acc_fma(Jx[3], Jx[4], acc_fma(Jx[0], Jx[1], pa[28]));
            pa[29] = acc_fma(Jx[3], Jx[5], acc_fma(Jx[0], Jx[2], pa[29]));
            pa[30] = acc_fma(Jx[4], Jx[4], acc_fma(Jx[1], Jx[1], pa[30]));
            pa[31] = acc_fma(Jx[4], Jx[5], acc_fma(Jx[1], Jx[2], pa[31]));
            pa[32] = acc_fma(Jx[5], Jx[5], acc_fma(Jx[2], Jx[2], pa[32]));
            pa[33] = acc_fma(Jx[3], r[1], acc_fma(Jx[0], r[0], pa[33]));
            pa[34] = acc_fma(Jx[4], r[1], acc_fma(Jx[1], r[0], pa[34]));
            pa[35] = acc_fma(Jx[5], r[1], acc_fma(Jx[2], r[0], pa[35]));
        }
#endif
        K2_STAMP(1);
        // ---- Gram of the staged rows on the matrix cores ----
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double *row = myrows + lane * AROW;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 6; ++i) row[i] = Jp[6 * h + i];
#pragma unroll
            for (int i = 0; i < 9; ++i) row[6 + i] = Jc[9 * h + i];
            row[15] = r[h];
            __builtin_amdgcn_wave_barrier();
            const double *src = myrows + (lane >> 4) * AROW + (lane & 15);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const double val = src[4 * s * AROW];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(val, val, acc, 0, 0, 0);
            }
        }
        // acc[i] = G[(lane>>4) + 4 i][lane & 15] for this wave's observations
#pragma unroll
        for (int i = 0; i < 4; ++i) tiles[wave * 256 + ((lane >> 4) + 4 * i) * 16 + (lane & 15)] = acc[i];
        acc_tot += acc;
        K2_STAMP(2);
        __syncthreads();
        for (int e = tid; e < 256; e += T) {
            double s = tiles[e];
#pragma unroll
            for (int w = 1; w < W; ++w) s += tiles[w * 256 + e];
            gsum[e] = s;
        }
        __syncthreads();

        K2_STAMP(3);
        // ---- pose block: damping, factor, inverse (every lane redundantly, operands broadcast) ----
        double A[21], gvv[6], Dd[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) A[TRI(i, j)] = gsum[i * 16 + j];
            gvv[i] = gsum[i * 16 + 15];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double d = A[TRI(i, i)];
            dmax = fmax(dmax, d);
            gmax = fmax(gmax, fabs(gvv[i]));
            if (MODE == 0) {
                const double s2 = as_uniform(a.s2pose)[6 * v + i];
                double c = s2 * d;
                c = fmin(fmax(c, a.min_diag), a.max_diag);
                Dd[i] = c * a.inv_radius / s2 + a.mu;
                A[TRI(i, i)] = d + Dd[i];
            } else {
                Dd[i] = d;
            }
        }
        if (MODE == 1) {
            if (a.s2pose_out && tid < 6) {
                const double s = 1.0 / (1.0 + sqrt(gsum[tid * 16 + tid]));
                a.s2pose_out[6 * v + tid] = s * s;
            }
        }
        if (MODE == 0) {
            const bool ok = chol6_factor(A);
            if (!ok) nbad = 1.0;
            K2_STAMP(4);
            double *Zv = a.Z + (size_t)6 * v * a.ZC;
            // point columns: Z[.][b P + pid] = L^-1 (Jp^T Jx)[.][b].  Z's point columns are coordinate-major (all x, then
            // all y, then all z): lanes hold consecutive points in the usual board order, so each of these stores is
            // one contiguous run of 8 bytes per lane; point-major (3 pid + b) they were 8-byte pieces 24 bytes apart,
            // three instructions to a line, and the kernel is partly bound by the rate at which such pieces are taken
            // (DESIGN.md section 4).  The column order is private to Z: k_assemble and the step's embedding map it.
            if (active && a.pts_free) {
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    double w[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) w[k] = acc_fma(Jp[6 + k], Jx[3 + b], Jp[k] * Jx[b]);
                    chol6_forward(A, w);
#pragma unroll
                    for (int i = 0; i < 6; ++i) Zv[(size_t)i * a.ZC + b * a.P + pid] = w[i];
                }
            }
            K2_STAMP(6);
            // camera columns (lanes 0..8), reduced-gradient column (lane 9)
            if (tid < 10) {
                const bool isy = (tid == 9);
                if (isy || a.cam_free) {
                    double w[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) w[k] = isy ? gvv[k] : gsum[k * 16 + 6 + tid];
                    const int col = isy ? a.ycol : a.camcol0 + tid;
                    chol6_forward(A, w);
#pragma unroll
                    for (int i = 0; i < 6; ++i) Zv[(size_t)i * a.ZC + col] = w[i];
                }
            }
            // per-view records: one lane of the LAST wave (which has nothing else to do here), so that the 33
            // single-lane stores do not sit behind the camera columns on wave 0's path to the next barrier
            if (tid == T - 1) {
#pragma unroll
                for (int i = 0; i < 21; ++i) a.Lfac[(size_t)21 * v + i] = A[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    a.Dv[6 * v + i] = Dd[i];
                    a.gv[6 * v + i] = gvv[i];
                }
            }
        }
        K2_STAMP(5);
    }
#ifdef K2_WALL
    w_loop = wall_clock64();
#endif
    asm volatile("" :: "s"(rt_touch));      // (keeps the touching loads alive)
    // ---- flush the group's accumulators ----
    double *sp = a.slab_pt + (size_t)g * a.P * PT_ACC;
#ifdef K2_NO_PACC
    if (false) {
#else
    if (a.pts_free) {
#endif
        if (ALIGNED) {
            // points the group never sees keep the zeros the slab was created with
            if (v0 < v1 && tid < a.view_off[v0 + 1] - a.view_off[v0]) {
#pragma unroll
                // (plain stores: as non-temporal 8-byte pieces 288 bytes apart they made the kernel 20 us slower)
                for (int i = 0; i < PT_ACC; ++i) sp[pid * PT_ACC + i] = pacc[i];
            }
        } else {
            for (int e = tid; e < a.P * PT_ACC; e += T) sp[e] = accpt[e];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) tiles[wave * 256 + ((lane >> 4) + 4 * i) * 16 + (lane & 15)] = acc_tot[i];
    __syncthreads();
    for (int e = tid; e < 256; e += T) {
        double s = tiles[e];
#pragma unroll
        for (int w = 1; w < W; ++w) s += tiles[w * 256 + e];
        a.slab_cam[(size_t)g * 256 + e] = s;
    }
    if (tid == 0) {
        a.grp_scal[4 * g + 0] = gmax;
        a.grp_scal[4 * g + 1] = dmax;
        a.grp_scal[4 * g + 2] = nbad;
#ifdef K2_WALL
        { const long long w_end = wall_clock64();
          a.grp_scal[4 * g + 3] = (g % 4 == 0) ? (double)w_start : (g % 4 == 1) ? (double)w_pro : (g % 4 == 2) ? (double)w_loop : (double)w_end; }
#elif defined(K2_PROF)
        a.grp_scal[4 * g + 3] = (double)ph[g % 8];
#else
        a.grp_scal[4 * g + 3] = 0.0;
#endif
    }
}

// -----------------------------------------------------------------------------------------
// K3: Schur contraction  S_local = Z^T Z  on fp64 MFMA (v_mfma_f64_16x16x4_f64): a SYRK with
// K = 6V rows and N = ZC columns.  The upper triangle is cut into 16x16 blocks (row-major over the
// triangle); a workgroup of 12 waves owns one chunk of Z rows and one "part" of at most
// 12 x 10 = 120 blocks, ten accumulators per wave.  The chunk's rows stream through LDS in
// double-buffered stages of KB rows (16-byte coalesced loads of whole rows, staged in registers
// while the previous stage is being multiplied), so every Z row is read from HBM once per part and
// every MFMA operand comes from LDS: lane (kq, cq) reads Z[4 ks + kq][16 blk + cq], conflict-free
// because the LDS row pitch is 16 mod 32 doubles.  Partial blocks go to a slab
// [chunk][block][16*16] and are summed in fixed order by k_group_sum.
// blockIdx -> (chunk, part): the parts of one chunk sit on one XCD (blockIdx % 8), which keeps the
// second read of the chunk's rows in that XCD's L2.
// -----------------------------------------------------------------------------------------
// Where entry (row, col) of a 16x16 partial block lies among the block's 256 doubles in the slab (and in the per-camera
// sums made from it): in the order in which k_schur_syrk4's lanes hold it, so that its result leaves the registers as
// two 16-byte stores per lane and block, 1 KB contiguous per instruction (row-major it was four 8-byte stores, 32-byte
// pieces of 16 different lines each).  Lane l = 16 i + 4 b + j of accumulator r holds row 4 b + i, column
// 4 ((b - r) & 3) + j; it stores accumulators 0, 1 at doubles 2 l, 2 l + 1 and accumulators 2, 3 at 128 + 2 l, + 1.
__host__ __device__ inline int k4_slot(int row, int col) {
    const int b = row >> 2, i = row & 3, r = (b - (col >> 2)) & 3, l = 16 * i + 4 * b + (col & 3);
    return 128 * (r >> 1) + 2 * l + (r & 1);
}

constexpr int K3_WAVES = 12, K3_MAXB = 10, K3_CAP = K3_WAVES * K3_MAXB;

struct K3Args {
    const double *Z;
    const int *chk_row0;   // [n_chunks + 1] first row of each chunk
    double *slab;          // [n_chunks][nblk][256]
    int ZC, ZCp, KB, n_chunks, nb, nblk, parts;
};

__global__ void __launch_bounds__(64 * K3_WAVES) k_schur_syrk(K3Args a) {
    extern __shared__ double lds[];      // [2][KB][ZCp]
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int part = slot % a.parts, chunk = (slot / a.parts) * 8 + xcd;
    if (chunk >= a.n_chunks) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane >> 4, cq = lane & 15;
    const int r0 = a.chk_row0[chunk], r1 = a.chk_row0[chunk + 1];
    const int ZC = a.ZC, ZCp = a.ZCp, KB = a.KB;
    // this wave's blocks: q0 .. q0 + cnt of this part
    const int pcap = (a.nblk + a.parts - 1) / a.parts;
    const int pq0 = part * pcap, pq1 = min(a.nblk, pq0 + pcap);
    const int per = (pq1 - pq0 + K3_WAVES - 1) / K3_WAVES;
    const int q0 = pq0 + wave * per;
    const int cnt = max(0, min(per, pq1 - q0));
    int offA[K3_MAXB], offB[K3_MAXB];
    {
        int bi = 0, rem = q0;
        while (bi < a.nb && rem >= a.nb - bi) { rem -= a.nb - bi; ++bi; }
        int bj = bi + rem;
#pragma unroll
        for (int i = 0; i < K3_MAXB; ++i) {
            offA[i] = min(bi, a.nb - 1) * 16; offB[i] = min(bj, a.nb - 1) * 16;
            if (++bj >= a.nb) { ++bi; bj = bi; }
        }
    }
    f64x4 acc[K3_MAXB];
#pragma unroll
    for (int i = 0; i < K3_MAXB; ++i) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};

    // staging: KB rows x ZC/2 double2 per stage, 4 per thread at most
    const int halfc = ZC >> 1, per_stage = KB * halfc;
    constexpr int NLD = 4;
    const int nstages = (r1 - r0 + KB - 1) / KB;
    double2 st[NLD];
    auto fetch = [&](int s) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + 64 * K3_WAVES * i;
            const int row = e / halfc, c2 = e - row * halfc;
            const int grow = r0 + s * KB + row;
            st[i] = double2{0.0, 0.0};
            if (e < per_stage && grow < r1) st[i] = *(const double2 *)(a.Z + (size_t)grow * ZC + 2 * c2);
        }
    };
    auto stash = [&](int buf) {
        double *dst = lds + (size_t)buf * KB * ZCp;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + 64 * K3_WAVES * i;
            const int row = e / halfc, c2 = e - row * halfc;
            if (e < per_stage) *(double2 *)(dst + row * ZCp + 2 * c2) = st[i];
        }
    };
    if (nstages > 0) { fetch(0); stash(0); }
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
        if (s + 1 < nstages) fetch(s + 1);
        const double *src = lds + (size_t)(s & 1) * KB * ZCp + kq * ZCp + cq;
        for (int ks = 0; ks < KB; ks += 4) {
            const double *p = src + ks * ZCp;
#pragma unroll
            for (int i = 0; i < K3_MAXB; ++i)
                if (i < cnt) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(p[offA[i]], p[offB[i]], acc[i], 0, 0, 0);
        }
        if (s + 1 < nstages) stash((s + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < K3_MAXB; ++i)
        if (i < cnt) {
            double *out = a.slab + ((size_t)chunk * a.nblk + q0 + i) * 256;
#pragma unroll
            for (int q = 0; q < 4; ++q) out[k4_slot(kq + 4 * q, cq)] = acc[i][q];
        }
}

// -----------------------------------------------------------------------------------------
// K3, second form: the same contraction on v_mfma_f64_4x4x4_4b_f64, which this part issues at 16 cycles per
// 512 flop (the 16x16x4 instruction: 103 cycles per 2048 flop, 62 % of that).  The instruction multiplies four
// independent 4x4x4 blocks and pairs block b of A with block b of B only (lane i + 4b + 16k of A, j + 4b + 16k of
// B, result in lane j + 4b + 16i), so an operand register loaded exactly like the 16x16x4 one -- lane (kq, cq) =
// Z[4 ks + kq][16 g + cq] -- gives the four diagonal 4x4 blocks of the 16x16 product of groups (gA, gB); the
// other twelve come from three DPP rotations of the B register by 4, 8, 12 lanes.  Operands would have to come
// from LDS twice per four instructions that way, which is slower than 16x16x4; instead a wave keeps a register
// tile of 4 x 4 groups (64 x 64 outputs, 64 accumulator registers): per k-step 4 A reads, 4 B reads and the B
// rotations feed 64 instructions.
// Tasks (one per wave, K4_WAVES per workgroup, `parts` workgroups per chunk, built by the host):
//   FULL  tile (I, J), I < J: groups 4I..4I+3 x 4J..4J+3, those below nb
//   DIAG  tile (I, I): the ten blocks pa <= qb; B operands are the A registers.  When nb = 1 mod 4 the last
//         group would make a column of nearly empty tiles: its blocks (4I + pa, nb - 1) ride in four of the six
//         unused accumulator slots of DIAG task I, and the corner block (nb - 1, nb - 1) in a fifth.
// -----------------------------------------------------------------------------------------
constexpr int K4_WAVES = 8;
constexpr int K4_BUFS = 3;        // stage buffers in LDS
constexpr int K4_KB = 16;         // rows per stage
constexpr int K4_PITCH = 400;     // LDS row pitch in doubles (16 mod 32: conflict-free operand reads); ZC <= 400

struct K4Task {
    int a0, b0;      // first column group of the A and B side
    int nA, nB;      // valid groups (1..4)
    int diag;        // 1: DIAG task (b0 == a0)
    int edge;        // DIAG: column group riding in the free slots, or -1
    int corner;      // DIAG: 1 if the block (edge, edge) rides too
    int copy;        // rows of every stage this wave copies into LDS: first | count << 8 (k4_build_tasks)
};

struct K4Args {
    const double *Z;
    const int *chk_row0;
    const K4Task *tasks;   // [parts * K4_WAVES], nA = 0 marks an idle wave
    double *slab;          // [n_chunks][nblk][256]
    int ZC, ZCp, KB, n_chunks, nb, nblk, parts;   // ZCp, KB: informational (the kernel uses K4_PITCH, K4_KB)
    int zero_row;          // first of the (at least two) zero rows behind Z's last row
};

#define K4_MFMA(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f64_4x4x4f64(A_, B_, C_, 0, 0, 0)

// one global_load_lds_dwordx4: lane l copies 16 bytes from g_base + v_off + OFF to LDS byte address l_base + OFF + 16 l
// (g_base and l_base wave-uniform: the global address is a scalar base plus a 32-bit lane offset, so a copy costs no
// vector address arithmetic; l_base sits in M0 for the duration of the instruction; M0 is the compiler's, so it is
// put back)
template <int OFF>
__device__ __forceinline__ void k4_glds16(const void *g_base, unsigned v_off, unsigned l_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v_off), "s"(g_base), "s"(l_base), "n"(OFF) : "memory");
}

// MODE 0: FULL 4 x 4 groups; 1: DIAG, 4 groups; 2: DIAG, 4 groups + the edge group: no branch in the k loop.
// 3 / 4: FULL / DIAG with fewer than 4 x 4 groups (tiles at the border), one scalar branch per block.
// The rotated B operands are read from LDS as such (lane (kq, cq) reads column 16 g + ((cq - 4r) & 15): the same
// banks as the unrotated read), not made with DPP moves: every VALU instruction between the MFMAs costs them issue
// slots (24 v_mov_dpp per k-step: 15 % of the kernel), LDS reads do not.  All LDS addresses are one VGPR per
// (group, rotation), advanced once per stage; the k-step and the row pitch are immediates (K4_PITCH, K4_KB).
template <int MODE>
__device__ __forceinline__ void k4_body(const K4Args &a, const K4Task t, int chunk, double *lds) {
    constexpr bool DIAG = (MODE == 1 || MODE == 2 || MODE == 4), GEN = MODE >= 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int kq = lane >> 4, cq = lane & 15;
    const int r0 = a.chk_row0[chunk], r1 = a.chk_row0[chunk + 1];
    const int ZC = a.ZC, nb = a.nb;
    const int nA = GEN ? t.nA : 4, nB = GEN ? t.nB : 4;
    const bool has_edge = GEN ? (DIAG && t.edge >= 0) : (MODE == 2);
    const bool has_corner = has_edge && t.corner;
    // element offsets inside a stage buffer: A groups unrotated, B groups (or, DIAG, the A groups again) in four
    // rotations, the edge group in four rotations
    // (the groups of a side are consecutive: one address register per rotation, the group is an immediate; groups
    // beyond nb are read too -- up to 48 columns past ZC, inside the pitch or the slack behind the buffers -- and
    // their blocks skipped)
    const int pA0 = kq * K4_PITCH + t.a0 * 16 + cq;
    int pBr[4], pEr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        pBr[r] = kq * K4_PITCH + t.b0 * 16 + ((cq - 4 * r) & 15);
        pEr[r] = kq * K4_PITCH + (has_edge ? t.edge : 0) * 16 + ((cq - 4 * r) & 15);
    }
    double acc[4][4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

    // staging: K4_KB rows of ZC doubles per stage, copied global -> LDS by the load unit itself (global_load_lds,
    // 16 bytes per lane: no staging registers, no ds_write pass).  One wave instruction fills 1 KB of LDS, lane l at
    // base + 16 l, so a row goes in pieces of 1 KB; which waves copy which rows is in the task list (k4_build_tasks).  The pieces are whole:
    // the last one of a row runs into the next row of Z (memory that exists: Z ends in eight rows of padding) and
    // lands in the pad columns of the LDS row, which no counted block reads.  Rows past the end of the chunk (last
    // stage only) are copied from Z's padding rows, which are zero.  The copies are issued from inline asm: told of
    // an LDS-DMA in flight, the compiler turns every wait for an LDS operand into lgkmcnt(0), which would stall the
    // operand pipeline below on its newest read; the wait for the copies (vmcnt) is written out before the barrier.
    const int nstages = (r1 - r0 + K4_KB - 1) / K4_KB;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
    const unsigned lane16 = 16u * (unsigned)lane;
    const int crow0 = t.copy & 255, crows = t.copy >> 8;      // this wave's rows of a stage (wave-uniform)
    // piece k (0..2; with k = 2 goes, for ZC > 384, the 128-byte rest) of stage row `row`
    auto copy_piece = [&](int s, int buf, int row, int k) {
        const int grow = r0 + s * K4_KB + row;
        const double *g = a.Z + (size_t)(grow < r1 ? grow : a.zero_row) * ZC;      // wave-uniform
        const unsigned l = lds0 + 8u * (unsigned)(buf * K4_KB * K4_PITCH + row * K4_PITCH);
        if (k == 0) k4_glds16<0>(g, lane16, l);
        if (k == 1 && ZC > 128) k4_glds16<1024>(g, lane16, l);
        if (k == 2 && ZC > 256) {
            k4_glds16<2048>(g, lane16, l);
            if (ZC > 384 && lane < 8) k4_glds16<3072>(g, lane16, l);      // pitch 400: 128 bytes are left
        }
    };
    // this wave's rows from `first` on, one after the other
    auto copy_rows = [&](int s, int buf, int first) {
        for (int i = first; i < crows; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) copy_piece(s, buf, crow0 + i, k);
    };
    auto kstep = [&](const double *p) {     // p = stage buffer + 4 ks rows
        double av[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = p[pA0 + 16 * i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = (DIAG && r == 0) ? av[j] : p[pBr[r] + 16 * j];
#pragma unroll
            for (int i = 0; i < (DIAG ? j + 1 : 4); ++i)
                if (!GEN || (i < nA && j < nB)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) K4_MFMA(av[i], bv[r], acc[i][j][r]);
                }
        }
        if (DIAG && has_edge) {
            double ev[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ev[r] = p[pEr[r]];
            // free slots (1,0) (2,0) (2,1) (3,0): blocks (a0 + 0..3, edge); (3,1): the corner block
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                K4_MFMA(av[0], ev[r], acc[1][0][r]);
                if (!GEN || nA > 1) K4_MFMA(av[1], ev[r], acc[2][0][r]);
                if (!GEN || nA > 2) K4_MFMA(av[2], ev[r], acc[2][1][r]);
                if (!GEN || nA > 3) K4_MFMA(av[3], ev[r], acc[3][0][r]);
            }
            if (has_corner) {
#pragma unroll
                for (int r = 0; r < 4; ++r) K4_MFMA(ev[0], ev[r], acc[3][1][r]);
            }
        }
    };
    // Whole tiles (MODE 0..2): the chunk as one pipelined instruction stream.  The two waves of a SIMD run the same
    // code from the same barrier and share one matrix pipe in turn, so they reach every wait for an LDS operand
    // together and stay in step: a read issued right before its use costs the pipe its whole latency, the other wave
    // hides nothing (the k-step form above: matrix pipe 62 % busy).  Here every operand register is read again, for
    // its next use sixteen or more MFMAs later, right after its last use -- one or two ds_read per four MFMAs, in
    // their issue shadow -- and the scheduling fences pin that order.  Three stage buffers and ONE barrier per
    // stage, three quarters into it, which does two things at once: every wave has waited for its own copies of
    // stage s + 1 (issued a stage earlier), so behind the barrier that stage is complete and a wave that finishes
    // stage s goes straight on; and every wave has left stage s - 1, so its buffer is free and the copies of stage
    // s + 2 go there, one at a time between the MFMAs of the last quarter.  The barrier is a bare s_barrier: no wave
    // has to drain its LDS reads for it.  Accumulation order per output is unchanged (rows ascending):
    // bit-identical partial sums.
#define K4_SB() __builtin_amdgcn_sched_barrier(0)
    auto stage_buf = [&](int bufi) -> const double * { return lds + (size_t)bufi * K4_KB * K4_PITCH; };
    auto stage_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's copies of stage s + 1 have landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // operand addresses of the current stage buffer (elements from lds), moved from buffer to buffer in place: an
    // invariant copy of each next to a per-stage one is five to nine registers this kernel does not have
    typedef const __attribute__((address_space(3))) double *lds_cptr;      // a 32-bit LDS address, as the ds_read takes it
    const lds_cptr lbase = (lds_cptr)lds;
    lds_cptr qA = lbase + pA0, qB[4] = {lbase + pBr[0], lbase + pBr[1], lbase + pBr[2], lbase + pBr[3]},
             qE[4] = {lbase + pEr[0], lbase + pEr[1], lbase + pEr[2], lbase + pEr[3]};
    auto stage_full = [&](int s, int buf_free) {
        double A[2][4], B[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) A[0][i] = qA[16 * i];
#pragma unroll
        for (int r = 0; r < 4; ++r) B[r] = qB[r][0];
        K4_SB();
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int ks = g >> 2, j = g & 3, ca = ks & 1;
            const int ob = ((g + 1) >> 2) * 4 * K4_PITCH, oa = (ks + 1) * 4 * K4_PITCH;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < 4; ++i) K4_MFMA(A[ca][i], B[r], acc[i][j][r]);
                K4_SB();
                // rotation r of the next column group takes the place of the one just used
                if (g < 15) B[r] = qB[r][ob + 16 * ((g + 1) & 3)];
                if (j == 2 && ks < 3) A[ca ^ 1][r] = qA[oa + 16 * r];
                // the six copies of stage s + 2, one at a time behind the barrier (a copy is a handful of scalar
                // instructions around the load: between MFMAs they cost nothing, in one block they cost 7 us of 135)
                if (g >= 12 && g <= 14 && (r & 1) && s + 2 < nstages) {
                    const int idx = 2 * (g - 12) + (r >> 1);      // 0..5: row idx / 3, piece idx % 3
                    if (idx / 3 < crows) copy_piece(s + 2, buf_free, crow0 + idx / 3, idx % 3);
                }
                K4_SB();
            }
            if (g == 11) {
                stage_sync();
                if (crows > 2 && s + 2 < nstages) copy_rows(s + 2, buf_free, 2);      // more than the six slots below hold
                K4_SB();
            }
        }
    };
    // DIAG tasks have 60 MFMAs per four rows against the 64 of a FULL one: they read their first operands at the
    // start of each stage (complete since the barrier of the stage before) and can afford that latency; carrying
    // their 24 operand registers across the stages as well spills
    auto stage_diag = [&](int s, int buf_free) {
        // R[j][r]: column group a0 + j rotated by 4 r (r = 0 is A[j] itself).  With the edge group (MODE 2) the A
        // registers are read again in place during the edge blocks, their last use in the four rows; without it
        // they alternate between two sets.
        constexpr int NA = MODE == 2 ? 1 : 2;
        double A[NA][4], R[4][4], E[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) A[0][i] = qA[16 * i];
#pragma unroll
        for (int j = 3; j >= 0; --j)
#pragma unroll
            for (int r = 1; r < 4; ++r) R[j][r] = qB[r][16 * j];
        K4_SB();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ca = MODE == 2 ? 0 : ks & 1;
            const int o0 = ks * 4 * K4_PITCH, o1 = o0 + 4 * K4_PITCH;
            // column groups 3, 2, 1, 0: blocks (i, j), i <= j
#pragma unroll
            for (int j = 3; j >= 0; --j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int i = 0; i <= j; ++i) K4_MFMA(A[ca][i], r == 0 ? A[ca][j] : R[j][r], acc[i][j][r]);
                    K4_SB();
                    if (ks < 3 && r > 0) R[j][r] = qB[r][o1 + 16 * j];
                    if (MODE != 2 && ks < 3 && j == 1) A[(ca ^ 1) % NA][r] = qA[o1 + 16 * r];
                    if (MODE == 2 && j == 1) E[r] = qE[r][o0];
                    K4_SB();
                }
            }
            if (MODE == 2) {
                // free slots (1,0) (2,0) (2,1) (3,0): blocks (a0 + 0..3, edge); (3,1): the corner block
                if (has_corner) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) K4_MFMA(E[0], E[r], acc[3][1][r]);
                }
                K4_SB();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) K4_MFMA(A[0][i], E[r], acc[i == 0 ? 1 : i == 1 ? 2 : i == 2 ? 2 : 3][i == 2 ? 1 : 0][r]);
                    K4_SB();
                    if (ks < 3) A[0][i] = qA[o1 + 16 * i];
                    K4_SB();
                }
            }
            if (ks == 2) {
                stage_sync();       // (a DIAG wave copies no rows: k4_build_tasks)
                K4_SB();
            }
        }
    };
    if (nstages > 0) copy_rows(0, 0, 0);
    if (nstages > 1) copy_rows(1, 1, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int s = 0; s < nstages; ++s) {
        const int buf_next = buf == 2 ? 0 : buf + 1, buf_free = buf == 0 ? 2 : buf - 1;
        if (!GEN) {
            if (DIAG) stage_diag(s, buf_free); else stage_full(s, buf_free);
            const int d = (buf_next - buf) * K4_KB * K4_PITCH;
            qA += d;
#pragma unroll
            for (int r = 0; r < 4; ++r) { qB[r] += d; qE[r] += d; }
        } else {
            const double *src = stage_buf(buf);
            if (nA > 0) {
#pragma unroll
                for (int ks = 0; ks < K4_KB - 4; ks += 4) kstep(src + ks * K4_PITCH);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stage_sync();
            if (s + 2 < nstages) copy_rows(s + 2, buf_free, 0);
            if (nA > 0) kstep(src + (K4_KB - 4) * K4_PITCH);
        }
        buf = buf_next;
    }
#undef K4_SB
    if (nA <= 0) return;
    // accumulator r of block (gA, gB), lane l: row 4b + i, column 4((b - r) & 3) + j with i = l >> 4, b = (l >> 2) & 3,
    // j = l & 3 (the rotation by 4r hands block b the B block b - r); the slab keeps that order (k4_slot)
    auto put = [&](int gA, int gB, const double (&v)[4]) {
        const int q = gA * nb - gA * (gA - 1) / 2 + (gB - gA);
        double *out = a.slab + ((size_t)chunk * a.nblk + q) * 256 + 2 * lane;      // k4_slot order
        // non-temporal: the slab (60 MB at config 4) is written once and read once by the sums; kept out of the caches it
        // leaves Z there for the back substitution and the observations for the next linearisation (measured in
        // bench.py: K3 -2 us, K2 -4 us, back substitution -2 us against plain stores)
        nt_store2(out, v[0], v[1]);
        nt_store2(out + 128, v[2], v[3]);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = DIAG ? i : 0; j < 4; ++j)
            if (i < nA && j < nB) put(t.a0 + i, t.b0 + j, acc[i][j]);
    if (DIAG && has_edge) {
        put(t.a0, t.edge, acc[1][0]);
        if (nA > 1) put(t.a0 + 1, t.edge, acc[2][0]);
        if (nA > 2) put(t.a0 + 2, t.edge, acc[2][1]);
        if (nA > 3) put(t.a0 + 3, t.edge, acc[3][0]);
        if (has_corner) put(t.edge, t.edge, acc[3][1]);
    }
}

#ifdef K4_WALL
__device__ long long g_k4wall[2 * 4096];      // development: start / end of every workgroup (100 MHz)
#endif

__global__ void __launch_bounds__(64 * K4_WAVES, 2) k_schur_syrk4(K4Args a) {
    extern __shared__ double lds[];      // [K4_BUFS][K4_KB][K4_PITCH]
    const int b = blockIdx.x;
#ifdef K4_WALL
    if (threadIdx.x == 0) g_k4wall[2 * b] = wall_clock64();
    struct Stamp { int b; __device__ ~Stamp() { if (threadIdx.x == 0) g_k4wall[2 * b + 1] = wall_clock64(); } } stamp{b};
#endif
    const int xcd = b & 7, slot = b >> 3;
    const int part = slot % a.parts, chunk = (slot / a.parts) * 8 + xcd;
    if (chunk >= a.n_chunks) return;
    K4Task t = a.tasks[part * K4_WAVES + (threadIdx.x >> 6)];
    // wave-uniform: keep the task in scalar registers, so that the per-block tests are scalar branches
    t.a0 = __builtin_amdgcn_readfirstlane(t.a0); t.b0 = __builtin_amdgcn_readfirstlane(t.b0);
    t.nA = __builtin_amdgcn_readfirstlane(t.nA); t.nB = __builtin_amdgcn_readfirstlane(t.nB);
    t.diag = __builtin_amdgcn_readfirstlane(t.diag); t.edge = __builtin_amdgcn_readfirstlane(t.edge);
    t.corner = __builtin_amdgcn_readfirstlane(t.corner); t.copy = __builtin_amdgcn_readfirstlane(t.copy);
    const bool whole = t.nA == 4 && t.nB == 4;
    // The waves of a workgroup run different k4_body instantiations and still meet at the same barriers: s_barrier
    // counts arrivals, not program counters.  The invariant every instantiation keeps is the NUMBER of barriers a
    // wave executes, nstages + 1 (one after the first stage is stashed, one per stage), and nstages depends on the
    // chunk alone; idle waves (nA = 0) take part in the staging and in every barrier before they return.
    if (!t.diag) {
        if (whole) k4_body<0>(a, t, chunk, lds); else k4_body<3>(a, t, chunk, lds);
    } else if (whole) {
        if (t.edge >= 0) k4_body<2>(a, t, chunk, lds); else k4_body<1>(a, t, chunk, lds);
    } else {
        k4_body<4>(a, t, chunk, lds);
    }
}

// host: the task list of k_schur_syrk4 for nb column groups, padded to a multiple of K4_WAVES with idle entries;
// returns the number of parts
inline int k4_build_tasks(int nb, std::vector<K4Task> &tasks) {
    tasks.clear();
    const bool merge = (nb % 4 == 1) && nb > 4;     // the last group rides in the DIAG tasks
    const int ng = merge ? nb - 1 : nb, TI = (ng + 3) / 4;
    // FULL tasks first: every part then holds tasks of the same (largest) size
    for (int I = 0; I < TI; ++I)
        for (int J = I + 1; J < TI; ++J)
            tasks.push_back(K4Task{4 * I, 4 * J, std::min(4, ng - 4 * I), std::min(4, ng - 4 * J), 0, -1, 0, 0});
    for (int I = 0; I < TI; ++I) {
        const int n = std::min(4, ng - 4 * I);
        tasks.push_back(K4Task{4 * I, 4 * I, n, n, 1, merge ? nb - 1 : -1, (merge && I == TI - 1) ? 1 : 0, 0});
    }
    const int parts = ((int)tasks.size() + K4_WAVES - 1) / K4_WAVES;
    while ((int)tasks.size() < parts * K4_WAVES) tasks.push_back(K4Task{0, 0, 0, 0, 0, -1, 0, 0});
    // who copies the K4_KB rows of a stage into LDS: the idle waves of the workgroup if it has any (they have nothing
    // else to do), else its FULL waves (a copy fits between their MFMAs); never a DIAG wave, which has no registers
    // to spare for it (at most six DIAG tasks exist, so every workgroup has another kind of wave); consecutive rows,
    // shared evenly
    for (int p = 0; p < parts; ++p) {
        K4Task *w = tasks.data() + (size_t)p * K4_WAVES;
        int n_idle = 0, n_full = 0;
        for (int i = 0; i < K4_WAVES; ++i) { n_idle += w[i].nA == 0; n_full += w[i].nA > 0 && !w[i].diag; }
        if (n_idle + n_full == 0) { tasks.clear(); return 0; }       // (cannot happen for ZC <= K4_PITCH)
        const int nf = n_idle ? n_idle : n_full;
        int row = 0, f = 0;
        for (int i = 0; i < K4_WAVES; ++i) {
            const bool mine = n_idle ? w[i].nA == 0 : (w[i].nA > 0 && !w[i].diag);
            w[i].copy = 0;
            if (!mine) continue;
            const int cnt = (K4_KB - row + (nf - f) - 1) / (nf - f);
            w[i].copy = row | cnt << 8;
            row += cnt; ++f;
        }
    }
    return parts;
}

// -----------------------------------------------------------------------------------------
// Slab reductions, fixed order.  group_sum_block: dst[c][e] = sum over the slabs listed for camera c
// of src[slab * stride + e]; a block is 64 consecutive outputs x 4 slices of the list, so every
// load is a coalesced 512 B row segment and the four partial sums are combined as (0+1)+(2+3).
// Used for the per-point accumulators (slab_pt -> Tpt[C][P*36]), the per-camera Gram tiles
// (slab_cam -> Gcam[C][256]) and the Schur tiles (slab_tile -> Tt[C][nblk*256]).
// -----------------------------------------------------------------------------------------
struct SumJob {
    const double *src;
    size_t stride;
    int n_out, blocks_x;
    int wide;          // 1: a block is 256 consecutive outputs, each thread sums its whole (short) list
    const int *list, *list_off;
    double *dst;
};

struct ReduceArgs {
    SumJob job[3];
    int njobs, C;
    // scalars of the linearisation (done by the last block): sum r^2 = entry 255 of every group's Gram
    // tile, pose blocks that failed to factor, max |g_pose|, max diag V
    const double *slab_cam, *grp_scal;
    int n_groups;
    // scal = { sum, bad, gmax[nranks], dmax[nranks] }: a rank fills its own max slots and zeroes the
    // others, so that the sum all-reduce of the exchange delivers every rank's maxima
    double *scal;
    int rank, nranks;
    int *fail;      // factorisation flag of the reduced solve, cleared here for this linear step
};

// the slabs are read exactly once: non-temporal loads (with the slab's non-temporal stores and the back substitution's
// non-temporal reads of Z: 0.400 -> 0.388 ms per LM iteration at config 4)
__device__ __forceinline__ double2 sum_load2(const double *p) { return nt_load2(p); }

__device__ __forceinline__ void group_sum_block(const SumJob &j, int bx, int c, double (*part)[128]) {
    if (j.wide) {
        // short lists (the Schur tiles: one entry per chunk of the camera): 2 KB coalesced rows, no LDS, sixteen
        // independent loads in flight per thread
        // (two adjacent outputs per thread: 16-byte loads, 1 KB per wave instruction; n_out is a multiple of 256)
        const int e = 2 * (bx * 256 + threadIdx.x);
        if (e >= j.n_out) return;
        const int q0 = j.list_off[c], q1 = j.list_off[c + 1];
        double2 s = {0.0, 0.0};
        int q = q0;
        for (; q + 16 <= q1; q += 16) {
            double2 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = sum_load2(j.src + (size_t)j.list[q + u] * j.stride + e);
            s.x += (((v[0].x + v[1].x) + (v[2].x + v[3].x)) + ((v[4].x + v[5].x) + (v[6].x + v[7].x))) +
                   (((v[8].x + v[9].x) + (v[10].x + v[11].x)) + ((v[12].x + v[13].x) + (v[14].x + v[15].x)));
            s.y += (((v[0].y + v[1].y) + (v[2].y + v[3].y)) + ((v[4].y + v[5].y) + (v[6].y + v[7].y))) +
                   (((v[8].y + v[9].y) + (v[10].y + v[11].y)) + ((v[12].y + v[13].y) + (v[14].y + v[15].y)));
        }
        for (; q < q1; ++q) {
            const double2 v = sum_load2(j.src + (size_t)j.list[q] * j.stride + e);
            s.x += v.x; s.y += v.y;
        }
        *(double2 *)(j.dst + (size_t)c * j.n_out + e) = s;
        return;
    }
    // 128 outputs (two per lane, 16-byte loads; n_out is even) x 4 slices of the list
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int e = bx * 128 + 2 * lane;
    const int q0 = j.list_off[c], q1 = j.list_off[c + 1];
    const int per = (q1 - q0 + 3) >> 2;
    const int a0 = q0 + sl * per, a1 = min(q1, a0 + per);
    double2 s = {0.0, 0.0};
    if (e < j.n_out) {
        int q = a0;
        for (; q + 8 <= a1; q += 8) {       // eight independent loads in flight
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sum_load2(j.src + (size_t)j.list[q + u] * j.stride + e);
            s.x += ((v[0].x + v[1].x) + (v[2].x + v[3].x)) + ((v[4].x + v[5].x) + (v[6].x + v[7].x));
            s.y += ((v[0].y + v[1].y) + (v[2].y + v[3].y)) + ((v[4].y + v[5].y) + (v[6].y + v[7].y));
        }
        for (; q < a1; ++q) {
            const double2 v = sum_load2(j.src + (size_t)j.list[q] * j.stride + e);
            s.x += v.x; s.y += v.y;
        }
    }
    part[sl][2 * lane] = s.x; part[sl][2 * lane + 1] = s.y;
    __syncthreads();
    if (sl == 0 && e < j.n_out) {
        double2 r;
        r.x = (part[0][2 * lane] + part[1][2 * lane]) + (part[2][2 * lane] + part[3][2 * lane]);
        r.y = (part[0][2 * lane + 1] + part[1][2 * lane + 1]) + (part[2][2 * lane + 1] + part[3][2 * lane + 1]);
        *(double2 *)(j.dst + (size_t)c * j.n_out + e) = r;
    }
}

// all slab sums of one linearisation in one launch: blocks [0, sum_j blocks_x_j * C) do the jobs,
// the last block the scalars
__global__ void __launch_bounds__(256) k_reduce_all(ReduceArgs a) {
    __shared__ double part[4][128];
    int b = blockIdx.x;
    for (int ji = 0; ji < a.njobs; ++ji) {
        const int nb = a.job[ji].blocks_x * a.C;
        if (b < nb) {
            group_sum_block(a.job[ji], b % a.job[ji].blocks_x, b / a.job[ji].blocks_x, part);
            return;
        }
        b -= nb;
    }
    double gm = 0.0, dm = 0.0, nb = 0.0, cs = 0.0;
    for (int q = threadIdx.x; q < a.n_groups; q += 256) {
        gm = fmax(gm, a.grp_scal[4 * q]);
        dm = fmax(dm, a.grp_scal[4 * q + 1]);
        nb += a.grp_scal[4 * q + 2];
        cs += a.slab_cam[(size_t)q * 256 + 255];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        gm = fmax(gm, __shfl_xor(gm, off, 64)); dm = fmax(dm, __shfl_xor(dm, off, 64));
        nb += __shfl_xor(nb, off, 64); cs += __shfl_xor(cs, off, 64);
    }
    if (lane == 0) { part[0][wave] = gm; part[1][wave] = dm; part[2][wave] = nb; part[3][wave] = cs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 0; q < 2 * a.nranks; ++q) a.scal[2 + q] = 0.0;
        a.scal[2 + a.rank] = fmax(fmax(part[0][0], part[0][1]), fmax(part[0][2], part[0][3]));
        a.scal[2 + a.nranks + a.rank] = fmax(fmax(part[1][0], part[1][1]), fmax(part[1][2], part[1][3]));
        a.scal[1] = (part[2][0] + part[2][1]) + (part[2][2] + part[2][3]);
        a.scal[0] = (part[3][0] + part[3][1]) + (part[3][2] + part[3][3]);
        *a.fail = 0;
    }
}

// -----------------------------------------------------------------------------------------
// Assemble the exchange buffer E (one thread per entry):
//   E = [ S = U - Z^T Z (ns*ns) | gred (ns) | gs (ns) | diagU (ns) | scalars... ]
// shared-parameter order (API): free intrinsics of camera 0, 1, ..., then points xyz.
// Inputs are the per-camera sums made by k_group_sum: Tpt[C][P][36] (Ucx 27 | Uxx 6 | gx 3),
// Gcam[C][256], Tt[C][nblk][16*16] (null when the Schur term is absent).
// -----------------------------------------------------------------------------------------
struct AsmArgs {
    const double *Tt, *Tpt, *Gcam;
    int ns, C, P, ncf, fix_k3, pts_free, ZC, camcol0, ycol, nb, nblk;
    double *E;
    // single-rank runs: the damped, padded matrix of the reduced solve is written in the same pass (what
    // cvxs::k_damp_prepare makes from E when an exchange step lies between); A == nullptr: E only
    double *A, *Ds;
    const double *s2;
    int npad;
    double inv_radius, mu, min_diag, max_diag;
    // view-sharded runs: what the ranks exchange, X = [lower triangle of S, packed i(i+1)/2 + j | gred | gs | diagU |
    // scalars] -- half of E; X == nullptr otherwise
    double *X;
    int n_scal;
    const LmDev *st;      // speculated steps: the damping comes from the record
};

struct SharedIdx {
    int is_cam, c, kk, p, b, zcol;
};

__device__ __forceinline__ SharedIdx shared_index(const AsmArgs &a, int i) {
    SharedIdx s;
    const int ncam = a.C * a.ncf;
    if (i < ncam) {
        s.is_cam = 1; s.c = i / a.ncf;
        const int k = i % a.ncf;
        s.kk = (a.fix_k3 && k >= 6) ? k + 1 : k;
        s.p = 0; s.b = 0; s.zcol = a.camcol0 + s.kk;
    } else {
        const int q = i - ncam;
        s.is_cam = 0; s.c = -1; s.kk = 0; s.p = q / 3; s.b = q % 3; s.zcol = s.b * a.P + s.p;
    }
    return s;
}

// offset of Z^T Z entry (zi, zj) inside one camera's block array [nblk][16*16] (upper triangle,
// row-major over blocks)
__device__ __forceinline__ size_t tile_offset(const AsmArgs &a, int zi, int zj) {
    int bi = zi >> 4, bj = zj >> 4;
    if (bi > bj) { int t = bi; bi = bj; bj = t; t = zi; zi = zj; zj = t; }
    const int q = bi * a.nb - bi * (bi - 1) / 2 + (bj - bi);
    return (size_t)q * 256 + k4_slot(zi & 15, zj & 15);
}

// sum of the cameras' tile entries: all cameras (cam < 0) or one
__device__ __forceinline__ double tile_sum(const AsmArgs &a, int cam, int zi, int zj) {
    if (!a.Tt) return 0.0;
    const size_t off = tile_offset(a, zi, zj), cs = (size_t)a.nblk * 256;
    if (cam >= 0) return a.Tt[cam * cs + off];
    double z = 0.0;
    for (int c = 0; c < a.C; ++c) z += a.Tt[c * cs + off];
    return z;
}

// Thread grid: pitch x pitch matrix entries, then ns vector entries; pitch = ns, or npad when A is written too
// (the entries outside ns x ns are then A's right-hand-side row, its 1e300 pivot and the identity padding).
__global__ void __launch_bounds__(256) k_assemble(AsmArgs a) {
    if (a.st) {
        const uniform_lm st = as_uniform(a.st);
        if (!st->active) return;
        a.inv_radius = st->inv_radius; a.mu = st->mu; a.min_diag = st->min_diag; a.max_diag = st->max_diag;
    }
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nn = (long long)a.ns * a.ns;
    const int pitch = a.A ? a.npad : a.ns;
    const long long mm = (long long)pitch * pitch;
    if (t >= mm + a.ns) return;
    const int sym6[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
    if (t < mm) {
        const int i = (int)(t / pitch), j = (int)(t % pitch);
        if (i >= a.ns || j >= a.ns) {
            // row ns below column ns is the right-hand side (the vector threads write it); the rest is constant
            if (!(i == a.ns && j < a.ns)) a.A[t] = (i == j) ? (i == a.ns ? 1e300 : 1.0) : 0.0;
            return;
        }
        const SharedIdx si = shared_index(a, i), sj = shared_index(a, j);
        double u = 0.0, z = 0.0;
        if (si.is_cam && sj.is_cam) {
            if (si.c == sj.c) {
                u = a.Gcam[(size_t)si.c * 256 + (6 + si.kk) * 16 + 6 + sj.kk];
                z = tile_sum(a, si.c, si.zcol, sj.zcol);
            }
        } else if (si.is_cam) {
            u = a.Tpt[((size_t)si.c * a.P + sj.p) * PT_ACC + si.kk * 3 + sj.b];
            z = tile_sum(a, si.c, si.zcol, sj.zcol);
        } else if (sj.is_cam) {
            u = a.Tpt[((size_t)sj.c * a.P + si.p) * PT_ACC + sj.kk * 3 + si.b];
            z = tile_sum(a, sj.c, si.zcol, sj.zcol);
        } else {
            if (si.p == sj.p)
                for (int c = 0; c < a.C; ++c) u += a.Tpt[((size_t)c * a.P + si.p) * PT_ACC + 27 + sym6[si.b * 3 + sj.b]];
            z = tile_sum(a, -1, si.zcol, sj.zcol);
        }
        const double e = u - z;
        a.E[(size_t)i * a.ns + j] = e;
        if (a.X && i >= j) a.X[(size_t)i * (i + 1) / 2 + j] = e;
        if (a.A) {
            double v = e;
            if (i == j) {       // D_i = clamp(s_i^2 U_ii, min, max) / (radius s_i^2) + mu, as k_damp_prepare
                const double s2 = a.s2[i];
                double c = s2 * u;
                c = fmin(fmax(c, a.min_diag), a.max_diag);
                c = c * a.inv_radius / s2 + a.mu;
                a.Ds[i] = c;
                v += c;
            }
            a.A[t] = v;
        }
    } else {
        const int i = (int)(t - mm);
        const SharedIdx si = shared_index(a, i);
        double g = 0.0, d = 0.0;
        if (si.is_cam) {
            g = a.Gcam[(size_t)si.c * 256 + (6 + si.kk) * 16 + 15];
            d = a.Gcam[(size_t)si.c * 256 + (6 + si.kk) * 16 + 6 + si.kk];
        } else {
            for (int c = 0; c < a.C; ++c) {
                const double *pa = a.Tpt + ((size_t)c * a.P + si.p) * PT_ACC;
                g += pa[33 + si.b];
                d += pa[27 + sym6[si.b * 4]];
            }
        }
        const double z = tile_sum(a, si.is_cam ? si.c : -1, si.zcol, a.ycol);
        a.E[nn + i] = g - z;
        a.E[nn + a.ns + i] = g;
        a.E[nn + 2 * a.ns + i] = d;
        if (a.A) a.A[(size_t)a.ns * a.npad + i] = -(g - z);
        if (a.X) {
            double *xt = a.X + (size_t)a.ns * (a.ns + 1) / 2;
            xt[i] = g - z; xt[a.ns + i] = g; xt[2 * a.ns + i] = d;
            if (i == 0)     // the scalars of the linearisation, left behind the vectors of E by k_reduce_all
                for (int q = 0; q < a.n_scal; ++q) xt[3 * a.ns + q] = a.E[nn + 3 * a.ns + q];
        }
    }
}

// -----------------------------------------------------------------------------------------
// final scalar reduction (single block of 1024, fixed order):
//   E2 = { sum cost partials + sum view4[.][0], sum view4[.][1], sum view4[.][2], sum view4[.][3] }
// cost_partial[n_partial]: the blocks of k_cost (the initial cost; n_partial = 0 in a step); view4[V][4]: what
// k_backsubst_cost_step leaves per view = {candidate cost, dp^T (D dp - g), |dp|^2, |pose|^2} (V = 0 at the start).
// A single workgroup is bound by the latency of its loads, so all of them are in flight at once: a thread takes every
// 1024th view, two 16-byte loads each, FS_PER views per round (one round up to 10 240 views).
// -----------------------------------------------------------------------------------------
constexpr int FS_PER = 10;
__device__ __forceinline__ void final_sums(const double *__restrict__ cost_partial, int n_partial,
                                           const double *__restrict__ view4, int V, double (*sh)[16]) {
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < n_partial; i += 1024) s[0] += cost_partial[i];
    for (int base = 0; base < V; base += FS_PER * 1024) {
        double2 a[FS_PER], b[FS_PER];
#pragma unroll
        for (int u = 0; u < FS_PER; ++u) {
            const int i = base + (int)threadIdx.x + u * 1024;
            const double2 *p = reinterpret_cast<const double2 *>(view4 + 4 * (size_t)(i < V ? i : 0));   // unconditional loads
            a[u] = p[0]; b[u] = p[1];
            if (i >= V) { a[u].x = 0.0; a[u].y = 0.0; b[u].x = 0.0; b[u].y = 0.0; }
        }
        static_assert(FS_PER == 10, "the sums below are written out for ten");
        s[0] += (((a[0].x + a[1].x) + (a[2].x + a[3].x)) + ((a[4].x + a[5].x) + (a[6].x + a[7].x))) + (a[8].x + a[9].x);
        s[1] += (((a[0].y + a[1].y) + (a[2].y + a[3].y)) + ((a[4].y + a[5].y) + (a[6].y + a[7].y))) + (a[8].y + a[9].y);
        s[2] += (((b[0].x + b[1].x) + (b[2].x + b[3].x)) + ((b[4].x + b[5].x) + (b[6].x + b[7].x))) + (b[8].x + b[9].x);
        s[3] += (((b[0].y + b[1].y) + (b[2].y + b[3].y)) + ((b[4].y + b[5].y) + (b[6].y + b[7].y))) + (b[8].y + b[9].y);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_xor(s[k], off, 64);
        if (lane == 0) sh[k][wave] = s[k];
    }
}
// one of the four sums from its sixteen wave partials
__device__ __forceinline__ double final_sum16(const double *q) {
    return (((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]))) +
           (((q[8] + q[9]) + (q[10] + q[11])) + ((q[12] + q[13]) + (q[14] + q[15])));
}

__global__ void __launch_bounds__(1024) k_final_scalars(const double *__restrict__ cost_partial, int n_partial,
                                                        const double *__restrict__ view4, int V,
                                                        double *__restrict__ E2, double *__restrict__ host_copy,
                                                        double seq) {
    __shared__ double sh[4][16];
    // the step's other scalars, left in E2 by workgroup 0 of k_backsubst_cost_step (or k_step_finish): fetched now, used last
    const double e2 = (threadIdx.x >= 4 && threadIdx.x < 16) ? E2[threadIdx.x] : 0.0;
    final_sums(cost_partial, n_partial, view4, V, sh);
    __syncthreads();
    if (threadIdx.x < 16) {
        double v = e2;
        if (threadIdx.x < 4) {
            v = final_sum16(sh[threadIdx.x]);
            E2[threadIdx.x] = v;
        }
        // single-rank runs: the 16 doubles go straight into host-mapped memory, then a sequence number the host
        // polls for (no copy kernel, no interrupt); the first wave holds all 16 writers, so no block barrier
        if (host_copy) {
            if (threadIdx.x < 15) host_copy[threadIdx.x] = v;
            __threadfence_system();
            __builtin_amdgcn_wave_barrier();
            if (threadIdx.x == 0) {
                host_copy[16] = seq;
                __threadfence_system();
            }
        }
    }
}

// view-sharded runs: the 16 doubles of the step, complete once the candidate's four sums have been all-reduced, go to
// host-mapped memory the same way k_final_scalars sends them on one rank (data, system-scope fence, sequence number)
__global__ void __launch_bounds__(64) k_publish_scalars(const double *__restrict__ E2, double *__restrict__ host_copy, double seq) {
    if (threadIdx.x < 15) host_copy[threadIdx.x] = E2[threadIdx.x];
    __threadfence_system();
    __builtin_amdgcn_wave_barrier();
    if (threadIdx.x == 0) {
        host_copy[16] = seq;
        __threadfence_system();
    }
}

// k_final_scalars of a speculated step, with the loop's decision behind the sums (trust_region.h: the host runs the same
// functions on the same 16 doubles).  The record moves on to the NEXT step: current buffer, damping, state; `active`
// stays 1 only on the loop's common path -- the step accepted (and no best point to save) or rejected or invalid, the
// solve not stopped, iterations left.  The 16 doubles go to the host as k_final_scalars sends them, R[15] = 1 when the
// kernels enqueued behind this one will run.
__global__ void __launch_bounds__(1024) k_final_scalars_decide(const double *__restrict__ cost_partial, int n_partial,
                                                               const double *__restrict__ view4, int V,
                                                               double *__restrict__ E2, double *__restrict__ host_copy,
                                                               double seq, LmDev *__restrict__ st) {
    if (!as_uniform(st)->active) {
        if (threadIdx.x == 0) st->last_cand = -1;
        return;
    }
    __shared__ double sh[4][16];
    __shared__ double rec[20];
    const double e2 = (threadIdx.x >= 4 && threadIdx.x < 16) ? E2[threadIdx.x] : 0.0;
    final_sums(cost_partial, n_partial, view4, V, sh);
    __syncthreads();
    if (threadIdx.x < 16) {
        double v = e2;
        if (threadIdx.x < 4) {
            v = final_sum16(sh[threadIdx.x]);
            E2[threadIdx.x] = v;
        }
        rec[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        cvx::TrState tr = st->tr;
        const cvx::TrOptions opt = st->opt;
        const int cand_buf = 1 - st->cur;
        double cost_now;
        const cvx::TrStep step = step_from_record(rec, tr.need_gcheck ? opt.gradient_tolerance : -1.0, &cost_now);
        cvx::TrTrace trace;
        int what = cvx::tr_begin_step(tr, step, trace), save_best = 0;
        if (what == cvx::TR_GO_ON)
            what = cvx::tr_finish_step(tr, opt, step.cand_cost, step.model_change, step.step_norm2, step.x_norm2, 0, trace, &save_best);
        const bool moves = what == cvx::TR_ACCEPTED && !save_best;
        const bool go_on = (moves || what == cvx::TR_REJECTED || what == cvx::TR_INVALID) && !tr.term && tr.it < opt.max_iterations;
        st->tr = tr;
        if (moves) st->cur = cand_buf;
        st->last_cand = cand_buf;
        st->inv_radius = 1.0 / tr.radius; st->mu = 0.0; st->min_diag = st->min_lm_diagonal; st->max_diag = st->max_lm_diagonal;
        st->active = go_on ? 1 : 0;
        rec[15] = go_on ? 1.0 : 0.0;
        E2[15] = rec[15];
        rec[16] = 0.0; rec[17] = st->inv_radius; rec[18] = (double)st->cur;     // what the next step will run with
    }
    __syncthreads();
    if (threadIdx.x < 19 && threadIdx.x != 16) {
        const double v = rec[threadIdx.x];
        if (host_copy) {
            host_copy[threadIdx.x] = v;
            __threadfence_system();
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (host_copy && threadIdx.x == 0) {
        host_copy[16] = seq;
        __threadfence_system();
    }
}

}  // namespace cvxk
