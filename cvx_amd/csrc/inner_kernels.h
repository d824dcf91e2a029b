// Inner iterations of the calibration solve (Solver::Options::use_inner_iterations = true at
// /root/reference/modules/calib/src/calibration.cpp:424; no inner_iteration_ordering is given).
// Included only by calib.hip.
//
// What ceres::Solve does with that option (Ceres is un-vendored: restated from its published algorithm, parity
// unpinned): after every trust-region step the candidate point goes through one pass of
// CoordinateDescentMinimizer::Minimize -- the parameter blocks are split into independent sets (no residual block
// touches two blocks of one set; host side: cvx_calib_inner_ordering), and set after set every block is minimised
// over its own residual blocks, all other blocks held, by a trust-region LM with Ceres' default options (dense
// solve, Jacobi scaling, monotonic steps, at most 50 iterations, function tolerance 1e-6).
//
// Here one set is one batch: every block of the set carries its own LM state (radius, cost, iteration count,
// termination) in HBM, and one round of the host loop advances every block that is still running by one LM
// iteration:
//   lin    normal equations H = J^T J, g = J^T r and the cost over the block's own observations
//            pose block:   one wave per view (rotation table of the block's pose made in registers)
//            point block:  pt_slices workgroups per point, each over its share of the point's observation list
//                          (view, camera and observed corner copied into the list's order: coalesced reads)
//            camera block: one workgroup per group of views of the camera (K2's groups)
//            k_inner_sum:  the partial sums of a camera or a point, added in a fixed order
//            (view-sharded runs: the sums of the shared blocks are all-reduced here, k_inner_xchg)
//   solve  one thread per block: Jacobi scales (first round), gradient test, damped Cholesky solve with the
//          invalid-step loop, parameter-tolerance test, candidate block
//   cost   the candidate's cost over the same observations
//   decide one thread per block: function-tolerance test, step quality, accept (the block's parameters are
//          overwritten in place) or reject, radius update; counts the blocks that go on
// Blocks of a set share no observation, so what one block reads of the others never changes during the set.
// All sums are taken in a fixed order (no atomics on data; the count of running blocks is an atomic sum of ones).
#pragma once
#include "calib_kernels.h"

namespace cvxi {

using cvxk::Intr;
using cvxk::load_intr;
using cvxk::RT_STRIDE;

constexpr int IB_STRIDE = 56;    // per block: upper triangle of H (<= 45) | g at 45 (<= 9) | cost at 54
constexpr int IB_G = 45, IB_COST = 54;
constexpr int PT_SLOT = 16;      // per point slice: H (6) | g (3) | cost
constexpr int IB_STATE = 16;     // per block: cost, radius, decrease factor, model change, first, pad, s2[9]
enum { ST_COST = 0, ST_RADIUS, ST_DECREASE, ST_MODEL, ST_FIRST, ST_PAD, ST_S2 };
enum { IB_CAM = 0, IB_POINT = 1, IB_POSE = 2 };
enum { PH_LIN = 0, PH_CAND = 1, PH_DONE = 2, PH_RETRY = 3 };
// flag[4 * id + ...]
enum { FL_PHASE = 0, FL_ITERS, FL_INVALID, FL_TERM };

struct InnerArgs {
    // problem
    const int *view_off, *cam_of_view, *point_id, *view_id;
    const double *obs_x, *obs_y;
    const int *pt_off, *pt_obs;              // observations by point
    const int *pt_view;                      // in the order of pt_obs: the observation's view, its camera,
    const int *pt_cam;
    const double *pt_ox, *pt_oy;             // and the observed corner: a point block reads coalesced
    const int *grp_v0, *glist, *glist_off;   // K2's view groups (one camera each) and the per-camera lists of them
    // the point being improved (the trust-region candidate), and its rotation tables
    double *intr, *points, *poses;
    const double *rt;
    // per block (ids: camera c -> c, point p -> C + p, pose v -> C + P + v)
    const int *blk_group;
    double *H, *st, *xc, *cc;
    int *flag;
    double *cam_slab;                        // [view groups][IB_STRIDE]
    double *pt_slab;                         // [points][pt_slices][PT_SLOT]
    int pt_slices;                           // workgroups per point
    double *active;                          // number of blocks that go on after `decide` (a double: it is all-reduced)
    int C, P, V, ncf, fix_k3;
    int cur;                                 // the set being minimised
};

__device__ __forceinline__ int tri(int n, int a, int b) { return a * n - a * (a - 1) / 2 + (b - a); }   // a <= b

// Sum over the wave, valid on WAVE_SUM_LANE (and its whole row of 16).  Data-parallel-primitive moves instead of
// ds_bpermute shuffles: lane ^ 1, lane ^ 2 inside a quad, the mirrored half row, the mirrored row, then lane 15 of
// rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2 / 3 (masked-off rows add zero).  A fixed order like any other.
constexpr int WAVE_SUM_LANE = 63;
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_take<0xB1, 0xF>(v);        // quad_perm [1,0,3,2]
    v += dpp_take<0x4E, 0xF>(v);        // quad_perm [2,3,0,1]
    v += dpp_take<0x141, 0xF>(v);       // row_half_mirror
    v += dpp_take<0x140, 0xF>(v);       // row_mirror
    v += dpp_take<0x142, 0xA>(v);       // row_bcast:15 into rows 1 and 3
    v += dpp_take<0x143, 0xC>(v);       // row_bcast:31 into rows 2 and 3
    return v;
}

// sum over the workgroup in a fixed order; the result is valid on thread 0.  sh: one double per wave
template <int WAVES>
__device__ __forceinline__ double group_sum(double v, double *sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == WAVE_SUM_LANE) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < WAVES; ++w) t += sh[w];
    return t;
}

// N sums over the workgroup at once, fixed order: on return thread t < N holds the t-th sum in acc[0].
// sh: WAVES * N doubles
template <int N, int WAVES>
__device__ __forceinline__ void group_sum_n(double (&acc)[N], double *sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = wave_sum(acc[i]);
    if (lane == WAVE_SUM_LANE) {
#pragma unroll
        for (int i = 0; i < N; ++i) sh[wave * N + i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < N) {
        double t = 0.0;
        for (int w = 0; w < WAVES; ++w) t += sh[w * N + threadIdx.x];
        acc[0] = t;
    }
}

__device__ __forceinline__ bool block_runs(const InnerArgs &a, int id, int want) {
    return a.blk_group[id] == a.cur && a.flag[4 * id + FL_PHASE] == want;
}

// ---------------------------------------------------------------------------------------------
// pose blocks: one wave per view.  MODE 0: H (21) | g (6) | cost at the block's pose; MODE 1: cost at the candidate
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(256) k_inner_pose(InnerArgs a) {        // (held to 168 registers for a third wave: spills, 70 us)
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= a.V) return;
    const int id = a.C + a.P + v;
    if (!block_runs(a, id, MODE == 0 ? PH_LIN : PH_CAND)) return;
    // the block's rotation table, made by every lane for itself (by lane 0 through the wave's LDS row: 52 instead of 42 us)
    double rt[RT_STRIDE];
    cvxk::pose_table(MODE == 0 ? a.poses + 6 * (size_t)v : a.xc + 9 * (size_t)id, rt);
    const Intr in = load_intr(a.intr + 9 * a.cam_of_view[v]);
    double acc[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) acc[i] = 0.0;
    for (int o = a.view_off[v] + lane; o < a.view_off[v + 1]; o += 64) {
        const double *X = a.points + 3 * a.point_id[o];
        double r0, r1;
        if (MODE == 0) {
            double Jp[12], Jc[18], Jx[6];
            cvxk::reproj_jacobian(in, (const double *)rt, X[0], X[1], X[2], a.obs_x[o], a.obs_y[o], r0, r1, Jp, Jc, Jx);
            int k = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = i; j < 6; ++j) acc[k++] += Jp[i] * Jp[j] + Jp[6 + i] * Jp[6 + j];
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[21 + i] += Jp[i] * r0 + Jp[6 + i] * r1;
        } else {
            cvxk::reproj_residual(in, (const double *)rt, X[0], X[1], X[2], a.obs_x[o], a.obs_y[o], r0, r1);
        }
        acc[27] += r0 * r0 + r1 * r1;
    }
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 28; ++i) acc[i] = wave_sum(acc[i]);
        if (lane == WAVE_SUM_LANE) {
            double *H = a.H + (size_t)id * IB_STRIDE;
            for (int i = 0; i < 21; ++i) H[i] = acc[i];
            for (int i = 0; i < 6; ++i) H[IB_G + i] = acc[21 + i];
            H[IB_COST] = 0.5 * acc[27];
        }
    } else {
        const double c = wave_sum(acc[27]);
        if (lane == WAVE_SUM_LANE) a.cc[id] = 0.5 * c;
    }
}

// ---------------------------------------------------------------------------------------------
// point blocks: pt_slices workgroups per point, each over its share of the point's observation list -> pt_slab;
// k_inner_sum adds the slices in order
// ---------------------------------------------------------------------------------------------
constexpr int IP_THREADS = 256;
template <int MODE>
__global__ void __launch_bounds__(IP_THREADS) k_inner_point(InnerArgs a) {
    constexpr int NA = MODE == 0 ? 10 : 1;
    __shared__ double sh[(IP_THREADS / 64) * NA];
    const int p = blockIdx.x, sl = blockIdx.y, id = a.C + p;
    if (!block_runs(a, id, MODE == 0 ? PH_LIN : PH_CAND)) return;        // the same for the whole workgroup
    const double *Xp = MODE == 0 ? a.points + 3 * p : a.xc + 9 * (size_t)id;
    const double X0 = Xp[0], X1 = Xp[1], X2 = Xp[2];
    double acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = 0.0;
    const long long off = a.pt_off[p], len = a.pt_off[p + 1] - off;
    const int k0 = (int)(off + len * sl / a.pt_slices), k1 = (int)(off + len * (sl + 1) / a.pt_slices);
    for (int k = k0 + threadIdx.x; k < k1; k += IP_THREADS) {
        const int v = a.pt_view[k];
        const double ox = a.pt_ox[k], oy = a.pt_oy[k];
        const Intr in = load_intr(a.intr + 9 * a.pt_cam[k]);
        const double *rt = a.rt + (size_t)v * RT_STRIDE;
        double r0, r1;
        if (MODE == 0) {
            double Jp[12], Jc[18], Jx[6];
            cvxk::reproj_jacobian(in, rt, X0, X1, X2, ox, oy, r0, r1, Jp, Jc, Jx);
            int q = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = i; j < 3; ++j) acc[q++] += Jx[i] * Jx[j] + Jx[3 + i] * Jx[3 + j];
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[6 + i] += Jx[i] * r0 + Jx[3 + i] * r1;
        } else {
            cvxk::reproj_residual(in, rt, X0, X1, X2, ox, oy, r0, r1);
        }
        acc[NA - 1] += r0 * r0 + r1 * r1;
    }
    group_sum_n<NA, IP_THREADS / 64>(acc, sh);
    if (threadIdx.x < NA) a.pt_slab[((size_t)p * a.pt_slices + sl) * PT_SLOT + (MODE == 0 ? threadIdx.x : 9)] = acc[0];
}

// ---------------------------------------------------------------------------------------------
// camera blocks: one workgroup per view group (all of one camera) -> cam_slab; k_inner_sum adds the groups of a
// camera in list order.  H is the full 9 x 9 triangle; `solve` takes the free columns out of it.
// ---------------------------------------------------------------------------------------------
constexpr int IC_THREADS = 256;
template <int MODE>
__global__ void __launch_bounds__(IC_THREADS) k_inner_cam(InnerArgs a) {
    constexpr int NA = MODE == 0 ? 55 : 1;
    __shared__ double sh[(IC_THREADS / 64) * NA];
    const int g = blockIdx.x, v0 = a.grp_v0[g], v1 = a.grp_v0[g + 1];
    const int c = a.cam_of_view[v0];
    if (!block_runs(a, c, MODE == 0 ? PH_LIN : PH_CAND)) return;
    const Intr in = load_intr(MODE == 0 ? a.intr + 9 * c : a.xc + 9 * (size_t)c);
    double acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = 0.0;
    for (int o = a.view_off[v0] + threadIdx.x; o < a.view_off[v1]; o += IC_THREADS) {
        const int v = a.view_id[o];
        const double *X = a.points + 3 * a.point_id[o];
        const double *rt = a.rt + (size_t)v * RT_STRIDE;
        double r0, r1;
        if (MODE == 0) {
            double Jp[12], Jc[18], Jx[6];
            cvxk::reproj_jacobian(in, rt, X[0], X[1], X[2], a.obs_x[o], a.obs_y[o], r0, r1, Jp, Jc, Jx);
            int q = 0;
#pragma unroll
            for (int i = 0; i < 9; ++i)
#pragma unroll
                for (int j = i; j < 9; ++j) acc[q++] += Jc[i] * Jc[j] + Jc[9 + i] * Jc[9 + j];
#pragma unroll
            for (int i = 0; i < 9; ++i) acc[45 + i] += Jc[i] * r0 + Jc[9 + i] * r1;
        } else {
            cvxk::reproj_residual(in, rt, X[0], X[1], X[2], a.obs_x[o], a.obs_y[o], r0, r1);
        }
        acc[NA - 1] += r0 * r0 + r1 * r1;
    }
    group_sum_n<NA, IC_THREADS / 64>(acc, sh);
    if (threadIdx.x < NA) a.cam_slab[(size_t)g * IB_STRIDE + (MODE == 0 ? threadIdx.x : IB_COST)] = acc[0];
}

// the partial sums of a shared block, added in list order (sixteen interleaved running sums, then those in order):
// workgroup b < C: camera b over its view groups; workgroup C + p: point p over its slices
constexpr int IS_THREADS = 1024, IS_WAYS = IS_THREADS / 64;
template <int MODE>
__global__ void __launch_bounds__(IS_THREADS) k_inner_sum(InnerArgs a) {
    __shared__ double sh[IS_WAYS][64];
    const int id = blockIdx.x, k = threadIdx.x & 63, j = threadIdx.x >> 6;
    if (!block_runs(a, id, MODE == 0 ? PH_LIN : PH_CAND)) return;
    const bool cam = id < a.C;
    const int nval = cam ? 55 : 10, last = nval - 1;
    double s = 0.0;
    if (MODE == 0 ? k < nval : k == last) {
        if (cam) {
#pragma unroll 4
            for (int i = a.glist_off[id] + j; i < a.glist_off[id + 1]; i += IS_WAYS) s += a.cam_slab[(size_t)a.glist[i] * IB_STRIDE + k];
        } else {
            const double *src = a.pt_slab + (size_t)(id - a.C) * a.pt_slices * PT_SLOT;
#pragma unroll 4
            for (int i = j; i < a.pt_slices; i += IS_WAYS) s += src[(size_t)i * PT_SLOT + k];
        }
    }
    sh[j][k] = s;
    __syncthreads();
    if (j != 0 || (MODE == 0 ? k >= nval : k != last)) return;
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < IS_WAYS; ++w) t += sh[w][k];
    if (MODE == 1) { a.cc[id] = 0.5 * t; return; }
    const int dst = cam ? k : (k < 6 ? k : k < 9 ? IB_G + (k - 6) : IB_COST);
    a.H[(size_t)id * IB_STRIDE + dst] = k == last ? 0.5 * t : t;
}

// ---------------------------------------------------------------------------------------------
// per-block LM control (TrustRegionMinimizer + LevenbergMarquardtStrategy with a default Minimizer::Options,
// as CoordinateDescentMinimizer::Solve sets them up)
// ---------------------------------------------------------------------------------------------
struct IbShape { int type, nf, ng; double *x; };
__device__ __forceinline__ IbShape ib_shape(const InnerArgs &a, int id) {
    IbShape s;
    if (id < a.C) { s.type = IB_CAM; s.nf = a.ncf; s.ng = 9; s.x = a.intr + 9 * (size_t)id; }
    else if (id < a.C + a.P) { s.type = IB_POINT; s.nf = s.ng = 3; s.x = a.points + 3 * (size_t)(id - a.C); }
    else { s.type = IB_POSE; s.nf = s.ng = 6; s.x = a.poses + 6 * (size_t)(id - a.C - a.P); }
    return s;
}
// free column k of a block -> its index in the block (the intrinsics block skips k3 when it is held)
__device__ __forceinline__ int ib_col(const InnerArgs &a, int type, int k) { return (type == IB_CAM && a.fix_k3 && k >= 6) ? k + 1 : k; }

__global__ void k_inner_reset(InnerArgs a) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id == 0) *a.active = 0.0;
    if (id >= a.C + a.P + a.V || a.blk_group[id] != a.cur) return;
    a.flag[4 * id + FL_PHASE] = PH_LIN; a.flag[4 * id + FL_ITERS] = 0; a.flag[4 * id + FL_INVALID] = 0; a.flag[4 * id + FL_TERM] = 0;
    double *st = a.st + (size_t)id * IB_STATE;
    st[ST_RADIUS] = 1e4; st[ST_DECREASE] = 2.0; st[ST_FIRST] = 1.0;
}

// One LM iteration's linear part for a block of NF free columns in NG parameters (SKIP6: the intrinsics block with
// k3 held -- free column k >= 6 is parameter k + 1).  Sizes are compile-time so that the small dense algebra stays in
// registers; blocks of one type are contiguous in id, so a wave runs one instantiation (but for the two at a seam).
template <int NF, int NG, bool SKIP6>
__device__ __forceinline__ void ib_solve_block(const InnerArgs &a, int id, int phase, double *xp) {
    auto col = [](int k) { return (SKIP6 && k >= 6) ? k + 1 : k; };
    int *fl = a.flag + 4 * id;
    const double *H = a.H + (size_t)id * IB_STRIDE;
    double *st = a.st + (size_t)id * IB_STATE;
    double g[NF], diag[NF], s2[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        g[i] = H[IB_G + col(i)];
        diag[i] = H[tri(NG, col(i), col(i))];
    }
    if (phase == PH_LIN) {
        if (st[ST_FIRST] != 0.0) {
#pragma unroll
            for (int i = 0; i < NF; ++i) { const double s = 1.0 / (1.0 + sqrt(diag[i])); st[ST_S2 + i] = s * s; }
            st[ST_FIRST] = 0.0;
        }
        st[ST_COST] = H[IB_COST];
        double gmax = 0.0;
#pragma unroll
        for (int i = 0; i < NF; ++i) gmax = fmax(gmax, fabs(g[i]));
        if (gmax <= 1e-10) { fl[FL_TERM] = 1; fl[FL_PHASE] = PH_DONE; return; }
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) s2[i] = st[ST_S2 + i];
    double radius = st[ST_RADIUS];
    int iters = fl[FL_ITERS], invalid = fl[FL_INVALID];
    double x[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) x[i] = xp[i];
    for (;;) {
        if (iters >= 50) { fl[FL_TERM] = 4; fl[FL_PHASE] = PH_DONE; break; }
        ++iters;
        double D[NF], L[NF * (NF + 1) / 2], d[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const double t = fmin(fmax(s2[i] * diag[i], 1e-6), 1e32);
            D[i] = t / (s2[i] * radius);
        }
        // Cholesky of H_free + D (lower triangle, row-major packed: L[i(i+1)/2 + j])
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double t = H[tri(NG, col(j), col(i))] + (i == j ? D[i] : 0.0);
#pragma unroll
                for (int k = 0; k < j; ++k) t -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
                if (i == j) {
                    if (!(t > 0.0) || !(t < 1.7e308)) { ok = false; t = 1.0; }
                    L[i * (i + 1) / 2 + i] = sqrt(t);
                } else {
                    L[i * (i + 1) / 2 + j] = t / L[j * (j + 1) / 2 + j];
                }
            }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            double t = -g[i];
#pragma unroll
            for (int k = 0; k < i; ++k) t -= L[i * (i + 1) / 2 + k] * d[k];
            d[i] = t / L[i * (i + 1) / 2 + i];
        }
#pragma unroll
        for (int i = NF - 1; i >= 0; --i) {
            double t = d[i];
#pragma unroll
            for (int k = i + 1; k < NF; ++k) t -= L[k * (k + 1) / 2 + i] * d[k];
            d[i] = t / L[i * (i + 1) / 2 + i];
        }
        double mc = 0.0;
#pragma unroll
        for (int i = 0; i < NF; ++i) mc += d[i] * (D[i] * d[i] - g[i]);
        mc *= 0.5;
        if (!ok || !(mc > 0.0)) {
            if (++invalid >= 5) { fl[FL_TERM] = 6; fl[FL_PHASE] = PH_DONE; break; }
            radius *= 0.5;
            continue;
        }
        invalid = 0;
        double xc[NG];
#pragma unroll
        for (int i = 0; i < NG; ++i) xc[i] = x[i];
#pragma unroll
        for (int i = 0; i < NF; ++i) xc[col(i)] += d[i];
        double sn = 0.0, xn = 0.0;
#pragma unroll
        for (int i = 0; i < NG; ++i) { const double e = x[i] - xc[i]; sn += e * e; xn += x[i] * x[i]; }
        sn = sqrt(sn); xn = sqrt(xn);
        if (sn <= 1e-8 * (xn + 1e-8)) { fl[FL_TERM] = 2; fl[FL_PHASE] = PH_DONE; break; }
#pragma unroll
        for (int i = 0; i < NG; ++i) a.xc[9 * (size_t)id + i] = xc[i];
        st[ST_MODEL] = mc;
        fl[FL_PHASE] = PH_CAND;
        break;
    }
    st[ST_RADIUS] = radius;
    fl[FL_ITERS] = iters; fl[FL_INVALID] = invalid;
}

__global__ void __launch_bounds__(64) k_inner_solve(InnerArgs a) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id == 0) *a.active = 0.0;       // counted anew by this round's `decide`
    if (id >= a.C + a.P + a.V || a.blk_group[id] != a.cur) return;
    const int phase = a.flag[4 * id + FL_PHASE];
    if (phase != PH_LIN && phase != PH_RETRY) return;
    const IbShape sh = ib_shape(a, id);
    if (sh.type == IB_POSE) ib_solve_block<6, 6, false>(a, id, phase, sh.x);
    else if (sh.type == IB_POINT) ib_solve_block<3, 3, false>(a, id, phase, sh.x);
    else if (a.ncf == 8) ib_solve_block<8, 9, true>(a, id, phase, sh.x);
    else ib_solve_block<9, 9, false>(a, id, phase, sh.x);
}

__global__ void __launch_bounds__(64) k_inner_decide(InnerArgs a) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= a.C + a.P + a.V || a.blk_group[id] != a.cur) return;
    int *fl = a.flag + 4 * id;
    if (fl[FL_PHASE] != PH_CAND) return;
    double *st = a.st + (size_t)id * IB_STATE;
    const double cost = st[ST_COST], cand = a.cc[id];
    if (fabs(cost - cand) <= 1e-6 * cost) { fl[FL_TERM] = 3; fl[FL_PHASE] = PH_DONE; return; }
    const double rho = (cost - cand) / st[ST_MODEL];
    if (rho > 1e-3) {
        const IbShape sh = ib_shape(a, id);
        for (int i = 0; i < sh.ng; ++i) sh.x[i] = a.xc[9 * (size_t)id + i];
        const double t = 2.0 * rho - 1.0;
        double f = 1.0 - t * t * t;
        if (f < 1.0 / 3.0) f = 1.0 / 3.0;
        double radius = st[ST_RADIUS] / f;
        if (radius > 1e32) radius = 1e32;       // TrustRegionStrategy::Options::max_radius (not Solver::Options' 1e16)
        st[ST_RADIUS] = radius; st[ST_DECREASE] = 2.0;
        fl[FL_PHASE] = PH_LIN;
    } else {
        const double radius = st[ST_RADIUS] / st[ST_DECREASE];
        st[ST_RADIUS] = radius; st[ST_DECREASE] *= 2.0;
        if (radius < 1e-32) { fl[FL_TERM] = 5; fl[FL_PHASE] = PH_DONE; return; }
        fl[FL_PHASE] = PH_RETRY;
    }
    atomicAdd(a.active, 1.0);           // integers in a double: exact in any order
}

// the round's count of blocks that go on, into host-mapped memory behind k_inner_decide: the host polls the sequence number
// instead of waiting for a copy (a stream synchronisation wakes up 10-20 us after the fact, and a pass has six of them)
__global__ void __launch_bounds__(64) k_inner_publish(const double *active, double *host, double seq) {
    if (threadIdx.x != 0) return;
    host[0] = *active;
    __threadfence_system();
    host[1] = seq;
    __threadfence_system();
}

// View-sharded runs: a shared block's sums are partial on every rank.  DIR 0 lays the rows of the running blocks
// (zeros for all others) into the exchange buffer X, DIR 1 takes the all-reduced rows back.  The LM state of a
// shared block is replicated -- every rank takes the same decisions from the same sums -- so "running" is the same
// everywhere.  MODE 0: rows of H | g | cost; MODE 1: candidate costs.
template <int MODE, int DIR>
__global__ void __launch_bounds__(64) k_inner_xchg(InnerArgs a, double *X) {
    const int id = blockIdx.x, k = threadIdx.x;
    const bool run = block_runs(a, id, MODE == 0 ? PH_LIN : PH_CAND);
    if (MODE == 0) {
        if (k >= IB_STRIDE) return;
        const size_t e = (size_t)id * IB_STRIDE + k;
        if (DIR == 0) X[e] = run ? a.H[e] : 0.0;
        else if (run) a.H[e] = X[e];
    } else {
        if (k != 0) return;
        if (DIR == 0) X[id] = run ? a.cc[id] : 0.0;
        else if (run) a.cc[id] = X[id];
    }
}

// |x - y|^2 over the parameter arrays (the step norm of the outer iteration after the pass), one workgroup
__global__ void __launch_bounds__(1024) k_inner_step_norm(const double *i0, const double *i1, long long ni, const double *p0,
                                                          const double *p1, long long np, const double *v0, const double *v1,
                                                          long long nv, double *out) {
    __shared__ double sh[16];
    double s = 0.0;
    for (long long i = threadIdx.x; i < ni; i += 1024) { const double e = i0[i] - i1[i]; s += e * e; }
    for (long long i = threadIdx.x; i < np; i += 1024) { const double e = p0[i] - p1[i]; s += e * e; }
    const double shared = group_sum<16>(s, sh);
    s = 0.0;
    for (long long i = threadIdx.x; i < nv; i += 1024) { const double e = v0[i] - v1[i]; s += e * e; }
    const double poses = group_sum<16>(s, sh);
    if (threadIdx.x == 0) { out[0] = shared; out[1] = poses; }      // shared blocks (replicated) | this rank's poses
}

}  // namespace cvxi
