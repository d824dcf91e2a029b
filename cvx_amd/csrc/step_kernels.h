// K6 + K4 of an LM linear step, one wave per view, with the step's bookkeeping folded in.
//   K6  t = Z_v * ds_emb[cam]  (the step embedded in Z's column order, 1 in the gradient column)  ->  dp = -L_v^-T t;
//       candidate pose = pose + dp;  out_view[v][1..3] = dp^T (D dp - g), |dp|^2, |pose|^2
//   K4  the candidate's cost over the view's own observations, with the candidate rotation held in registers (same
//       expressions as k_pose_prepare, so the value equals what k_cost computes from the tables): out_view[v][0]
//       (one record of four doubles per view: the final sums read it as two 16-byte loads)
// What a one-workgroup k_step_finish (solve_kernels.h) used to make in front of this kernel -- ds_emb, the candidate
// intrinsics and points -- is a handful of adds on the solution x of the reduced system, so every wave makes what it
// needs from x itself (same single additions: same values), and workgroup 0 does the rest of k_step_finish's work on
// the side: the candidate arrays the NEXT kernels read, and the scalars of the step.  One launch and 6 us less per
// iteration.  Reference: /root/reference/modules/calib/src/calibration.cpp:420-433 (after the linear solve the
// candidate point x + delta is evaluated and compared with the model's prediction).
// Included only by calib.hip, after calib_kernels.h and solve_kernels.h.
#pragma once
#include "calib_kernels.h"
#include "solve_kernels.h"

namespace cvxk {

// entry `col` of camera `cam`'s embedded step (k_step_finish's ds_emb[cam][col]), without a branch: one unconditional
// load of x at a clamped index and two selects, so that a wave's six of them are in flight together
__device__ __forceinline__ double step_emb(const cvxs::FinishArgs &a, const double *__restrict__ x, int cam, int col) {
    const bool is_pt = a.pts_free && col < 3 * a.P;
    const int kk = col - a.camcol0;
    const bool is_cam = !is_pt && a.cam_free && kk >= 0 && kk < 9;
    const bool use_cam = is_cam && !(a.fix_k3 && kk == 6);
    const int cb = (col >= a.P) + (col >= 2 * a.P);      // Z's point columns are coordinate-major: column b P + p is unknown 3 p + b
    const int idx = is_pt ? a.C * a.ncf + 3 * (col - cb * a.P) + cb : use_cam ? cam * a.ncf + ((a.fix_k3 && kk > 6) ? kk - 1 : kk) : 0;
    const double v = x[idx];      // x holds at least one element
    return (is_pt || use_cam) ? v : (!is_cam && col == a.ycol) ? 1.0 : 0.0;
}

constexpr int BSC_THREADS = 256;
// (five waves per SIMD: 96 registers, one spilled -- 40.1 us against 41.7 at the four the compiler picks on its own; six
//  waves spill 21 registers, 50.5 us; the view's first observations loaded ahead of the pass over Z: 42.4-44.2 us)

__global__ void __launch_bounds__(BSC_THREADS, 5) k_backsubst_cost_step(const double *__restrict__ Z, const double *__restrict__ Lfac,
                                                                     const double *__restrict__ Dv, const double *__restrict__ gv,
                                                                     const int *__restrict__ cam_of_view,
                                                                     const double *__restrict__ poses, double *__restrict__ poses_cand,
                                                                     double *__restrict__ dposes, double *__restrict__ out_view, int V,
                                                                     const int *__restrict__ view_off,
                                                                     const int *__restrict__ point_id, const double *__restrict__ obs_x,
                                                                     const double *__restrict__ obs_y, cvxs::FinishArgs f) {
    extern __shared__ double sm[];
    if (f.st) {
        // a speculated step: which buffer is current is the record's to say (uniform loads: the whole grid agrees)
        const uniform_lm st = as_uniform(f.st);
        if (!st->active) return;
        const long long cur = st->cur, nxt = 1 - cur;
        f.intr_cand = const_cast<double *>(f.intr) + nxt * f.intr_step; f.intr += cur * f.intr_step;
        f.points_cand = const_cast<double *>(f.points) + nxt * f.points_step; f.points += cur * f.points_step;
        poses_cand = const_cast<double *>(poses) + nxt * f.poses_step; poses += cur * f.poses_step;
    }
    const double *__restrict__ x = f.delta_s;      // the solution of the reduced system (last k_chol_step launch)
    if (blockIdx.x == 0) {
        // k_step_finish's part, before any wave of the workgroup can leave
        double *xs = sm, *red = xs + ((f.n + 7) & ~7);
        for (int i = threadIdx.x; i < f.n; i += BSC_THREADS) xs[i] = x[i];
        __syncthreads();
        cvxs::finish_tail<BSC_THREADS>(f, xs, red);
    }
    // the candidate points (k_step_finish's points_cand), once per workgroup, in LDS behind the finish area
    double *Xc = sm + ((f.n + 7) & ~7) + 64;
    for (int q = threadIdx.x; q < 3 * f.P; q += BSC_THREADS) {
        const double dx = x[f.pts_free ? f.C * f.ncf + q : 0];
        const double val = f.points[q];
        Xc[q] = f.pts_free ? val + dx : val;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= V) return;
    const int cam = cam_of_view[v];
    const int ZC = f.ZC;
    const double *Zv = Z + (size_t)6 * v * ZC;
    // the view's observations do not depend on the step: fetch them first
    const int o0 = view_off[v], o1 = view_off[v + 1];
    double t[6] = {0, 0, 0, 0, 0, 0};
    for (int c = 2 * lane; c < ZC; c += 128) {       // ZC is a multiple of 16: 16-byte loads, 1 KB per wave instruction
        const double dx = step_emb(f, x, cam, c), dy = step_emb(f, x, cam, c + 1);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double2 z = nt_load2(Zv + (size_t)i * ZC + c);      // Z's last reader: non-temporal (40.3 -> 36.9 us)
            t[i] += z.x * dx + z.y * dy;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t[i] += __shfl_xor(t[i], off, 64);
    // every lane carries the (wave-uniform) substitution: the candidate pose is needed by all of them
    const double *Lf = Lfac + (size_t)21 * v;         // off-diagonal L, diagonal slots 1 / L_ii
    double xp[6], pc[6];
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = t[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= Lf[TRI(k, i)] * xp[k];
        xp[i] = s * Lf[TRI(i, i)];
    }
    double m = 0.0, s2 = 0.0, x2 = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double dp = -xp[i];
        const double p = poses[6 * v + i];
        pc[i] = p + dp;
        m += dp * (Dv[6 * v + i] * dp - gv[6 * v + i]);
        s2 += dp * dp;
        x2 += p * p;
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            poses_cand[6 * v + i] = pc[i];
            if (dposes) dposes[6 * v + i] = -xp[i];
        }
        out_view[4 * (size_t)v + 1] = m;
        out_view[4 * (size_t)v + 2] = s2;
        out_view[4 * (size_t)v + 3] = x2;
    }
    // rotation of the candidate pose: the R part of k_pose_prepare, in registers
    double rtl[RT_STRIDE];
    {
        const double w0 = pc[0], w1 = pc[1], w2 = pc[2];
        const double th2 = w0 * w0 + w1 * w1 + w2 * w2;
        if (th2 > 2.220446049250313e-16) {
            const double th = sqrt(th2), c = cos(th), s = sin(th), ti = 1.0 / th, omc = 1.0 - c;
            const double a[3] = {w0 * ti, w1 * ti, w2 * ti};
            const double K[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k) rtl[i * 3 + k] = (i == k ? c : 0.0) + s * K[i * 3 + k] + omc * a[i] * a[k];
        } else {
            const double Rs[9] = {1, -w2, w1, w2, 1, -w0, -w1, w0, 1};
#pragma unroll
            for (int i = 0; i < 9; ++i) rtl[i] = Rs[i];
        }
        rtl[36] = pc[3]; rtl[37] = pc[4]; rtl[38] = pc[5];
    }
    // the camera's candidate intrinsics (k_step_finish's intr_cand)
    double ic[9];
#pragma unroll
    for (int kk = 0; kk < 9; ++kk) {
        const bool moves = f.cam_free && !(f.fix_k3 && kk == 6);
        const double dx = x[moves ? cam * f.ncf + ((f.fix_k3 && kk > 6) ? kk - 1 : kk) : 0];
        const double val = f.intr[9 * cam + kk];
        ic[kk] = moves ? val + dx : val;
    }
    const Intr in = load_intr(ic);
    double acc = 0.0;
    for (int o = o0 + lane; o < o1; o += 64) {
        const double *X = Xc + 3 * point_id[o];
        double r0, r1;
        reproj_residual(in, rtl, X[0], X[1], X[2], obs_x[o], obs_y[o], r0, r1);
        acc += r0 * r0 + r1 * r1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) out_view[4 * (size_t)v] = acc;
}

}  // namespace cvxk
