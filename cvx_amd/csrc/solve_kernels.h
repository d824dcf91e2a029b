// HIP kernels (gfx950 / CDNA4, wave64) of the damped reduced solve of one LM linear step:
//   (S + D) delta_s = -g_red      with S the ns x ns Schur complement assembled by k_assemble.
// Replaces, on the device, what Ceres does after the Schur eliminator for
// /root/reference/modules/calib/src/calibration.cpp:420-433 (LevenbergMarquardtStrategy: diagonal
// D = clamp(diag) / radius, then a linear solve) and what levmar does at
// /root/reference/modules/util/src/3rdparty/levmar/lm_core.c:290-305 (augment the diagonal by mu,
// solve); the factorisation is a tiled right-looking Cholesky (32 x 32 tiles, one launch per tile
// column, trailing update on fp64 MFMA), the right-hand side rides along as an extra matrix row so
// the forward substitution costs nothing, and one workgroup finishes: back substitution, candidate
// shared parameters, the embedded step the pose back-substitution multiplies by, and the scalars of
// the step (model change, norms).  Nothing in the iteration returns to the host in between.
// Included only by calib.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace cvxs {

typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int TB = 32;

// -----------------------------------------------------------------------------------------
// A (npad x npad, row-major) = [ S + D  .  ] rows/cols < n
//                              [ -g_red 1 ] row n (the right-hand side), identity beyond
// D_j = clamp(s_j^2 d_j, min_diag, max_diag) * inv_radius / s_j^2 + mu,  d_j = U_jj
// -----------------------------------------------------------------------------------------
struct DampArgs {
    const double *E;    // S (n*n) | gred (n) | gs (n) | diagU (n)
    const double *s2;   // Jacobi scales squared
    int n, npad;
    double inv_radius, mu, min_diag, max_diag;
    double *A, *Ds;
};

__global__ void __launch_bounds__(256) k_damp_prepare(DampArgs a) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)a.npad * a.npad) return;
    const int i = (int)(t / a.npad), j = (int)(t % a.npad);
    const int n = a.n;
    double v = 0.0;
    if (i < n && j < n) {
        v = a.E[(size_t)i * n + j];
        if (i == j) {
            const double s2 = a.s2[i];
            double c = s2 * a.E[(size_t)n * n + 2 * n + i];
            c = fmin(fmax(c, a.min_diag), a.max_diag);
            c = c * a.inv_radius / s2 + a.mu;
            a.Ds[i] = c;
            v += c;
        }
    } else if (i == n && j < n) {
        v = -a.E[(size_t)n * n + j];
    } else if (i == j) {
        v = 1.0;
    }
    a.A[t] = v;
}

// -----------------------------------------------------------------------------------------
// One tile column k of the factorisation.  Block b owns trailing tile (ti, tj), k < tj <= ti:
// it factors the diagonal tile A_kk (every block redundantly: 32 columns, one wave), solves its
// two panel tiles L_ik = A_ik L_kk^-T, L_jk, and updates A_ij -= L_ik L_jk^T with MFMA.  Panel k of
// A is only read in this launch; factor tiles are written to the separate array L (block 0 the
// diagonal tile, the blocks with tj == k + 1 their L_ik), so no block reads what another writes.
// Rows with global index >= n (the right-hand-side row and the padding) take diagonal 1.
// -----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_chol_step(double *A, double *L, int ld, int n, int k, int nT, int *fail) {
    __shared__ double D[TB][TB + 1];
    __shared__ double Pi[TB][TB + 1], Pj[TB][TB + 1];
    __shared__ double invd[TB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntr = nT - 1 - k;
    int ti = -1, tj = -1;
    if (ntr > 0) {
        int rem = blockIdx.x, r = 0;
        while (rem > r) { rem -= r + 1; ++r; }
        ti = k + 1 + r; tj = k + 1 + rem;
    }
    for (int e = tid; e < TB * TB; e += 256) {
        const int r = e >> 5, c = e & 31;
        D[r][c] = A[(size_t)(k * TB + r) * ld + k * TB + c];
        if (ti >= 0) {
            Pi[r][c] = A[(size_t)(ti * TB + r) * ld + k * TB + c];
            Pj[r][c] = A[(size_t)(tj * TB + r) * ld + k * TB + c];
        }
    }
    __syncthreads();
    if (wave == 0) {
        const int r = lane & 31;
        double row[TB];
        bool bad = false;
#pragma unroll
        for (int c = 0; c < TB; ++c) row[c] = D[r][c];
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            // left-looking column j: t = A[r][j] - sum_{m<j} L[r][m] L[j][m]
            double t0 = row[j], t1 = 0.0;
#pragma unroll
            for (int m = 0; m < j; ++m) {
                const double ljm = D[j][m];
                if (m & 1) t1 += row[m] * ljm; else t0 -= row[m] * ljm;
            }
            const double t = t0 - t1;
            const double d = __shfl(t, j, 64);
            const bool special = (k * TB + j) >= n;
            bad = bad || (!special && !(d > 0.0 && d < 1.7e308));
            const double l = special ? 1.0 : sqrt(d);
            const double inv = 1.0 / l;
            row[j] = (r == j) ? l : t * inv;
            __builtin_amdgcn_wave_barrier();
            D[r][j] = row[j];
            if (r == j) invd[j] = inv;
            __builtin_amdgcn_wave_barrier();
        }
        if (ti >= 0) {
            // panel rows: lanes 0..31 solve a row of A_ik, lanes 32..63 a row of A_jk:  x L_kk^T = a
            double (*Pm)[TB + 1] = (lane < 32) ? Pi : Pj;
            double x[TB];
#pragma unroll
            for (int c = 0; c < TB; ++c) x[c] = Pm[r][c];
#pragma unroll
            for (int c = 0; c < TB; ++c) {
                x[c] *= invd[c];
#pragma unroll
                for (int m = c + 1; m < TB; ++m) x[m] -= x[c] * D[m][c];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < TB; ++c) Pm[r][c] = x[c];
        }
        if (__any(bad) && lane == 0) *fail = 1;
    }
    __syncthreads();
    if (blockIdx.x == 0)
        for (int e = tid; e < TB * TB; e += 256) {
            const int r = e >> 5, c = e & 31;
            L[(size_t)(k * TB + r) * ld + k * TB + c] = (c <= r) ? D[r][c] : 0.0;
        }
    if (ti < 0) return;
    if (tj == k + 1)
        for (int e = tid; e < TB * TB; e += 256) {
            const int r = e >> 5, c = e & 31;
            L[(size_t)(ti * TB + r) * ld + k * TB + c] = Pi[r][c];
        }
    // A_ij -= L_ik L_jk^T: wave w owns the 16x16 block (w >> 1, w & 1)
    const int bi = wave >> 1, bj = wave & 1, kq = lane >> 4, cq = lane & 15;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < TB / 4; ++ks)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pi[bi * 16 + cq][ks * 4 + kq], Pj[bj * 16 + cq][ks * 4 + kq], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const size_t idx = (size_t)(ti * TB + bi * 16 + kq + 4 * q) * ld + tj * TB + bj * 16 + cq;
        A[idx] = A[idx] - acc[q];
    }
}

// -----------------------------------------------------------------------------------------
// Finish (one workgroup): back substitution L^T x = y with y = row n of L, then everything the
// step needs from delta_s = x:
//   delta_s, candidate intrinsics / points, ds_emb[C][ZC] (per camera: the step laid out in Z's
//   column order with 1 in the gradient column), and R[4..14]:
//   R[4] sum ds (Ds ds - gs)   R[5] |ds|^2   R[6] |x|^2 over non-constant shared blocks (Ceres)
//   R[7] |x|^2 over free shared scalars (levmar)   R[8] 1 if the factorisation failed or ds is not finite
//   R[9] max |gs|  R[10] max diag U  R[11] sum r^2  R[12] bad pose blocks  R[13] max |g_pose|  R[14] max diag V
// -----------------------------------------------------------------------------------------
struct FinishArgs {
    const double *L;
    int ld, n, npad;
    const double *E;        // S | gred | gs | diagU | scalars at E_scal
    long long E_scal;
    int nranks;
    const double *Ds;
    const int *fail;
    int C, P, ncf, fix_k3, cam_free, pts_free, ZC, camcol0, ycol;
    const double *intr, *points;
    double *intr_cand, *points_cand, *ds_emb, *delta_s, *R;
};

__device__ __forceinline__ double block_sum(double v, double *red, int tid) {
    red[tid] = v;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ double block_max(double v, double *red, int tid) {
    red[tid] = v;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] = fmax(red[tid], red[tid + off]);
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(256) k_solve_finish(FinishArgs a) {
    extern __shared__ double sm[];
    double *xs = sm;                       // [npad]
    double *tile = xs + a.npad;            // [32][33]
    double *part = tile + TB * (TB + 1);   // [8][32]
    double *red = part + 8 * TB;           // [256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n, ld = a.ld;
    for (int i = tid; i < a.npad; i += 256) xs[i] = (i < n) ? a.L[(size_t)n * ld + i] : 0.0;
    __syncthreads();
    const int nTn = (n + TB - 1) / TB;
    for (int kt = nTn - 1; kt >= 0; --kt) {
        for (int e = tid; e < TB * TB; e += 256) {
            const int r = e >> 5, c = e & 31;
            tile[r * (TB + 1) + c] = a.L[(size_t)(kt * TB + r) * ld + kt * TB + c];
        }
        {
            const int g = tid >> 5, c = tid & 31;
            double s = 0.0;
            for (int r = (kt + 1) * TB + g; r < n; r += 8) s += a.L[(size_t)r * ld + kt * TB + c] * xs[r];
            part[g * TB + c] = s;
        }
        __syncthreads();
        if (wave == 0) {
            const int m = lane & 31;
            double rhs = xs[kt * TB + m] - (((part[m] + part[TB + m]) + (part[2 * TB + m] + part[3 * TB + m])) +
                                            ((part[4 * TB + m] + part[5 * TB + m]) + (part[6 * TB + m] + part[7 * TB + m])));
#pragma unroll
            for (int c = TB - 1; c >= 0; --c) {
                const double xc = __shfl(rhs, c, 64) / tile[c * (TB + 1) + c];
                if (m == c) rhs = xc;
                else if (m < c) rhs -= tile[c * (TB + 1) + m] * xc;
            }
            if (lane < 32) xs[kt * TB + m] = rhs;
        }
        __syncthreads();
    }
    // ---- the step and its scalars ----
    const double *gs = a.E + (size_t)n * n + n, *dU = gs + n;
    double ms = 0.0, ss2 = 0.0, gm = 0.0, dm = 0.0, bad = (double)(*a.fail != 0);
    for (int i = tid; i < n; i += 256) {
        const double ds = xs[i];
        a.delta_s[i] = ds;
        if (!(fabs(ds) < 1.7e308)) bad = 1.0;
        ms += ds * (a.Ds[i] * ds - gs[i]);
        ss2 += ds * ds;
        gm = fmax(gm, fabs(gs[i]));
        dm = fmax(dm, dU[i]);
    }
    const int ncam = a.C * a.ncf;
    double xs2 = 0.0, fs2 = 0.0;
    for (int idx = tid; idx < 9 * a.C; idx += 256) {
        const int c = idx / 9, kk = idx % 9;
        double v = a.intr[idx];
        if (a.cam_free) {
            xs2 += v * v;
            if (!(a.fix_k3 && kk == 6)) {
                fs2 += v * v;
                const int kf = (a.fix_k3 && kk > 6) ? kk - 1 : kk;
                v += xs[c * a.ncf + kf];
            }
        }
        a.intr_cand[idx] = v;
    }
    for (int q = tid; q < 3 * a.P; q += 256) {
        double v = a.points[q];
        if (a.pts_free) { xs2 += v * v; fs2 += v * v; v += xs[ncam + q]; }
        a.points_cand[q] = v;
    }
    for (int e = tid; e < a.C * a.ZC; e += 256) {
        const int c = e / a.ZC, col = e % a.ZC;
        double v = 0.0;
        if (a.pts_free && col < 3 * a.P) v = xs[ncam + col];
        else if (a.cam_free && col >= a.camcol0 && col < a.camcol0 + 9) {
            const int kk = col - a.camcol0;
            if (!(a.fix_k3 && kk == 6)) v = xs[c * a.ncf + ((a.fix_k3 && kk > 6) ? kk - 1 : kk)];
        } else if (col == a.ycol) v = 1.0;
        a.ds_emb[e] = v;
    }
    ms = block_sum(ms, red, tid);
    ss2 = block_sum(ss2, red, tid);
    xs2 = block_sum(xs2, red, tid);
    fs2 = block_sum(fs2, red, tid);
    gm = block_max(gm, red, tid);
    dm = block_max(dm, red, tid);
    bad = block_max(bad, red, tid);
    if (tid == 0) {
        const double *scal = a.E + a.E_scal;
        double gp = 0.0, dp = 0.0;
        for (int q = 0; q < a.nranks; ++q) { gp = fmax(gp, scal[2 + q]); dp = fmax(dp, scal[2 + a.nranks + q]); }
        a.R[4] = ms; a.R[5] = ss2; a.R[6] = xs2; a.R[7] = fs2; a.R[8] = bad; a.R[9] = gm; a.R[10] = dm;
        a.R[11] = scal[0]; a.R[12] = scal[1]; a.R[13] = gp; a.R[14] = dp; a.R[15] = 0.0;
    }
}

}  // namespace cvxs
