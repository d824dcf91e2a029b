// HIP kernels (gfx950 / CDNA4, wave64) of the damped reduced solve of one LM linear step:
//   (S + D) delta_s = -g_red      with S the ns x ns Schur complement assembled by k_assemble.
// Replaces, on the device, what Ceres does after the Schur eliminator for
// /root/reference/modules/calib/src/calibration.cpp:420-433 (LevenbergMarquardtStrategy: diagonal
// D = clamp(diag) / radius, then a linear solve) and what levmar does at
// /root/reference/modules/util/src/3rdparty/levmar/lm_core.c:290-305 (augment the diagonal by mu,
// solve); the factorisation is a tiled right-looking Cholesky (32 x 32 tiles, one launch per tile
// column, trailing update on fp64 MFMA).  Neither substitution exists as a pass of its own: the
// right-hand side rides along as an extra matrix row (forward substitution), and the identity rides
// along as a block of extra rows BELOW it -- what the elimination leaves in the right-hand side's
// column of those rows is -x L_nn, the solution itself (see k_chol_step).  One small workgroup then
// makes the candidate shared parameters, the embedded step the pose back-substitution multiplies
// by, and the scalars of the step (model change, norms).  Nothing in the iteration returns to the
// host in between.
// Included only by calib.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace cvxk { struct LmDev; }

namespace cvxs {

typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int TB = 32;

// -----------------------------------------------------------------------------------------
// A (npad x npad, row-major) = [ S + D  .  ] rows/cols < n
//                              [ -g_red c ] row n (the right-hand side; c = 1e300), identity beyond
// D_j = clamp(s_j^2 d_j, min_diag, max_diag) * inv_radius / s_j^2 + mu,  d_j = U_jj
// -----------------------------------------------------------------------------------------
struct DampArgs {
    const double *E;    // S (n*n) | gred (n) | gs (n) | diagU (n)
    const double *Sp;   // view-sharded runs: the all-reduced lower triangle of S, packed i(i+1)/2 + j (else nullptr)
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
        v = a.Sp ? (i >= j ? a.Sp[(size_t)i * (i + 1) / 2 + j] : a.Sp[(size_t)j * (j + 1) / 2 + i]) : a.E[(size_t)i * n + j];
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
        v = (i == n) ? 1e300 : 1.0;     // keeps the right-hand-side row's pivot positive whatever |y|^2 is
    }
    a.A[t] = v;
}

#ifdef CVX_STAMPS
__device__ long long g_cstamps[64];
#define CSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && k == 3) g_cstamps[i] = wall_clock64(); } while (0)
#else
#define CSTAMP(i) do {} while (0)
#endif

__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, srclane);
    hi = __builtin_amdgcn_readlane(hi, srclane);
    return __hiloint2double(hi, lo);
}

// Sixteen columns [J0, J0 + 16) of the left-looking factor loop over the 64 rows a wave holds, software-pipelined:
// while column j is being finished (its diagonal -> rsqrt -> scale chain), the dot product of column j + 1 over
// every term of this half but the newest is already accumulated from LDS broadcasts of pivot row j + 1 (all
// published at least one column earlier); the newest term, L[j+1][j], travels by readlane.  The LDS round trip is
// off the serial chain.  Terms of earlier halves must already be folded into row[].
template <int J0>
__device__ __forceinline__ void chol_columns(double (&row)[TB], const double (*Dw)[TB + 1], double *mine, int lane, bool &bad) {
    (void)lane;
    double tpart = row[J0];     // row[j] - sum_{J0 <= m <= j-2} L[r][m] L[j][m]
    double valprev = 0.0;
    // The column's serial chain is a lane read, the rsqrt and five dependent fp64 operations; a wave issues in order, so the independent
    // dot-product fmas are placed by hand into the chain's stalls, one group after each link (the compiler's
    // scheduler puts them all behind the chain).  CHOL_FENCE keeps that order.
#define CHOL_FENCE() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
    for (int j = J0; j < J0 + 16; ++j) {
        const int nd = (j + 1 < J0 + 16) ? j - J0 : 0;     // terms of column j + 1's dot product available now
        double pl[16];          // pivot row j + 1, every read issued before the first use
#pragma unroll
        for (int i = 0; i < nd; ++i) pl[i] = Dw[j + 1][J0 + i];
        double acc[4] = {(j + 1 < J0 + 16) ? row[j + 1] : 0.0, 0.0, 0.0, 0.0};
        auto dots = [&](int g) {    // group g of five
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (i < nd && i >= g * nd / 5 && i < (g + 1) * nd / 5)
                    acc[i & 3] = fma((i & 3) ? row[J0 + i] : -row[J0 + i], pl[i], acc[i & 3]);
        };
        CHOL_FENCE();
        // the pivot does not wait for the lanes' own update: d = tpart[j] - L[j][j-1]^2 from two lane reads, of which
        // tpart[j] was ready a column ago; and 1 / sqrt(d) is never formed: val = t y0 (1 + e h) with the Newton
        // correction e h applied to t y0, which is ready before e is
        const double a = (j > J0) ? readlane_f64(valprev, j) : 0.0;      // L[j][j-1]
        const double d = fma(-a, a, readlane_f64(tpart, j));
        const double t = fma(-valprev, a, tpart);
        // hardware estimate of 1 / sqrt(d) + one Newton-Raphson step of third order (no special-case
        // handling: d <= 0 or non-finite gives NaN, which the flag reports)
        const double y0 = __builtin_amdgcn_rsq(d);
        CHOL_FENCE();
        dots(0);
        CHOL_FENCE();
        const double dy = -d * y0, ty = t * y0;
        CHOL_FENCE();
        dots(1);
        CHOL_FENCE();
        const double e = fma(dy, y0, 1.0);
        CHOL_FENCE();
        dots(2);
        CHOL_FENCE();
        const double tye = ty * e, h = fma(e, 0.375, 0.5);
        CHOL_FENCE();
        dots(3);
        CHOL_FENCE();
        const double val = fma(tye, h, ty);     // the diagonal lane has t == d: val = sqrt(d)
        CHOL_FENCE();
        dots(4);
        CHOL_FENCE();
        bad = bad || !(d > 0.0 && d < 1.7e308);
        row[j] = val;
        valprev = val;
        mine[j] = val;
        tpart = (acc[0] - acc[1]) - (acc[2] + acc[3]);
        CHOL_FENCE();
    }
#undef CHOL_FENCE
}

// -----------------------------------------------------------------------------------------
// One tile column k of the factorisation.  A block owns one trailing tile (ti, tj), k < tj.
// Waves 0 and 1 each run the left-looking column loop over 64 rows: lanes 0..31 hold the rows of
// the diagonal tile A_kk (factored redundantly by both waves and by every block: it is the serial
// part either way), lanes 32..63 the rows of the panel tile A_ik (wave 0) or A_jk (wave 1) -- a
// panel row obeys the same recurrence L[r][j] = (A[r][j] - sum_{m<j} L[r][m] L[j][m]) / L[j][j], so
// the triangular solve rides along for free.  Then all four waves update A_ij -= L_ik L_jk^T with
// MFMA.  Panel k of A is only read in this launch: no block reads what another writes.
// Rows with global index >= n (the right-hand-side row and the padding) need no special case: k_damp_prepare
// gives them diagonal 1e300 and 1.
//
// The solution without a back substitution.  Below the npad rows of the matrix lie npad EXTRA rows that
// start as the identity: extra row tile r holds I in tile column r and zeros elsewhere.  They are panel rows
// like any other (never pivots), so after columns 0 .. m-1 are eliminated tile (r', m) holds a block of
// L^-T, and the trailing updates accumulate, in the right-hand side's own column n (the one with the 1e300
// pivot), -sum_k (L^-T)[r][k] y[k] = -x_r.  The ride-along division of that column by L_nn is undone when the
// last launch writes x.  Extra tile (r', j) takes part from launch k = r on (before that it is all zero), and
// launch r synthesises its initial value instead of loading it, so the extra rows are never initialised in
// memory; the blocks of extra tiles sit behind the symmetric ones: rows r <= k, columns j > k.  The last
// launch (k = nT - 1) has one block per extra row tile: ride-along only, then x_r = -L[r][n] L_nn.
// A: (2 npad) x ld, rows >= npad the extra rows.  x_out: n doubles.
// -----------------------------------------------------------------------------------------
__host__ __device__ inline int chol_step_blocks(int nT, int k) {
    const int ntr = nT - 1 - k;
    return ntr > 0 ? ntr * (ntr + 1) / 2 + (k + 1) * ntr : nT;
}

__global__ void __launch_bounds__(256) k_chol_step(double *A, int ld, int n, int k, int nT, int *fail, double *x_out) {
    __shared__ double D[2][TB][TB + 1];
    __shared__ double P[2][TB][TB + 1];
    __shared__ double X[2][64][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    CSTAMP(0);
    const int ntr = nT - 1 - k;
    // task: (ti, tj) tile rows of the two panels; tj < 0: no trailing tile (last launch); fresh: extra row tile k,
    // whose tiles still have their initial values (identity in column k, zero right of it)
    int ti, tj = -1;
    bool fresh = false;
    if (ntr > 0) {
        const int nsym = ntr * (ntr + 1) / 2;
        int b = blockIdx.x;
        if (b < nsym) {
            int r = 0;
            while (b > r) { b -= r + 1; ++r; }
            ti = k + 1 + r; tj = k + 1 + b;
        } else {
            b -= nsym;
            const int r = b / ntr;
            ti = nT + r; tj = k + 1 + b % ntr;
            fresh = (r == k);
        }
    } else {
        ti = nT + blockIdx.x;
        fresh = ((int)blockIdx.x == k);
    }
    // this wave's 16x16 block of A_ij, fetched now so the latency hides behind the factorisation
    const int bi = wave >> 1, bj = wave & 1, kq = lane >> 4, cq = lane & 15;
    double a_old[4] = {0.0, 0.0, 0.0, 0.0};
    if (tj >= 0 && !fresh) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a_old[q] = A[(size_t)(ti * TB + bi * 16 + kq + 4 * q) * ld + tj * TB + bj * 16 + cq];
    }
    {
        double vd[4], vi[4], vj[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q, r = e >> 5, c = e & 31;
            vd[q] = A[(size_t)(k * TB + r) * ld + k * TB + c];
            vi[q] = fresh ? ((r == c) ? 1.0 : 0.0) : A[(size_t)(ti * TB + r) * ld + k * TB + c];
            vj[q] = (tj >= 0) ? A[(size_t)(tj * TB + r) * ld + k * TB + c] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q, r = e >> 5, c = e & 31;
            D[0][r][c] = vd[q]; D[1][r][c] = vd[q];
            P[0][r][c] = vi[q]; P[1][r][c] = vj[q];
        }
    }
    __syncthreads();
    CSTAMP(1);
    if (wave < 2) {
        // 64 rows per wave (lanes 0..31 the diagonal tile, 32..63 this wave's panel tile), one lane per row, the
        // row in registers.  Two half-tiles of 16 columns (chol_columns), and between them the rank-16 update of
        // columns 16..31 by the finished columns 0..15 on MFMA -- done lane by lane it would be 256 of the 496
        // fma + LDS-broadcast pairs of the whole tile, which a single wave issues at ~12 cycles each.
        const int r = lane & 31;
        const bool panel = lane >= 32;
        double (*Dw)[TB + 1] = D[wave];
        double (*Pw)[TB + 1] = P[wave];
        double *mine = panel ? &Pw[r][0] : &Dw[r][0];   // every lane publishes its row: no branch in the loops
        double row[TB];
        bool bad = false;
#pragma unroll
        for (int c = 0; c < TB; ++c) row[c] = mine[c];
        chol_columns<0>(row, Dw, mine, lane, bad);
        {
            // C[R][16 + c] = sum_{m < 16} L[R][m] L[16 + c][m] for the rows R >= 16 of the wave's 64 (row block 0 is
            // the upper triangle of the diagonal tile: not needed).  The wave's own LDS writes above are ordered
            // before these reads by the LDS queue; the asm only stops the compiler from moving them.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            f64x4 acc1 = {0.0, 0.0, 0.0, 0.0}, acc2 = acc1, acc3 = acc1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const double b = Dw[16 + cq][4 * ks + kq];
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Pw[cq][4 * ks + kq], b, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(Pw[16 + cq][4 * ks + kq], b, acc3, 0, 0, 0);
            }
            // accumulator layout (lane = cq + 16 kq holds rows kq + 4 q of column cq) -> one row per lane, through LDS
            double (*Xw)[17] = X[wave];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                Xw[16 + kq + 4 * q][cq] = acc1[q];
                Xw[32 + kq + 4 * q][cq] = acc2[q];
                Xw[48 + kq + 4 * q][cq] = acc3[q];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane >= 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) row[16 + c] -= Xw[lane][c];
            }
        }
        chol_columns<16>(row, Dw, mine, lane, bad);
        if (wave == 0 && blockIdx.x == 0 && __any(bad) && lane == 0) *fail = 1;
    }
    __syncthreads();
    CSTAMP(2);
    if (tj < 0) {
        // last launch: extra row tile r = blockIdx.x rode along as panel 0.  Column nl of it holds
        // (-x_r) / L_nn, L_nn = D[nl][nl]
        const int nl = n - k * TB, r0 = (int)blockIdx.x * TB;
        if (tid < TB && r0 + tid < n) x_out[r0 + tid] = -(P[0][tid][nl] * D[0][nl][nl]);
        return;
    }
    // A_ij -= L_ik L_jk^T: wave w owns the 16x16 block (w >> 1, w & 1)
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < TB / 4; ++ks)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(P[0][bi * 16 + cq][ks * 4 + kq], P[1][bj * 16 + cq][ks * 4 + kq], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        A[(size_t)(ti * TB + bi * 16 + kq + 4 * q) * ld + tj * TB + bj * 16 + cq] = a_old[q] - acc[q];
    CSTAMP(3);
}

// -----------------------------------------------------------------------------------------
// Finish (one workgroup): everything the step needs from delta_s = x, which the last k_chol_step launch left
// in delta_s:
//   candidate intrinsics / points, ds_emb[C][ZC] (per camera: the step laid out in Z's
//   column order with 1 in the gradient column), and R[4..14]:
//   R[4] sum ds (Ds ds - gs)   R[5] |ds|^2   R[6] |x|^2 over non-constant shared blocks (Ceres)
//   R[7] |x|^2 over free shared scalars (levmar)   R[8] 1 if the factorisation failed or ds is not finite
//   R[9] max |gs|  R[10] max diag U  R[11] sum r^2  R[12] bad pose blocks  R[13] max |g_pose|  R[14] max diag V
// -----------------------------------------------------------------------------------------
struct FinishArgs {
    int n;
    const double *E;        // S | gred | gs | diagU | scalars at E_scal
    long long E_scal;
    int nranks;
    const double *Ds;
    const int *fail;
    int C, P, ncf, fix_k3, cam_free, pts_free, ZC, camcol0, ycol;
    const double *intr, *points;
    double *intr_cand, *points_cand, *ds_emb, *delta_s, *R;
    // speculated steps (calib_kernels.h): the four parameter pointers are buffer 0's, buffer 1 lies `*_step` doubles
    // behind, and the record says which one is current
    const cvxk::LmDev *st;
    long long intr_step, points_step, poses_step;
};

// block reductions of the finish kernel: wave butterflies, then four partials through LDS
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// the step and its scalars (shared tail of both finish kernels); NT = threads of the workgroup
template <int NT>
__device__ __forceinline__ void finish_tail(const FinishArgs &a, const double *xs, double *red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n;
    const double *gs = a.E + (size_t)n * n + n, *dU = gs + n;
    double ms = 0.0, ss2 = 0.0, gm = 0.0, dm = 0.0, bad = (double)(*a.fail != 0);
    for (int i = tid; i < n; i += NT) {
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
    for (int idx = tid; idx < 9 * a.C; idx += NT) {
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
    for (int q = tid; q < 3 * a.P; q += NT) {
        double v = a.points[q];
        if (a.pts_free) { xs2 += v * v; fs2 += v * v; v += xs[ncam + q]; }
        a.points_cand[q] = v;
    }
    for (int e = tid; e < a.C * a.ZC; e += NT) {
        const int c = e / a.ZC, col = e % a.ZC;
        double v = 0.0;
        const int cb = (col >= a.P) + (col >= 2 * a.P);      // Z's point columns are coordinate-major: column b P + p is unknown 3 p + b
        if (a.pts_free && col < 3 * a.P) v = xs[ncam + 3 * (col - cb * a.P) + cb];
        else if (a.cam_free && col >= a.camcol0 && col < a.camcol0 + 9) {
            const int kk = col - a.camcol0;
            if (!(a.fix_k3 && kk == 6)) v = xs[c * a.ncf + ((a.fix_k3 && kk > 6) ? kk - 1 : kk)];
        } else if (col == a.ycol) v = 1.0;
        a.ds_emb[e] = v;
    }
    ms = wave_sum(ms); ss2 = wave_sum(ss2); xs2 = wave_sum(xs2); fs2 = wave_sum(fs2);
    gm = wave_max(gm); dm = wave_max(dm); bad = wave_max(bad);
    if (lane == 0) {
        double *rw = red + 8 * wave;
        rw[0] = ms; rw[1] = ss2; rw[2] = xs2; rw[3] = fs2; rw[4] = gm; rw[5] = dm; rw[6] = bad;
    }
    __syncthreads();
    if (tid == 0) {
        ms = 0.0; ss2 = 0.0; xs2 = 0.0; fs2 = 0.0; gm = 0.0; dm = 0.0; bad = 0.0;
        for (int wv = 0; wv < NT / 64; ++wv) {
            const double *rw = red + 8 * wv;
            ms += rw[0]; ss2 += rw[1]; xs2 += rw[2]; fs2 += rw[3];
            gm = fmax(gm, rw[4]); dm = fmax(dm, rw[5]); bad = fmax(bad, rw[6]);
        }
    }
    if (tid == 0) {
        const double *scal = a.E + a.E_scal;
        double gp = 0.0, dp = 0.0;
        for (int q = 0; q < a.nranks; ++q) { gp = fmax(gp, scal[2 + q]); dp = fmax(dp, scal[2 + a.nranks + q]); }
        a.R[4] = ms; a.R[5] = ss2; a.R[6] = xs2; a.R[7] = fs2; a.R[8] = bad; a.R[9] = gm; a.R[10] = dm;
        a.R[11] = scal[0]; a.R[12] = scal[1]; a.R[13] = gp; a.R[14] = dp; a.R[15] = 0.0;
    }
}

constexpr int FIN_THREADS = 512;

__global__ void __launch_bounds__(FIN_THREADS) k_step_finish(FinishArgs a) {
    extern __shared__ double sm[];
    double *xs = sm;                       // [n]
    double *red = xs + ((a.n + 7) & ~7);   // [64]
    for (int i = threadIdx.x; i < a.n; i += FIN_THREADS) xs[i] = a.delta_s[i];
    __syncthreads();
    finish_tail<FIN_THREADS>(a, xs, red);
}
__host__ __device__ inline size_t finish_lds_doubles(int n) { return (size_t)((n + 7) & ~7) + 64; }

}  // namespace cvxs
