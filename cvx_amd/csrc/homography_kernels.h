// Batched homography refinement on gfx950: one LM problem (8 parameters, 2N residuals) per 64-lane wave.
// Path: HomographyEstimator::solveLM (/root/reference/modules/calib/src/homography.cpp:159-236) =
// LMHomographyEstimator::errors / jacobian (:166-207) driven by levmar's dlevmar_der
// (modules/util/src/3rdparty/levmar/lm_core.c:60-423) through LMSolver::minimizeDer (lm.hpp:60-64), with the
// no-LAPACK LU solve (Axb_core.c:992-1139).
//
// Every floating-point sum is evaluated in the reference's order (the row order and 32-row blocking of
// lm_core.c:217-255 / misc_core.c:80-132, the four interleaved chains of misc_core.c:712-798, Crout LU with
// implicit-scaling pivoting), so a problem's iterates are bit-identical to the CPU reference's; parallelism
// comes from one lane per normal-equation entry (29 distinct sums because of the Jacobian's block structure),
// the 8x8 system held one entry per lane, and thousands of independent problems in flight.
#pragma once
#include <cfloat>

#include "common.h"
#include "nielsen.h"

namespace cvxhom {

struct HomArgs {
    const int *off;                         // [n_problems + 1] offsets into the point arrays
    const float *x1, *y1, *x2, *y2;         // gathered point pairs (cv::Point2f)
    double *H;                              // [n_problems][9] row-major, in/out
    double *info;                           // [n_problems][10] levmar info[] (may be null)
    int *ret;                               // [n_problems] dlevmar_der return value (may be null)
    int n_problems, itmax;
    double tau, eps1, eps2, eps3;           // opts[0..3] as lm.hpp:61 builds them
    long long *prof;                        // -DHOM_PROF builds only: [n_problems][4] cycles in eval, sums, norm, LU
};

#ifdef HOM_PROF
#define HOM_T(slot, stmt) { const long long t__ = clock64(); stmt; prof_acc[slot] += clock64() - t__; }
#else
#define HOM_T(slot, stmt) { stmt; }
#endif

constexpr int HOM_SCRATCH = 32;                                   // doubles of per-wave exchange space
__host__ __device__ inline size_t hom_lds_doubles(int n_points) { return (size_t)14 * n_points + HOM_SCRATCH; }

__device__ __forceinline__ double bcast(double v, int lane /* wave-uniform */) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Which products lane L sums.  Per point the Jacobian rows are (homography.cpp:204-205)
//   even row: a0 a1 a2 0 0 0 b00 b01      odd row: 0 0 0 a0 a1 a2 b10 b11
// stored as Q[pt] = {a0, a1, a2, b00, b01, b10, b11, e_even, e_odd, 0}.  {c0, d0} = columns of Q multiplied for the
// even row's term, {c1, d1} for the odd row's, -1 = structurally zero (read from the zero column: the
// reference adds those +0 products too, lm_core.c:228 / misc_core.c:113, so the sums stay bit-identical).  Lanes 0-20: the distinct entries of
// J^T J; lanes 21-28: J^T e.
__constant__ signed char HOM_TERMS[29][4] = {
    {0, 0, -1, -1}, {0, 1, -1, -1}, {0, 2, -1, -1}, {1, 1, -1, -1}, {1, 2, -1, -1}, {2, 2, -1, -1},   // a_i a_j
    {0, 3, -1, -1}, {0, 4, -1, -1}, {1, 3, -1, -1}, {1, 4, -1, -1}, {2, 3, -1, -1}, {2, 4, -1, -1},   // a_i b0_j
    {-1, -1, 0, 5}, {-1, -1, 0, 6}, {-1, -1, 1, 5}, {-1, -1, 1, 6}, {-1, -1, 2, 5}, {-1, -1, 2, 6},   // a_i b1_j
    {3, 3, 5, 5}, {3, 4, 5, 6}, {4, 4, 6, 6},                                                           // b_j b_k
    {0, 7, -1, -1}, {1, 7, -1, -1}, {2, 7, -1, -1}, {-1, -1, 0, 8}, {-1, -1, 1, 8}, {-1, -1, 2, 8},   // J^T e 0..5
    {3, 7, 5, 8}, {4, 7, 6, 8}};                                                                        // J^T e 6,7

// index into the 29 sums of entry (r, c) of J^T J, -1 for the structural zeros
__device__ __forceinline__ int hom_entry(int r, int c) {
    if (r > c) { const int t = r; r = c; c = t; }
    if (r < 3) {
        if (c < 3) return r == 0 ? c : (r == 1 ? 2 + c : 5);
        if (c < 6) return -1;
        return 6 + r * 2 + (c - 6);
    }
    if (r < 6) {
        if (c < 6) { const int i = r - 3, j = c - 3; return i == 0 ? j : (i == 1 ? 2 + j : 5); }
        return 12 + (r - 3) * 2 + (c - 6);
    }
    return r == 6 ? 18 + (c - 6) : 20;
}

// errors() / jacobian() at h over the wave's points (homography.cpp:166-207); e = 0 - f because levmar forms
// e = x - hx with x = NULL (lm_core.c:163-167)
template <bool JAC>
__device__ __forceinline__ void hom_eval(const float4 *P, int N, const double h[8], double *Q, double *E, int lane) {
    for (int i = lane; i < N; i += 64) {
        const float4 pt = P[i];
        const float x1 = pt.x, y1 = pt.y, x2 = pt.z, y2 = pt.w;
        double w = h[6] * x1 + h[7] * y1 + 1.0;
        w = (w > DBL_EPSILON) ? 1.0 / w : 0.0;
        const double xi = (h[0] * x1 + h[1] * y1 + h[2]) * w;
        const double yi = (h[3] * x1 + h[4] * y1 + h[5]) * w;
        const double e0 = 0.0 - (xi - x2), e1 = 0.0 - (yi - y2);
        if (JAC) {
            double *q = Q + (size_t)i * 10;
            q[0] = x1 * w; q[1] = y1 * w; q[2] = w;
            q[3] = -x1 * xi * w; q[4] = -y1 * xi * w;
            q[5] = -x1 * yi * w; q[6] = -y1 * yi * w;
            q[7] = e0; q[8] = e1; q[9] = 0.0;
        } else {
            E[2 * i] = e0; E[2 * i + 1] = e1;
        }
    }
}

// ||e||^2 in the order of LEVMAR_L2NRMXMY (misc_core.c:712-798): four interleaved chains over the blocks of 8
// from the top down, then the tail; wave-uniform result
__device__ __forceinline__ double hom_norm2(const double *E, int n, int lane) {
    const int blockn = (n >> 3) << 3;
    double s = 0.0;
    if (lane < 4) {
        int i = blockn - 1 - lane;
        // eight loads in flight per step of the chain (the adds stay in order)
        for (; i >= 28; i -= 32) {
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = E[i - 4 * j];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j] * v[j];
        }
        for (; i >= 0; i -= 4) { const double v = E[i]; s += v * v; }
        for (i = blockn; i < n; ++i)
            if (((7 - (n - i)) & 3) == lane) { const double v = E[i]; s += v * v; }
    }
    return bcast(s, 0) + bcast(s, 1) + bcast(s, 2) + bcast(s, 3);
}

// (A + mu I) x = b for the 8x8 system held one entry per lane (lane = 8 r + c), b[r] replicated over row r.
// Crout LU with implicitly scaled partial pivoting in the operation order of Axb_core.c:992-1139, forward
// substitution folded into the elimination steps, back substitution on wave-uniform values.  Returns false
// only for an all-zero row, as the reference does.
__device__ __forceinline__ bool hom_lu_solve(double a, double bv, int lane, double x[8]) {
    const int r = lane >> 3, c = lane & 7;
    double big = fabs(a);
    big = fmax(big, __shfl_xor(big, 1, 64));
    big = fmax(big, __shfl_xor(big, 2, 64));
    big = fmax(big, __shfl_xor(big, 4, 64));
    if (__any(big == 0.0)) return false;
    double sc = 1.0 / big;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double t = sc * fabs(a);
        double best = 0.0;
        int piv = -1;
#pragma unroll
        for (int rr = k; rr < 8; ++rr) {
            const double tr = bcast(t, rr * 8 + k);
            if (tr >= best) { best = tr; piv = rr; }
        }
        if (piv < 0) piv = k;
        if (piv != k) {
            const int srow = (r == k) ? piv : (r == piv ? k : r);
            const int src = srow * 8 + c;
            a = __shfl(a, src, 64);
            bv = __shfl(bv, src, 64);
            sc = __shfl(sc, src, 64);
        }
        double pv = bcast(a, k * 9);
        if (pv == 0.0) {
            pv = DBL_EPSILON;
            if (lane == k * 9) a = pv;
        }
        if (k != 7) {
            const double inv = 1.0 / pv;
            if (c == k && r > k) a *= inv;
            const double lrk = __shfl(a, (lane & 56) | k, 64);
            const double ukc = __shfl(a, k * 8 + c, 64);
            const double bk = bcast(bv, k * 8);
            if (r > k) {
                if (c > k) a -= lrk * ukc;
                bv -= lrk * bk;
            }
        }
    }
#pragma unroll
    for (int rr = 7; rr >= 0; --rr) {
        double acc = bcast(bv, rr * 8);
#pragma unroll
        for (int kk = rr + 1; kk < 8; ++kk) acc -= bcast(a, rr * 8 + kk) * x[kk];
        x[rr] = acc / bcast(a, rr * 9);
    }
    return true;
}

// one block = one wave = one problem; dynamic LDS = hom_lds_doubles(max points of any problem) * 8 bytes
__global__ __launch_bounds__(64) void k_homography_lm(HomArgs a) {
    extern __shared__ double4 hom_lds_raw[];
    double *lds = reinterpret_cast<double *>(hom_lds_raw);
    const int prob = blockIdx.x, lane = threadIdx.x;
    const int o0 = a.off[prob], N = a.off[prob + 1] - o0, n = 2 * N;
    double *Hp = a.H + (size_t)prob * 9;
    if (n < 8) {      // lm_core.c:117-120: fewer measurements than unknowns -> LM_ERROR, parameters untouched
        if (lane == 0 && a.ret) a.ret[prob] = -1;
        if (lane < 10 && a.info) a.info[(size_t)prob * 10 + lane] = 0.0;
        return;
    }
    float4 *P = reinterpret_cast<float4 *>(lds);
    double *Q = lds + 2 * (size_t)N, *E = Q + 10 * (size_t)N, *S = E + 2 * (size_t)N;
    for (int i = lane; i < N; i += 64) P[i] = make_float4(a.x1[o0 + i], a.y1[o0 + i], a.x2[o0 + i], a.y2[o0 + i]);

    const int r = lane >> 3, c = lane & 7;
    const int entry = hom_entry(r, c);
    int c0 = -1, d0 = -1, c1 = -1, d1 = -1;
    if (lane < 29) { c0 = HOM_TERMS[lane][0]; d0 = HOM_TERMS[lane][1]; c1 = HOM_TERMS[lane][2]; d1 = HOM_TERMS[lane][3]; }
    const bool is_jte = lane >= 21;
    const bool small = (long)n * 8 < 32 * 32;          // lm_core.c:206 (nm < __BLOCKSZ__SQ)

    double p[8], pDp[8], Dp[8], jte[8], diag[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { p[i] = Hp[i]; diag[i] = 0.0; jte[i] = 0.0; Dp[i] = 0.0; pDp[i] = 0.0; }
    cvx::NielsenDamping damp = {0.0, 2};
    double grad_inf = 0.0, param_sq = 0.0, step_sq = DBL_MAX, err_sq, cand_sq, ajj = 0.0;
    int stop = cvx::NS_RUNNING, n_eval = 1, n_jac = 0, n_solve = 0, outer;
#ifdef HOM_PROF
    long long prof_acc[4] = {0, 0, 0, 0};
    const long long prof_t0 = clock64();
#endif

    __syncthreads();
    hom_eval<false>(P, N, p, Q, E, lane);
    __syncthreads();
    err_sq = hom_norm2(E, n, lane);
    const double first_err_sq = err_sq;
    if (!isfinite(err_sq)) stop = cvx::NS_NOT_FINITE;

    // outer loop: one linearisation per pass; inner loop: raise mu until a step reduces the error.  All decisions are
    // wave-uniform (the same values on every lane); damping schedule and step tests: nielsen.h
    for (outer = 0; outer < a.itmax && stop == cvx::NS_RUNNING; ++outer) {
        if (err_sq <= a.eps3) { stop = cvx::NS_SMALL_RESIDUAL; break; }
        __syncthreads();
        HOM_T(0, hom_eval<true>(P, N, p, Q, E, lane); __syncthreads())
        ++n_jac;
        __syncthreads();
#ifdef HOM_PROF
        const long long prof_ts = clock64();
#endif
        if (lane < 29) {
            // every lane reads both of its operand pairs for a group of points first, then adds the products in the
            // reference's order; a structurally zero term multiplies the zero column and adds +0
            const double *q0 = Q + (c0 >= 0 ? c0 : 9), *r0 = Q + (c0 >= 0 ? d0 : 9);
            const double *q1 = Q + (c1 >= 0 ? c1 : 9), *r1 = Q + (c1 >= 0 ? d1 : 9);
            double res;
            if (small) {            // lm_core.c:217-237: rows from the last to the first, one running sum
                double s = 0.0;
                int pt = N - 1;
                for (; pt >= 7; pt -= 8) {
                    double t0[8], t1[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int o = (pt - j) * 10;
                        t0[j] = q0[o] * r0[o]; t1[j] = q1[o] * r1[o];
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) { s += t1[j]; s += t0[j]; }
                }
                for (; pt >= 0; --pt) {
                    const int o = pt * 10;
                    const double t0 = q0[o] * r0[o], t1 = q1[o] * r1[o];
                    s += t1; s += t0;
                }
                res = s;
            } else {                // misc_core.c:80-132: 32-row partial sums; J^T e row by row (lm_core.c:246-254)
                double tot = 0.0, part = 0.0;
                for (int b0 = 0; b0 < N; b0 += 16) {
                    if (!is_jte) part = 0.0;
                    if (b0 + 16 <= N) {
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            double t0[8], t1[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int o = (b0 + hh * 8 + j) * 10;
                                t0[j] = q0[o] * r0[o]; t1[j] = q1[o] * r1[o];
                            }
#pragma unroll
                            for (int j = 0; j < 8; ++j) { part += t0[j]; part += t1[j]; }
                        }
                    } else {
                        for (int pt = b0; pt < N; ++pt) {
                            const int o = pt * 10;
                            part += q0[o] * r0[o];
                            part += q1[o] * r1[o];
                        }
                    }
                    if (!is_jte) tot += part;
                }
                res = is_jte ? part : tot;
            }
            S[lane] = res;
        }
#ifdef HOM_PROF
        __syncthreads();
        prof_acc[1] += clock64() - prof_ts;
#endif
        __syncthreads();
        ajj = entry >= 0 ? S[entry] : 0.0;
        param_sq = 0.0; grad_inf = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            jte[i] = S[21 + i]; diag[i] = S[hom_entry(i, i)];
            grad_inf = (grad_inf < fabs(jte[i])) ? fabs(jte[i]) : grad_inf;
            param_sq += p[i] * p[i];
        }
        if (grad_inf <= a.eps1) { step_sq = 0.0; stop = cvx::NS_SMALL_GRADIENT; break; }
        if (outer == 0) {
            double widest = -DBL_MAX;
#pragma unroll
            for (int i = 0; i < 8; ++i) widest = (diag[i] > widest) ? diag[i] : widest;
            damp.mu = a.tau * widest;
        }
        const double brow = S[21 + r];
        for (;;) {
            const double aug = (r == c) ? ajj + damp.mu : ajj;
            bool solved;
            HOM_T(3, solved = hom_lu_solve(aug, brow, lane, Dp))
            ++n_solve;
            if (solved) {
                step_sq = 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i) { pDp[i] = p[i] + Dp[i]; step_sq += Dp[i] * Dp[i]; }
                stop = cvx::nielsen_step_test(step_sq, param_sq, a.eps2);
                if (stop != cvx::NS_RUNNING) break;
                __syncthreads();
                HOM_T(0, hom_eval<false>(P, N, pDp, Q, E, lane); __syncthreads())
                ++n_eval;
                __syncthreads();
                HOM_T(2, cand_sq = hom_norm2(E, n, lane))
                if (!isfinite(cand_sq)) { stop = cvx::NS_NOT_FINITE; break; }
                double predicted = 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i) predicted += Dp[i] * (damp.mu * Dp[i] + jte[i]);
                const double actual = err_sq - cand_sq;
                if (predicted > 0.0 && actual > 0.0) {
                    damp.relax(actual, predicted);
#pragma unroll
                    for (int i = 0; i < 8; ++i) p[i] = pDp[i];
                    err_sq = cand_sq;
                    break;
                }
            }
            if (!damp.tighten()) { stop = cvx::NS_NO_REDUCTION; break; }
        }
    }
    if (outer >= a.itmax && stop == cvx::NS_RUNNING) stop = cvx::NS_ITERATION_LIMIT;

#ifdef HOM_PROF
    if (lane == 0 && a.prof) {
        for (int i = 0; i < 4; ++i) a.prof[(size_t)prob * 5 + i] = prof_acc[i];
        a.prof[(size_t)prob * 5 + 4] = clock64() - prof_t0;
    }
#endif
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) Hp[i] = p[i];
        Hp[8] = 1.0;                                   // homography.cpp:233
        if (a.ret) a.ret[prob] = (stop != cvx::NS_NEAR_SINGULAR && stop != cvx::NS_NOT_FINITE) ? outer : -1;
        if (a.info) {
            double mx = DBL_MIN;
#pragma unroll
            for (int i = 0; i < 8; ++i) if (diag[i] > mx) mx = diag[i];
            double *o = a.info + (size_t)prob * 10;
            // levmar's info[0..9] (levmar.h:60-75)
            o[0] = first_err_sq; o[1] = err_sq; o[2] = grad_inf; o[3] = step_sq; o[4] = damp.mu / mx;
            o[5] = (double)outer; o[6] = (double)stop; o[7] = (double)n_eval; o[8] = (double)n_jac; o[9] = (double)n_solve;
        }
    }
}

// -----------------------------------------------------------------------------------------
// HomographyEstimator::solve (the normalised DLT, /root/reference/modules/calib/src/homography.cpp:17-93) and
// HomographyEstimator::residuals (:95-110) for the same batches -- the steps either side of solveLM in the reference's
// flow (solveRansac -> inliers -> solveLM).  Two launches:
//   k_homography_dlt_reduce   one wave per problem: normalize_points in the reference's float arithmetic and summation
//                             order, the 2N x 9 constraint matrix in LDS, Householder QR down to its 9 x 9 triangle
//   k_homography_dlt_finish   one LANE per problem: one-sided Jacobi SVD of the triangle (the right singular vector of
//                             the smallest singular value = JacobiSVD(A, ComputeFullV).matrixV().col(8) up to sign and
//                             rounding; Eigen is un-vendored), H = T2^-1 Hn T1, H /= H(2,2)
// -----------------------------------------------------------------------------------------
struct DltArgs {
    const int *off;
    const float *x1, *y1, *x2, *y2;
    double *W;          // [n_problems][DLT_W] : R (81, row-major, zero below the diagonal) | cx1 cy1 s1 cx2 cy2 s2
    double *H;          // [n_problems][9]
    int n_problems;
};
constexpr int DLT_W = 88;
constexpr int DLT_PER_LANE = 16;            // 64 x 16 = CVX_HOMOGRAPHY_MAX_POINTS
__host__ __device__ inline size_t dlt_lds_doubles(int n_points) { return (size_t)18 * n_points + HOM_SCRATCH; }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double hom_dpp_take(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
// sum over the wave by data-parallel-primitive moves, the same value on every lane
__device__ __forceinline__ double hom_wave_sum(double v) {
    v += hom_dpp_take<0xB1, 0xF>(v);        // quad_perm [1,0,3,2]
    v += hom_dpp_take<0x4E, 0xF>(v);        // quad_perm [2,3,0,1]
    v += hom_dpp_take<0x141, 0xF>(v);       // row_half_mirror
    v += hom_dpp_take<0x140, 0xF>(v);       // row_mirror
    v += hom_dpp_take<0x142, 0xA>(v);       // row_bcast:15 into rows 1 and 3
    v += hom_dpp_take<0x143, 0xC>(v);       // row_bcast:31 into rows 2 and 3
    return bcast(v, 63);
}

// normalize_points' sums for one coordinate array (homography.cpp:20-30), by ONE lane in the reference's order: cv::Point2f
// arithmetic is float, so the order is part of the result.  Eight loads in flight per step of the chain.
__device__ __forceinline__ float dlt_serial_sum(const float *v, int stride, int N) {
    float s = 0.0f;
    int i = 0;
    for (; i + 8 <= N; i += 8) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = v[(i + j) * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) s = s + t[j];
    }
    for (; i < N; ++i) s = s + v[i * stride];
    return s;
}

__global__ __launch_bounds__(64) void k_homography_dlt_reduce(DltArgs a) {
    extern __shared__ double4 hom_lds_raw[];
    double *lds = reinterpret_cast<double *>(hom_lds_raw);
    const int prob = blockIdx.x, lane = threadIdx.x;
    const int o0 = a.off[prob], N = a.off[prob + 1] - o0, m = 2 * N;
    if (N < 4) return;
    float *P = reinterpret_cast<float *>(lds);                       // x1 y1 x2 y2 per point, overlaid by A later
    double *S = lds + 18 * (size_t)N;                                // scratch behind A
    float4 mine[DLT_PER_LANE];
#pragma unroll
    for (int k = 0; k < DLT_PER_LANE; ++k) {
        const int i = lane + 64 * k;
        mine[k] = (i < N) ? make_float4(a.x1[o0 + i], a.y1[o0 + i], a.x2[o0 + i], a.y2[o0 + i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < N) reinterpret_cast<float4 *>(P)[i] = mine[k];
    }
    __syncthreads();
    // centres: lanes 0..3 each sum one coordinate in point order; then the mean squared distances: lanes 0 and 1
    float *Sf = reinterpret_cast<float *>(S);
    if (lane < 4) Sf[lane] = dlt_serial_sum(P + lane, 4, N) / (float)N;
    __syncthreads();
    if (lane < 2) {
        // float products and sums one by one, as the host compiler leaves them without FMA contraction (the intrinsics of
        // the __f*_rn family are plain operators here, which the device compiler would fuse)
#pragma clang fp contract(off)
        const float cx = Sf[2 * lane], cy = Sf[2 * lane + 1];
        const float *px = P + 2 * lane;
        float nrm = 0.0f;
        int i = 0;
        for (; i + 8 <= N; i += 8) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dx = px[4 * (i + j)] - cx, dy = px[4 * (i + j) + 1] - cy;
                const float xx = dx * dx, yy = dy * dy;
                t[j] = xx + yy;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) nrm = nrm + t[j];
        }
        for (; i < N; ++i) {
            const float dx = px[4 * i] - cx, dy = px[4 * i + 1] - cy;
            const float xx = dx * dx, yy = dy * dy;
            const float d = xx + yy;
            nrm = nrm + d;
        }
        nrm = nrm / (float)N;
        // sqrt(2.0) / sqrt(nrm): the float overload (:32); correctly rounded square roots and divisions (the compiler's
        // default for HIP, -fhip-fp32-correctly-rounded-divide-sqrt)
        S[4 + lane] = sqrt(2.0) / (double)sqrtf(nrm);
    }
    __syncthreads();
    const float cx1 = Sf[0], cy1 = Sf[1], cx2 = Sf[2], cy2 = Sf[3];
    const double s1 = S[4], s2 = S[5];
    __syncthreads();        // P is dead from here on: A takes its place
    double *A = lds;
#pragma unroll
    for (int k = 0; k < DLT_PER_LANE; ++k) {
        const int i = lane + 64 * k;
        if (i < N) {
            // (p - center) * scale: a float difference, a double product rounded to float (homography.cpp:40)
            const double u1 = (double)(float)((double)(mine[k].x - cx1) * s1), v1 = (double)(float)((double)(mine[k].y - cy1) * s1);
            const double u2 = (double)(float)((double)(mine[k].z - cx2) * s2), v2 = (double)(float)((double)(mine[k].w - cy2) * s2);
            double *r0 = A + 18 * (size_t)i, *r1 = r0 + 9;
            r0[0] = u1; r0[1] = v1; r0[2] = 1.0; r0[3] = 0.0; r0[4] = 0.0; r0[5] = 0.0; r0[6] = -u2 * u1; r0[7] = -u2 * v1; r0[8] = -u2;
            r1[0] = 0.0; r1[1] = 0.0; r1[2] = 0.0; r1[3] = -u1; r1[4] = -v1; r1[5] = -1.0; r1[6] = v2 * u1; r1[7] = v2 * v1; r1[8] = v2;
        }
    }
    __syncthreads();
    // Householder QR, rows dealt to the lanes (row i on lane i % 64): reflector k from column k's rows >= k, applied to
    // the columns right of it; one wave sum for the norm, one per remaining column for the dot products
    const int kmax = m < 9 ? m : 9;
    for (int k = 0; k < kmax; ++k) {
        double part = 0.0;
        for (int i = k + ((lane - k) & 63); i < m; i += 64) { const double v = A[9 * (size_t)i + k]; part += v * v; }
        const double ss = hom_wave_sum(part);
        const double akk = A[9 * (size_t)k + k];
        __syncthreads();
        if (ss > 0.0) {
            const double alpha = akk > 0.0 ? -sqrt(ss) : sqrt(ss);
            const double beta = 1.0 / (ss - akk * alpha);                 // 2 / v^T v with v = x - alpha e_k
            if (lane == (k & 63)) A[9 * (size_t)k + k] = akk - alpha;    // v_k (the other v_i are the column's entries)
            __syncthreads();
            double d[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = 0.0;
            for (int i = k + ((lane - k) & 63); i < m; i += 64) {
                const double *row = A + 9 * (size_t)i;
                const double v = row[k];
#pragma unroll
                for (int j = 0; j < 8; ++j) if (k + 1 + j < 9) d[j] += v * row[k + 1 + j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) if (k + 1 + j < 9) d[j] = hom_wave_sum(d[j]) * beta;
            for (int i = k + ((lane - k) & 63); i < m; i += 64) {
                double *row = A + 9 * (size_t)i;
                const double v = row[k];
#pragma unroll
                for (int j = 0; j < 8; ++j) if (k + 1 + j < 9) row[k + 1 + j] -= d[j] * v;
            }
            __syncthreads();
            if (lane == (k & 63)) A[9 * (size_t)k + k] = alpha;
        }
        __syncthreads();
    }
    double *W = a.W + (size_t)prob * DLT_W;
    for (int e = lane; e < 81; e += 64) {
        const int r = e / 9, c = e - 9 * r;
        W[e] = (r < m && c >= r) ? A[9 * (size_t)r + c] : 0.0;
    }
    if (lane == 0) { W[81] = cx1; W[82] = cy1; W[83] = s1; W[84] = cx2; W[85] = cy2; W[86] = s2; W[87] = 0.0; }
}

// 3 x 3 inverse as cofactors times the reciprocal of the determinant (Eigen's fixed-size path)
__device__ __forceinline__ void hom_inverse3(const double M[9], double R[9]) {
    const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const double det = M[0] * c00 + M[1] * c01 + M[2] * c02, id = 1.0 / det;
    R[0] = c00 * id; R[1] = (M[2] * M[7] - M[1] * M[8]) * id; R[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    R[3] = c01 * id; R[4] = (M[0] * M[8] - M[2] * M[6]) * id; R[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    R[6] = c02 * id; R[7] = (M[1] * M[6] - M[0] * M[7]) * id; R[8] = (M[0] * M[4] - M[1] * M[3]) * id;
}
__device__ __forceinline__ void hom_mul3(const double A[9], const double B[9], double C[9]) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}

// one lane per problem; R and V live in LDS element-major ([element][lane]: consecutive lanes, consecutive banks)
constexpr int DLT_FIN_THREADS = 64;
constexpr size_t DLT_FIN_LDS = sizeof(double) * 2 * 81 * DLT_FIN_THREADS;
__global__ __launch_bounds__(DLT_FIN_THREADS) void k_homography_dlt_finish(DltArgs a) {
    extern __shared__ double4 hom_lds_raw[];       // 2 x 81 x 64 doubles: above the static limit
    double (*Rm)[DLT_FIN_THREADS] = reinterpret_cast<double (*)[DLT_FIN_THREADS]>(hom_lds_raw);
    double (*Vm)[DLT_FIN_THREADS] = Rm + 81;
    const int lane = threadIdx.x, prob = blockIdx.x * DLT_FIN_THREADS + lane;
    const bool live = prob < a.n_problems && (a.off[prob + 1] - a.off[prob]) >= 4;
    if (!live) return;              // no barrier below: lanes are independent
    const double *W = a.W + (size_t)prob * DLT_W;
    double fro = 0.0;
    for (int e = 0; e < 81; ++e) {
        const double v = W[e];
        Rm[e][lane] = v; Vm[e][lane] = (e % 10 == 0) ? 1.0 : 0.0;
        fro += v * v;
    }
    const double tiny = 1e-28 * fro;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 8; ++p)
            for (int q = p + 1; q < 9; ++q) {
                double cp[9], cq[9], al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
                for (int i = 0; i < 9; ++i) { cp[i] = Rm[9 * i + p][lane]; cq[i] = Rm[9 * i + q][lane]; }
#pragma unroll
                for (int i = 0; i < 9; ++i) { al += cp[i] * cp[i]; be += cq[i] * cq[i]; ga += cp[i] * cq[i]; }
                if (!(fabs(ga) > 1e-15 * sqrt(al * be)) || al <= tiny || be <= tiny) continue;
                rotated = true;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    Rm[9 * i + p][lane] = c * cp[i] - sn * cq[i]; Rm[9 * i + q][lane] = sn * cp[i] + c * cq[i];
                    const double vp = Vm[9 * i + p][lane], vq = Vm[9 * i + q][lane];
                    Vm[9 * i + p][lane] = c * vp - sn * vq; Vm[9 * i + q][lane] = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    int best = 0;
    double least = 0.0;
    for (int j = 0; j < 9; ++j) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) { const double v = Rm[9 * i + j][lane]; s += v * v; }
        if (j == 0 || s < least) { least = s; best = j; }
    }
    double v[9], T1[9], T2[9], T2i[9], M[9], Hh[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = Vm[9 * i + best][lane];
    const double cx1 = W[81], cy1 = W[82], s1 = W[83], cx2 = W[84], cy2 = W[85], s2 = W[86];
    T1[0] = s1; T1[1] = 0.0; T1[2] = -s1 * cx1; T1[3] = 0.0; T1[4] = s1; T1[5] = -s1 * cy1; T1[6] = 0.0; T1[7] = 0.0; T1[8] = 1.0;
    T2[0] = s2; T2[1] = 0.0; T2[2] = -s2 * cx2; T2[3] = 0.0; T2[4] = s2; T2[5] = -s2 * cy2; T2[6] = 0.0; T2[7] = 0.0; T2[8] = 1.0;
    hom_inverse3(T2, T2i);
    hom_mul3(T2i, v, M);
    hom_mul3(M, T1, Hh);
    const double h22 = Hh[8];
    double *Ho = a.H + (size_t)prob * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) Ho[i] = Hh[i] / h22;
}

// HomographyEstimator::residuals (homography.cpp:95-110): one wave per problem, a point pair per lane and turn
__global__ __launch_bounds__(64) void k_homography_residuals(DltArgs a, double *res) {
    const int prob = blockIdx.x, lane = threadIdx.x;
    const int o0 = a.off[prob], N = a.off[prob + 1] - o0;
    double H[9], Hi[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) H[i] = a.H[(size_t)prob * 9 + i];
    hom_inverse3(H, Hi);
    for (int i = lane; i < N; i += 64) {
        const double x = a.x1[o0 + i], y = a.y1[o0 + i], z = a.x2[o0 + i], w = a.y2[o0 + i];
        const double w1 = H[6] * x + H[7] * y + H[8], w2 = Hi[6] * z + Hi[7] * w + Hi[8];
        const double e0 = (H[0] * x + H[1] * y + H[2]) / w1 - z, e1 = (H[3] * x + H[4] * y + H[5]) / w1 - w;
        const double f0 = (Hi[0] * z + Hi[1] * w + Hi[2]) / w2 - x, f1 = (Hi[3] * z + Hi[4] * w + Hi[5]) / w2 - y;
        res[o0 + i] = sqrt(e0 * e0 + e1 * e1) + sqrt(f0 * f0 + f1 * f1);
    }
}

// -----------------------------------------------------------------------------------------
// HomographyEstimator::solveRansac (/root/reference/modules/calib/src/homography.cpp:112-156) = cvx::util::RANSAC::estimate
// (modules/util/include/cvx/util/math/ransac.hpp:28-78) with the minimal model of four pairs, for every problem of a
// batch.  A trial's outcome -- its model, its inliers, their mean residual -- does not depend on the trials before it; only
// the loop's bookkeeping does (the best so far, and max_trials, which the first acceptable model usually cuts to a
// handful).  So all max_num_trials hypotheses of all problems are fitted and scored at once, and one lane per problem
// then walks the trials in order exactly as the loop would, stopping where it would:
//   k_hom_ransac_gather    the four pairs of every hypothesis, from the caller's subsets (what RNG::sample draws, rng.hpp:62-75)
//                          or drawn here from a seed (a counter-based generator: the reference seeds std::mt19937_64 from
//                          std::random_device, rng.cpp:6-9, so there is no sequence of its own to reproduce)
//   k_hom_ransac_fit4      the models: solve() on four pairs, one lane per hypothesis
//   k_hom_ransac_score     findInliers (homography.cpp:131-144): one lane per hypothesis, the residuals added in index order
//   k_hom_ransac_select    the loop of ransac.hpp:44-73 on the scores
//   k_hom_ransac_offsets   where each problem's inliers go in the refit batch (exclusive scan)
//   k_hom_ransac_compact   best_subset, in index order, as a new batch; the refit (:77) is a DLT on that batch
// -----------------------------------------------------------------------------------------
struct RansacArgs {
    const int *off;
    const float *x1, *y1, *x2, *y2;
    int n_problems, T;                       // T = max_num_trials
    const int *subsets;                      // [n_problems][T][4] or null
    unsigned long long seed;
    double max_error, log_probability;
    int min_inliers;
    float *hx1, *hy1, *hx2, *hy2;            // [4 M], M = n_problems * T
    int *subsets_used;                       // [M][4]
    const double *hyp_H;                     // [M][9]
    int *cnt;                                // [M]
    double *mean;                            // [M]
    int *best, *trials_run, *n_inl;          // [n_problems]
    int *flags;                              // [0]: a subset index out of range or repeated
    // the refit batch
    int *new_off;                            // [n_problems + 1]
    float *nx1, *ny1, *nx2, *ny2;
    unsigned char *mask;                     // [total] (zeroed by the caller)
};

__device__ __forceinline__ unsigned long long hom_mix64(unsigned long long z) {      // splitmix64's output function
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_hom_ransac_gather(RansacArgs a) {
    const long long M = (long long)a.n_problems * a.T;
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const int prob = (int)(m / a.T);
    const int o0 = a.off[prob], N = a.off[prob + 1] - o0;
    int sub[4] = {0, 0, 0, 0};
    bool ok = N >= 4;
    if (ok && a.subsets) {
#pragma unroll
        for (int k = 0; k < 4; ++k) sub[k] = a.subsets[4 * m + k];
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bad = bad || sub[k] < 0 || sub[k] >= N;
#pragma unroll
            for (int j = 0; j < k; ++j) bad = bad || sub[j] == sub[k];
        }
        if (bad) { a.flags[0] = 1; ok = false; }
    } else if (ok) {
        // RNG::sample's swaps (rng.hpp:66-74) on an index vector that is never made: the few entries a draw has moved
        int pos[8], val[8], nm = 0;
        auto get = [&](int i) { int v = i; for (int j = 0; j < nm; ++j) if (pos[j] == i) v = val[j]; return v; };
        auto set = [&](int i, int v) {
            for (int j = 0; j < nm; ++j) if (pos[j] == i) { val[j] = v; return; }
            pos[nm] = i; val[nm] = v; ++nm;
        };
        for (int k = 0; k < 4; ++k) {
            const int max_idx = N - 1 - k;
            const unsigned long long r = hom_mix64(a.seed + 0x9E3779B97F4A7C15ULL * (unsigned long long)(4 * m + k + 1));
            const int index = (int)__umul64hi(r, (unsigned long long)(max_idx + 1));       // uniform on [0, max_idx]
            const int vi = get(index), vm = get(max_idx);
            set(index, vm); set(max_idx, vi);
            sub[k] = vi;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = o0 + sub[k];
        a.hx1[4 * m + k] = ok ? a.x1[i] : 0.f; a.hy1[4 * m + k] = ok ? a.y1[i] : 0.f;
        a.hx2[4 * m + k] = ok ? a.x2[i] : 0.f; a.hy2[4 * m + k] = ok ? a.y2[i] : 0.f;
        if (a.subsets_used) a.subsets_used[4 * m + k] = ok ? sub[k] : -1;
    }
}

// The model of a trial: HomographyEstimator::solve on FOUR pairs, one lane per hypothesis, everything in registers.
// normalize_points and the constraint matrix as k_homography_dlt_reduce makes them (the same float arithmetic in the same
// order); the null vector of the 8 x 9 matrix -- for a full-rank matrix the same direction whatever finds it -- by
// Householder QR with column pivoting and a back substitution with the column left over set to one (rank revealing: the
// free column is the most dependent one, so the triangle it leaves is as well conditioned as the subset allows); swaps are
// selects, so that no index is dynamic.  (The general kernels on a batch of a million four-pair problems: 21 ms of
// a 25 ms call -- a wave and 83 KB of LDS per 64 problems are sized for hundreds of pairs; this: 0.3 ms.)
template <int k>
__device__ __forceinline__ void hom_qrcp_step(double (&A)[8][9], int (&perm)[9]) {
    double best = -1.0;
    int p = k;
#pragma unroll
    for (int j = k; j < 9; ++j) {
        double ss = 0.0;
#pragma unroll
        for (int i = k; i < 8; ++i) ss += A[i][j] * A[i][j];
        if (ss > best) { best = ss; p = j; }
    }
#pragma unroll
    for (int j = k + 1; j < 9; ++j) {
        const bool sw = (j == p);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double t = A[i][k], u = A[i][j];
            A[i][k] = sw ? u : t; A[i][j] = sw ? t : u;
        }
        const int tp = perm[k], up = perm[j];
        perm[k] = sw ? up : tp; perm[j] = sw ? tp : up;
    }
    const double akk = A[k][k];
    if (best > 0.0) {
        const double alpha = akk > 0.0 ? -sqrt(best) : sqrt(best);
        const double beta = 1.0 / (best - akk * alpha);
        const double vk = akk - alpha;
#pragma unroll
        for (int j = k + 1; j < 9; ++j) {
            double d = vk * A[k][j];
#pragma unroll
            for (int i = k + 1; i < 8; ++i) d += A[i][k] * A[i][j];
            d *= beta;
            A[k][j] -= d * vk;
#pragma unroll
            for (int i = k + 1; i < 8; ++i) A[i][j] -= d * A[i][k];
        }
        A[k][k] = alpha;
    }
}

__device__ __forceinline__ void hom_null_vector_8x9(double (&A)[8][9], double v[9]) {
    int perm[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) perm[c] = c;
    // (eight instantiations, not a loop: every index of A must be a constant for A to live in registers)
    hom_qrcp_step<0>(A, perm); hom_qrcp_step<1>(A, perm); hom_qrcp_step<2>(A, perm); hom_qrcp_step<3>(A, perm);
    hom_qrcp_step<4>(A, perm); hom_qrcp_step<5>(A, perm); hom_qrcp_step<6>(A, perm); hom_qrcp_step<7>(A, perm);
    double w[9];
    w[8] = 1.0;
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        double sum = 0.0;
#pragma unroll
        for (int j = i + 1; j < 9; ++j) sum += A[i][j] * w[j];
        w[i] = -sum / A[i][i];
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) v[c] = 0.0;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int c = 0; c < 9; ++c) v[c] = (perm[j] == c) ? w[j] : v[c];
}

__global__ __launch_bounds__(64) void k_hom_ransac_fit4(RansacArgs a, double *hyp_H) {
    const long long M = (long long)a.n_problems * a.T;
    const long long m = (long long)blockIdx.x * 64 + threadIdx.x;
    if (m >= M) return;
    const float4 X1 = reinterpret_cast<const float4 *>(a.hx1)[m], Y1 = reinterpret_cast<const float4 *>(a.hy1)[m];
    const float4 X2 = reinterpret_cast<const float4 *>(a.hx2)[m], Y2 = reinterpret_cast<const float4 *>(a.hy2)[m];
    const float px[4][4] = {{X1.x, X1.y, X1.z, X1.w}, {Y1.x, Y1.y, Y1.z, Y1.w}, {X2.x, X2.y, X2.z, X2.w}, {Y2.x, Y2.y, Y2.z, Y2.w}};
    float ctr[4];
    double scale[2];
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int q = 0; q < 4; ++q) {           // homography.cpp:20-24: the sums in point order, in float
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) sum = sum + px[q][i];
            ctr[q] = sum / 4.0f;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {           // :26-32
            float nrm = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dx = px[2 * q][i] - ctr[2 * q], dy = px[2 * q + 1][i] - ctr[2 * q + 1];
                const float xx = dx * dx, yy = dy * dy;
                const float d = xx + yy;
                nrm = nrm + d;
            }
            nrm = nrm / 4.0f;
            scale[q] = sqrt(2.0) / (double)sqrtf(nrm);
        }
    }
    double A[8][9];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double u1 = (double)(float)((double)(px[0][i] - ctr[0]) * scale[0]), v1 = (double)(float)((double)(px[1][i] - ctr[1]) * scale[0]);
        const double u2 = (double)(float)((double)(px[2][i] - ctr[2]) * scale[1]), v2 = (double)(float)((double)(px[3][i] - ctr[3]) * scale[1]);
        double *r0 = A[2 * i], *r1 = A[2 * i + 1];
        r0[0] = u1; r0[1] = v1; r0[2] = 1.0; r0[3] = 0.0; r0[4] = 0.0; r0[5] = 0.0; r0[6] = -u2 * u1; r0[7] = -u2 * v1; r0[8] = -u2;
        r1[0] = 0.0; r1[1] = 0.0; r1[2] = 0.0; r1[3] = -u1; r1[4] = -v1; r1[5] = -1.0; r1[6] = v2 * u1; r1[7] = v2 * v1; r1[8] = v2;
    }
    double v[9], T1[9], T2[9], T2i[9], Mx[9], Hh[9];
    hom_null_vector_8x9(A, v);
    const double s1 = scale[0], s2 = scale[1], cx1 = ctr[0], cy1 = ctr[1], cx2 = ctr[2], cy2 = ctr[3];
    T1[0] = s1; T1[1] = 0.0; T1[2] = -s1 * cx1; T1[3] = 0.0; T1[4] = s1; T1[5] = -s1 * cy1; T1[6] = 0.0; T1[7] = 0.0; T1[8] = 1.0;
    T2[0] = s2; T2[1] = 0.0; T2[2] = -s2 * cx2; T2[3] = 0.0; T2[4] = s2; T2[5] = -s2 * cy2; T2[6] = 0.0; T2[7] = 0.0; T2[8] = 1.0;
    hom_inverse3(T2, T2i);
    hom_mul3(T2i, v, Mx);
    hom_mul3(Mx, T1, Hh);
    const double h22 = Hh[8];
#pragma unroll
    for (int i = 0; i < 9; ++i) hyp_H[9 * m + i] = Hh[i] / h22;
}

// the residual of homography.cpp:105-108 for one pair
__device__ __forceinline__ double hom_transfer_residual(const double H[9], const double Hi[9], double x, double y, double z, double w) {
    const double w1 = H[6] * x + H[7] * y + H[8], w2 = Hi[6] * z + Hi[7] * w + Hi[8];
    const double e0 = (H[0] * x + H[1] * y + H[2]) / w1 - z, e1 = (H[3] * x + H[4] * y + H[5]) / w1 - w;
    const double f0 = (Hi[0] * z + Hi[1] * w + Hi[2]) / w2 - x, f1 = (Hi[3] * z + Hi[4] * w + Hi[5]) / w2 - y;
    return sqrt(e0 * e0 + e1 * e1) + sqrt(f0 * f0 + f1 * f1);
}

__global__ __launch_bounds__(256) void k_hom_ransac_score(RansacArgs a) {
    const long long M = (long long)a.n_problems * a.T;
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const int prob = (int)(m / a.T);
    const int o0 = a.off[prob], N = a.off[prob + 1] - o0;
    int cnt = 0;
    double total = 0.0;
    if (N >= 4) {
        double H[9], Hi[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) H[i] = a.hyp_H[9 * m + i];
        hom_inverse3(H, Hi);
        for (int i = 0; i < N; ++i) {       // (a wave's lanes belong to one or two problems: the loads are broadcasts)
            const double r = hom_transfer_residual(H, Hi, a.x1[o0 + i], a.y1[o0 + i], a.x2[o0 + i], a.y2[o0 + i]);
            if (r < a.max_error) { ++cnt; total += r; }
        }
    }
    a.cnt[m] = cnt;
    a.mean[m] = total / (double)cnt;        // 0 / 0 with no inlier, as the reference (:142); never used then (ransac.hpp:55)
}

__global__ __launch_bounds__(256) void k_hom_ransac_select(RansacArgs a) {
    const int prob = blockIdx.x * 256 + threadIdx.x;
    if (prob >= a.n_problems) return;
    const int N = a.off[prob + 1] - a.off[prob];
    int best = -1, nbest = 0;
    size_t trial = 0;
    if (N >= 4) {
        const double one_over_indices = 1.0 / (double)N;
        const size_t max_num = (size_t)a.T;
        size_t max_trials = max_num;
        double min_residual = DBL_MAX;
        while (trial < (max_num < max_trials ? max_num : max_trials)) {
            const long long m = (long long)prob * a.T + (long long)trial;
            const int c = a.cnt[m];
            const double r = a.mean[m];
            if (c > a.min_inliers && r < min_residual) {
                min_residual = r; best = (int)trial; nbest = c;
                const double w = (double)c * one_over_indices;
                double p_no_outliers = 1.0 - pow(w, 4.0);
                p_no_outliers = p_no_outliers > DBL_EPSILON ? p_no_outliers : DBL_EPSILON;
                p_no_outliers = p_no_outliers < 1.0 - DBL_EPSILON ? p_no_outliers : 1.0 - DBL_EPSILON;
                max_trials = (size_t)(a.log_probability / log(p_no_outliers));
            }
            ++trial;
        }
    }
    a.best[prob] = best; a.n_inl[prob] = nbest; a.trials_run[prob] = (int)trial;
}

// exclusive scan of n_inl (one workgroup of 1024: a contiguous run per thread, the runs' sums scanned by thread 0)
__global__ __launch_bounds__(1024) void k_hom_ransac_offsets(RansacArgs a) {
    __shared__ int run[1024];
    const int n = a.n_problems, per = (n + 1023) / 1024;
    const int i0 = min(n, (int)threadIdx.x * per), i1 = min(n, i0 + per);
    int s = 0;
    for (int i = i0; i < i1; ++i) s += a.n_inl[i];
    run[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int t = 0; t < 1024; ++t) { const int v = run[t]; run[t] = acc; acc += v; }
        a.new_off[n] = acc;
    }
    __syncthreads();
    int acc = run[threadIdx.x];
    for (int i = i0; i < i1; ++i) { a.new_off[i] = acc; acc += a.n_inl[i]; }
}

__global__ __launch_bounds__(64) void k_hom_ransac_compact(RansacArgs a) {
    const int prob = blockIdx.x, lane = threadIdx.x;
    const int best = a.best[prob];
    if (best < 0) return;
    const int o0 = a.off[prob], N = a.off[prob + 1] - o0;
    double H[9], Hi[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) H[i] = a.hyp_H[9 * ((long long)prob * a.T + best) + i];
    hom_inverse3(H, Hi);
    int base = a.new_off[prob];
    for (int i0 = 0; i0 < N; i0 += 64) {
        const int i = i0 + lane;
        float x = 0.f, y = 0.f, z = 0.f, w = 0.f;
        bool in = false;
        if (i < N) {
            x = a.x1[o0 + i]; y = a.y1[o0 + i]; z = a.x2[o0 + i]; w = a.y2[o0 + i];
            in = hom_transfer_residual(H, Hi, x, y, z, w) < a.max_error;
            a.mask[o0 + i] = in ? 1 : 0;
        }
        const unsigned long long b = __ballot(in);
        if (in) {
            const int p = base + __popcll(b & ((1ULL << lane) - 1ULL));
            a.nx1[p] = x; a.ny1[p] = y; a.nx2[p] = z; a.ny2[p] = w;
        }
        base += __popcll(b);
    }
}

}  // namespace cvxhom
