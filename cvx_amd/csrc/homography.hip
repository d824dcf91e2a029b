// Batched homography refinement: handle + C ABI around k_homography_lm.
// Replaces HomographyEstimator::solveLM (/root/reference/modules/calib/src/homography.cpp:218-236) for many
// independent views at once.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include "common.h"
#include "homography_kernels.h"

using namespace cvxhom;

struct cvx_homography {
    int n = 0, device = 0, max_points = 0;
    long long total = 0;             // point pairs in use
    size_t tp = 1;                   // stride of the four coordinate planes in d_pts (>= total: a batch built on the device
                                     // -- the inliers of cvx_homography_ransac -- is sized before its total is known)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int *d_off = nullptr, *d_ret = nullptr;
    float *d_pts = nullptr;          // x1 | y1 | x2 | y2, `total` each
    double *d_H = nullptr, *d_info = nullptr;
    double *d_W = nullptr, *d_res = nullptr;     // DLT work records / transfer residuals, made on first use
    char *d_ransac = nullptr;                    // cvx_homography_ransac's work block, kept for the next call (180 B per hypothesis)
    size_t ransac_cap = 0;
    size_t lds_bytes = 0;
    ~cvx_homography() {
        (void)hipSetDevice(device);
        // d_pts heads the handle's one device block (offsets, estimates, info, return values behind it)
        for (void *p : {(void *)d_pts, (void *)d_W, (void *)d_res, (void *)d_ransac})
            if (p) (void)hipFree(p);
        if (own_stream) cvx::own_stream_end(device, stream);
    }
};

// a handle with room for `capacity` point pairs; nothing uploaded, H zeroed
static int make_handle(int n_problems, long long capacity, int max_points, int device, void *stream, cvx_homography **out) {
    std::unique_ptr<cvx_homography> h(new cvx_homography);
    h->n = n_problems; h->device = device; h->max_points = max_points; h->total = capacity;
    if (stream) h->stream = (hipStream_t)stream;
    else { h->own_stream = true; const int sst = cvx::own_stream_begin(h->device, &h->stream); if (sst) return sst; }
    h->lds_bytes = hom_lds_doubles(max_points < 4 ? 4 : max_points) * sizeof(double);
    const size_t tp = (size_t)(capacity > 0 ? capacity : 1);
    h->tp = tp;
    {
        // one device block, carved on 256-byte boundaries: points | H | info | offsets | return values
        auto up = [](size_t b) { return (b + 255) / 256 * 256; };
        const size_t b_pts = up(sizeof(float) * 4 * tp), b_H = up(sizeof(double) * 9 * n_problems), b_info = up(sizeof(double) * 10 * n_problems),
                     b_off = up(sizeof(int) * ((size_t)n_problems + 1)), b_ret = up(sizeof(int) * (size_t)n_problems);
        char *base = nullptr;
        CVX_HIP(hipMalloc((void **)&base, b_pts + b_H + b_info + b_off + b_ret));
        h->d_pts = reinterpret_cast<float *>(base); base += b_pts;
        h->d_H = reinterpret_cast<double *>(base); base += b_H;
        h->d_info = reinterpret_cast<double *>(base); base += b_info;
        h->d_off = reinterpret_cast<int *>(base); base += b_off;
        h->d_ret = reinterpret_cast<int *>(base);
    }
    CVX_HIP(hipMemsetAsync(h->d_H, 0, sizeof(double) * 9 * n_problems, h->stream));
    *out = h.release();
    return CVX_OK;
}

extern "C" int cvx_homography_create(int n_problems, const int *off, const float *x1, const float *y1,
                                     const float *x2, const float *y2, int device, void *stream,
                                     cvx_homography **out) {
    if (!out) return CVX_ERR_INVALID_ARG;
    *out = nullptr;
    if (n_problems <= 0 || !off || off[0] != 0) { cvx::set_last_error("homography: need n_problems > 0 and off[0] == 0"); return CVX_ERR_INVALID_ARG; }
    int max_points = 0;
    for (int i = 0; i < n_problems; ++i) {
        const int c = off[i + 1] - off[i];
        if (c < 0) { cvx::set_last_error("homography: offsets must not decrease"); return CVX_ERR_INVALID_ARG; }
        if (c > max_points) max_points = c;
    }
    const long long total = off[n_problems];
    if (total > 0 && (!x1 || !y1 || !x2 || !y2)) return CVX_ERR_INVALID_ARG;
    if (max_points > CVX_HOMOGRAPHY_MAX_POINTS) {
        cvx::set_last_error("homography: more than CVX_HOMOGRAPHY_MAX_POINTS point pairs in one problem");
        return CVX_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { cvx::set_last_error("no HIP device"); return CVX_ERR_NO_DEVICE; }
    CVX_HIP(hipSetDevice(device));
    cvx_homography *raw = nullptr;
    const int mst = make_handle(n_problems, total, max_points, device, stream, &raw);
    if (mst) return mst;
    std::unique_ptr<cvx_homography> h(raw);
    const size_t tp = h->tp;
    CVX_HIP(hipMemcpyAsync(h->d_off, off, sizeof(int) * (n_problems + 1), hipMemcpyHostToDevice, h->stream));
    if (total > 0) {
        const float *src[4] = {x1, y1, x2, y2};
        for (int k = 0; k < 4; ++k)
            CVX_HIP(hipMemcpyAsync(h->d_pts + k * tp, src[k], sizeof(float) * total, hipMemcpyHostToDevice, h->stream));
    }
    CVX_HIP(hipStreamSynchronize(h->stream));
    *out = h.release();
    return CVX_OK;
}

extern "C" void cvx_homography_destroy(cvx_homography *h) { delete h; }

extern "C" int cvx_homography_set(cvx_homography *h, const double *H) {
    if (!h || !H) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    CVX_HIP(hipMemcpyAsync(h->d_H, H, sizeof(double) * 9 * h->n, hipMemcpyHostToDevice, h->stream));
    CVX_HIP(hipStreamSynchronize(h->stream));
    return CVX_OK;
}

// launches the batch on the handle's stream and waits for it
extern "C" int cvx_homography_run(cvx_homography *h, int max_iter, const double *opts) {
    if (!h || max_iter < 0) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    const size_t tp = h->tp;
    HomArgs a;
    a.off = h->d_off;
    a.x1 = h->d_pts; a.y1 = h->d_pts + tp; a.x2 = h->d_pts + 2 * tp; a.y2 = h->d_pts + 3 * tp;
    a.H = h->d_H; a.info = h->d_info; a.ret = h->d_ret;
    a.n_problems = h->n; a.itmax = max_iter;
    // LMSolver::minimizeDer (lm.hpp:61): {factor_, g_tol_, x_tol_, g_tol_} = {1e-3, eps, eps, eps}
    a.tau = opts ? opts[0] : 1e-3;
    a.eps1 = opts ? opts[1] : DBL_EPSILON;
    a.eps2 = opts ? opts[2] : DBL_EPSILON;
    a.eps3 = opts ? opts[3] : DBL_EPSILON;
    a.prof = nullptr;
#ifdef HOM_PROF
    CVX_HIP(hipMalloc(&a.prof, sizeof(long long) * 5 * h->n));
#endif
    CVX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_homography_lm),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
    hipLaunchKernelGGL(k_homography_lm, dim3(h->n), dim3(64), h->lds_bytes, h->stream, a);
    CVX_HIP(hipGetLastError());
    CVX_HIP(hipStreamSynchronize(h->stream));
#ifdef HOM_PROF
    {
        std::vector<long long> pr((size_t)5 * h->n);
        CVX_HIP(hipMemcpy(pr.data(), a.prof, sizeof(long long) * pr.size(), hipMemcpyDeviceToHost));
        double s[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < h->n; ++i) for (int k = 0; k < 5; ++k) s[k] += (double)pr[(size_t)i * 5 + k];
        std::fprintf(stderr, "HOM_PROF mean cycles per problem: eval %.0f sums %.0f norm %.0f lu %.0f total %.0f\n", s[0] / h->n, s[1] / h->n, s[2] / h->n, s[3] / h->n, s[4] / h->n);
        (void)hipFree(a.prof);
    }
#endif
    return CVX_OK;
}

extern "C" int cvx_homography_get(cvx_homography *h, double *H, double *info, int *ret) {
    if (!h) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    if (H) CVX_HIP(hipMemcpyAsync(H, h->d_H, sizeof(double) * 9 * h->n, hipMemcpyDeviceToHost, h->stream));
    if (info) CVX_HIP(hipMemcpyAsync(info, h->d_info, sizeof(double) * 10 * h->n, hipMemcpyDeviceToHost, h->stream));
    if (ret) CVX_HIP(hipMemcpyAsync(ret, h->d_ret, sizeof(int) * h->n, hipMemcpyDeviceToHost, h->stream));
    CVX_HIP(hipStreamSynchronize(h->stream));
    return CVX_OK;
}

extern "C" int cvx_homography_solve(cvx_homography *h, double *H, int max_iter, const double *opts, double *info,
                                    int *ret) {
    int st = cvx_homography_set(h, H);
    if (st) return st;
    st = cvx_homography_run(h, max_iter, opts);
    if (st) return st;
    return cvx_homography_get(h, H, info, ret);
}

static DltArgs dlt_args(cvx_homography *h) {
    const size_t tp = h->tp;
    DltArgs a;
    a.off = h->d_off;
    a.x1 = h->d_pts; a.y1 = h->d_pts + tp; a.x2 = h->d_pts + 2 * tp; a.y2 = h->d_pts + 3 * tp;
    a.W = h->d_W; a.H = h->d_H; a.n_problems = h->n;
    return a;
}

// HomographyEstimator::solve (homography.cpp:46-93) for every problem of the handle; the estimates stay on the device
// as the start of cvx_homography_run
// the two launches of the normalised DLT over a batch described by `a`, no problem larger than max_points
static int launch_dlt(const DltArgs &a, int max_points, hipStream_t stream) {
    const size_t lds = dlt_lds_doubles(max_points < 4 ? 4 : max_points) * sizeof(double);
    CVX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_homography_dlt_reduce),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CVX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_homography_dlt_finish),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)DLT_FIN_LDS));
    hipLaunchKernelGGL(k_homography_dlt_reduce, dim3(a.n_problems), dim3(64), lds, stream, a);
    hipLaunchKernelGGL(k_homography_dlt_finish, dim3((a.n_problems + DLT_FIN_THREADS - 1) / DLT_FIN_THREADS), dim3(DLT_FIN_THREADS),
                       DLT_FIN_LDS, stream, a);
    CVX_HIP(hipGetLastError());
    return CVX_OK;
}

extern "C" int cvx_homography_dlt(cvx_homography *h, double *H) {
    if (!h) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    if (!h->d_W) CVX_HIP(hipMalloc(&h->d_W, sizeof(double) * DLT_W * h->n));
    const int lst = launch_dlt(dlt_args(h), h->max_points, h->stream);
    if (lst) return lst;
    if (H) CVX_HIP(hipMemcpyAsync(H, h->d_H, sizeof(double) * 9 * h->n, hipMemcpyDeviceToHost, h->stream));
    CVX_HIP(hipStreamSynchronize(h->stream));
    return CVX_OK;
}

// HomographyEstimator::residuals (homography.cpp:95-110) of every point pair under its problem's estimate
extern "C" int cvx_homography_residuals(cvx_homography *h, const double *H, double *res) {
    if (!h || !res) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    if (h->total == 0) return CVX_OK;
    if (!h->d_res) CVX_HIP(hipMalloc(&h->d_res, sizeof(double) * h->total));
    if (H) CVX_HIP(hipMemcpyAsync(h->d_H, H, sizeof(double) * 9 * h->n, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_homography_residuals, dim3(h->n), dim3(64), 0, h->stream, dlt_args(h), h->d_res);
    CVX_HIP(hipGetLastError());
    CVX_HIP(hipMemcpyAsync(res, h->d_res, sizeof(double) * h->total, hipMemcpyDeviceToHost, h->stream));
    CVX_HIP(hipStreamSynchronize(h->stream));
    return CVX_OK;
}

extern "C" void cvx_ransac_default_params(cvx_ransac_params *p) {
    if (!p) return;
    p->max_error = 0.1; p->outlier_probability = 0.1; p->max_num_trials = 100; p->min_inliers = 5;     // ransac.hpp:14-22
}

// HomographyEstimator::solveRansac (homography.cpp:151-156) for every problem of the handle: homography_kernels.h
extern "C" int cvx_homography_ransac(cvx_homography *h, const cvx_ransac_params *params, const int *subsets,
                                     unsigned long long seed, double *H, int *n_inliers, unsigned char *inlier_mask,
                                     int *trials_run, int *subsets_used, cvx_homography **inlier_batch) {
    if (inlier_batch) *inlier_batch = nullptr;
    if (!h) return CVX_ERR_INVALID_ARG;
    cvx_ransac_params p;
    cvx_ransac_default_params(&p);
    if (params) p = *params;
    if (p.max_num_trials < 1 || p.min_inliers < 0 || !(p.outlier_probability >= 0.0 && p.outlier_probability < 1.0) || !(p.max_error == p.max_error)) {
        cvx::set_last_error("ransac: max_num_trials >= 1, min_inliers >= 0, 0 <= outlier_probability < 1");
        return CVX_ERR_INVALID_ARG;
    }
    const long long M = (long long)h->n * p.max_num_trials;
    if (M > CVX_RANSAC_MAX_HYPOTHESES) {
        cvx::set_last_error("ransac: problems x max_num_trials beyond CVX_RANSAC_MAX_HYPOTHESES (run the batch in parts)");
        return CVX_ERR_UNSUPPORTED;
    }
    CVX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const size_t n = (size_t)h->n, Mz = (size_t)M, tot = (size_t)(h->total > 0 ? h->total : 1);
    // the refit batch: room for every pair (its total is known only after the selection)
    cvx_homography *raw = nullptr;
    // (on the caller's stream if the source handle runs on one, else on a stream of its own: it may outlive the source)
    int st = make_handle(h->n, h->total, h->max_points, h->device, h->own_stream ? nullptr : (void *)h->stream, &raw);
    if (st) return st;
    std::unique_ptr<cvx_homography> nb(raw);
    if (nb->stream != s) CVX_HIP(hipStreamSynchronize(nb->stream));      // its zeroed estimates, before this stream writes them
    CVX_HIP(hipMemcpyAsync(nb->d_H, h->d_H, sizeof(double) * 9 * n, hipMemcpyDeviceToDevice, s));      // no model: the estimate stays
    CVX_HIP(hipMalloc(&nb->d_W, sizeof(double) * DLT_W * n));
    // everything of the hypothesis batch in one block
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_pts = up(sizeof(float) * 16 * Mz), b_H = up(sizeof(double) * 9 * Mz), b_cnt = up(sizeof(int) * Mz), b_mean = up(sizeof(double) * Mz),
                 b_sub = up(sizeof(int) * 4 * Mz), b_per = up(sizeof(int) * n), b_mask = up(tot), b_flags = 256;
    const size_t need = b_pts + b_H + b_cnt + b_mean + 2 * b_sub + 3 * b_per + b_mask + b_flags;
    if (h->ransac_cap < need) {
        if (h->d_ransac) { CVX_HIP(hipStreamSynchronize(s)); (void)hipFree(h->d_ransac); h->d_ransac = nullptr; h->ransac_cap = 0; }
        CVX_HIP(hipMalloc((void **)&h->d_ransac, need));
        h->ransac_cap = need;
    }
    char *q = h->d_ransac;
    RansacArgs a;
    a.off = h->d_off;
    a.x1 = h->d_pts; a.y1 = h->d_pts + h->tp; a.x2 = h->d_pts + 2 * h->tp; a.y2 = h->d_pts + 3 * h->tp;
    a.n_problems = h->n; a.T = p.max_num_trials;
    a.seed = seed;
    a.max_error = p.max_error; a.log_probability = std::log(1.0 - p.outlier_probability); a.min_inliers = p.min_inliers;
    a.hx1 = reinterpret_cast<float *>(q); a.hy1 = a.hx1 + 4 * Mz; a.hx2 = a.hx1 + 8 * Mz; a.hy2 = a.hx1 + 12 * Mz; q += b_pts;
    double *hyp_H = reinterpret_cast<double *>(q); q += b_H;
    a.hyp_H = hyp_H;
    a.cnt = reinterpret_cast<int *>(q); q += b_cnt;
    a.mean = reinterpret_cast<double *>(q); q += b_mean;
    int *d_subsets = reinterpret_cast<int *>(q); q += b_sub;
    a.subsets_used = reinterpret_cast<int *>(q); q += b_sub;
    a.best = reinterpret_cast<int *>(q); q += b_per;
    a.trials_run = reinterpret_cast<int *>(q); q += b_per;
    a.n_inl = reinterpret_cast<int *>(q); q += b_per;
    a.mask = reinterpret_cast<unsigned char *>(q); q += b_mask;
    a.flags = reinterpret_cast<int *>(q);
    a.subsets = nullptr;
    if (subsets) {
        CVX_HIP(hipMemcpyAsync(d_subsets, subsets, sizeof(int) * 4 * Mz, hipMemcpyHostToDevice, s));
        a.subsets = d_subsets;
    }
    if (!subsets_used) a.subsets_used = nullptr;
    a.new_off = nb->d_off;
    a.nx1 = nb->d_pts; a.ny1 = nb->d_pts + nb->tp; a.nx2 = nb->d_pts + 2 * nb->tp; a.ny2 = nb->d_pts + 3 * nb->tp;
    CVX_HIP(hipMemsetAsync(a.flags, 0, sizeof(int), s));
    CVX_HIP(hipMemsetAsync(a.mask, 0, tot, s));
    hipLaunchKernelGGL(k_hom_ransac_gather, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_hom_ransac_fit4, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, s, a, hyp_H);
    hipLaunchKernelGGL(k_hom_ransac_score, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_hom_ransac_select, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_hom_ransac_offsets, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(k_hom_ransac_compact, dim3(h->n), dim3(64), 0, s, a);
    CVX_HIP(hipGetLastError());
    // the model of the best trial's inliers (ransac.hpp:77)
    st = launch_dlt(dlt_args(nb.get()), nb->max_points, s);
    if (st) return st;
    int flag = 0;
    std::vector<int> counts(n);
    CVX_HIP(hipMemcpyAsync(&flag, a.flags, sizeof(int), hipMemcpyDeviceToHost, s));
    CVX_HIP(hipMemcpyAsync(counts.data(), a.n_inl, sizeof(int) * n, hipMemcpyDeviceToHost, s));
    if (H) CVX_HIP(hipMemcpyAsync(H, nb->d_H, sizeof(double) * 9 * n, hipMemcpyDeviceToHost, s));
    if (inlier_mask && h->total > 0) CVX_HIP(hipMemcpyAsync(inlier_mask, a.mask, (size_t)h->total, hipMemcpyDeviceToHost, s));
    if (trials_run) CVX_HIP(hipMemcpyAsync(trials_run, a.trials_run, sizeof(int) * n, hipMemcpyDeviceToHost, s));
    if (subsets_used) CVX_HIP(hipMemcpyAsync(subsets_used, a.subsets_used, sizeof(int) * 4 * Mz, hipMemcpyDeviceToHost, s));
    CVX_HIP(hipStreamSynchronize(s));
    if (flag) { cvx::set_last_error("ransac: a subset names a pair outside its problem, or the same pair twice"); return CVX_ERR_INVALID_ARG; }
    long long used = 0;
    for (size_t i = 0; i < n; ++i) used += counts[i];
    nb->total = used;
    if (n_inliers) std::memcpy(n_inliers, counts.data(), sizeof(int) * n);
    if (inlier_batch) *inlier_batch = nb.release();
    return CVX_OK;
}

