// Batched homography refinement: handle + C ABI around k_homography_lm.
// Replaces HomographyEstimator::solveLM (/root/reference/modules/calib/src/homography.cpp:218-236) for many
// independent views at once.
#include <cfloat>
#include <cstring>
#include <memory>
#include <vector>

#include "common.h"
#include "homography_kernels.h"

using namespace cvxhom;

struct cvx_homography {
    int n = 0, device = 0, max_points = 0;
    long long total = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int *d_off = nullptr, *d_ret = nullptr;
    float *d_pts = nullptr;          // x1 | y1 | x2 | y2, `total` each
    double *d_H = nullptr, *d_info = nullptr;
    double *d_W = nullptr, *d_res = nullptr;     // DLT work records / transfer residuals, made on first use
    size_t lds_bytes = 0;
    ~cvx_homography() {
        (void)hipSetDevice(device);
        // d_pts heads the handle's one device block (offsets, estimates, info, return values behind it)
        for (void *p : {(void *)d_pts, (void *)d_W, (void *)d_res})
            if (p) (void)hipFree(p);
        if (own_stream) cvx::own_stream_end(device, stream);
    }
};

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
    std::unique_ptr<cvx_homography> h(new cvx_homography);
    h->n = n_problems; h->device = device; h->max_points = max_points; h->total = total;
    if (stream) h->stream = (hipStream_t)stream;
    else { h->own_stream = true; const int sst = cvx::own_stream_begin(h->device, &h->stream); if (sst) return sst; }
    h->lds_bytes = hom_lds_doubles(max_points < 4 ? 4 : max_points) * sizeof(double);
    const size_t tp = (size_t)(total > 0 ? total : 1);
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
    const size_t tp = (size_t)(h->total > 0 ? h->total : 1);
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
    const size_t tp = (size_t)(h->total > 0 ? h->total : 1);
    DltArgs a;
    a.off = h->d_off;
    a.x1 = h->d_pts; a.y1 = h->d_pts + tp; a.x2 = h->d_pts + 2 * tp; a.y2 = h->d_pts + 3 * tp;
    a.W = h->d_W; a.H = h->d_H; a.n_problems = h->n;
    return a;
}

// HomographyEstimator::solve (homography.cpp:46-93) for every problem of the handle; the estimates stay on the device
// as the start of cvx_homography_run
extern "C" int cvx_homography_dlt(cvx_homography *h, double *H) {
    if (!h) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    if (!h->d_W) CVX_HIP(hipMalloc(&h->d_W, sizeof(double) * DLT_W * h->n));
    const DltArgs a = dlt_args(h);
    const size_t lds = dlt_lds_doubles(h->max_points < 4 ? 4 : h->max_points) * sizeof(double);
    CVX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_homography_dlt_reduce),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CVX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_homography_dlt_finish),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)DLT_FIN_LDS));
    hipLaunchKernelGGL(k_homography_dlt_reduce, dim3(h->n), dim3(64), lds, h->stream, a);
    hipLaunchKernelGGL(k_homography_dlt_finish, dim3((h->n + DLT_FIN_THREADS - 1) / DLT_FIN_THREADS), dim3(DLT_FIN_THREADS),
                       DLT_FIN_LDS, h->stream, a);
    CVX_HIP(hipGetLastError());
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
