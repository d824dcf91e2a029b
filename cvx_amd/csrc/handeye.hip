// Hand-eye AX = XB refinement on MI355X: handle, HIP evaluator, C ABI.
// Replaces the body of HandEyeCalibration::solveNonLinear between "p[7] packed" and
// "X rebuilt" (/root/reference/modules/calib/src/handeye.cpp:282-324).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "common.h"
#include "handeye_kernels.h"
#include "lm_driver.h"

using namespace cvxh;

struct cvx_handeye {
    long long n = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_blocks = 1;
    double *d_pairs = nullptr, *d_p7 = nullptr, *d_partial = nullptr, *d_out = nullptr;
    double *h_out = nullptr, *h_p7 = nullptr;
    double *d_eval = nullptr;
    long long lm_seq = 0;
    HeLmOut *d_lm = nullptr, *h_lm = nullptr;    // result of the one-launch solve (k_he_lm): host-mapped, inside the pinned block
    // exchange
    cvx_allreduce_fn xfn = nullptr;
    void *xctx = nullptr;
    int rank = 0, nranks = 1, xdev = 0;
    ~cvx_handeye() {
        (void)hipSetDevice(device);
        // d_pairs heads the handle's one device block (p7, partial sums and outputs behind it), h_out its one pinned
        // block (h_p7 and the one-launch solve's record behind it)
        for (void *p : {(void *)d_pairs, (void *)d_eval})
            if (p) (void)hipFree(p);
        if (h_out) (void)hipHostFree(h_out);
        if (own_stream) cvx::own_stream_end(device, stream);
    }
};

namespace {

// Ceres QuaternionParameterization::Plus (handeye.cpp:305-308)
void quaternion_plus(const double x[4], const double d[3], double out[4]) {
    const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (n > 0.0) {
        const double sn = std::sin(n) / n;
        const double q[4] = {std::cos(n), sn * d[0], sn * d[1], sn * d[2]};
        out[0] = q[0] * x[0] - q[1] * x[1] - q[2] * x[2] - q[3] * x[3];
        out[1] = q[0] * x[1] + q[1] * x[0] + q[2] * x[3] - q[3] * x[2];
        out[2] = q[0] * x[2] - q[1] * x[3] + q[2] * x[0] + q[3] * x[1];
        out[3] = q[0] * x[3] + q[1] * x[2] - q[2] * x[1] + q[3] * x[0];
    } else {
        for (int i = 0; i < 4; ++i) out[i] = x[i];
    }
}

void plus7(const double x[7], const double d[6], double out[7]) {
    quaternion_plus(x, d, out);
    for (int a = 0; a < 3; ++a) out[4 + a] = x[4 + a] + d[3 + a];
}

int upload_p(cvx_handeye *h, const double p[7]) {
    std::memcpy(h->h_p7, p, 7 * sizeof(double));
    CVX_HIP(hipMemcpyAsync(h->d_p7, h->h_p7, 7 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return CVX_OK;
}

// reduce `width` doubles of the block partials, exchange, bring to h_out
int finish(cvx_handeye *h, int width) {
    hipLaunchKernelGGL(k_he_final, dim3(1), dim3(64), 0, h->stream, h->d_partial, h->n_blocks, width, h->d_out);
    CVX_HIP(hipGetLastError());
    if (h->nranks > 1 && h->xfn && h->xdev) {
        if (h->xfn(h->xctx, h->d_out, (size_t)width, 1, (void *)h->stream)) return CVX_ERR_EXCHANGE;
    }
    CVX_HIP(hipMemcpyAsync(h->h_out, h->d_out, width * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    CVX_HIP(hipStreamSynchronize(h->stream));
    if (h->nranks > 1 && h->xfn && !h->xdev) {
        if (h->xfn(h->xctx, h->h_out, (size_t)width, 0, nullptr)) return CVX_ERR_EXCHANGE;
    }
    return CVX_OK;
}

int device_cost(cvx_handeye *h, const double p[7], double *cost) {
    int st = upload_p(h, p);
    if (st) return st;
    hipLaunchKernelGGL(k_he_cost, dim3(h->n_blocks), dim3(256), 0, h->stream, h->d_pairs, h->n, h->d_p7, h->d_partial);
    CVX_HIP(hipGetLastError());
    st = finish(h, 1);
    if (st) return st;
    *cost = 0.5 * h->h_out[0];
    return CVX_OK;
}

// H (6x6 row-major, symmetric), g, cost at p
int device_normal_eq(cvx_handeye *h, const double p[7], double H[36], double g[6], double *cost) {
    int st = upload_p(h, p);
    if (st) return st;
    hipLaunchKernelGGL(k_he_normal_eq<PlaneSrc>, dim3(h->n_blocks), dim3(256), 0, h->stream,
                       PlaneSrc{h->d_pairs, h->n}, h->n, h->d_p7, h->d_partial);
    CVX_HIP(hipGetLastError());
    st = finish(h, HE_ACC);
    if (st) return st;
    int e = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) { H[a * 6 + b] = h->h_out[e]; H[b * 6 + a] = h->h_out[e]; ++e; }
    for (int a = 0; a < 6; ++a) g[a] = h->h_out[21 + a];
    *cost = 0.5 * h->h_out[27];
    return CVX_OK;
}

// the whole trust-region solve in one launch (k_he_lm): problems of at most HE_LM_MAX_PAIRS pairs on one rank
int device_lm(cvx_handeye *h, double p[7], const cvx_lm_options &opt, cvx_lm_summary &sum) {
    const auto t0 = std::chrono::steady_clock::now();
    std::memset(&sum, 0, sizeof sum);
    // the kernel writes its result straight into pinned host memory (posted writes, 33 KB at most): one
    // synchronisation and no copy between the launch and the answer
    if (!h->d_lm) {
        CVX_HIP(hipHostGetDevicePointer((void **)&h->d_lm, h->h_lm, 0));
        h->h_lm->seq = 0;
    }
    HeLmStart start;
    std::memcpy(start.p, p, sizeof start.p);
    start.seq = ++h->lm_seq;
    const int threads = (int)std::min<long long>(HE_LM_MAX_THREADS, std::max<long long>(64, (h->n + 63) / 64 * 64));
    hipLaunchKernelGGL(k_he_lm, dim3(1), dim3(threads), 0, h->stream, PlaneSrc{h->d_pairs, h->n}, start, opt, h->d_lm);
    CVX_HIP(hipGetLastError());
    // the result is polled (a stream synchronisation's wake-up costs more than the solve's first iterations)
    {
        volatile const long long *flag = &h->h_lm->seq;
        const auto tp = std::chrono::steady_clock::now();
        long long spins = 0;
        while (*flag != start.seq) {
            if ((++spins & 0xffff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - tp).count() > 30.0) {
                CVX_HIP(hipStreamSynchronize(h->stream));       // surfaces a kernel fault as a HIP error
                if (*flag != start.seq) { cvx::set_last_error("hand-eye solve result never arrived in host memory"); return CVX_ERR_HIP; }
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    const HeLmOut &r = *h->h_lm;
    if (r.status) return r.status;
    const int len = std::min(std::max(r.trace_len, 0), CVX_TRACE_MAX);
    std::memcpy(p, r.p, 7 * sizeof(double));
    sum.iterations = r.iterations; sum.num_successful = r.num_successful; sum.num_linear_solves = r.num_linear_solves;
    sum.termination = r.termination; sum.initial_cost = r.initial_cost; sum.final_cost = r.final_cost;
    sum.trace_len = len;
    std::memcpy(sum.trace_cost, r.trace[0], len * sizeof(double));
    std::memcpy(sum.trace_accepted, r.trace[1], len * sizeof(double));
    std::memcpy(sum.trace_damping, r.trace[2], len * sizeof(double));
    std::memcpy(sum.trace_gradient, r.trace[3], len * sizeof(double));
#ifdef HE_LM_PROF
    sum.trace_gradient[1000] = r.trace[3][1000]; sum.trace_gradient[1001] = r.trace[3][1001];      // 100 MHz ticks: evaluations | scalar sections
#endif
    sum.seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return CVX_OK;
}

class HandEyeEvaluator : public cvx::Evaluator {
public:
    HandEyeEvaluator(cvx_handeye *h_, double *p) : h(h_) { std::memcpy(x, p, sizeof x); }
    cvx_handeye *h;
    double x[7], xc[7];
    double H[36], g[6], cost = 0;
    double s2[6] = {1, 1, 1, 1, 1, 1};
    double Hc[36], gc[6], cand_cost_kept = 0;     // the candidate's linearisation (try_step), taken over by accept()
    bool cand_lin = false;
    bool likely_accept = true;                    // the last step was accepted (or none was taken yet)
    bool lin_valid = false, projected = true;

    void set_strategy(int strategy) override { projected = (strategy == 0); }

    // Ceres: max-norm of x - Plus(x, -g); levmar: max |g|
    double grad_norm() const {
        double m = 0;
        if (projected) {
            double d[6], xo[7];
            for (int a = 0; a < 6; ++a) d[a] = -g[a];
            plus7(x, d, xo);
            for (int i = 0; i < 7; ++i) m = std::max(m, std::fabs(x[i] - xo[i]));
        } else {
            for (int a = 0; a < 6; ++a) m = std::max(m, std::fabs(g[a]));
        }
        return m;
    }

    int linearize() {
        if (lin_valid) return CVX_OK;
        int st = device_normal_eq(h, x, H, g, &cost);
        if (st) return st;
        lin_valid = true;
        return CVX_OK;
    }

    cvx::InitInfo init(bool jacobi) override {
        cvx::InitInfo out;
        out.status = linearize();
        if (out.status) return out;
        double dm = 0;
        for (int a = 0; a < 6; ++a) {
            if (jacobi) { const double s = 1.0 / (1.0 + std::sqrt(H[a * 6 + a])); s2[a] = s * s; }
            else s2[a] = 1.0;
            dm = std::max(dm, H[a * 6 + a]);
        }
        out.cost = cost; out.grad_max = grad_norm(); out.diag_max = dm;
        return out;
    }

    cvx::StepInfo try_step(const cvx_damping &d, double gtol, bool gradient_only) override {
        cvx::StepInfo out;
        cand_lin = false;
        out.status = linearize();
        if (out.status) return out;
        out.cost = cost; out.grad_max = grad_norm();
        if (out.grad_max <= gtol) { out.grad_converged = true; return out; }
        if (gradient_only) return out;
        double A[36], delta[6], D[6];
        std::memcpy(A, H, sizeof A);
        for (int a = 0; a < 6; ++a) {
            double c = s2[a] * H[a * 6 + a];
            c = std::min(std::max(c, d.min_diag), d.max_diag);
            D[a] = c * d.inv_radius / s2[a] + d.mu;
            A[a * 6 + a] += D[a];
            delta[a] = -g[a];
        }
        bool solved = cvx::host_cholesky_solve(6, A, delta);
        for (int a = 0; a < 6 && solved; ++a) solved = std::isfinite(delta[a]);
        if (!solved) { out.ok = false; return out; }
        plus7(x, delta, xc);
        // After an accepted step (and at the start) the candidate is evaluated with the NORMAL-EQUATION kernel, not the cost
        // kernel: its 28th sum is the cost, and if this step is accepted too -- the usual case while the solve still
        // moves -- the other 27 are the next iteration's linearisation, which then costs no launch and no round trip to
        // the host (solveNonLinear's own options, 3 iterations: 1 738 pairs 154 -> 121 us per solve, 19 548 pairs
        // 214 -> 185).  After a rejected step the cost kernel is used (the normal-equation kernel is 8 us longer).
        double cand = 0;
        const bool with_lin = likely_accept;
        likely_accept = false;
        if (with_lin) out.status = device_normal_eq(h, xc, Hc, gc, &cand);
        else out.status = device_cost(h, xc, &cand);
        if (out.status) return out;
        cand_lin = with_lin;
        double m = 0, sn = 0, xn = 0;
        for (int a = 0; a < 6; ++a) m += delta[a] * (D[a] * delta[a] - g[a]);
        for (int i = 0; i < 7; ++i) { sn += (x[i] - xc[i]) * (x[i] - xc[i]); xn += x[i] * x[i]; }
        out.cand_cost = cand; out.model_change = 0.5 * m;
        cand_cost_kept = cand;
        // levmar's ||Dp||^2 is over the 6 tangent coordinates, Ceres' step norm over the 7 global ones
        double d2 = 0;
        for (int a = 0; a < 6; ++a) d2 += delta[a] * delta[a];
        out.step_norm2 = projected ? sn : d2;
        out.x_norm2 = xn; out.free_norm2 = xn;
        out.ok = std::isfinite(out.model_change);
        return out;
    }

    int accept() override {
        std::memcpy(x, xc, sizeof x);
        likely_accept = true;
        lin_valid = cand_lin;
        if (cand_lin) {
            std::memcpy(H, Hc, sizeof H); std::memcpy(g, gc, sizeof g);
            cost = cand_cost_kept;
        }
        cand_lin = false;
        return CVX_OK;
    }
    double xbest[7];
    int save_best() override { std::memcpy(xbest, x, sizeof x); return CVX_OK; }
    int restore_best() override {
        std::memcpy(x, xbest, sizeof x);
        lin_valid = false;
        return CVX_OK;
    }
};

}  // namespace

extern "C" int cvx_handeye_create(int n_pairs, const double *qa, const double *ta, const double *qb,
                                  const double *tb, int device, void *stream, cvx_handeye **out) {
    if (!out) return CVX_ERR_INVALID_ARG;
    *out = nullptr;
    if (n_pairs < 0 || (n_pairs > 0 && (!qa || !ta || !qb || !tb))) return CVX_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        cvx::set_last_error("no HIP device visible");
        return CVX_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) return CVX_ERR_NO_DEVICE;
    CVX_HIP(hipSetDevice(device));
    cvx_handeye *h = new cvx_handeye;
    std::unique_ptr<cvx_handeye> guard(h);
    h->n = n_pairs; h->device = device;
    if (stream) h->stream = (hipStream_t)stream;
    else { h->own_stream = true; const int sst = cvx::own_stream_begin(h->device, &h->stream); if (sst) return sst; }
    h->n_blocks = (int)std::min<long long>(1024, std::max<long long>(1, (h->n + 255) / 256));
    const size_t n = (size_t)n_pairs;
    std::vector<double> planes(14 * std::max<size_t>(n, 1));
    for (size_t i = 0; i < n; ++i) {
        for (int k = 0; k < 4; ++k) { planes[k * n + i] = qa[4 * i + k]; planes[(7 + k) * n + i] = qb[4 * i + k]; }
        for (int k = 0; k < 3; ++k) { planes[(4 + k) * n + i] = ta[3 * i + k]; planes[(11 + k) * n + i] = tb[3 * i + k]; }
    }
    // one device block and one pinned block per handle (an allocation costs more than a small solve's iteration)
    auto up32 = [](size_t c) { return (c + 31) / 32 * 32; };
    const size_t o_p7 = up32(planes.size()), o_partial = o_p7 + 32, o_out = o_partial + up32((size_t)h->n_blocks * HE_ACC);
    CVX_HIP(hipMalloc((void **)&h->d_pairs, (o_out + 32) * sizeof(double)));
    h->d_p7 = h->d_pairs + o_p7; h->d_partial = h->d_pairs + o_partial; h->d_out = h->d_pairs + o_out;
    CVX_HIP(hipMemcpyAsync(h->d_pairs, planes.data(), planes.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    CVX_HIP(hipHostMalloc((void **)&h->h_out, 64 * sizeof(double) + sizeof(HeLmOut), hipHostMallocMapped));
    h->h_p7 = h->h_out + 32;
    h->h_lm = reinterpret_cast<HeLmOut *>(h->h_out + 64);
    CVX_HIP(hipStreamSynchronize(h->stream));
    *out = guard.release();
    return CVX_OK;
}

extern "C" void cvx_handeye_destroy(cvx_handeye *h) { delete h; }

extern "C" int cvx_handeye_cost(cvx_handeye *h, const double p[7], double *cost) {
    if (!h || !p || !cost) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    return device_cost(h, p, cost);
}

extern "C" int cvx_handeye_normal_eq(cvx_handeye *h, const double p[7], double H[36], double g[6], double *cost) {
    if (!h || !p || !H || !g) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    double c = 0;
    int st = device_normal_eq(h, p, H, g, &c);
    if (st) return st;
    if (cost) *cost = c;
    return CVX_OK;
}

extern "C" int cvx_handeye_evaluate(cvx_handeye *h, const double p[7], double *r, double *Jq, double *Jt) {
    if (!h || !p) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    const size_t n = (size_t)h->n;
    if (n == 0) return CVX_OK;
    if (!h->d_eval) CVX_HIP(hipMalloc((void **)&h->d_eval, n * 96 * sizeof(double)));
    int st = upload_p(h, p);
    if (st) return st;
    double *dr = h->d_eval, *dq = dr + 12 * n, *dt = dq + 48 * n;
    hipLaunchKernelGGL(k_he_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_pairs, h->n, h->d_p7,
                       dr, dq, dt);
    CVX_HIP(hipGetLastError());
    if (r) CVX_HIP(hipMemcpyAsync(r, dr, 12 * n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (Jq) CVX_HIP(hipMemcpyAsync(Jq, dq, 48 * n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (Jt) CVX_HIP(hipMemcpyAsync(Jt, dt, 36 * n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    CVX_HIP(hipStreamSynchronize(h->stream));
    return CVX_OK;
}

// -----------------------------------------------------------------------------------------
// cvx_handeye_normal_eq for callers that cannot allocate.  The block partials of k_he_normal_eq, which the handle
// keeps in memory it got from hipMalloc, live in a file-scope device array instead: 1024 blocks x 28 doubles per
// GPU for as long as the process runs.  Its device address differs per GPU and is looked up once per device.
// -----------------------------------------------------------------------------------------
namespace {
constexpr int HE_STATIC_BLOCKS = CVX_HANDEYE_STATIC_MAX_PAIRS / 256;
constexpr int HE_STATIC_MAX_DEVICES = 64;
static __device__ __align__(256) double scratch[(size_t)HE_STATIC_BLOCKS * HE_ACC];
std::atomic<double *> g_scratch_ptr[HE_STATIC_MAX_DEVICES];
std::mutex g_scratch_mutex;
}  // namespace

extern "C" int cvx_handeye_normal_eq_static(int n_pairs, const double *d_qa, const double *d_ta, const double *d_qb,
                                            const double *d_tb, const double *d_p7, double *d_out, int device,
                                            void *stream) {
    if (n_pairs < 0 || !d_p7 || !d_out || (n_pairs > 0 && (!d_qa || !d_ta || !d_qb || !d_tb))) return CVX_ERR_INVALID_ARG;
    if (n_pairs > CVX_HANDEYE_STATIC_MAX_PAIRS) {
        cvx::set_last_error("cvx_handeye_normal_eq_static: n_pairs exceeds CVX_HANDEYE_STATIC_MAX_PAIRS (" +
                            std::to_string(CVX_HANDEYE_STATIC_MAX_PAIRS) + "), the size its fixed scratch array holds");
        return CVX_ERR_UNSUPPORTED;
    }
    if (device < 0) return CVX_ERR_NO_DEVICE;
    if (device >= HE_STATIC_MAX_DEVICES) return CVX_ERR_UNSUPPORTED;
    // The calling thread's current device is left as it was found.  After the first call on a device the call takes
    // no lock and asks the runtime nothing but the current device.
    int prev = -1;
    CVX_HIP(hipGetDevice(&prev));
    double *partial = g_scratch_ptr[device].load(std::memory_order_acquire);
    if (prev != device || !partial) {
        if (!partial) {
            int ndev = 0;
            if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
                cvx::set_last_error("no HIP device visible");
                return CVX_ERR_NO_DEVICE;
            }
            if (device >= ndev) return CVX_ERR_NO_DEVICE;
        }
        CVX_HIP(hipSetDevice(device));
    }
    struct RestoreDevice {
        int prev, now;
        ~RestoreDevice() { if (prev != now && prev >= 0) (void)hipSetDevice(prev); }
    } restore{prev, device};
    if (!partial) {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        partial = g_scratch_ptr[device].load(std::memory_order_relaxed);
        if (!partial) {
            // first call on this device (make it outside any stream capture): this GPU's copy of the array
            double *ptr = nullptr;
            CVX_HIP(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
            if (!ptr) { cvx::set_last_error("hipGetSymbolAddress returned a null address"); return CVX_ERR_HIP; }
            g_scratch_ptr[device].store(ptr, std::memory_order_release);
            partial = ptr;
        }
    }
    // same grid as the handle's launch, so the sums (and their order) are the same
    const long long n = n_pairs;
    const int n_blocks = (int)std::min<long long>(HE_STATIC_BLOCKS, std::max<long long>(1, (n + 255) / 256));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_he_normal_eq<AosSrc>, dim3(n_blocks), dim3(256), 0, s, AosSrc{d_qa, d_ta, d_qb, d_tb}, n, d_p7,
                       partial);
    CVX_HIP(hipGetLastError());
    // every block below n_blocks has written its 28 partials; nothing of an earlier call is read
    hipLaunchKernelGGL(k_he_final_expand, dim3(1), dim3(64), 0, s, partial, n_blocks, d_out);
    CVX_HIP(hipGetLastError());
    return CVX_OK;
}

extern "C" int cvx_handeye_set_exchange(cvx_handeye *h, cvx_allreduce_fn fn, void *ctx, int rank, int nranks,
                                        int device_buffers) {
    if (!h || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !fn)) return CVX_ERR_INVALID_ARG;
    h->xfn = fn; h->xctx = ctx; h->rank = rank; h->nranks = nranks; h->xdev = device_buffers ? 1 : 0;
    return CVX_OK;
}

extern "C" int cvx_handeye_solve(cvx_handeye *h, double p[7], const cvx_lm_options *opt, cvx_lm_summary *summary) {
    if (!h || !p || !opt || !summary) return CVX_ERR_INVALID_ARG;
    if (opt->strategy == 0 && opt->inner_iterations) {
        // one parameter block pair (q, t) with every residual on both: no independent sets to descend over
        cvx::set_last_error("hand-eye has one parameter block: inner iterations unsupported");
        return CVX_ERR_UNSUPPORTED;
    }
    CVX_HIP(hipSetDevice(h->device));
    // strategy 0 on one rank, up to HE_LM_MAX_PAIRS pairs: the whole loop runs on the device in one launch;
    // CVX_HANDEYE_HOST_LOOP=1 keeps the host-driven loop (the same control flow, lm_driver.cpp), as the tests compare
    const char *hl = std::getenv("CVX_HANDEYE_HOST_LOOP");
    const bool host_loop = hl && hl[0] == '1';
    if (opt->strategy == 0 && h->nranks == 1 && h->n <= HE_LM_MAX_PAIRS && !host_loop) return device_lm(h, p, *opt, *summary);
    HandEyeEvaluator ev(h, p);
    int st = cvx::lm_minimize(ev, *opt, *summary);
    std::memcpy(p, ev.x, 7 * sizeof(double));
    return st;
}
