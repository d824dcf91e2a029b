// Tool-level bundle adjustment: handle, evaluator and C ABI around toolba_kernels.h.
// Replaces the body of bundleAdjustment between "p[7] packed" and "pose rebuilt"
// (/root/reference/modules/calib/tools/hand_eye_registration.cpp:158-229, camera_ba.cpp:160-231): the device
// reduces the observations to the arrowhead normal equations; the host eliminates the 3x3 point blocks, solves
// the 6x6 pose system and runs the trust-region loop of lm_driver.cpp with the options of :201-210.
#include <algorithm>
#include <cmath>
#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <vector>

#include "common.h"
#include "lm_driver.h"
#include "toolba_kernels.h"

using namespace cvxt;

struct cvx_toolba {
    int V = 0, P = 0, device = 0, n_jobs = 0;
    long long N = 0;
    double huber = 1.0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    double *d_obs = nullptr;       // ox | oy, point-major
    int *d_view = nullptr, *d_jobs = nullptr, *d_job_first = nullptr;
    double *d_trans = nullptr, *d_params = nullptr, *d_intr = nullptr, *d_partial = nullptr, *d_cpartial = nullptr, *d_out = nullptr;
    double *h_out = nullptr, *h_params = nullptr;
    double *d_out_map = nullptr;     // device view of h_out (host-mapped): results are written there and polled for
    long long seq = 0;
    // exchange (view-sharded runs: every rank holds its own views' observations, the pose and the points are shared)
    cvx_allreduce_fn xfn = nullptr;
    void *xctx = nullptr;
    int rank = 0, nranks = 1, xdev = 0;
    ~cvx_toolba() {
        (void)hipSetDevice(device);
        // d_obs heads the handle's one device block, h_out its one pinned block
        if (d_obs) (void)hipFree(d_obs);
        if (h_out) (void)hipHostFree(h_out);
        if (own_stream) cvx::own_stream_end(device, stream);
    }
};

namespace {

TbaArgs make_args(cvx_toolba *h) {
    TbaArgs a;
    const size_t n = (size_t)std::max<long long>(h->N, 1);
    a.ox = h->d_obs; a.oy = h->d_obs + n; a.view = h->d_view; a.trans = h->d_trans; a.params = h->d_params;
    a.intr = h->d_intr; a.huber = h->huber;
    a.job_point = h->d_jobs; a.job_start = h->d_jobs + h->n_jobs; a.job_count = h->d_jobs + 2 * h->n_jobs;
    a.partial = h->d_partial;
    return a;
}

int upload(cvx_toolba *h, const double *x) {
    std::memcpy(h->h_params, x, sizeof(double) * (7 + 3 * (size_t)h->P));
    CVX_HIP(hipMemcpyAsync(h->d_params, h->h_params, sizeof(double) * (7 + 3 * (size_t)h->P), hipMemcpyHostToDevice, h->stream));
    return CVX_OK;
}

// d_out[first .. first + cnt) -> h_out, and wait for it: written into host-mapped memory by a kernel behind the ones that
// made it and polled for (toolba_kernels.h: k_tba_publish); without a device view of h_out, a copy and a synchronisation
int fetch(cvx_toolba *h, size_t first, size_t cnt) {
    if (!h->d_out_map) {
        CVX_HIP(hipMemcpyAsync(h->h_out + first, h->d_out + first, sizeof(double) * cnt, hipMemcpyDeviceToHost, h->stream));
        CVX_HIP(hipStreamSynchronize(h->stream));
        return CVX_OK;
    }
    const size_t flag_at = (size_t)h->P * TBA_ACC + 4;
    const double want = (double)++h->seq;
    hipLaunchKernelGGL(k_tba_publish, dim3(1), dim3(256), 0, h->stream, h->d_out + first, h->d_out_map + first, (long long)cnt,
                       h->d_out_map + flag_at, want);
    CVX_HIP(hipGetLastError());
    volatile const double *flag = h->h_out + flag_at;
    const auto t0 = std::chrono::steady_clock::now();
    long long spins = 0;
    while (*flag != want) {
        if ((++spins & 0xffff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0) {
            CVX_HIP(hipStreamSynchronize(h->stream));       // surfaces a kernel fault as a HIP error
            if (*flag != want) { cvx::set_last_error("tool bundle adjustment: a result never arrived in host memory"); return CVX_ERR_HIP; }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return CVX_OK;
}

// 1/2 sum rho at x
int device_cost(cvx_toolba *h, const double *x, double *cost) {
    int st = upload(h, x);
    if (st) return st;
    if (h->n_jobs > 0) {
        TbaArgs a = make_args(h);
        a.partial = h->d_cpartial;
        hipLaunchKernelGGL(k_tba_cost, dim3(h->n_jobs), dim3(TBA_THREADS), 0, h->stream, a);
        CVX_HIP(hipGetLastError());
    }
    // the scalar lives behind the per-point blocks, which stay valid across rejected steps
    const size_t slot = (size_t)h->P * TBA_ACC;
    hipLaunchKernelGGL(k_tba_cost_final, dim3(1), dim3(TBA_THREADS), 0, h->stream, h->d_cpartial, h->n_jobs, h->d_out + slot);
    CVX_HIP(hipGetLastError());
    const bool sharded = h->nranks > 1 && h->xfn;
    if (sharded && h->xdev && h->xfn(h->xctx, h->d_out + slot, 1, 1, (void *)h->stream)) return CVX_ERR_EXCHANGE;
    int fst = fetch(h, slot, 1);
    if (fst) return fst;
    if (sharded && !h->xdev && h->xfn(h->xctx, h->h_out + slot, 1, 0, nullptr)) return CVX_ERR_EXCHANGE;
    *cost = 0.5 * h->h_out[slot];
    return CVX_OK;
}

// per-point sums [P][TBA_ACC] at x -> h_out
int device_blocks(cvx_toolba *h, const double *x) {
    int st = upload(h, x);
    if (st) return st;
    if (h->n_jobs > 0) {
        hipLaunchKernelGGL(k_tba_normal_eq, dim3(h->n_jobs), dim3(TBA_THREADS), 0, h->stream, make_args(h));
        CVX_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_tba_reduce, dim3(h->P), dim3(64), 0, h->stream, h->d_partial, h->d_job_first, h->d_out);
    CVX_HIP(hipGetLastError());
    // the one exchange step of a linearisation: the P x 55 block sums, summed over the ranks' view shards; every
    // rank then eliminates the points and solves the same 6 x 6 system
    const size_t cnt = TBA_ACC * (size_t)h->P;
    const bool sharded = h->nranks > 1 && h->xfn;
    if (sharded && h->xdev && h->xfn(h->xctx, h->d_out, cnt, 1, (void *)h->stream)) return CVX_ERR_EXCHANGE;
    int fst = fetch(h, 0, cnt);
    if (fst) return fst;
    if (sharded && !h->xdev && h->xfn(h->xctx, h->h_out, cnt, 0, nullptr)) return CVX_ERR_EXCHANGE;
    return CVX_OK;
}

void quaternion_plus(const double x[4], const double d[3], double out[4]) {     // Ceres QuaternionParameterization::Plus
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

// 3x3 SPD solve helpers: A (upper triangle a00 a01 a02 a11 a12 a22) -> Cholesky factor in place; false if not PD
bool chol3(double a[6]) {
    if (!(a[0] > 0.0)) return false;
    const double l00 = std::sqrt(a[0]), l10 = a[1] / l00, l20 = a[2] / l00;
    const double d1 = a[3] - l10 * l10;
    if (!(d1 > 0.0)) return false;
    const double l11 = std::sqrt(d1), l21 = (a[4] - l20 * l10) / l11;
    const double d2 = a[5] - l20 * l20 - l21 * l21;
    if (!(d2 > 0.0)) return false;
    a[0] = l00; a[1] = l10; a[2] = l20; a[3] = l11; a[4] = l21; a[5] = std::sqrt(d2);
    return true;
}
void chol3_solve(const double l[6], double b[3]) {
    b[0] /= l[0];
    b[1] = (b[1] - l[1] * b[0]) / l[3];
    b[2] = (b[2] - l[2] * b[0] - l[4] * b[1]) / l[5];
    b[2] /= l[5];
    b[1] = (b[1] - l[4] * b[2]) / l[3];
    b[0] = (b[0] - l[1] * b[1] - l[2] * b[2]) / l[0];
}

// ---- inner iterations (use_inner_iterations = true, hand_eye_registration.cpp:205) ----
// The parameter blocks of the reference's problem are q (4 numbers, QuaternionParameterization), t (3) and the board
// points (:176-181), all with three degrees of freedom; every residual block touches q, t and one point.  One LM
// state per block, the state machine of inner_kernels.h (k_inner_solve / k_inner_decide) on the host: the block
// sums come from the device kernels the outer iteration uses (all-reduced there when the views are sharded, so
// every rank walks through the same decisions).
struct BlockLM3 {
    double cost = 0, radius = 1e4, decrease = 2.0, model = 0, s2[3] = {1, 1, 1};
    double H[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};     // upper triangle a00 a01 a02 a11 a12 a22
    double xc[4] = {0, 0, 0, 0};
    int iters = 0, invalid = 0, term = 0;
    bool first = true, has_candidate = false, needs_lin = true, done = false;

    // after a linearisation at x: tests, then the damped step; x has ng numbers, plus() makes the candidate
    template <class Plus>
    void solve(const double *x, int ng, Plus plus) {
        has_candidate = false;
        if (needs_lin) {
            if (first) {
                const double dg[3] = {H[0], H[3], H[5]};
                for (int i = 0; i < 3; ++i) { const double s = 1.0 / (1.0 + std::sqrt(dg[i])); s2[i] = s * s; }
                first = false;
            }
            double gmax = 0;
            for (int i = 0; i < 3; ++i) gmax = std::max(gmax, std::fabs(g[i]));
            if (gmax <= 1e-10) { term = 1; done = true; return; }
            needs_lin = false;
        }
        const double dg[3] = {H[0], H[3], H[5]};
        for (;;) {
            if (iters >= 50) { term = 4; done = true; return; }
            ++iters;
            double D[3], l[6] = {H[0], H[1], H[2], H[3], H[4], H[5]}, d[3] = {-g[0], -g[1], -g[2]};
            for (int i = 0; i < 3; ++i) D[i] = std::min(std::max(s2[i] * dg[i], 1e-6), 1e32) / (s2[i] * radius);
            l[0] += D[0]; l[3] += D[1]; l[5] += D[2];
            double mc = 0;
            const bool ok = chol3(l);
            if (ok) {
                chol3_solve(l, d);
                for (int i = 0; i < 3; ++i) mc += d[i] * (D[i] * d[i] - g[i]);
                mc *= 0.5;
            }
            if (!ok || !(mc > 0.0)) {
                if (++invalid >= 5) { term = 6; done = true; return; }
                radius *= 0.5;
                continue;
            }
            invalid = 0;
            plus(x, d, xc);
            double sn = 0, xn = 0;
            for (int i = 0; i < ng; ++i) { sn += (x[i] - xc[i]) * (x[i] - xc[i]); xn += x[i] * x[i]; }
            if (std::sqrt(sn) <= 1e-8 * (std::sqrt(xn) + 1e-8)) { term = 2; done = true; return; }
            model = mc;
            has_candidate = true;
            return;
        }
    }
    // with the candidate's cost: true when the block moves to its candidate
    bool decide(double cand) {
        has_candidate = false;
        if (std::fabs(cost - cand) <= 1e-6 * cost) { term = 3; done = true; return false; }
        const double rho = (cost - cand) / model;
        if (rho > 1e-3) {
            const double t = 2.0 * rho - 1.0;
            radius = std::min(radius / std::max(1.0 / 3.0, 1.0 - t * t * t), 1e32);    // TrustRegionStrategy::Options::max_radius
            decrease = 2.0;
            needs_lin = true;
            return true;
        }
        radius /= decrease; decrease *= 2.0;
        if (radius < 1e-32) { term = 5; done = true; }
        return false;
    }
};

class ToolBAEvaluator : public cvx::Evaluator {
public:
    ToolBAEvaluator(cvx_toolba *h_, const double *p, const double *pts) : h(h_), P(h_->P), nf(6 + 3 * h_->P), ng(7 + 3 * h_->P) {
        x.resize(ng); xc.resize(ng); s2.assign(nf, 1.0); diag.resize(nf); grad.resize(nf); delta.resize(nf); D.resize(nf);
        std::memcpy(x.data(), p, 7 * sizeof(double));
        std::memcpy(x.data() + 7, pts, sizeof(double) * 3 * (size_t)P);
    }
    cvx_toolba *h;
    int P, nf, ng;
    std::vector<double> x, xc, s2, diag, grad, delta, D;
    double Hpp[36], cost = 0;
    bool lin_valid = false, projected = true;

    void set_strategy(int strategy) override { projected = (strategy == 0); }

    const double *blk(int p) const { return h->h_out + (size_t)p * TBA_ACC; }

    int linearize() {
        if (lin_valid) return CVX_OK;
        int st = device_blocks(h, x.data());
        if (st) return st;
        double up[28];
        for (int k = 0; k < 28; ++k) up[k] = 0.0;
        double c = 0.0;
        for (int p = 0; p < P; ++p) {                       // pose block = sum over points, fixed order
            const double *b = blk(p);
            for (int k = 0; k < 27; ++k) up[k] += b[k];
            c += b[54];
            diag[6 + 3 * p] = b[45]; diag[6 + 3 * p + 1] = b[48]; diag[6 + 3 * p + 2] = b[50];
            for (int k = 0; k < 3; ++k) grad[6 + 3 * p + k] = b[51 + k];
        }
        int e = 0;
        for (int a = 0; a < 6; ++a)
            for (int b = a; b < 6; ++b) { Hpp[a * 6 + b] = up[e]; Hpp[b * 6 + a] = up[e]; ++e; }
        for (int a = 0; a < 6; ++a) { diag[a] = Hpp[a * 6 + a]; grad[a] = up[21 + a]; }
        cost = 0.5 * c;
        lin_valid = true;
        return CVX_OK;
    }

    void plus(const double *d, double *out) const {
        quaternion_plus(x.data(), d, out);
        for (int a = 0; a < 3; ++a) out[4 + a] = x[4 + a] + d[3 + a];
        for (int a = 0; a < 3 * P; ++a) out[7 + a] = x[7 + a] + d[6 + a];
    }

    double grad_norm() {
        double m = 0;
        if (projected) {                                    // Ceres: max-norm of x - Plus(x, -g)
            double d[3] = {-grad[0], -grad[1], -grad[2]}, q[4];
            quaternion_plus(x.data(), d, q);
            for (int i = 0; i < 4; ++i) m = std::max(m, std::fabs(x[i] - q[i]));
            for (int a = 3; a < nf; ++a) m = std::max(m, std::fabs(grad[a]));
        } else {
            for (int a = 0; a < nf; ++a) m = std::max(m, std::fabs(grad[a]));
        }
        return m;
    }

    cvx::InitInfo init(bool jacobi) override {
        cvx::InitInfo out;
        out.status = linearize();
        if (out.status) return out;
        double dm = 0;
        for (int a = 0; a < nf; ++a) {
            if (jacobi) { const double s = 1.0 / (1.0 + std::sqrt(diag[a])); s2[a] = s * s; }
            else s2[a] = 1.0;
            dm = std::max(dm, diag[a]);
        }
        out.cost = cost; out.grad_max = grad_norm(); out.diag_max = dm;
        return out;
    }

    // (H + D) delta = -g by eliminating the point blocks
    bool solve_arrowhead() {
        double S[36], b[6];
        std::memcpy(S, Hpp, sizeof S);
        for (int a = 0; a < 6; ++a) { S[a * 6 + a] += D[a]; b[a] = -grad[a]; }
        std::vector<double> L((size_t)6 * P);
        for (int p = 0; p < P; ++p) {
            const double *k = blk(p);
            double *l = L.data() + 6 * (size_t)p;
            l[0] = k[45] + D[6 + 3 * p]; l[1] = k[46]; l[2] = k[47];
            l[3] = k[48] + D[6 + 3 * p + 1]; l[4] = k[49]; l[5] = k[50] + D[6 + 3 * p + 2];
            if (!chol3(l)) return false;
            double y[6][3];                                  // rows of W (U + D)^-1
            for (int a = 0; a < 6; ++a) {
                for (int c = 0; c < 3; ++c) y[a][c] = k[27 + 3 * a + c];
                chol3_solve(l, y[a]);
            }
            for (int a = 0; a < 6; ++a) {
                for (int c = 0; c < 6; ++c)
                    S[a * 6 + c] -= y[a][0] * k[27 + 3 * c] + y[a][1] * k[27 + 3 * c + 1] + y[a][2] * k[27 + 3 * c + 2];
                b[a] += y[a][0] * k[51] + y[a][1] * k[52] + y[a][2] * k[53];
            }
        }
        if (!cvx::host_cholesky_solve(6, S, b)) return false;
        for (int a = 0; a < 6; ++a) delta[a] = b[a];
        for (int p = 0; p < P; ++p) {
            const double *k = blk(p);
            double r[3];
            for (int c = 0; c < 3; ++c) {
                double s = k[51 + c];
                for (int a = 0; a < 6; ++a) s += k[27 + 3 * a + c] * delta[a];
                r[c] = -s;
            }
            chol3_solve(L.data() + 6 * (size_t)p, r);
            for (int c = 0; c < 3; ++c) delta[6 + 3 * p + c] = r[c];
        }
        for (int a = 0; a < nf; ++a) if (!std::isfinite(delta[a])) return false;
        return true;
    }

    cvx::StepInfo try_step(const cvx_damping &d, double gtol, bool gradient_only) override {
        cvx::StepInfo out;
        out.status = linearize();
        if (out.status) return out;
        out.cost = cost; out.grad_max = grad_norm();
        if (out.grad_max <= gtol) { out.grad_converged = true; return out; }
        if (gradient_only) return out;
        for (int a = 0; a < nf; ++a) {
            double c = s2[a] * diag[a];
            c = std::min(std::max(c, d.min_diag), d.max_diag);
            D[a] = c * d.inv_radius / s2[a] + d.mu;
        }
        if (!solve_arrowhead()) { out.ok = false; return out; }
        plus(delta.data(), xc.data());
        double cand = 0;
        out.status = device_cost(h, xc.data(), &cand);
        if (out.status) return out;
        double m = 0, sn = 0, xn = 0, d2 = 0;
        for (int a = 0; a < nf; ++a) { m += delta[a] * (D[a] * delta[a] - grad[a]); d2 += delta[a] * delta[a]; }
        for (int i = 0; i < ng; ++i) { sn += (x[i] - xc[i]) * (x[i] - xc[i]); xn += x[i] * x[i]; }
        out.cand_cost = cand; out.model_change = 0.5 * m;
        out.step_norm2 = projected ? sn : d2;
        out.x_norm2 = xn; out.free_norm2 = xn;
        out.ok = std::isfinite(out.model_change);
        return out;
    }

    int accept() override {
        x = xc;
        lin_valid = false;
        return CVX_OK;
    }

    // ---- one pass of CoordinateDescentMinimizer::Minimize over the candidate xc ----
    // pose-side sums of the per-point blocks in h_out: upper triangle of the 6 x 6 block (21) | gradient (6) | 2 cost
    void pose_sums(double up[27], double *c) const {
        for (int k = 0; k < 27; ++k) up[k] = 0.0;
        double s = 0.0;
        for (int p = 0; p < P; ++p) { const double *b = blk(p); for (int k = 0; k < 27; ++k) up[k] += b[k]; s += b[54]; }
        *c = 0.5 * s;
    }
    // block q (which = 0: tangent 0..2 of the pose) or t (which = 1: tangent 3..5), minimised alone
    int inner_pose_block(int which) {
        BlockLM3 lm;
        double *xb = xc.data() + (which ? 4 : 0);
        const int ng = which ? 3 : 4;
        for (int round = 0; round < 64 && !lm.done; ++round) {
            if (lm.needs_lin) {
                int st = device_blocks(h, xc.data());
                if (st) return st;
                double up[27];
                pose_sums(up, &lm.cost);
                // upper-triangle index of (a, b), a <= b, in a 6 x 6: a * 6 - a (a - 1) / 2 + (b - a)
                auto ut = [&](int a, int b) { return up[a * 6 - a * (a - 1) / 2 + (b - a)]; };
                const int o = which ? 3 : 0;
                lm.H[0] = ut(o, o); lm.H[1] = ut(o, o + 1); lm.H[2] = ut(o, o + 2);
                lm.H[3] = ut(o + 1, o + 1); lm.H[4] = ut(o + 1, o + 2); lm.H[5] = ut(o + 2, o + 2);
                for (int i = 0; i < 3; ++i) lm.g[i] = up[21 + o + i];
            }
            if (which) lm.solve(xb, ng, [](const double *x0, const double *d, double *out) { for (int i = 0; i < 3; ++i) out[i] = x0[i] + d[i]; });
            else lm.solve(xb, ng, [](const double *x0, const double *d, double *out) { quaternion_plus(x0, d, out); });
            if (!lm.has_candidate) continue;
            std::vector<double> y = xc;
            for (int i = 0; i < ng; ++i) y[(which ? 4 : 0) + i] = lm.xc[i];
            double cand = 0;
            int st = device_cost(h, y.data(), &cand);
            if (st) return st;
            if (lm.decide(cand)) xc = y;
        }
        return CVX_OK;
    }
    // all point blocks at once, each with its own LM state; a round costs two block evaluations
    int inner_points() {
        std::vector<BlockLM3> lm(P);
        std::vector<double> y;
        for (int p = 0; p < P; ++p) lm[p].done = !used[p];
        for (int round = 0; round < 64; ++round) {
            bool any = false;
            for (int p = 0; p < P; ++p) any = any || !lm[p].done;
            if (!any) break;
            int st = device_blocks(h, xc.data());
            if (st) return st;
            y = xc;
            bool cands = false;
            for (int p = 0; p < P; ++p) {
                BlockLM3 &b = lm[p];
                if (b.done) continue;
                if (b.needs_lin) {
                    const double *k = blk(p);
                    for (int i = 0; i < 6; ++i) b.H[i] = k[45 + i];
                    for (int i = 0; i < 3; ++i) b.g[i] = k[51 + i];
                    b.cost = 0.5 * k[54];
                }
                b.solve(xc.data() + 7 + 3 * p, 3, [](const double *x0, const double *d, double *out) { for (int i = 0; i < 3; ++i) out[i] = x0[i] + d[i]; });
                if (b.has_candidate) { for (int i = 0; i < 3; ++i) y[7 + 3 * p + i] = b.xc[i]; cands = true; }
            }
            if (!cands) continue;
            st = device_blocks(h, y.data());        // a point's cost depends on its own coordinates only
            if (st) return st;
            for (int p = 0; p < P; ++p) {
                BlockLM3 &b = lm[p];
                if (b.done || !b.has_candidate) continue;
                if (b.decide(0.5 * blk(p)[54]))
                    for (int i = 0; i < 3; ++i) xc[7 + 3 * p + i] = y[7 + 3 * p + i];
            }
        }
        return CVX_OK;
    }
    std::vector<char> used;      // points that occur in a residual block
    int n_used = -1;
    int inner_iterations(double *cost_out, double *step_norm2) override {
        if (n_used < 0) {
            used.assign(P, 0);
            n_used = 0;
            int st = device_blocks(h, xc.data());
            if (st) return st;
            // a point without observations has an empty block: its diagonal and cost sums are exactly zero
            for (int p = 0; p < P; ++p) { const double *k = blk(p); used[p] = (k[45] != 0.0 || k[48] != 0.0 || k[50] != 0.0 || k[54] != 0.0); n_used += used[p]; }
        }
        // Ceres' automatic ordering on this graph (greedy independent sets by ascending degree, reversed: the points
        // have degree 2, q and t degree 1 + points, q is named before t): t, q, points; with a single point in use
        // all degrees tie and the naming order gives: the point, t, q
        int st = CVX_OK;
        if (n_used == 1) st = inner_points();
        if (!st && n_used > 0) st = inner_pose_block(1);
        if (!st && n_used > 0) st = inner_pose_block(0);
        if (!st && n_used > 1) st = inner_points();
        if (st) return st;
        st = device_cost(h, xc.data(), cost_out);
        if (st) return st;
        double sn = 0;
        for (int i = 0; i < ng; ++i) sn += (x[i] - xc[i]) * (x[i] - xc[i]);
        *step_norm2 = sn;
        lin_valid = false;       // h_out holds the blocks of another point now
        return CVX_OK;
    }
    std::vector<double> xbest;
    int save_best() override { xbest = x; return CVX_OK; }
    int restore_best() override {
        x = xbest;
        lin_valid = false;
        return CVX_OK;
    }

};

}  // namespace

extern "C" void cvx_lm_options_toolba(cvx_lm_options *o) {     // hand_eye_registration.cpp:201-210
    cvx_lm_options_refine(o);      // use_inner_iterations (:205) as in refine
    o->nonmonotonic = 0;
    o->function_tolerance = o->gradient_tolerance = o->parameter_tolerance = 1e-20;
}

extern "C" int cvx_toolba_create(const cvx_toolba_desc *d, cvx_toolba **out) {
    if (!out) return CVX_ERR_INVALID_ARG;
    *out = nullptr;
    if (!d || d->n_views < 0 || d->n_points <= 0 || d->n_obs < 0 || !d->view_off || (d->n_views > 0 && !d->trans) ||
        (d->n_obs > 0 && (!d->point_id || !d->obs_x || !d->obs_y)))
        return CVX_ERR_INVALID_ARG;
    if (d->n_points > CVX_MAX_POINTS) { cvx::set_last_error("toolba: more than CVX_MAX_POINTS board points"); return CVX_ERR_UNSUPPORTED; }
    if (d->view_off[0] != 0 || d->view_off[d->n_views] != d->n_obs) { cvx::set_last_error("toolba: view_off does not span n_obs"); return CVX_ERR_INVALID_ARG; }
    const int V = d->n_views, P = d->n_points;
    const long long N = d->n_obs;
    // point-major order: observations of point 0 (by view), then point 1, ...
    std::vector<long long> first((size_t)P + 1, 0);
    for (int v = 0; v < V; ++v) {
        if (d->view_off[v + 1] < d->view_off[v]) return CVX_ERR_INVALID_ARG;
        for (int o = d->view_off[v]; o < d->view_off[v + 1]; ++o) {
            const int p = d->point_id[o];
            if (p < 0 || p >= P) { cvx::set_last_error("toolba: point id out of range"); return CVX_ERR_INVALID_ARG; }
            ++first[(size_t)p + 1];
        }
    }
    for (int p = 0; p < P; ++p) first[(size_t)p + 1] += first[p];
    const size_t n = (size_t)std::max<long long>(N, 1);
    std::vector<double> obs(2 * n, 0.0);
    std::vector<int> view(n, 0);
    {
        std::vector<long long> cur(first.begin(), first.end() - 1);
        for (int v = 0; v < V; ++v)
            for (int o = d->view_off[v]; o < d->view_off[v + 1]; ++o) {
                const long long k = cur[d->point_id[o]]++;
                obs[k] = d->obs_x[o]; obs[n + k] = d->obs_y[o]; view[k] = v;
            }
    }
    // observations of one point per workgroup: whole multiples of the block, sized for >= ~1000 workgroups
    const long long chunk = TBA_THREADS * std::max<long long>(1, std::min<long long>(TBA_CHUNK / TBA_THREADS, N / (TBA_THREADS * 1024LL)));
    std::vector<int> jp, js, jc, jfirst((size_t)P + 1, 0);
    for (int p = 0; p < P; ++p) {
        jfirst[p] = (int)jp.size();
        for (long long s = first[p]; s < first[(size_t)p + 1]; s += chunk) {
            jp.push_back(p); js.push_back((int)s); jc.push_back((int)std::min<long long>(chunk, first[(size_t)p + 1] - s));
        }
    }
    jfirst[P] = (int)jp.size();
    const int n_jobs = (int)jp.size();

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { cvx::set_last_error("no HIP device visible"); return CVX_ERR_NO_DEVICE; }
    if (d->device < 0 || d->device >= ndev) return CVX_ERR_NO_DEVICE;
    CVX_HIP(hipSetDevice(d->device));
    std::unique_ptr<cvx_toolba> h(new cvx_toolba);
    h->V = V; h->P = P; h->N = N; h->device = d->device; h->n_jobs = n_jobs; h->huber = d->huber;
    if (d->stream) h->stream = (hipStream_t)d->stream;
    else { h->own_stream = true; const int sst = cvx::own_stream_begin(h->device, &h->stream); if (sst) return sst; }
    std::vector<int> jobs(3 * (size_t)std::max(n_jobs, 1), 0);
    for (int j = 0; j < n_jobs; ++j) { jobs[j] = jp[j]; jobs[n_jobs + j] = js[j]; jobs[2 * (size_t)n_jobs + j] = jc[j]; }
    const size_t nv = (size_t)std::max(V, 1);
    {
        // one device block and one pinned block per handle, carved on 256-byte boundaries (ten hipMalloc / hipFree pairs
        // cost more than the tool's whole solve at its usual sizes)
        auto up = [](size_t b) { return (b + 255) / 256 * 256; };
        const size_t nj = (size_t)std::max(n_jobs, 1), n_out = (size_t)P * TBA_ACC + 8, n_par = 7 + 3 * (size_t)P;
        const size_t sizes[10] = {2 * n * sizeof(double), n * sizeof(int), jobs.size() * sizeof(int), ((size_t)P + 1) * sizeof(int),
                                  nv * 12 * sizeof(double), n_par * sizeof(double), 9 * sizeof(double), nj * TBA_ACC * sizeof(double),
                                  nj * sizeof(double), n_out * sizeof(double)};
        size_t total = 0;
        for (size_t b : sizes) total += up(b);
        char *base = nullptr;
        CVX_HIP(hipMalloc((void **)&base, total));
        void **slots[10] = {(void **)&h->d_obs, (void **)&h->d_view, (void **)&h->d_jobs, (void **)&h->d_job_first, (void **)&h->d_trans,
                            (void **)&h->d_params, (void **)&h->d_intr, (void **)&h->d_partial, (void **)&h->d_cpartial, (void **)&h->d_out};
        for (int k = 0; k < 10; ++k) { *slots[k] = base; base += up(sizes[k]); }
        CVX_HIP(hipHostMalloc((void **)&h->h_out, (n_out + n_par) * sizeof(double), hipHostMallocMapped));
        h->h_params = h->h_out + n_out;
        std::memset(h->h_out + (size_t)P * TBA_ACC, 0, 8 * sizeof(double));     // [4] is a polled sequence number
        if (hipHostGetDevicePointer((void **)&h->d_out_map, h->h_out, 0) != hipSuccess) { (void)hipGetLastError(); h->d_out_map = nullptr; }
    }
    CVX_HIP(hipMemcpyAsync(h->d_obs, obs.data(), 2 * n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    CVX_HIP(hipMemcpyAsync(h->d_view, view.data(), n * sizeof(int), hipMemcpyHostToDevice, h->stream));
    CVX_HIP(hipMemcpyAsync(h->d_jobs, jobs.data(), jobs.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    CVX_HIP(hipMemcpyAsync(h->d_job_first, jfirst.data(), jfirst.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (V > 0) CVX_HIP(hipMemcpyAsync(h->d_trans, d->trans, (size_t)V * 12 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    CVX_HIP(hipMemcpyAsync(h->d_intr, d->intrinsics, 9 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    CVX_HIP(hipStreamSynchronize(h->stream));
    *out = h.release();
    return CVX_OK;
}

extern "C" int cvx_toolba_set_exchange(cvx_toolba *h, cvx_allreduce_fn fn, void *ctx, int rank, int nranks, int device_buffers) {
    if (!h || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !fn)) return CVX_ERR_INVALID_ARG;
    h->xfn = fn; h->xctx = ctx; h->rank = rank; h->nranks = nranks; h->xdev = device_buffers ? 1 : 0;
    return CVX_OK;
}

extern "C" void cvx_toolba_destroy(cvx_toolba *h) { delete h; }

extern "C" int cvx_toolba_cost(cvx_toolba *h, const double p[7], const double *points, double *cost) {
    if (!h || !p || !points || !cost) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    std::vector<double> x(7 + 3 * (size_t)h->P);
    std::memcpy(x.data(), p, 7 * sizeof(double));
    std::memcpy(x.data() + 7, points, sizeof(double) * 3 * (size_t)h->P);
    return device_cost(h, x.data(), cost);
}

extern "C" int cvx_toolba_normal_eq(cvx_toolba *h, const double p[7], const double *points, double Hpp[36], double gp[6],
                                    double *W, double *U, double *gx, double *cost) {
    if (!h || !p || !points) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    ToolBAEvaluator ev(h, p, points);
    int st = ev.linearize();
    if (st) return st;
    if (Hpp) std::memcpy(Hpp, ev.Hpp, sizeof ev.Hpp);
    if (gp) for (int a = 0; a < 6; ++a) gp[a] = ev.grad[a];
    for (int q = 0; q < h->P; ++q) {
        const double *b = ev.blk(q);
        if (W) for (int k = 0; k < 18; ++k) W[18 * (size_t)q + k] = b[27 + k];
        if (U) {
            double *u = U + 9 * (size_t)q;
            u[0] = b[45]; u[1] = u[3] = b[46]; u[2] = u[6] = b[47]; u[4] = b[48]; u[5] = u[7] = b[49]; u[8] = b[50];
        }
        if (gx) for (int k = 0; k < 3; ++k) gx[3 * (size_t)q + k] = b[51 + k];
    }
    if (cost) *cost = ev.cost;
    return CVX_OK;
}

extern "C" int cvx_toolba_inner_pass(cvx_toolba *h, double p[7], double *points, double *cost) {
    if (!h || !p || !points) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    ToolBAEvaluator ev(h, p, points);
    ev.xc = ev.x;
    double c = 0, n2 = 0;
    int st = ev.inner_iterations(&c, &n2);
    if (st) return st;
    std::memcpy(p, ev.xc.data(), 7 * sizeof(double));
    std::memcpy(points, ev.xc.data() + 7, sizeof(double) * 3 * (size_t)h->P);
    if (cost) *cost = c;
    return CVX_OK;
}

extern "C" int cvx_toolba_solve(cvx_toolba *h, double p[7], double *points, const cvx_lm_options *opt,
                                cvx_lm_summary *summary) {
    if (!h || !p || !points || !opt || !summary) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    ToolBAEvaluator ev(h, p, points);
    int st = cvx::lm_minimize(ev, *opt, *summary);
    std::memcpy(p, ev.x.data(), 7 * sizeof(double));
    std::memcpy(points, ev.x.data() + 7, sizeof(double) * 3 * (size_t)h->P);
    return st;
}
