// Host-only test of the LM control flow (cvx_amd/csrc/lm_driver.cpp) with a scripted evaluator: no device work.
// Built by cvx_amd/csrc/Makefile against libcvx_amd.so, run by tests/test_capi_and_host.py.
// Prints one JSON object per scenario.
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../cvx_amd/csrc/lm_driver.h"

// An evaluator over one scalar "position" that walks a scripted list of candidate costs: try_step offers the next
// candidate with model_change = 1, accept() moves there.  It records what the driver asks for.
struct Scripted : cvx::Evaluator {
    std::vector<double> cand;      // candidate cost of the k-th linear solve
    std::vector<double> inner;     // cost of that candidate after an inner-iteration pass (empty: no passes offered)
    double cur_cost, last_cost = 0;
    int inner_calls = 0;
    int pos = 0, at = -1, best_at = -2, saved = 0, restored = 0;   // `at`: index of the accepted candidate we stand on
    explicit Scripted(double c0, std::vector<double> c) : cand(std::move(c)), cur_cost(c0) {}
    cvx::InitInfo init(bool) override {
        cvx::InitInfo o; o.cost = cur_cost; o.grad_max = 1.0; o.diag_max = 1.0; return o;
    }
    cvx::StepInfo try_step(const cvx_damping &, double, bool gradient_only) override {
        cvx::StepInfo s; s.cost = cur_cost; s.grad_max = 1.0;
        if (gradient_only) return s;
        s.ok = true; s.cand_cost = cand[pos < (int)cand.size() ? pos : (int)cand.size() - 1];
        s.model_change = 1.0; s.step_norm2 = 1.0; s.x_norm2 = 1.0; s.free_norm2 = 1.0;
        last_cost = s.cand_cost;
        ++pos;
        return s;
    }
    int inner_iterations(double *cost, double *step_norm2) override {
        if (inner.empty()) return CVX_ERR_UNSUPPORTED;
        ++inner_calls;
        last_cost = inner[pos - 1 < (int)inner.size() ? pos - 1 : (int)inner.size() - 1];
        *cost = last_cost; *step_norm2 = 1.0;
        return 0;
    }
    int accept() override { at = pos - 1; cur_cost = last_cost; return 0; }
    int save_best() override { best_at = at; ++saved; return 0; }
    int restore_best() override { at = best_at; ++restored; return 0; }
};

static void run(const char *name, double c0, std::vector<double> c, int nonmonotonic, int iters,
                std::vector<double> inner = {}) {
    cvx_lm_options o;
    cvx_lm_options_refine(&o);
    o.inner_iterations = inner.empty() ? 0 : 1;
    o.nonmonotonic = nonmonotonic;
    o.max_iterations = iters;
    o.function_tolerance = o.parameter_tolerance = -1.0;
    o.gradient_tolerance = -1.0;
    Scripted ev(c0, c);
    ev.inner = inner;
    cvx_lm_summary s;
    const int st = cvx::lm_minimize(ev, o, s);
    std::printf("{\"name\": \"%s\", \"status\": %d, \"iterations\": %d, \"successful\": %d, \"final_cost\": %.17g, "
                "\"at\": %d, \"saved\": %d, \"restored\": %d, \"termination\": %d, \"inner_calls\": %d, \"inner_steps\": %d}\n",
                name, st, s.iterations, s.num_successful, s.final_cost, ev.at, ev.saved, ev.restored, s.termination,
                ev.inner_calls, s.num_inner_iteration_steps);
}

int main() {
    // monotone descent: nothing saved, nothing restored, ends on the last candidate
    run("monotone", 100.0, {90, 80, 70, 60}, 1, 4);
    // non-monotonic: 100 -> 50 -> 40 (min) -> 45 (uphill, accepted against the reference cost) -> 44: the run must hand
    // back the point with cost 40 (candidate index 1), saved once before the uphill accept
    run("uphill_then_end", 100.0, {50, 40, 45, 44}, 1, 4);
    // the same script with monotonic steps: 45 and 44 are rejected, the run stays at 40
    run("monotonic_rejects", 100.0, {50, 40, 45, 44}, 0, 4);
    // uphill, then a new minimum: the saved copy is stale and must not be restored
    run("uphill_then_better", 100.0, {50, 40, 45, 30}, 1, 4);
    // inner iterations (TrustRegionMinimizer::DoInnerIterationsIfNeeded / IsStepSuccessful), monotonic steps:
    // solve 0: the step goes uphill (120) but the pass ends below the current cost (95): accepted whatever the ratio;
    // solve 1: uphill (96) and the pass (95.5, a gain above the tolerance) stays above the current 95: rejected, the
    //          gain being credited to the model ((95 - 95.5) / (1 + 0.5) < 0);
    // solve 2: 94 -> 93.99 is a gain below 1e-3 of the candidate: accepted (93.99 < 95), and the passes stop;
    // solve 3: 93 without a pass
    run("inner_iterations", 100.0, {120, 96, 94, 93}, 0, 4, {95, 95.5, 93.99, 1});
    return 0;
}
