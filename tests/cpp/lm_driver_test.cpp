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
    double cur_cost;
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
        ++pos;
        return s;
    }
    int accept() override { at = pos - 1; cur_cost = cand[at]; return 0; }
    int save_best() override { best_at = at; ++saved; return 0; }
    int restore_best() override { at = best_at; ++restored; return 0; }
};

static void run(const char *name, double c0, std::vector<double> c, int nonmonotonic, int iters) {
    cvx_lm_options o;
    cvx_lm_options_refine(&o);
    o.nonmonotonic = nonmonotonic;
    o.max_iterations = iters;
    o.function_tolerance = o.parameter_tolerance = -1.0;
    o.gradient_tolerance = -1.0;
    Scripted ev(c0, c);
    cvx_lm_summary s;
    const int st = cvx::lm_minimize(ev, o, s);
    std::printf("{\"name\": \"%s\", \"status\": %d, \"iterations\": %d, \"successful\": %d, \"final_cost\": %.17g, "
                "\"at\": %d, \"saved\": %d, \"restored\": %d, \"termination\": %d}\n",
                name, st, s.iterations, s.num_successful, s.final_cost, ev.at, ev.saved, ev.restored, s.termination);
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
    return 0;
}
