#include "lm_driver.h"
#include "nielsen.h"

#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace cvx {

static void trace_push(cvx_lm_summary &s, double cost, int acc, double damp, double grad) {
    if (s.trace_len < CVX_TRACE_MAX) {
        s.trace_cost[s.trace_len] = cost;
        s.trace_accepted[s.trace_len] = acc;
        s.trace_damping[s.trace_len] = damp;
        s.trace_gradient[s.trace_len] = grad;
        s.trace_len++;
    }
}

// ---------------------------------------------------------------------------------------
// strategy 0: Ceres-style trust-region Levenberg-Marquardt
// ---------------------------------------------------------------------------------------
static int lm_trust_region(Evaluator &ev, const cvx_lm_options &o, cvx_lm_summary &sum) {
    InitInfo ini = ev.init(o.jacobi_scaling != 0);
    if (ini.status) return ini.status;
    if (!std::isfinite(ini.cost)) return CVX_ERR_NUMERIC;
    sum.initial_cost = ini.cost;
    // the decisions themselves are trust_region.h's (shared with the device: speculated steps of the calibration path)
    TrOptions to;
    to.max_iterations = o.max_iterations;
    to.max_nonmono = o.nonmonotonic ? o.max_consecutive_nonmonotonic_steps : 0;
    to.function_tolerance = o.function_tolerance; to.gradient_tolerance = o.gradient_tolerance;
    to.parameter_tolerance = o.parameter_tolerance;
    to.max_radius = o.max_radius; to.min_radius = o.min_radius; to.min_relative_decrease = o.min_relative_decrease;
    TrState t;
    tr_init(t, ini.cost, ini.grad_max, o.initial_radius, o.inner_iterations != 0);
    if (t.gmax <= o.gradient_tolerance) t.term = 1;

    while (!t.term) {
        if (t.it >= o.max_iterations) {
            if (t.need_gcheck) {   // Ceres tests the gradient right after the accepted step
                cvx_damping d0 = {0, 0, 0, 0};
                StepInfo g = ev.try_step(d0, o.gradient_tolerance, true);
                if (g.status) return g.status;
                if (g.grad_converged) { t.term = 1; break; }
            }
            t.term = 4;
            break;
        }
        cvx_damping d = {1.0 / t.radius, 0.0, o.min_lm_diagonal, o.max_lm_diagonal};
        ev.lm_state(t, to, d);
        StepInfo s = ev.try_step(d, t.need_gcheck ? o.gradient_tolerance : -1.0, false);
        if (s.status) return s.status;
        TrStep st;
        st.grad_converged = s.grad_converged; st.ok = s.ok; st.grad_max = s.grad_max; st.cand_cost = s.cand_cost;
        st.model_change = s.model_change; st.step_norm2 = s.step_norm2; st.x_norm2 = s.x_norm2;
        TrTrace tr;
        int what = tr_begin_step(t, st, tr);
        if (tr.push) trace_push(sum, tr.cost, tr.accepted, tr.damping, tr.grad);
        if (what != TR_GO_ON) continue;         // stopped (the loop condition sees it) or an invalid step: next try
        double cand = s.cand_cost, model_change = s.model_change, step_norm2 = s.step_norm2;
        int inner_useful = 0;
        if (t.inner_enabled && std::isfinite(cand)) {
            // TrustRegionMinimizer::DoInnerIterationsIfNeeded: what the pass gains is credited to the model, and
            // the passes stop for good once one gains less than the tolerance
            double inner_cost = 0, n2 = 0;
            const int ist = ev.inner_iterations(&inner_cost, &n2);
            if (ist) return ist;
            sum.num_inner_iteration_steps++;
            model_change += cand - inner_cost;
            inner_useful = inner_cost < t.cost;
            t.inner_enabled = (1.0 - inner_cost / cand) > o.inner_iteration_tolerance;
            cand = inner_cost;
            step_norm2 = n2;
        }
        int save_best = 0;
        what = tr_finish_step(t, to, cand, model_change, step_norm2, s.x_norm2, inner_useful, tr, &save_best);
        if (tr.push) trace_push(sum, tr.cost, tr.accepted, tr.damping, tr.grad);
        if (what == TR_ACCEPTED) {
            if (save_best) {
                int sb = ev.save_best();
                if (sb) return sb;
            }
            int ast = ev.accept();
            if (ast) return ast;
        }
    }
    double cost = t.cost;
    if (!t.best_is_current) {
        int rb = ev.restore_best();
        if (rb) return rb;
        cost = t.best_cost;
    }
    sum.iterations = t.it;
    sum.num_successful = t.num_successful;
    sum.num_linear_solves = t.num_linear_solves;
    sum.termination = t.term;
    sum.final_cost = cost;
    return CVX_OK;
}

// ---------------------------------------------------------------------------------------
// strategy 1: the Nielsen-damped loop of the reference's levmar (dlevmar_der, lm_core.c:163-409): same tests in the
// same order on the same quantities, so that a run reproduces the reference's info[] -- damping schedule and step
// tests in nielsen.h
// ---------------------------------------------------------------------------------------
static int lm_nielsen(Evaluator &ev, const cvx_lm_options &o, cvx_lm_summary &sum) {
    InitInfo ini = ev.init(false);
    if (ini.status) return ini.status;
    sum.initial_cost = ini.cost;
    double err_sq = 2.0 * ini.cost;                 // |e|^2 at the current point
    NielsenDamping damp = {0.0, 2};
    int stop = std::isfinite(err_sq) ? NS_RUNNING : NS_NOT_FINITE;
    int outer = 0;
    for (; outer < o.max_iterations && stop == NS_RUNNING; ++outer) {
        if (err_sq <= o.eps3) { stop = NS_SMALL_RESIDUAL; break; }
        if (outer == 0) damp.mu = o.tau * ini.diag_max;
        // inner loop: raise mu until a step reduces the error (or a test fires)
        for (bool first = true;; first = false) {
            const cvx_damping d = {0.0, damp.mu, 0.0, 0.0};
            const StepInfo s = ev.try_step(d, first ? o.eps1 : -1.0, false);
            if (s.status) return s.status;
            if (first && s.grad_converged) { stop = NS_SMALL_GRADIENT; break; }
            sum.num_linear_solves++;
            if (!s.ok) {
                trace_push(sum, NAN, 0, damp.mu, s.grad_max);
            } else {
                stop = nielsen_step_test(s.step_norm2, s.free_norm2, o.eps2);
                if (stop != NS_RUNNING) break;
                const double cand_sq = 2.0 * s.cand_cost;
                if (!std::isfinite(cand_sq)) { stop = NS_NOT_FINITE; break; }
                const double predicted = 2.0 * s.model_change;      // sum delta (mu delta - g)
                const double actual = err_sq - cand_sq;
                if (predicted > 0.0 && actual > 0.0) {
                    trace_push(sum, s.cand_cost, 1, damp.mu, s.grad_max);
                    damp.relax(actual, predicted);
                    const int st = ev.accept();
                    if (st) return st;
                    err_sq = cand_sq;
                    sum.num_successful++;
                    break;
                }
                trace_push(sum, s.cand_cost, 0, damp.mu, s.grad_max);
            }
            if (!damp.tighten()) { stop = NS_NO_REDUCTION; break; }
        }
        if (stop == NS_SMALL_GRADIENT) break;       // the reference tests the gradient before its inner loop: no count
    }
    if (outer >= o.max_iterations && stop == NS_RUNNING) stop = NS_ITERATION_LIMIT;
    sum.iterations = outer;
    sum.termination = stop;
    sum.final_cost = 0.5 * err_sq;
    return CVX_OK;
}

int lm_minimize(Evaluator &ev, const cvx_lm_options &opt, cvx_lm_summary &sum) {
    std::memset(&sum, 0, sizeof sum);
    auto t0 = std::chrono::steady_clock::now();
    ev.set_strategy(opt.strategy == 1 ? 1 : 0);
    int st = (opt.strategy == 1) ? lm_nielsen(ev, opt, sum) : lm_trust_region(ev, opt, sum);
    sum.seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return st;
}

}  // namespace cvx

// ---------------------------------------------------------------------------------------
// option presets (C ABI)
// ---------------------------------------------------------------------------------------
static void lm_defaults(cvx_lm_options *o) {
    std::memset(o, 0, sizeof *o);
    o->strategy = 0;
    o->max_iterations = 50;          // Ceres defaults
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->initial_radius = 1e4;
    o->max_radius = 1e16;
    o->min_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->jacobi_scaling = 1;
    o->nonmonotonic = 0;
    o->max_consecutive_nonmonotonic_steps = 5;
    o->tau = 1e-3;                   // levmar.h:90-91
    o->eps1 = o->eps2 = o->eps3 = 1e-17;
    o->inner_iterations = 0;
    o->inner_iteration_tolerance = 1e-3;
}

extern "C" void cvx_lm_options_refine(cvx_lm_options *o) {
    lm_defaults(o);
    o->nonmonotonic = 1;
    o->max_iterations = 100;
    o->function_tolerance = 1e-10;
    o->gradient_tolerance = 1e-10;
    o->inner_iterations = 1;         // calibration.cpp:424
}

extern "C" void cvx_lm_options_handeye(cvx_lm_options *o) {
    lm_defaults(o);
    o->max_iterations = 100;
}

extern "C" void cvx_lm_options_levmar(cvx_lm_options *o) {
    lm_defaults(o);
    o->strategy = 1;
    o->max_iterations = 100;
    o->eps1 = o->eps2 = o->eps3 = DBL_EPSILON;
}
