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
    double cost = ini.cost;
    sum.initial_cost = cost;
    double gmax = ini.grad_max;
    double radius = o.initial_radius, decrease_factor = 2.0;
    // TrustRegionStepEvaluator
    double ev_min = cost, ev_cur = cost, ev_ref = cost, ev_cand = cost, ev_acc_ref = 0, ev_acc_cand = 0;
    int ev_nonmono = 0;
    const int max_nonmono = o.nonmonotonic ? o.max_consecutive_nonmonotonic_steps : 0;
    int term = 0, invalid_run = 0, it = 0;
    bool need_gcheck = false;   // the gradient at the current point has not been tested yet
    bool inner_enabled = o.inner_iterations != 0;
    // lowest-cost point visited: the current one until an uphill step is accepted
    double best_cost = cost;
    bool best_is_current = true;
    if (gmax <= o.gradient_tolerance) term = 1;

    while (!term) {
        if (it >= o.max_iterations) {
            if (need_gcheck) {   // Ceres tests the gradient right after the accepted step
                cvx_damping d0 = {0, 0, 0, 0};
                StepInfo g = ev.try_step(d0, o.gradient_tolerance, true);
                if (g.status) return g.status;
                if (g.grad_converged) { term = 1; break; }
            }
            term = 4;
            break;
        }
        cvx_damping d = {1.0 / radius, 0.0, o.min_lm_diagonal, o.max_lm_diagonal};
        StepInfo s = ev.try_step(d, need_gcheck ? o.gradient_tolerance : -1.0, false);
        if (s.status) return s.status;
        if (s.grad_converged) { term = 1; break; }
        need_gcheck = false;
        gmax = s.grad_max;
        ++it;
        sum.num_linear_solves++;
        if (!s.ok || !(s.model_change > 0.0)) {
            // HandleInvalidStep + LevenbergMarquardtStrategy::StepIsInvalid
            trace_push(sum, NAN, 0, radius, gmax);
            if (++invalid_run >= 5) { term = 6; break; }
            radius *= 0.5;
            continue;
        }
        invalid_run = 0;
        double cand = s.cand_cost, model_change = s.model_change, step_norm2 = s.step_norm2;
        bool inner_useful = false;
        if (inner_enabled && std::isfinite(cand)) {
            // TrustRegionMinimizer::DoInnerIterationsIfNeeded: what the pass gains is credited to the model, and
            // the passes stop for good once one gains less than the tolerance
            double inner_cost = 0, n2 = 0;
            const int st = ev.inner_iterations(&inner_cost, &n2);
            if (st) return st;
            sum.num_inner_iteration_steps++;
            model_change += cand - inner_cost;
            inner_useful = inner_cost < cost;
            inner_enabled = (1.0 - inner_cost / cand) > o.inner_iteration_tolerance;
            cand = inner_cost;
            step_norm2 = n2;
        }
        const double sn = std::sqrt(step_norm2), xn = std::sqrt(s.x_norm2);
        if (sn <= o.parameter_tolerance * (xn + o.parameter_tolerance)) {
            trace_push(sum, cand, 0, radius, gmax);
            term = 2;
            break;
        }
        if (std::fabs(cost - cand) <= o.function_tolerance * cost) {
            trace_push(sum, cand, 0, radius, gmax);
            term = 3;
            break;
        }
        const double rel = (ev_cur - cand) / model_change;
        const double hist = (ev_ref - cand) / (ev_acc_ref + model_change);
        const double rho = (max_nonmono > 0 && hist > rel) ? hist : rel;
        if (inner_useful || rho > o.min_relative_decrease) {      // IsStepSuccessful
            trace_push(sum, cand, 1, radius, gmax);
            if (cand < best_cost) {
                best_cost = cand; best_is_current = true;
            } else if (best_is_current) {
                int sb = ev.save_best();
                if (sb) return sb;
                best_is_current = false;
            }
            int st = ev.accept();
            if (st) return st;
            cost = cand;
            double t = 2.0 * rho - 1.0;
            double f = 1.0 - t * t * t;
            if (f < 1.0 / 3.0) f = 1.0 / 3.0;
            radius = radius / f;
            if (radius > o.max_radius) radius = o.max_radius;
            decrease_factor = 2.0;
            ev_cur = cand; ev_acc_cand += model_change; ev_acc_ref += model_change;
            if (ev_cur < ev_min) { ev_min = ev_cur; ev_nonmono = 0; ev_cand = ev_cur; ev_acc_cand = 0; }
            else { ++ev_nonmono; if (ev_cur > ev_cand) { ev_cand = ev_cur; ev_acc_cand = 0; } }
            if (ev_nonmono == max_nonmono) { ev_ref = ev_cand; ev_acc_ref = ev_acc_cand; }
            sum.num_successful++;
            need_gcheck = true;
        } else {
            trace_push(sum, cand, 0, radius, gmax);
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
            if (radius < o.min_radius) { term = 5; break; }
        }
    }
    if (!best_is_current) {
        int rb = ev.restore_best();
        if (rb) return rb;
        cost = best_cost;
    }
    sum.iterations = it;
    sum.termination = term;
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
