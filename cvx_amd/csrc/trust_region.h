// The per-iteration decisions of the Ceres-style trust-region loop (lm_driver.cpp, strategy 0) as functions of a plain
// state record, compiled for the host AND the device: lm_trust_region() runs them on the host, and the calibration path's
// last kernel of a step runs the very same code on the device, so that the next step's kernels can be enqueued before
// the host has seen this step's result (calib.hip, "speculated steps").  Same operations in the same order on both
// sides (IEEE +, -, *, /, sqrt; no contraction), hence the same decisions bit for bit.
// Restates TrustRegionMinimizer / LevenbergMarquardtStrategy / TrustRegionStepEvaluator as ceres::Solve runs them for
// /root/reference/modules/calib/src/calibration.cpp:420-433 (Ceres is un-vendored: from its published algorithm).
#pragma once
#include <cmath>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define CVX_TR_HD __host__ __device__
#else
#define CVX_TR_HD
#endif

namespace cvx {

struct TrOptions {
    int max_iterations, max_nonmono;
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    double max_radius, min_radius, min_relative_decrease;
};

struct TrState {
    double cost, gmax, radius, decrease_factor;
    // TrustRegionStepEvaluator
    double ev_min, ev_cur, ev_ref, ev_cand, ev_acc_ref, ev_acc_cand;
    double best_cost;           // lowest cost visited
    int ev_nonmono;
    int term;                   // 0 while the solve goes on
    int invalid_run, it;
    int need_gcheck;            // the gradient at the current point has not been tested yet
    int best_is_current;        // the current point is the lowest-cost one
    int inner_enabled;
    int num_successful, num_linear_solves;
};

// what a linear step hands to the loop (StepInfo of lm_driver.h, the part the decisions read)
struct TrStep {
    int grad_converged, ok;
    double grad_max, cand_cost, model_change, step_norm2, x_norm2;
};

// one line of the solve's trace
struct TrTrace {
    int push;
    double cost;
    int accepted;
    double damping, grad;
};

enum { TR_GO_ON = 0, TR_ACCEPTED = 1, TR_REJECTED = 2, TR_INVALID = 3, TR_STOP = 4 };

CVX_TR_HD inline void tr_init(TrState &s, double cost, double gmax, double initial_radius, int inner_enabled) {
    s.cost = cost; s.gmax = gmax; s.radius = initial_radius; s.decrease_factor = 2.0;
    s.ev_min = s.ev_cur = s.ev_ref = s.ev_cand = cost; s.ev_acc_ref = 0.0; s.ev_acc_cand = 0.0;
    s.best_cost = cost;
    s.ev_nonmono = 0; s.term = 0; s.invalid_run = 0; s.it = 0; s.need_gcheck = 0; s.best_is_current = 1;
    s.inner_enabled = inner_enabled; s.num_successful = 0; s.num_linear_solves = 0;
}

// First half, right behind the linear step: the gradient test Ceres makes after an accepted step, the iteration
// count, and HandleInvalidStep + LevenbergMarquardtStrategy::StepIsInvalid.  TR_GO_ON: the candidate is valid
// (inner iterations, if any, run next).
CVX_TR_HD inline int tr_begin_step(TrState &s, const TrStep &st, TrTrace &tr) {
    tr.push = 0;
    if (st.grad_converged) { s.term = 1; return TR_STOP; }
    s.need_gcheck = 0;
    s.gmax = st.grad_max;
    ++s.it;
    ++s.num_linear_solves;
    if (!st.ok || !(st.model_change > 0.0)) {
        tr.push = 1; tr.cost = NAN; tr.accepted = 0; tr.damping = s.radius; tr.grad = s.gmax;
        if (++s.invalid_run >= 5) { s.term = 6; return TR_STOP; }
        s.radius *= 0.5;
        return TR_INVALID;
    }
    s.invalid_run = 0;
    return TR_GO_ON;
}

// Second half, with the candidate's cost (after an inner-iteration pass: its cost, the model change credited with
// what the pass gained, the step norm recomputed): tolerances, step quality, accept / reject, radius.
// *save_best: the step is accepted uphill while the current point is the best one -- the caller copies the current
// point aside BEFORE it moves to the candidate.
CVX_TR_HD inline int tr_finish_step(TrState &s, const TrOptions &o, double cand, double model_change, double step_norm2,
                                 double x_norm2, int inner_useful, TrTrace &tr, int *save_best) {
    *save_best = 0;
    tr.push = 1; tr.cost = cand; tr.accepted = 0; tr.damping = s.radius; tr.grad = s.gmax;
    const double sn = sqrt(step_norm2), xn = sqrt(x_norm2);
    if (sn <= o.parameter_tolerance * (xn + o.parameter_tolerance)) { s.term = 2; return TR_STOP; }
    if (fabs(s.cost - cand) <= o.function_tolerance * s.cost) { s.term = 3; return TR_STOP; }
    const double rel = (s.ev_cur - cand) / model_change;
    const double hist = (s.ev_ref - cand) / (s.ev_acc_ref + model_change);
    const double rho = (o.max_nonmono > 0 && hist > rel) ? hist : rel;
    if (inner_useful || rho > o.min_relative_decrease) {      // IsStepSuccessful
        tr.accepted = 1;
        if (cand < s.best_cost) {
            s.best_cost = cand; s.best_is_current = 1;
        } else if (s.best_is_current) {
            *save_best = 1;
            s.best_is_current = 0;
        }
        s.cost = cand;
        const double t = 2.0 * rho - 1.0;
        double f = 1.0 - t * t * t;
        if (f < 1.0 / 3.0) f = 1.0 / 3.0;
        s.radius = s.radius / f;
        if (s.radius > o.max_radius) s.radius = o.max_radius;
        s.decrease_factor = 2.0;
        s.ev_cur = cand; s.ev_acc_cand += model_change; s.ev_acc_ref += model_change;
        if (s.ev_cur < s.ev_min) { s.ev_min = s.ev_cur; s.ev_nonmono = 0; s.ev_cand = s.ev_cur; s.ev_acc_cand = 0.0; }
        else { ++s.ev_nonmono; if (s.ev_cur > s.ev_cand) { s.ev_cand = s.ev_cur; s.ev_acc_cand = 0.0; } }
        if (s.ev_nonmono == o.max_nonmono) { s.ev_ref = s.ev_cand; s.ev_acc_ref = s.ev_acc_cand; }
        s.num_successful++;
        s.need_gcheck = 1;
        return TR_ACCEPTED;
    }
    s.radius = s.radius / s.decrease_factor;
    s.decrease_factor *= 2.0;
    if (s.radius < o.min_radius) { s.term = 5; return TR_STOP; }
    return TR_REJECTED;
}

}  // namespace cvx
