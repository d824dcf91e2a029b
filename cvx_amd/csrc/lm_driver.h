// Host-side Levenberg-Marquardt control flow.  Pure scalar logic: every vector the
// iteration touches lives behind the Evaluator (in HBM for the product).
//
// Restates, in terms of whole-problem scalars,
//   strategy 0: the trust-region LM that ceres::Solve runs for
//               /root/reference/modules/calib/src/calibration.cpp:420-433 and handeye.cpp:310-322
//               (TrustRegionMinimizer + LevenbergMarquardtStrategy + TrustRegionStepEvaluator;
//                Ceres is un-vendored, restated from its published algorithm; inner iterations as
//                DoInnerIterationsIfNeeded / IsStepSuccessful use them, the pass itself is the
//                evaluator's);
//   strategy 1: levmar's dlevmar_der loop,
//               /root/reference/modules/util/src/3rdparty/levmar/lm_core.c:163-409.
#pragma once
#include "../../include/cvx_amd.h"
#include "trust_region.h"

namespace cvx {

struct InitInfo {
    int status = 0;        // cvx_status
    double cost = 0;       // 1/2 sum r^2 at x0 (all ranks)
    double grad_max = 0;   // max-norm of the (projected) gradient at x0
    double diag_max = 0;   // max_j (J^T J)_jj, levmar's mu0 scale
};

struct StepInfo {
    int status = 0;
    bool grad_converged = false;  // grad_max <= gtol: nothing else was computed
    bool ok = false;              // the damped system was solved (all blocks positive definite)
    double cost = 0;              // at the current point
    double grad_max = 0;
    double cand_cost = 0;         // at the candidate point
    double model_change = 0;      // 1/2 delta^T (D delta - g)
    double step_norm2 = 0;        // |x - x_cand|^2 in global coordinates
    double x_norm2 = 0;           // |x|^2 over non-constant parameter blocks (Ceres)
    double free_norm2 = 0;        // |p|^2 over free scalars (levmar)
};

class Evaluator {
public:
    virtual ~Evaluator() {}
    // 0: Ceres-style (projected gradient norm, global step norm); 1: levmar (tangent-space norms)
    virtual void set_strategy(int) {}
    // linearise at the current point; when jacobi_scaling, fix s_j = 1/(1+sqrt(d_j)) from it
    virtual InitInfo init(bool jacobi_scaling) = 0;
    // linearise at the current point, stop early when grad_max <= gtol (gtol < 0: never) or when
    // gradient_only; otherwise solve with damping d, form the candidate and evaluate it
    virtual StepInfo try_step(const cvx_damping &d, double gtol, bool gradient_only) = 0;
    // current point <- last candidate
    virtual int accept() = 0;
    // Non-monotonic runs may accept uphill steps and must hand back the lowest-cost point they visited (what
    // ceres::Solve documents for use_nonmonotonic_steps, set at calibration.cpp:421): save_best() copies the
    // current point aside -- called before an uphill accept leaves the best point behind -- and restore_best()
    // makes that copy the current point again.
    virtual int save_best() = 0;
    virtual int restore_best() = 0;
    // One inner-iteration pass (CoordinateDescentMinimizer::Minimize) on the last candidate, in place: *cost = its
    // cost afterwards, *step_norm2 = |x - candidate|^2 afterwards.  Evaluators without one refuse.
    // Strategy 0 tells the evaluator, right before every try_step that solves a linear step, the loop's state and the
    // damping it is about to ask for (an evaluator may use it to run the loop's decisions ahead of the host: calib.hip)
    virtual void lm_state(const TrState &, const TrOptions &, const cvx_damping &) {}
    virtual int inner_iterations(double *cost, double *step_norm2) { (void)cost; (void)step_norm2; return CVX_ERR_UNSUPPORTED; }
};

int lm_minimize(Evaluator &ev, const cvx_lm_options &opt, cvx_lm_summary &sum);

}  // namespace cvx
