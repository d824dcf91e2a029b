// Damping schedule and stopping tests of the Nielsen-style Levenberg-Marquardt loop, shared by the host driver
// (lm_driver.cpp, strategy 1) and the batched homography kernel (homography_kernels.h).
//
// WHAT it computes is fixed by the reference: results must equal, bit for bit, those of the levmar 2.5 build the
// reference links (modules/util/src/3rdparty/levmar/lm_core.c:257-391, constants lm.c:35-36), so the arithmetic --
// every operation and its order -- is that algorithm's (H. B. Nielsen, "Damping parameter in Marquardt's method",
// IMM-REP-1999-05).  HOW it is written is this file's own: a damping object and a test object, no code of levmar.
#pragma once

#include <hip/hip_runtime.h>
#define CVX_HD __host__ __device__ inline

namespace cvx {

// termination codes as the reference's info[6] reports them (levmar.h:60-75)
enum NielsenStop {
    NS_RUNNING = 0,
    NS_SMALL_GRADIENT = 1,     // |J^T e|_inf <= eps1
    NS_SMALL_STEP = 2,         // |dp|^2 <= eps2^2 |p|^2
    NS_ITERATION_LIMIT = 3,
    NS_NEAR_SINGULAR = 4,      // the step is enormous against the parameters: restart with larger mu advised
    NS_NO_REDUCTION = 5,       // the growth factor overflowed without an acceptable step
    NS_SMALL_RESIDUAL = 6,     // |e|^2 <= eps3
    NS_NOT_FINITE = 7
};

struct NielsenDamping {
    double mu;
    int growth;

    CVX_HD void start(double tau, double largest_diagonal) { mu = tau * largest_diagonal; growth = 2; }

    // accepted step with gain ratio actual / predicted: mu shrinks by max(1/3, 1 - (2 gain - 1)^3)
    CVX_HD void relax(double actual, double predicted) {
        const double floor_ = 0.3333333334;                 // the build's "one third" (lm.c:36)
        double s = 2.0 * actual / predicted - 1.0;
        s = 1.0 - s * s * s;
        mu = mu * ((s >= floor_) ? s : floor_);
        growth = 2;
    }

    // rejected step: mu grows by the factor, the factor doubles; false once the factor has overflowed
    CVX_HD bool tighten() {
        mu *= growth;
        const int next = (int)((unsigned)growth << 1);
        if (next <= growth) return false;
        growth = next;
        return true;
    }
};

// the two tests on a freshly solved step (squared norms of the step and of the free parameters)
CVX_HD int nielsen_step_test(double step_sq, double param_sq, double eps2) {
    const double tiny = 1E-12;                              // lm.c:35
    if (step_sq <= eps2 * eps2 * param_sq) return NS_SMALL_STEP;
    if (step_sq >= (param_sq + eps2) / (tiny * tiny)) return NS_NEAR_SINGULAR;
    return NS_RUNNING;
}

}  // namespace cvx
