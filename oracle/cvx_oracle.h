/*
 * cvx_oracle.h -- CPU restatement of the cvx_calibration NLLS inner loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cvx_amd/ (the product) may include,
 * link, load or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker / the timed CPU port.
 *
 * What it restates (all citations relative to /root/reference):
 *   - OpenCVReprojectionError + ApplyRadialDistortionCameraIntrinsics
 *       modules/calib/src/calibration.cpp:219-315
 *   - problem structure of CameraCalibration::refine  calibration.cpp:354-464
 *   - PoseError2 and solveNonLinear                   modules/calib/src/handeye.cpp:219-333
 *   - levmar 2.5 dlevmar_der control flow             modules/util/src/3rdparty/levmar/lm_core.c:60-423
 *   - Ceres Solver pieces the reference calls but does not vendor
 *     (AngleAxisRotatePoint, QuaternionParameterization, the Levenberg-Marquardt
 *     trust-region loop, inner iterations with their automatic ordering): restated
 *     from Ceres' published algorithm; version is
 *     unpinned by the reference (libceres-dev, modules/calib/README.md:7).
 *
 * Parity pinning status:
 *   - LM engine (orc_lm_nielsen_der): pinned against the reference's own levmar
 *     known answers (modules/util/test/test_lm.cpp:62, levmar/lmdemo.c) and
 *     against levmar itself compiled from /root/reference into oracle/_ref.
 *   - Schur normal equations + Nielsen damping on the calibration functor:
 *     pinned against levmar (dense) driven by this file's callbacks
 *     (tests/golden/levmar_calib_trace.json).
 *   - The two cost functors themselves and the Ceres trust-region loop (inner
 *     iterations included):
 *     PARITY UNPINNED -- the reference holds no golden vectors for them and
 *     Ceres/Eigen/OpenCV are absent from this image.  They are cross-checked
 *     three ways instead (analytic vs. forward-mode dual numbers vs. mpmath
 *     50-digit fixtures).
 */
#ifndef CVX_ORACLE_H
#define CVX_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* intrinsics block order, calibration.cpp:219-229 */
enum { ORC_FX = 0, ORC_FY, ORC_CX, ORC_CY, ORC_K1, ORC_K2, ORC_K3, ORC_P1, ORC_P2 };

/* ---- per-observation functor ------------------------------------------- */
/* analytic residual + Jacobian blocks, row-major: Jc[2][9], Jp[2][6], Jx[2][3] */
void orc_reproj_analytic(const double intr[9], const double pose[6], const double X[3],
                         double ox, double oy, double r[2],
                         double Jc[18], double Jp[12], double Jx[6]);
/* same through forward-mode dual numbers on the literal reference expression
 * (what ceres::AutoDiffCostFunction<...,2,9,6,3> evaluates) */
void orc_reproj_autodiff(const double intr[9], const double pose[6], const double X[3],
                         double ox, double oy, double r[2],
                         double Jc[18], double Jp[12], double Jx[6]);

/* hand-eye PoseError2 (handeye.cpp:228-261): q = (w,x,y,z) unnormalised, r[12],
 * Jq[12][4] (global), Jt[12][3] */
void orc_handeye_analytic(const double q[4], const double t[3],
                          const double qa[4], const double ta[3],
                          const double qb[4], const double tb[3],
                          double r[12], double Jq[48], double Jt[36]);
void orc_handeye_autodiff(const double q[4], const double t[3],
                          const double qa[4], const double ta[3],
                          const double qb[4], const double tb[3],
                          double r[12], double Jq[48], double Jt[36]);

/* Ceres rotation helpers used by pack/unpack (calibration.cpp:318-350).
 * R is COLUMN-major 3x3, as Eigen's &R(0,0). */
void orc_angle_axis_to_rotation(const double w[3], double Rcolmajor[9]);
void orc_rotation_to_angle_axis(const double Rcolmajor[9], double w[3]);
/* Ceres QuaternionParameterization::Plus */
void orc_quaternion_plus(const double x[4], const double delta[3], double out[4]);

/* ---- calibration problem ------------------------------------------------ */
typedef struct {
    int C, V, P, N;
    const int *cam_of_view;   /* [V] */
    const int *view_off;      /* [V+1] CSR into the observation arrays */
    const int *point_id;      /* [N] */
    const double *obs_x, *obs_y; /* [N] */
    int fix_k3;               /* calibration.cpp:412 -> 1 */
    int fix_intrinsics;       /* refine(..., intrinsics=false) */
    int fix_points;           /* not in the reference (points are free there) */
} orc_calib_problem;

typedef struct {
    int strategy;             /* 0 = Ceres-style LM trust region, 1 = levmar/Nielsen */
    int max_iterations;
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    double initial_radius, max_radius, min_radius, min_relative_decrease;
    double min_lm_diagonal, max_lm_diagonal;
    int jacobi_scaling;
    int nonmonotonic;
    int max_consecutive_nonmonotonic_steps;
    double tau, eps1, eps2, eps3;   /* Nielsen: lm_core.c:128-141 */
    int inner_iterations;           /* Solver::Options::use_inner_iterations (calibration.cpp:424) */
    double inner_iteration_tolerance;   /* Ceres default 1e-3 */
} orc_lm_options;

#define ORC_TRACE_MAX 1024
typedef struct {
    int iterations;           /* outer iterations performed (successful + unsuccessful for Ceres-style) */
    int num_successful;
    int num_linear_solves;
    int termination;          /* Ceres-style: 1 gradient tol, 2 parameter tol, 3 function tol,
                                 4 max iterations, 5 radius too small, 6 solve failure.
                                 Nielsen: levmar's info[6] codes 1..7 */
    double initial_cost, final_cost;   /* 1/2 sum r^2 */
    int trace_len;
    double trace_cost[ORC_TRACE_MAX];      /* cost of the CANDIDATE evaluated at each linear solve */
    double trace_accepted[ORC_TRACE_MAX];  /* 1/0 */
    double trace_damping[ORC_TRACE_MAX];   /* radius (Ceres) or mu (Nielsen) used for the solve */
    double trace_gradient[ORC_TRACE_MAX];  /* max-norm of gradient at the linearisation point */
    int num_inner_iteration_steps;         /* outer iterations whose candidate went through inner iterations */
} orc_lm_summary;

void orc_lm_options_refine(orc_lm_options *o);   /* calibration.cpp:420-428 + Ceres defaults */
void orc_lm_options_handeye(orc_lm_options *o);  /* handeye.cpp:310-319 + Ceres defaults */
void orc_lm_options_levmar(orc_lm_options *o);   /* lm.hpp:43-50,61 */

int orc_calib_num_shared(const orc_calib_problem *pb);  /* n_s = 8C(or 0) + 3P(or 0) */

/* 1/2 sum r^2 */
double orc_calib_cost(const orc_calib_problem *pb, const double *intr, const double *poses,
                      const double *points);
/* r[2N] interleaved (rx,ry per obs), Jc[N*18], Jp[N*12], Jx[N*6]; any output may be NULL */
void orc_calib_eval(const orc_calib_problem *pb, const double *intr, const double *poses,
                    const double *points, double *r, double *Jc, double *Jp, double *Jx);
/* dense Jacobian [2N x m] row-major over FREE parameters, ordered
 * [intrinsics (8 or 9 or 0 per camera) | poses 6V | points 3P (or 0)] */
int orc_calib_num_free(const orc_calib_problem *pb);
void orc_calib_dense_jacobian(const orc_calib_problem *pb, const double *intr,
                              const double *poses, const double *points, double *J);
/* undamped Schur pieces at the given parameters:
 *   Vv[V*36] row-major pose Hessians, gv[V*6], U[ns*ns], gs[ns],
 *   W[V*6*ns] (dense embedding, may be NULL), cost
 * shared-block ordering: [cam0 free intrinsics | cam1 ... | point0 xyz | point1 ...] */
void orc_calib_normal_blocks(const orc_calib_problem *pb, const double *intr, const double *poses,
                             const double *points, double *Vv, double *gv, double *U, double *gs,
                             double *W, double *cost);
/* reduced system for per-parameter additive damping Dp[6V], Ds[ns]:
 *   S = U + diag(Ds) - sum_v W_v^T (V_v + diag(Dp_v))^-1 W_v,  gred = gs - sum_v W_v^T (..)^-1 g_v */
void orc_calib_reduced_system(const orc_calib_problem *pb, const double *intr, const double *poses,
                              const double *points, const double *Dp, const double *Ds,
                              double *S, double *gred, double *cost);
/* full LM; intr[C*9], poses[V*6], points[P*3] are in/out */
int orc_calib_solve(const orc_calib_problem *pb, double *intr, double *poses, double *points,
                    const orc_lm_options *opt, orc_lm_summary *sum);
/* the value CameraCalibration::refine returns: sqrt(total_cost / (2N)), calibration.cpp:463 */
double orc_calib_refine_return(double total_cost, int N);
/* Inner iterations (use_inner_iterations = true, calibration.cpp:424; no inner_iteration_ordering is given, so
 * Ceres builds one: CoordinateDescentMinimizer::CreateOrdering = recursive greedy independent sets of the Hessian
 * graph by ascending degree, reversed).  Restated from Ceres' published algorithm, PARITY UNPINNED.
 * group[C + P + V]: block ids are camera c -> c, point p -> C + p, pose v -> C + P + v; the value is the position
 * of the block's independent set in the order the sets are minimised (0 first), -1 for a block that is constant or
 * in no residual.  Ties of degree are broken by the order in which refine's AddResidualBlock calls first name the
 * blocks (Ceres breaks them by address).  Returns the number of sets. */
int orc_calib_inner_ordering(const orc_calib_problem *pb, int *group);
/* one pass of CoordinateDescentMinimizer::Minimize over x = (intr, poses, points) in place: every block of every
 * set in turn is minimised over its own residuals, all other blocks held, by the trust-region LM with Ceres'
 * default options; returns the cost of the whole problem afterwards */
double orc_calib_inner_pass(const orc_calib_problem *pb, double *intr, double *poses, double *points);

/* ---- hand-eye ------------------------------------------------------------ */
typedef struct {
    int n;                                  /* motion pairs */
    const double *qa, *ta, *qb, *tb;        /* [n*4],[n*3],[n*4],[n*3], q = (w,x,y,z) */
} orc_handeye_problem;

double orc_handeye_cost(const orc_handeye_problem *pb, const double p[7]);
/* local (6-dof) normal equations at p: H[36], g[6], cost */
void orc_handeye_normal_eq(const orc_handeye_problem *pb, const double p[7],
                           double H[36], double g[6], double *cost);
int orc_handeye_solve(const orc_handeye_problem *pb, double p[7],
                      const orc_lm_options *opt, orc_lm_summary *sum);

/* ---- levmar restatement (dense), lm_core.c:60-423 ------------------------ */
typedef void (*orc_lm_func)(double *p, double *hx, int m, int n, void *adata);
int orc_lm_nielsen_der(orc_lm_func func, orc_lm_func jacf, double *p, const double *x,
                       int m, int n, int itmax, const double opts[4], double info[10],
                       void *adata);

/* ---- tool-level bundle adjustment: hand_eye_registration.cpp:48-238 / camera_ba.cpp:47-240 ---- */
/* One global pose p = {q.w,q.x,q.y,q.z,t} (QuaternionParameterization), free board points, constant intrinsics,
 * a fixed rigid pre-transform per view (gripper_to_base[i], row-major 3x4), HuberLoss(huber) on every 2-residual
 * block (hand_eye_registration.cpp:168,181-188); huber <= 0: no loss function. */
typedef struct {
    int V, P, N;
    const int *view_off, *point_id;
    const double *obs_x, *obs_y;
    const double *trans;      /* [V][12] */
    double intr[9];           /* fx,fy,cx,cy,k1,k2,k3,p1,p2 (constants: hand_eye_registration.cpp:67-77) */
    double huber;             /* 1.0 in the reference */
} orc_toolba_problem;
/* functor hand_eye_registration.cpp:60-140: r[2], Jq[2][4] (global quaternion derivative), Jt[2][3], Jx[2][3] */
void orc_toolba_analytic(const double intr[9], const double trans[12], const double q[4], const double t[3],
                         const double X[3], double ox, double oy, double r[2], double Jq[8], double Jt[6], double Jx[6]);
void orc_toolba_autodiff(const double intr[9], const double trans[12], const double q[4], const double t[3],
                         const double X[3], double ox, double oy, double r[2], double Jq[8], double Jt[6], double Jx[6]);
/* ceres::HuberLoss::Evaluate: rho[0..2] at s = |r|^2 */
void orc_huber(double a, double s, double rho[3]);
double orc_toolba_cost(const orc_toolba_problem *pb, const double p[7], const double *points);   /* 1/2 sum rho(s) */
/* robustified normal equations over the tangent space [pose 6 | points 3P]: H[(6+3P)^2], g[6+3P] */
void orc_toolba_normal_eq(const orc_toolba_problem *pb, const double p[7], const double *points, double *H, double *g,
                          double *cost);
void orc_lm_options_toolba(orc_lm_options *o);   /* hand_eye_registration.cpp:201-210 */
/* one inner-iteration pass (:205; blocks q, t and every point, in Ceres' automatic order) on (p, points) in place;
 * returns the cost afterwards */
double orc_toolba_inner_pass(const orc_toolba_problem *pb, double p[7], double *points);
int orc_toolba_solve(const orc_toolba_problem *pb, double p[7], double *points, const orc_lm_options *opt,
                     orc_lm_summary *sum);

/* ---- HomographyEstimator::solveLM, homography.cpp:159-236 ------------------ */
/* errors (homography.cpp:166-186) and jacobian (:188-207) of LMHomographyEstimator over n point pairs
 * (already gathered by the inlier list); h[8] in/out; opts as LMSolver::minimizeDer builds them
 * (lm.hpp:61: {factor, g_tol, x_tol, g_tol}); x = NULL.  engine: NULL = this file's restatement
 * (orc_lm_nielsen_der), else the address of the reference's dlevmar_der (oracle/_ref).  Returns what
 * dlevmar_der returns; -1 and h untouched when 2n < 8 (lm_core.c:117-120). */
int orc_homography_lm(void *engine, int n, const float *x1, const float *y1, const float *x2, const float *y2,
                      double h[8], int itmax, const double opts[4], double info[10]);
void orc_homography_errors(int n, const float *x1, const float *y1, const float *x2, const float *y2,
                           const double h[8], double *e /* 2n */);
void orc_homography_jacobian(int n, const float *x1, const float *y1, const double h[8], double *J /* 2n x 8 */);

/* ---- HomographyEstimator::solve (normalised DLT) and ::residuals, homography.cpp:17-110 ---- */
/* PARITY UNPINNED: the reference holds no vectors for these, and the SVD is Eigen's (un-vendored): restated as a
 * one-sided Jacobi SVD, pinned against LAPACK's SVD of the same matrix in tests/test_homography_oracle.py. */
int orc_homography_dlt(int n, const float *x1, const float *y1, const float *x2, const float *y2, double H[9]);
void orc_homography_dlt_matrix(int n, const float *x1, const float *y1, const float *x2, const float *y2,
                               double *A /* 2n x 9 */, double T1[9], double T2[9]);
int orc_null_vector9(int m, double *A /* m x 9, destroyed */, double v[9]);
void orc_homography_residuals(int n, const float *x1, const float *y1, const float *x2, const float *y2,
                              const double H[9], double *res /* n */);

/* ---- HomographyEstimator::solveRansac, homography.cpp:112-156 on cvx::util::RANSAC::estimate, ransac.hpp:28-78 ---- */
/* PARITY UNPINNED (the reference holds no vectors; its generator is seeded from std::random_device, rng.cpp:6-9, so no
 * two runs of the reference agree either): the loop is restated on a GIVEN sequence of subsets -- what RNG::sample would
 * hand it trial after trial -- so that the same sequence gives the same model. */
typedef struct orc_ransac_params {   /* cvx::util::RANSAC::Parameters, ransac.hpp:12-23 */
    double max_error, outlier_probability;
    int max_num_trials, min_inliers;
} orc_ransac_params;
void orc_ransac_defaults(orc_ransac_params *p);     /* {0.1, 0.1, 100, 5}: solveRansac builds `RANSAC solver;` (homography.cpp:152) */
/* RNG::sample(4, N, subset) (rng.hpp:62-75) given the integers its four uniform_int_distribution(0, max_idx) draws return */
void orc_rng_sample4(int N, const int draw[4], int subset[4]);
/* subsets: [max_num_trials][4].  Returns estimate()'s value; H (row-major) = the refit on the best trial's inliers,
 * untouched on failure; inliers[n] (capacity n), *n_inliers; *trials_run = trial_count when the loop ended. */
int orc_homography_ransac(int n, const float *x1, const float *y1, const float *x2, const float *y2,
                          const orc_ransac_params *p, const int *subsets, double H[9], int *inliers, int *n_inliers,
                          int *trials_run);

/* LU solve of the no-LAPACK levmar build (Axb_core.c:992-1139); A, b preserved */
int orc_lu_solve(int n, const double *A, const double *b, double *x);
/* dense SPD solve used by the drivers (Cholesky); returns 0 on failure */
int orc_cholesky_solve(int n, double *A /* destroyed */, double *b /* in: rhs, out: x */);

/* CPU-baseline knob (bench.py only): OpenMP threads for the per-view loops of the calibration model; 1 (the
 * default, and what every test uses) keeps the single-thread summation order. */
void orc_set_threads(int n);
int orc_get_threads(void);

#ifdef __cplusplus
}
#endif
#endif
