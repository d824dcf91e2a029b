/*
 * cvx_amd.h -- C ABI of the MI355X-native cvx_calibration NLLS inner loop.
 *
 * One shared library, libcvx_amd.so, built from cvx_amd/csrc (HIP for gfx950 +
 * host C++).  Plain pointers and sizes only; no C++ or torch types cross this
 * boundary.  The reference (angeltop/cvx) has no FFI layer of its own for this
 * path: the functions below are what a header-only adaptor behind
 *   double CameraCalibration::refine(const Data&, PinholeCamera&, std::vector<Eigen::Matrix4d>&, bool)
 *       modules/calib/include/cvx/calib/calibration.hpp:33, impl modules/calib/src/calibration.cpp:354-464
 *   bool HandEyeCalibration::solveFixed / solveMoving(...)
 *       modules/calib/include/cvx/calib/handeye.hpp:25-30, impl modules/calib/src/handeye.cpp:665-764
 * binds (see INTEGRATION.md and include/cvx_amd_adaptor.hpp).
 *
 * Every function returns a cvx_status (0 = ok) unless stated.  All host
 * pointers; device memory is owned by the handles.  Handles are
 * thread-compatible (one thread at a time per handle), no global state.
 * The library never falls back to a CPU path: without a usable HIP device
 * every compute entry point returns CVX_ERR_NO_DEVICE.
 */
#ifndef CVX_AMD_H
#define CVX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    CVX_OK = 0,
    CVX_ERR_INVALID_ARG = 1,
    CVX_ERR_NO_DEVICE = 2,
    CVX_ERR_HIP = 3,
    CVX_ERR_UNSUPPORTED = 4,       /* size beyond a documented kernel limit */
    CVX_ERR_DUPLICATE_MARKER = 5,  /* a view lists the same board point twice */
    CVX_ERR_EXCHANGE = 6,          /* the all-reduce hook failed */
    CVX_ERR_NUMERIC = 7            /* non-finite cost at the initial point */
} cvx_status;

const char *cvx_status_string(int status);
/* last HIP / RCCL error text seen on this thread ("" if none) */
const char *cvx_last_error(void);
/* number of visible HIP devices (0 without a GPU; never fails) */
int cvx_device_count(void);

/* ------------------------------------------------------------------------ */
/* Levenberg-Marquardt control (host side, shared by both problems)          */
/* ------------------------------------------------------------------------ */
/* strategy 0: the trust-region LM ceres::Solve runs with the options set at
 *   calibration.cpp:420-428 / handeye.cpp:310-319 (Ceres defaults elsewhere);
 * strategy 1: levmar's Nielsen damping, lm_core.c:257-409 (what LMSolver,
 *   modules/util/include/cvx/util/math/solvers/lm.hpp:39-110, drives). */
typedef struct {
    int strategy;
    int max_iterations;
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    double initial_radius, max_radius, min_radius, min_relative_decrease;
    double min_lm_diagonal, max_lm_diagonal;
    int jacobi_scaling;
    int nonmonotonic;
    int max_consecutive_nonmonotonic_steps;
    double tau, eps1, eps2, eps3;
    int inner_iterations;               /* Solver::Options::use_inner_iterations (calibration.cpp:424,
                                           hand_eye_registration.cpp:205); strategy 0 of cvx_calib_solve and
                                           cvx_toolba_solve: cvx_handeye_solve returns CVX_ERR_UNSUPPORTED with it
                                           (one block, handeye.cpp:315), strategy 1 ignores it */
    double inner_iteration_tolerance;   /* Ceres default 1e-3 */
} cvx_lm_options;

#define CVX_TRACE_MAX 1024
typedef struct {
    int iterations, num_successful, num_linear_solves;
    int termination;   /* strategy 0: 1 gradient tol, 2 parameter tol, 3 function tol, 4 max iterations,
                                      5 radius below minimum, 6 repeated invalid steps;
                          strategy 1: levmar info[6] codes 1..7 */
    double initial_cost, final_cost;   /* 1/2 sum r^2 */
    int trace_len;
    double trace_cost[CVX_TRACE_MAX];      /* candidate cost of every linear solve (NaN: solve failed) */
    double trace_accepted[CVX_TRACE_MAX];
    double trace_damping[CVX_TRACE_MAX];   /* radius (strategy 0) or mu (strategy 1) */
    double trace_gradient[CVX_TRACE_MAX];  /* max-norm gradient at the linearisation point */
    double seconds_total;                  /* wall time inside the solve */
    int num_inner_iteration_steps;         /* linear solves whose candidate went through an inner-iteration pass */
} cvx_lm_summary;

/* calibration.cpp:420-428, use_inner_iterations (:424) included.  Deviation from what ceres::Solve does with those
 * options (the reference holds no vectors for this loop, parity unpinned): linear solves are
 * exact (the reference's ITERATIVE_SCHUR + SCHUR_JACOBI conjugate gradients are inexact).  Non-monotonic steps are
 * restated, including that the solve hands back the lowest-cost point it visited; inner iterations are, with the
 * ordering Ceres builds when none is given (cvx_calib_inner_ordering); on view-sharded handles too (the ordering
 * is then made from the whole problem's structure, gathered through the exchange hook at the first pass). */
void cvx_lm_options_refine(cvx_lm_options *o);
void cvx_lm_options_handeye(cvx_lm_options *o);  /* handeye.cpp:310-319 */
void cvx_lm_options_levmar(cvx_lm_options *o);   /* lm.hpp:43-50,61 */

/* ------------------------------------------------------------------------ */
/* Exchange step for view-sharded multi-GPU runs                             */
/* ------------------------------------------------------------------------ */
/* Sums `count` doubles in place over all ranks.  `on_device` tells whether buf
 * is device memory (then `stream` is the hipStream_t the data is ordered on)
 * or host memory.  Return 0 on success. */
typedef int (*cvx_allreduce_fn)(void *ctx, double *buf, size_t count, int on_device, void *stream);

/* built-in RCCL implementation of the hook (one communicator per process/GPU) */
#define CVX_RCCL_ID_BYTES 128
int cvx_rccl_unique_id(char id[CVX_RCCL_ID_BYTES]);
int cvx_rccl_create(int rank, int nranks, const char id[CVX_RCCL_ID_BYTES], int device, void **comm);
int cvx_rccl_allreduce(void *comm, double *buf, size_t count, int on_device, void *stream);
void cvx_rccl_destroy(void *comm);

/* contiguous view ranges balanced by observation count: first_view[nranks+1] */
int cvx_partition_views(int n_views, const int *view_off, int nranks, int *first_view);

/* ------------------------------------------------------------------------ */
/* Camera calibration bundle adjustment (CameraCalibration::refine)          */
/* ------------------------------------------------------------------------ */
typedef struct cvx_calib cvx_calib;

typedef struct {
    int n_cameras;            /* the reference has exactly one (calibration.hpp:31-33) */
    int n_views;              /* (camera, view) pose blocks held by THIS rank */
    int n_points;             /* board points, Data::coords_ (calibration.hpp:20) */
    int64_t n_obs;            /* sum of markers_[i].size() */
    const int *cam_of_view;   /* [n_views] */
    const int *view_off;      /* [n_views + 1] CSR offsets into the observation arrays */
    const int *point_id;      /* [n_obs]  Marker::id_  (pattern.hpp:23) */
    const double *obs_x;      /* [n_obs]  Marker::pt_.x promoted to double (calibration.cpp:393); or NULL: obs_xf / obs_yf */
    const double *obs_y;
    int fix_k3;               /* 1 in the reference (calibration.cpp:410-415) */
    int fix_intrinsics;       /* refine(..., intrinsics = false), calibration.cpp:407-408 */
    int fix_points;           /* 0 in the reference (board points are free blocks, calibration.cpp:397) */
    int device;               /* HIP device ordinal */
    void *stream;             /* hipStream_t to run on, or NULL: the handle creates its own */
    const float *obs_xf;      /* [n_obs] Marker::pt_ as the reference holds it (cv::Point2f), read when obs_x is NULL: */
    const float *obs_yf;      /* half the upload, the promotion to double (calibration.cpp:312-313) made on the device */
    int prepare_inner_iterations; /* 1: the host half of the inner iterations' first use (Ceres' ordering of the blocks
                               * into independent sets, the points' observation lists: ~2 ms at 1 M observations) starts
                               * on a host thread inside cvx_calib_create, beside its uploads -- for callers that go on
                               * to solve with the reference's options (use_inner_iterations, calibration.cpp:424) */
} cvx_calib_desc;

/* limits of this build (cvx_calib_create returns CVX_ERR_UNSUPPORTED beyond them).  A view lists a board point
 * at most once (CVX_ERR_DUPLICATE_MARKER), so its marker count is bounded by the point count; the normal-equation
 * kernel runs one observation per lane in workgroups of up to 512 lanes. */
#define CVX_MAX_POINTS 448
#define CVX_MAX_MARKERS_PER_VIEW CVX_MAX_POINTS
#define CVX_MAX_CAMERAS 64

int cvx_calib_create(const cvx_calib_desc *desc, cvx_calib **out);
/* the same preparation for a handle made without the flag: starts it on a host thread and returns at once (a no-op when
 * it is under way or done, and on view-sharded handles, whose first use is a collective).  cvx_calib_solve starts it
 * itself when its options ask for inner iterations, beside the initial linearisation. */
int cvx_calib_prepare_inner_iterations(cvx_calib *h);
/* cvx_calib_destroy keeps the handle's device memory (one block per device, at most 1 GiB) for the next
 * cvx_calib_create of a similar size instead of freeing it: the reference's refine() is one call, i.e. create -> solve ->
 * destroy here, and un-mapping / mapping a few hundred MB is 0.4 ms of it.  A handle of any kind that made its own stream
 * (desc.stream NULL) likewise leaves it, idle, for the next handle on the device (one per device): creating a stream and
 * the first copies on it cost 0.1-5 ms.  This hands the kept blocks and streams back to the runtime; CVX_ARENA_CACHE=0
 * in the environment switches the keeping off. */
int cvx_release_cached_memory(void);
void cvx_calib_destroy(cvx_calib *h);

/* parameter blocks in the reference's layouts: intrinsics[n_cameras][9] in the order
 * fx,fy,cx,cy,k1,k2,k3,p1,p2 (calibration.cpp:219-229); poses[n_views][6] = angle-axis, t
 * (calibration.cpp:318-332); points[n_points][3].  set_params returns once the caller's arrays have been copied into
 * the handle's pinned staging block (they may be freed or changed at once); the upload itself is ordered on the
 * handle's stream ahead of whatever the handle does next. */
int cvx_calib_set_params(cvx_calib *h, const double *intrinsics, const double *poses, const double *points);
int cvx_calib_get_params(cvx_calib *h, double *intrinsics, double *poses, double *points);

/* size of the reduced (Schur) system: free intrinsics of every camera, then free points */
int cvx_calib_num_shared(const cvx_calib *h);

/* 1/2 sum r^2 over this rank's observations (Problem::Evaluate, calibration.cpp:438-442) */
int cvx_calib_cost(cvx_calib *h, double *cost);

/* residuals and Jacobian blocks exactly as the AutoDiffCostFunction<..,2,9,6,3> of
 * calibration.cpp:391-398 yields them, per observation, row-major:
 *   r[n_obs][2], Jc[n_obs][2][9], Jp[n_obs][2][6], Jx[n_obs][2][3]; any may be NULL. */
int cvx_calib_evaluate(cvx_calib *h, double *cost, double *r, double *Jc, double *Jp, double *Jx);

/* The same evaluation left in HBM as component planes (SoA): r[2][n_obs], Jc[18][n_obs],
 * Jp[12][n_obs], Jx[6][n_obs] written into the handle's own device buffer (allocated on
 * first use, 36 doubles per observation + residuals).  Asynchronous on the handle's stream;
 * *device_buffer receives the device address (layout: r | Jc | Jp | Jx planes). */
int cvx_calib_evaluate_device(cvx_calib *h, void **device_buffer);

/* damping of one linear solve: D_j = clamp(s_j^2 d_j, min_diag, max_diag) * inv_radius / s_j^2 + mu
 * with d_j = (J^T J)_jj and s_j the Jacobi scale (1 unless cvx_calib_init_scaling ran) */
typedef struct {
    double inv_radius, mu, min_diag, max_diag;
} cvx_damping;

/* fixes s_j = 1 / (1 + sqrt(d_j)) from the Jacobian at the current parameters (Ceres
 * jacobi_scaling); enable = 0 resets s_j = 1 */
int cvx_calib_init_scaling(cvx_calib *h, int enable);

/* One damped linear solve at the current parameters, without changing them.
 * Outputs (any may be NULL): S[ns*ns] damped reduced matrix, gred[ns] reduced gradient,
 * delta_shared[ns], delta_poses[n_views*6], cost (at the current parameters),
 * cand_cost (at parameters + delta), model_change = 1/2 delta^T (D delta - g). */
int cvx_calib_linear_step(cvx_calib *h, const cvx_damping *d, double *S, double *gred,
                          double *delta_shared, double *delta_poses, double *cost,
                          double *cand_cost, double *model_change);

/* multi-GPU: this rank holds views [its shard]; shared blocks are replicated */
int cvx_calib_set_exchange(cvx_calib *h, cvx_allreduce_fn fn, void *ctx, int rank, int nranks,
                           int device_buffers);

/* the whole LM solve (ceres::Solve at calibration.cpp:433); parameters are updated in place */
int cvx_calib_solve(cvx_calib *h, const cvx_lm_options *opt, cvx_lm_summary *summary);

/* Inner iterations (use_inner_iterations, calibration.cpp:424).  The reference gives no inner_iteration_ordering,
 * so Ceres makes one (CoordinateDescentMinimizer::CreateOrdering: greedy independent sets of the Hessian graph by
 * ascending degree, recursively on what is left, then reversed).  group[n_cameras + n_points + n_views] receives,
 * for camera c at [c], point p at [n_cameras + p], pose v at [n_cameras + n_points + v], the position of the
 * block's set in the order the sets are minimised, or -1 for a block that is constant or occurs in no residual;
 * *n_sets the number of sets.  Degree ties go to the block refine's AddResidualBlock loop (:384-399) names first
 * (Ceres: lowest address).  Host only. */
int cvx_calib_inner_ordering(int n_cameras, int n_views, int n_points, const int *cam_of_view, const int *view_off,
                             const int *point_id, int fix_intrinsics, int fix_points, int *group, int *n_sets);
/* One pass of CoordinateDescentMinimizer::Minimize on the handle's current parameters, in place: set after set,
 * every block is minimised over its own residual blocks by an LM with Ceres' default options, all blocks of a set
 * at once on the device (inner_kernels.h).  *cost (may be NULL) receives 1/2 sum r^2 afterwards (over this rank's
 * observations, as cvx_calib_cost).  What cvx_calib_solve runs on every candidate while opt->inner_iterations is set
 * and the passes still pay.  On a view-sharded handle it is a collective: every rank calls it. */
int cvx_calib_inner_pass(cvx_calib *h, double *cost);

/* value CameraCalibration::refine returns: sqrt(total_cost / (2 n_obs)), calibration.cpp:463 */
double cvx_calib_refine_return(double total_cost, int64_t n_obs_total);

/* Per-phase device time, measured with hipEvents on the handle's stream while timing is enabled.
 * ms[0..7]: accumulated milliseconds of [0] normal-equation kernel, [1] Schur contraction,
 * [2] reduce + assemble, [3] back-substitution, [4] candidate cost, [5] damped reduced solve
 * (damp, tiled Cholesky, finish); ms[8..15]: launches counted per phase.  cvx_calib_enable_timing resets the accumulators. */
int cvx_calib_enable_timing(cvx_calib *h, int enable);
int cvx_calib_phase_times(cvx_calib *h, double ms[16]);
/* average duration (ms) of the Schur-contraction kernel k_schur_syrk over `reps` back-to-back launches on the
 * handle's stream at the current parameters, bracketed by two hipEvents: the per-launch figure bench.py's
 * `roofline` uses (per-phase events in a running iteration add stream bubbles of their own) */
int cvx_calib_time_schur(cvx_calib *h, int reps, double *ms_per_launch);

/* launch geometry, for stating algorithmic work: [0] Z columns, [1] reduced-system size,
 * [2] normal-equation workgroups, [3] Schur row chunks, [4] 16x16 blocks of the upper triangle of Z^T Z,
 * [5] threads per normal-equation workgroup, [6] Z rows (6 * n_views), [7] 32-column tile columns of the reduced
 * solve (launches of k_chol_step per LM iteration) */
int cvx_calib_geometry(const cvx_calib *h, int64_t out[8]);
/* Speculated steps since the handle was made.  With CVX_CALIB_SPECULATE=1 in the environment cvx_calib_solve (strategy
 * 0, one rank, once the inner iterations have switched themselves off) enqueues the next LM step before the host has
 * seen the current one's result: the step's last kernel takes the loop's decision on the device with the host's own
 * code (trust_region.h), and the host repeats it -- results are the same bit for bit.  Off by default: on MI355X it is
 * 1 % slower than waiting for the host.
 * out[0]: steps that ran that way; out[1]: steps enqueued ahead that the loop switched off (a stopping test, an uphill
 * step whose best point had to be saved, the iteration limit) -- their kernels return at once. */
int cvx_calib_speculation_counts(const cvx_calib *h, int64_t out[2]);

/* ------------------------------------------------------------------------ */
/* Hand-eye AX = XB refinement (HandEyeCalibration::solveNonLinear)          */
/* ------------------------------------------------------------------------ */
typedef struct cvx_handeye cvx_handeye;

/* motion pairs as unit quaternions (w,x,y,z) + translations, handeye.cpp:286-291 */
int cvx_handeye_create(int n_pairs, const double *qa, const double *ta, const double *qb,
                       const double *tb, int device, void *stream, cvx_handeye **out);
void cvx_handeye_destroy(cvx_handeye *h);
/* p = {q.w,q.x,q.y,q.z,tx,ty,tz} (handeye.cpp:279-280) */
int cvx_handeye_cost(cvx_handeye *h, const double p[7], double *cost);
/* 6-dof normal equations through QuaternionParameterization: H[36], g[6], cost */
int cvx_handeye_normal_eq(cvx_handeye *h, const double p[7], double H[36], double g[6], double *cost);
/* residuals r[n][12] and global Jacobians Jq[n][12][4], Jt[n][12][3] (any may be NULL) */
int cvx_handeye_evaluate(cvx_handeye *h, const double p[7], double *r, double *Jq, double *Jt);
/* cvx_handeye_normal_eq without a handle and without any allocation.  No counterpart in the reference: it is for
 * callers that cannot allocate -- code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  Every pointer is DEVICE memory the caller owns: the pairs in the layouts cvx_handeye_create takes
 * (d_qa[n][4], d_ta[n][3], d_qb[n][4], d_tb[n][3]), d_p7[7], and d_out[43] = H[36] row-major | g[6] | cost.
 * The two kernels are enqueued on `stream` (NULL: the default stream) and the call returns without synchronising;
 * it calls no hipMalloc / hipFree or allocating routine.  The block partials live in a fixed array in the
 * library's device image (229 376 bytes on every GPU the library runs on, held until the process ends), hence the
 * limit: n_pairs > CVX_HANDEYE_STATIC_MAX_PAIRS returns CVX_ERR_UNSUPPORTED.  All calls on one GPU share that
 * array: calls that overlap on one GPU (two streams, two threads) race; calls ordered on one stream do not.
 * The first call on a device looks up the array's address there: make it outside any stream capture.  `device` is
 * made current for the launches and the calling thread's current device is restored before the call returns;
 * `stream` must belong to `device` (not checked).
 * Results are bit-identical to cvx_handeye_normal_eq on the same pairs. */
#define CVX_HANDEYE_STATIC_MAX_PAIRS 262144
int cvx_handeye_normal_eq_static(int n_pairs, const double *d_qa, const double *d_ta, const double *d_qb,
                                 const double *d_tb, const double *d_p7, double *d_out, int device, void *stream);
int cvx_handeye_set_exchange(cvx_handeye *h, cvx_allreduce_fn fn, void *ctx, int rank, int nranks,
                             int device_buffers);
/* solveNonLinear's ceres::Solve (handeye.cpp:310-322): p in-out.  Strategy 0 on one rank with at most 512 motion
 * pairs runs as ONE kernel launch (the whole trust-region loop on the device; same decisions as the host-driven loop,
 * which larger problems, view-sharded handles and strategy 1 take; CVX_HANDEYE_HOST_LOOP=1 in the environment forces
 * the host loop).  opt->inner_iterations is refused with CVX_ERR_UNSUPPORTED (one parameter block pair). */
int cvx_handeye_solve(cvx_handeye *h, double p[7], const cvx_lm_options *opt, cvx_lm_summary *summary);

/* ------------------------------------------------------------------------ */
/* Host-side conversions the adaptor needs (no device work)                  */
/* ------------------------------------------------------------------------ */
/* ceres::RotationMatrixToAngleAxis / AngleAxisToRotationMatrix on COLUMN-major 3x3
 * (calibration.cpp:328,344) */
void cvx_rotation_to_angle_axis(const double R_colmajor[9], double w[3]);
void cvx_angle_axis_to_rotation(const double w[3], double R_colmajor[9]);
/* Eigen::Quaterniond(R) on a column-major 3x3; q = (w,x,y,z) */
void cvx_rotation_to_quaternion(const double R_colmajor[9], double q[4]);
/* 4x4 column-major transforms -> (q, t) of handeye.cpp:286-291: q = Quaterniond(A.rotation()), where rotation() is the
 * orthogonal polar factor of the linear part (Eigen 3.3's computeRotationScaling); a linear part that is orthonormal
 * to rounding is converted as it is */
void cvx_affine_to_quat_trans(const double T_colmajor[16], double q[4], double t[3]);
/* X = Translation(t) * q.toRotationMatrix(), handeye.cpp:326-330 */
void cvx_quat_trans_to_affine(const double p[7], double T_colmajor[16]);

/* motion-pair construction of solveFixed (handeye.cpp:681-693) and solveMoving
 * (handeye.cpp:738-746 with sortStationMovements :615-663).  Inputs are n column-major 4x4
 * transforms; outputs up to `cap` pairs of 4x4 transforms; *n_pairs receives the count
 * (CVX_ERR_INVALID_ARG if cap is too small). */
int cvx_handeye_pairs_fixed(int n, const double *gripper_to_base, const double *target_to_sensor,
                            double *A, double *B, int cap, int *n_pairs);
int cvx_handeye_pairs_moving(int n, const double *gripper_to_base, const double *target_to_sensor,
                             double *A, double *B, int cap, int *n_pairs);

/* closed-form initial estimate of X from motion pairs A_i X = X B_i (n column-major 4x4 each), the step
 * solveFixed / solveMoving run before solveNonLinear (handeye.cpp:697-702, 750-755):
 * method 0 dual quaternions, Daniilidis (handeye.cpp:335-473; Parameters::method_ default, handeye.hpp:16),
 * method 1 Tsai (handeye.cpp:475-533; needs n > 2), method 2 Horaud-Dornaika (handeye.cpp:537-599).
 * Host only.  X: column-major 4x4. */
int cvx_handeye_linear(int method, int n, const double *A, const double *B, double X[16]);

/* ------------------------------------------------------------------------ */
/* Tool-level bundle adjustment                         */
/* ------------------------------------------------------------------------ */
/* bundleAdjustment of tools/hand_eye_registration.cpp:144-238 (= tools/camera_ba.cpp:146-240): one global pose
 * p = {q.w,q.x,q.y,q.z,t} with QuaternionParameterization (:153-158,196-199), free board points (:161-166,187),
 * constant intrinsics, the fixed rigid transform gripper_to_base[i] applied to the board point before the pose
 * (functor :48-141), ceres::HuberLoss(1.0) on every residual block (:168,184), options :201-210. */
typedef struct cvx_toolba cvx_toolba;
typedef struct {
    int n_views, n_points;
    int64_t n_obs;
    const int *view_off;      /* [n_views + 1] */
    const int *point_id;      /* [n_obs] Marker::id_ */
    const double *obs_x;      /* [n_obs] Marker::pt_ promoted to double (cv::Point2d img_, :51) */
    const double *obs_y;
    const double *trans;      /* [n_views][12]: top 3 rows of gripper_to_base[i].matrix(), row-major */
    double intrinsics[9];     /* fx,fy,cx,cy,k1,k2,k3,p1,p2 read from the PinholeCamera (:67-77) */
    double huber;             /* 1.0 in the reference; <= 0: no loss function */
    int device;
    void *stream;
} cvx_toolba_desc;
int cvx_toolba_create(const cvx_toolba_desc *desc, cvx_toolba **out);
void cvx_toolba_destroy(cvx_toolba *h);
void cvx_lm_options_toolba(cvx_lm_options *o);   /* hand_eye_registration.cpp:201-210 */
/* view-sharded runs: each rank's handle holds a range of the views (its observations and pre-transforms), pose and
 * points are shared; one all-reduce of the n_points x 55 block sums per linearisation, one double per cost */
int cvx_toolba_set_exchange(cvx_toolba *h, cvx_allreduce_fn fn, void *ctx, int rank, int nranks, int device_buffers);
/* 1/2 sum rho(|r|^2), what Problem::Evaluate returns at :222 */
int cvx_toolba_cost(cvx_toolba *h, const double p[7], const double *points, double *cost);
/* robustified normal equations over the tangent space, as arrowhead blocks: Hpp[36], gp[6], W[n_points][6][3],
 * U[n_points][3][3], gx[n_points][3]; any may be NULL */
int cvx_toolba_normal_eq(cvx_toolba *h, const double p[7], const double *points, double Hpp[36], double gp[6],
                         double *W, double *U, double *gx, double *cost);
/* one inner-iteration pass (use_inner_iterations, :205) on (p, points) in place: the blocks q, t and every point in
 * the order Ceres' automatic ordering gives them (t, q, then all points; with one point in use: the point, t, q),
 * each minimised alone by an LM with Ceres' default options; *cost (may be NULL) = 1/2 sum rho afterwards.  What
 * cvx_toolba_solve runs on every candidate while the passes pay (cvx_lm_options_toolba sets inner_iterations). */
int cvx_toolba_inner_pass(cvx_toolba *h, double p[7], double *points, double *cost);
/* ceres::Solve of :213-214; p and points in/out (the caller inverts the pose as :229-230 does) */
int cvx_toolba_solve(cvx_toolba *h, double p[7], double *points, const cvx_lm_options *opt, cvx_lm_summary *summary);

/* ------------------------------------------------------------------------ */
/* Batched homography refinement                        */
/* ------------------------------------------------------------------------ */
/* HomographyEstimator::solveLM (homography.cpp:218-236) for n_problems independent views in one launch:
 * functor LMHomographyEstimator (homography.cpp:166-207: 2 residuals and the analytic 2x8 Jacobian rows per
 * point pair), engine levmar's dlevmar_der (lm_core.c:60-423, LU solve Axb_core.c:992-1139) with the options
 * LMSolver::minimizeDer builds (lm.hpp:61).  One problem per 64-lane wave; all sums in the reference's order,
 * so each problem's result is bit-identical to the CPU reference's.
 * Point pairs are cv::Point2f, already gathered by the inlier list (homography.cpp:170), problem i owning
 * [off[i], off[i+1]); a problem with fewer than 4 pairs is left untouched and reports ret = -1
 * (lm_core.c:117-120). */
#define CVX_HOMOGRAPHY_MAX_POINTS 1024
typedef struct cvx_homography cvx_homography;
int cvx_homography_create(int n_problems, const int *off, const float *x1, const float *y1, const float *x2,
                          const float *y2, int device, void *stream, cvx_homography **out);
void cvx_homography_destroy(cvx_homography *h);
/* H: n_problems x 9 row-major, in: initial estimates with H(2,2) = 1 assumed (homography.cpp:229), out:
 * {h0..h7, 1} (:233).  opts: NULL = {1e-3, eps, eps, eps} (lm.hpp:43-46,61) else levmar's opts[0..3].
 * info: n_problems x 10 levmar info[] (or NULL); ret: dlevmar_der's return value per problem (or NULL). */
int cvx_homography_solve(cvx_homography *h, double *H, int max_iter, const double *opts, double *info, int *ret);
/* the same in three steps, so that a caller can keep the estimates resident and time the launch alone */
int cvx_homography_set(cvx_homography *h, const double *H);
int cvx_homography_run(cvx_homography *h, int max_iter, const double *opts);
int cvx_homography_get(cvx_homography *h, double *H, double *info, int *ret);
/* The steps either side of solveLM in the reference's flow (solveRansac -> inliers -> solveLM), on the same handle:
 * HomographyEstimator::solve (homography.cpp:46-93): the normalised DLT of every problem -- normalize_points
 * (:17-42) in the reference's float arithmetic and summation order, the 2N x 9 constraint matrix (:58-83), the right
 * singular vector of its smallest singular value (:86-88; Eigen's JacobiSVD is un-vendored: Householder QR to the 9 x 9
 * triangle, then a one-sided Jacobi SVD), H = T2^-1 Hn T1, H /= H(2,2) (:91-92).  The estimates stay on the device as
 * the start of cvx_homography_run; H (n_problems x 9 row-major, or NULL) receives them.  A problem with fewer than 4
 * pairs keeps its current estimate (zeros after create). */
int cvx_homography_dlt(cvx_homography *h, double *H);
/* HomographyEstimator::residuals (homography.cpp:95-110): per point pair |hn(H p1) - p2| + |hn(H^-1 p2) - p1|, res in
 * the order of the point arrays (off[n_problems] doubles).  H: n_problems x 9 row-major, or NULL = the handle's current
 * estimates (what cvx_homography_dlt / _run / _set left). */
int cvx_homography_residuals(cvx_homography *h, const double *H, double *res);
/* HomographyEstimator::solveRansac (homography.cpp:112-156): cvx::util::RANSAC::estimate (modules/util/include/cvx/util/
 * math/ransac.hpp:28-78) with the minimal model of four pairs (fit = solve on the subset, :119-129; findInliers = the
 * pairs whose residual is below max_error and their mean residual, :131-144), for every problem of the handle.  A
 * trial's outcome does not depend on the trials before it, so all max_num_trials hypotheses of all problems are fitted
 * and scored at once, and the loop of ransac.hpp:44-73 -- keep the trial with more than min_inliers inliers and the
 * smallest mean residual, cut max_trials to log(1 - outlier_probability) / log(1 - w^4) -- is then walked per problem
 * over the scores, stopping where the reference's loop stops; the returned model is solve() on ALL inliers of the
 * kept trial (:77).
 * params: NULL = RANSAC::Parameters' defaults {0.1, 0.1, 100, 5} (what solveRansac runs on: `RANSAC solver;`, :152).
 * subsets: [n_problems][max_num_trials][4] indices into each problem's own pairs = what RNG::sample(4, n, .) draws trial
 * after trial (rng.hpp:62-75; the adaptor draws them with the reference's own generator type), or NULL: drawn on the
 * device from `seed` by a counter-based generator (the reference seeds its std::mt19937_64 from std::random_device,
 * rng.cpp:6-9: it has no sequence of its own to reproduce).
 * Out, each may be NULL: H (n_problems x 9 row-major; a problem for which estimate() returns false keeps the handle's
 * current estimate), n_inliers[n_problems] (0 = estimate() false), inlier_mask[off[n_problems]] (1 = in best_subset),
 * trials_run[n_problems] (trial_count when the loop ended), subsets_used[n_problems][max_num_trials][4] (-1 for problems
 * with fewer than 4 pairs), inlier_batch = a new handle (the caller destroys it) whose problems are the inliers in index
 * order with the refit as current estimates: cvx_homography_run on it is solveLM(pts1, pts2, inliers, H) (:218-236).
 * n_problems x max_num_trials beyond CVX_RANSAC_MAX_HYPOTHESES: CVX_ERR_UNSUPPORTED (180 bytes of work space each). */
#define CVX_RANSAC_MAX_HYPOTHESES (1 << 23)
typedef struct cvx_ransac_params {
    double max_error;             /* 0.1 */
    double outlier_probability;   /* 0.1 */
    int max_num_trials;           /* 100 */
    int min_inliers;              /* 5   */
} cvx_ransac_params;
void cvx_ransac_default_params(cvx_ransac_params *p);
int cvx_homography_ransac(cvx_homography *h, const cvx_ransac_params *params, const int *subsets, unsigned long long seed,
                          double *H, int *n_inliers, unsigned char *inlier_mask, int *trials_run, int *subsets_used,
                          cvx_homography **inlier_batch);

/* ------------------------------------------------------------------------ */
/* On-disk formats either side of the path (host only)   */
/* ------------------------------------------------------------------------ */
/* CameraCalibration::Data::load (calibration.cpp:49-79): "w h", n_coords + "x y z" lines, n_views, then per
 * view image_path (one whitespace-free token), n_markers and "x y id" lines.  Values are read as float
 * (cv::Point3f / Point2f) and handed out as double in the flat layout cvx_calib_desc uses. */
typedef struct cvx_calib_data cvx_calib_data;
int cvx_calib_data_load(const char *path, cvx_calib_data **out);
void cvx_calib_data_free(cvx_calib_data *d);
int cvx_calib_data_info(const cvx_calib_data *d, int *width, int *height, int *n_coords, int *n_views,
                        int64_t *n_obs);
/* coords[n_coords*3], view_off[n_views+1], point_id/obs_x/obs_y[n_obs]; any may be NULL */
int cvx_calib_data_copy(const cvx_calib_data *d, double *coords, int *view_off, int *point_id, double *obs_x,
                        double *obs_y);
const char *cvx_calib_data_path(const cvx_calib_data *d, int view);   /* valid until cvx_calib_data_free */
/* CameraCalibration::Data::save (calibration.cpp:22-47): same text, floats at the stream's default precision */
int cvx_calib_data_save(const char *path, int width, int height, int n_coords, const double *coords,
                        int n_views, const char *const *image_paths, const int *view_off,
                        const int *point_id, const double *obs_x, const double *obs_y);

/* PinholeCamera::read / write (camera.hpp:138-180): YAML keys image_width, image_height, camera_matrix and
 * distortion_coefficients, matrices as {rows, cols, data} in row-major order (camera.hpp:19-72).
 * K = {fx, fy, cx, cy}; dist = OpenCV order {k1, k2, p1, p2, k3}. */
int cvx_camera_yaml_read(const char *path, int *width, int *height, double K[4], double dist[5]);
int cvx_camera_yaml_write(const char *path, int width, int height, const double K[4], const double dist[5]);

/* CameraRobotConfiguration::readRobotConfiguration (configuration.hpp:44-91).  Frame names are copied into
 * buffers of `cap` bytes; initial_cam is the column-major 4x4 Translation(x,y,z) * q.normalized(); when the
 * q* keys are absent, roll/pitch/yaw are accepted as Rx(roll) Ry(pitch) Rz(yaw) (the form the reference's
 * sample scripts/configuration.yaml uses). */
int cvx_robot_config_read(const char *path, char *base_frame, char *tool_frame, char *camera_frame, size_t cap,
                          double base_to_tag[3], double initial_cam[16]);
/* CameraRobotConfiguration::writeCameraTranformationToFile (configuration.hpp:31-43): ROS launch file with
 * one static_transform_publisher "x y z qx qy qz qw tool camera 100" */
int cvx_robot_config_write_launch(const char *path, const char *tool_frame, const char *camera_frame,
                                  const double X_colmajor[16]);

/* ------------------------------------------------------------------------ */
/* Synthetic workloads of SURVEY.md section 8(d) (host only, deterministic)  */
/* ------------------------------------------------------------------------ */
typedef struct {
    int n_cameras, views_per_camera, board_w, board_h;   /* 9x6 -> 54 points, 12x9 -> 108 */
    double tile;                                         /* 25 */
    uint64_t seed;
    double pixel_noise;                                  /* 0.1 */
    int ragged;                                          /* 0: every marker visible; else drop ~1/ragged of them */
} cvx_synth_calib_spec;

/* sizes: n_views = n_cameras * views_per_camera, n_points = board_w * board_h */
int64_t cvx_synth_calib_count(const cvx_synth_calib_spec *s);   /* number of observations */
/* fills the problem arrays plus ground-truth and perturbed initial parameters */
int cvx_synth_calib(const cvx_synth_calib_spec *s, int *cam_of_view, int *view_off, int *point_id,
                    double *obs_x, double *obs_y, double *intr_gt, double *poses_gt, double *points_gt,
                    double *intr_init, double *poses_init, double *points_init);

/* n_poses gripper/target pose pairs obeying AX = XB up to noise, for the fixed-camera
 * (moving = 0, solveFixed) or the moving-camera (moving = 1, solveMoving) pairing; X_gt, X_init as p[7] */
int cvx_synth_handeye(int n_poses, uint64_t seed, double rot_noise, double trans_noise, int moving,
                      double *gripper_to_base /* n*16 */, double *target_to_sensor /* n*16 */,
                      double x_gt[7], double x_init[7]);

#ifdef __cplusplus
}
#endif
#endif
