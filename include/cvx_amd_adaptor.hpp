// cvx_amd_adaptor.hpp -- header-only C++ adaptor that keeps the signatures of
//   double cvx::camera::CameraCalibration::refine(const Data&, PinholeCamera&, std::vector<Eigen::Matrix4d>&, bool)
//       /root/reference/modules/calib/include/cvx/calib/calibration.hpp:33, impl calibration.cpp:354-464
//   bool   cvx::camera::HandEyeCalibration::solveNonLinear(A, B, X)
//       /root/reference/modules/calib/include/cvx/calib/handeye.hpp:43, impl handeye.cpp:272-333
// and runs them on libcvx_amd.so through the C ABI of include/cvx_amd.h.
//
// The bodies are templates over the reference's own types, touched only through the member
// names the reference uses (Data::coords_, Data::markers_, Marker::pt_, Marker::id_,
// PinholeCamera::fx()/getDistortion()/setMatrix()..., Matrix4d::operator()(r,c),
// Affine3d::matrix()).  OpenCV and Eigen are absent from the build image, so the camera
// accessors go through `camera_traits`, whose cv::Mat specialisation is compiled only when
// <opencv2/core.hpp> exists; tests/test_adaptor.py instantiates the templates with small
// stand-in structs of the same shape.
//
// Error behaviour mirrors the reference: refine returns the RMS-like double
// and hand-eye returns bool; a failure of the device path throws std::runtime_error carrying
// cvx_status_string + cvx_last_error -- there is no CPU fallback.
#pragma once
#include <cmath>
#include <stdexcept>
#include <string>
#include <random>
#include <vector>

#include "cvx_amd.h"

#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define CVX_AMD_HAVE_OPENCV 1
#endif
#endif

namespace cvx_amd {

inline void check(int status, const char *what) {
    if (status != CVX_OK)
        throw std::runtime_error(std::string(what) + ": " + cvx_status_string(status) + " [" + cvx_last_error() + "]");
}

// intrinsics block in the reference's order fx,fy,cx,cy,k1,k2,k3,p1,p2 (calibration.cpp:219-229)
template <class Camera>
struct camera_traits {
    // generic form: Camera exposes fx(),fy(),cx(),cy() and a 5-element distortion in OpenCV order
    // k1,k2,p1,p2,k3 through dist(i) / set_dist(i, v) / set_pinhole(fx,fy,cx,cy)
    static void get(const Camera &cam, double k[9]) {
        k[0] = cam.fx(); k[1] = cam.fy(); k[2] = cam.cx(); k[3] = cam.cy();
        k[4] = cam.dist(0); k[5] = cam.dist(1); k[7] = cam.dist(2); k[8] = cam.dist(3); k[6] = cam.dist(4);
    }
    static void set(Camera &cam, const double k[9]) {
        cam.set_pinhole(k[0], k[1], k[2], k[3]);
        cam.set_dist(0, k[4]); cam.set_dist(1, k[5]); cam.set_dist(2, k[7]); cam.set_dist(3, k[8]); cam.set_dist(4, k[6]);
    }
};

#ifdef CVX_AMD_HAVE_OPENCV
// cvx::util::PinholeCamera (modules/util/include/cvx/util/camera/camera.hpp:77-188): the same reads
// and writes as calibration.cpp:365-375 and :444-459
template <class Camera>
struct opencv_camera_traits {
    static void get(const Camera &cam, double k[9]) {
        k[0] = cam.fx(); k[1] = cam.fy(); k[2] = cam.cx(); k[3] = cam.cy();
        cv::Mat_<double> dist(cam.getDistortion());
        k[4] = dist(0, 0); k[5] = dist(0, 1); k[7] = dist(0, 2); k[8] = dist(0, 3); k[6] = dist(0, 4);
    }
    static void set(Camera &cam, const double k[9]) {
        cv::Mat_<double> P = cv::Mat_<double>::zeros(3, 3);
        P(0, 0) = k[0]; P(1, 1) = k[1]; P(2, 2) = 1.0; P(0, 2) = k[2]; P(1, 2) = k[3];
        cv::Mat_<double> dist(cam.getDistortion());
        dist(0, 0) = k[4]; dist(0, 1) = k[5]; dist(0, 2) = k[7]; dist(0, 3) = k[8]; dist(0, 4) = k[6];
        cam.setDistortion(dist);
        cam.setMatrix(P);
    }
};
#endif

struct RefineReport {
    cvx_lm_summary summary;
    double rms = 0.0;
};

// CameraCalibration::refine.  `Traits` reads/writes the camera; Mat4 is indexed (row, col).
// Like the reference, the refined board points are NOT written back (calibration.cpp:377-381
// copies coords_ into a local vector) and views without markers keep their extrinsics.
template <class Data, class Camera, class Mat4, class Traits = camera_traits<Camera>>
double refine(const Data &data, Camera &cam, std::vector<Mat4> &extrinsics, bool intrinsics = true,
              int device = 0, void *stream = nullptr, RefineReport *report = nullptr) {
    const int V = (int)data.markers_.size(), P = (int)data.coords_.size();
    if ((int)extrinsics.size() < V) throw std::runtime_error("refine: fewer extrinsics than views");
    std::vector<int> cam_of_view(V, 0), view_off(V + 1, 0), point_id;
    std::vector<float> ox, oy;                  // cv::Point2f as it is: promoted to double (calibration.cpp:393) on the device
    size_t n_obs = 0;
    for (int i = 0; i < V; ++i) n_obs += data.markers_[i].size();
    point_id.reserve(n_obs); ox.reserve(n_obs); oy.reserve(n_obs);
    for (int i = 0; i < V; ++i) {
        for (const auto &m : data.markers_[i]) {
            point_id.push_back(m.id_);
            ox.push_back((float)m.pt_.x);
            oy.push_back((float)m.pt_.y);
        }
        view_off[i + 1] = (int)point_id.size();
    }
    double k[9];
    Traits::get(cam, k);
    std::vector<double> poses(6 * (size_t)V), points(3 * (size_t)P);
    for (int i = 0; i < V; ++i) {       // packCamerasRotationAndTranslation, calibration.cpp:318-332
        double R[9];
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 3; ++r) R[3 * c + r] = extrinsics[i](r, c);
        cvx_rotation_to_angle_axis(R, &poses[6 * i]);
        for (int r = 0; r < 3; ++r) poses[6 * i + 3 + r] = extrinsics[i](r, 3);
    }
    for (int p = 0; p < P; ++p) {
        points[3 * p] = (double)data.coords_[p].x; points[3 * p + 1] = (double)data.coords_[p].y;
        points[3 * p + 2] = (double)data.coords_[p].z;
    }
    cvx_calib_desc d;
    d.n_cameras = 1; d.n_views = V; d.n_points = P; d.n_obs = (int64_t)point_id.size();
    d.cam_of_view = cam_of_view.data(); d.view_off = view_off.data(); d.point_id = point_id.data();
    d.obs_x = nullptr; d.obs_y = nullptr; d.obs_xf = ox.data(); d.obs_yf = oy.data();
    d.fix_k3 = 1; d.fix_intrinsics = intrinsics ? 0 : 1; d.fix_points = 0;
    d.device = device; d.stream = stream;
    d.prepare_inner_iterations = 1;                 // the solve below runs with use_inner_iterations (calibration.cpp:424)
    cvx_calib *h = nullptr;
    check(cvx_calib_create(&d, &h), "cvx_calib_create");
    struct Guard { cvx_calib *h; ~Guard() { cvx_calib_destroy(h); } } guard{h};
    check(cvx_calib_set_params(h, k, poses.data(), points.data()), "cvx_calib_set_params");
    cvx_lm_options opt;
    cvx_lm_options_refine(&opt);
    RefineReport local;
    RefineReport &rep = report ? *report : local;
    check(cvx_calib_solve(h, &opt, &rep.summary), "cvx_calib_solve");
    check(cvx_calib_get_params(h, k, poses.data(), points.data()), "cvx_calib_get_params");
    for (int i = 0; i < V; ++i) {       // unpackCamerasRotationAndTranslation, calibration.cpp:335-350
        double R[9];
        cvx_angle_axis_to_rotation(&poses[6 * i], R);
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 3; ++r) extrinsics[i](r, c) = R[3 * c + r];
        for (int r = 0; r < 3; ++r) extrinsics[i](r, 3) = poses[6 * i + 3 + r];
    }
    if (intrinsics) Traits::set(cam, k);
    rep.rms = cvx_calib_refine_return(rep.summary.final_cost, d.n_obs);   // calibration.cpp:463
    return rep.rms;
}

// HandEyeCalibration::solveNonLinear(A, B, X): Affine is indexed through .matrix()(r, c)
template <class Affine>
bool solve_nonlinear(const std::vector<Affine> &A, const std::vector<Affine> &B, Affine &X, int device = 0,
                     void *stream = nullptr, cvx_lm_summary *summary = nullptr) {
    const int n = (int)A.size();
    if ((int)B.size() != n) return false;
    std::vector<double> qa(4 * (size_t)n), ta(3 * (size_t)n), qb(4 * (size_t)n), tb(3 * (size_t)n);
    auto flat = [](const Affine &T, double M[16]) {
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) M[4 * c + r] = T.matrix()(r, c);
    };
    double M[16];
    for (int i = 0; i < n; ++i) {
        flat(A[i], M); cvx_affine_to_quat_trans(M, &qa[4 * i], &ta[3 * i]);
        flat(B[i], M); cvx_affine_to_quat_trans(M, &qb[4 * i], &tb[3 * i]);
    }
    double p[7], q[4], t[3];
    flat(X, M);
    cvx_affine_to_quat_trans(M, q, t);
    p[0] = q[0]; p[1] = q[1]; p[2] = q[2]; p[3] = q[3]; p[4] = t[0]; p[5] = t[1]; p[6] = t[2];
    cvx_handeye *h = nullptr;
    check(cvx_handeye_create(n, qa.data(), ta.data(), qb.data(), tb.data(), device, stream, &h), "cvx_handeye_create");
    struct Guard { cvx_handeye *h; ~Guard() { cvx_handeye_destroy(h); } } guard{h};
    cvx_lm_options opt;
    cvx_lm_options_handeye(&opt);
    cvx_lm_summary local;
    check(cvx_handeye_solve(h, p, &opt, summary ? summary : &local), "cvx_handeye_solve");
    cvx_quat_trans_to_affine(p, M);     // X = Translation(t) * q.toRotationMatrix(), handeye.cpp:326-330
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) X.matrix()(r, c) = M[4 * c + r];
    return true;
}

// HandEyeCalibration::solveFixed / solveMoving (handeye.cpp:665-764) in full: motion pairs exactly as the
// reference builds them, the closed-form initial estimate selected by `method` (HandEyeCalibration::Method:
// 0 DualQuaternion -- the default, handeye.hpp:16 --, 1 Tsai, 2 Horaud; handeye.cpp:335-599) and, when
// `refine` (Parameters::refine_, default true), solveNonLinear on the GPU.  X is output only.
template <class Affine>
bool solve_handeye(const std::vector<Affine> &gripper_to_base, const std::vector<Affine> &target_to_sensor,
                   Affine &X, bool moving, int method = 0, bool refine = true, int device = 0,
                   void *stream = nullptr) {
    const int n = (int)gripper_to_base.size();
    if ((int)target_to_sensor.size() != n || n < 2) return false;
    std::vector<double> G(16 * (size_t)n), T(16 * (size_t)n);
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) {
                G[16 * i + 4 * c + r] = gripper_to_base[i].matrix()(r, c);
                T[16 * i + 4 * c + r] = target_to_sensor[i].matrix()(r, c);
            }
    auto pairs = moving ? cvx_handeye_pairs_moving : cvx_handeye_pairs_fixed;
    int cnt = 0;
    check(pairs(n, G.data(), T.data(), nullptr, nullptr, 0, &cnt), "pair count");
    if (cnt < 1) return false;
    std::vector<double> A(16 * (size_t)cnt), B(A.size());
    check(pairs(n, G.data(), T.data(), A.data(), B.data(), cnt, &cnt), "pairs");
    double X0[16];
    if (cvx_handeye_linear(method, cnt, A.data(), B.data(), X0) != CVX_OK) return false;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) X.matrix()(r, c) = X0[4 * c + r];
    if (!refine) return true;
    std::vector<Affine> Av(cnt, X), Bv(cnt, X);
    for (int i = 0; i < cnt; ++i)
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) {
                Av[i].matrix()(r, c) = A[16 * i + 4 * c + r];
                Bv[i].matrix()(r, c) = B[16 * i + 4 * c + r];
            }
    return solve_nonlinear(Av, Bv, X, device, stream);
}

// ------------------------------------------------------------------------------------------
// bool HomographyEstimator::solveLM(pts1, pts2, inliers, H)      homography.cpp:218-236
// for a batch of views in one launch.  Point: anything with float members x, y (cv::Point2f); Mat3: operator()(r, c)
// (Eigen::Matrix3d).  views[i] = {pts1, pts2, inliers}; H[i] in: the linear / RANSAC estimate, out: the refined one
// with H(2,2) = 1.  max_iter = HomographyEstimator::Parameters::lm_params_.max_iter_ (the reference passes its
// ftol_ too, but LMSolver::minimizeDer never reads it, lm.hpp:61).  Returns true like the reference.
// ------------------------------------------------------------------------------------------
template <class Point, class Mat3>
bool homography_solve_lm_batch(const std::vector<const std::vector<Point> *> &pts1,
                               const std::vector<const std::vector<Point> *> &pts2,
                               const std::vector<const std::vector<size_t> *> &inliers, std::vector<Mat3> &H,
                               int max_iter = 100, int device = 0, void *stream = nullptr) {
    const size_t n = H.size();
    if (pts1.size() != n || pts2.size() != n || inliers.size() != n) throw std::invalid_argument("homography batch: size mismatch");
    if (n == 0) return true;
    std::vector<int> off(n + 1, 0);
    for (size_t i = 0; i < n; ++i) off[i + 1] = off[i] + (int)inliers[i]->size();
    std::vector<float> x1(off[n]), y1(off[n]), x2(off[n]), y2(off[n]);
    std::vector<double> h(9 * n);
    for (size_t i = 0; i < n; ++i) {
        int o = off[i];
        for (size_t idx : *inliers[i]) {                       // homography.cpp:170-174
            x1[o] = (*pts1[i])[idx].x; y1[o] = (*pts1[i])[idx].y;
            x2[o] = (*pts2[i])[idx].x; y2[o] = (*pts2[i])[idx].y;
            ++o;
        }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) h[9 * i + 3 * r + c] = H[i](r, c);   // :229
    }
    cvx_homography *hh = nullptr;
    check(cvx_homography_create((int)n, off.data(), x1.data(), y1.data(), x2.data(), y2.data(), device, stream, &hh),
          "cvx_homography_create");
    std::vector<int> ret(n);
    const int st = cvx_homography_solve(hh, h.data(), max_iter, nullptr, nullptr, ret.data());
    cvx_homography_destroy(hh);
    check(st, "cvx_homography_solve");
    for (size_t i = 0; i < n; ++i) {
        if (ret[i] < 0 && inliers[i]->size() < 4) continue;    // levmar refuses 2N < 8 and leaves h alone; :233 then
        for (int r = 0; r < 3; ++r)                            // rewrites the same numbers with H(2,2) = 1
            for (int c = 0; c < 3; ++c) H[i](r, c) = h[9 * i + 3 * r + c];
    }
    for (size_t i = 0; i < n; ++i)
        if (inliers[i]->size() < 4) H[i](2, 2) = 1.0;
    return true;
}

// the reference's single-view signature
template <class Point, class Mat3>
bool homography_solve_lm(const std::vector<Point> &pts1, const std::vector<Point> &pts2,
                         const std::vector<size_t> &inliers, Mat3 &H, int max_iter = 100, int device = 0,
                         void *stream = nullptr) {
    std::vector<Mat3> Hs(1, H);
    const bool ok = homography_solve_lm_batch<Point, Mat3>({&pts1}, {&pts2}, {&inliers}, Hs, max_iter, device, stream);
    H = Hs[0];
    return ok;
}

// ------------------------------------------------------------------------------------------
// void HomographyEstimator::solve(pts1, pts2, H)                 homography.cpp:46-93   (the normalised DLT)
// void HomographyEstimator::residuals(pts1, pts2, H, res)        homography.cpp:95-110
// for a batch of views.  With refine_iterations > 0, solve_batch goes on to solveLM over ALL pairs of every view
// (inliers = everything) without the estimates leaving the device.
// ------------------------------------------------------------------------------------------
namespace detail {
template <class Point>
inline cvx_homography *homography_handle(const std::vector<const std::vector<Point> *> &pts1,
                                         const std::vector<const std::vector<Point> *> &pts2, std::vector<int> &off,
                                         int device, void *stream) {
    const size_t n = pts1.size();
    if (pts2.size() != n) throw std::invalid_argument("homography batch: size mismatch");
    off.assign(n + 1, 0);
    for (size_t i = 0; i < n; ++i) {
        if (pts1[i]->size() != pts2[i]->size()) throw std::invalid_argument("homography batch: pts1 / pts2 differ in length");   // :47
        off[i + 1] = off[i] + (int)pts1[i]->size();
    }
    std::vector<float> x1(off[n]), y1(off[n]), x2(off[n]), y2(off[n]);
    for (size_t i = 0; i < n; ++i)
        for (size_t k = 0; k < pts1[i]->size(); ++k) {
            const int o = off[i] + (int)k;
            x1[o] = (*pts1[i])[k].x; y1[o] = (*pts1[i])[k].y; x2[o] = (*pts2[i])[k].x; y2[o] = (*pts2[i])[k].y;
        }
    cvx_homography *hh = nullptr;
    check(cvx_homography_create((int)n, off.data(), x1.data(), y1.data(), x2.data(), y2.data(), device, stream, &hh),
          "cvx_homography_create");
    return hh;
}
}  // namespace detail

template <class Point, class Mat3>
void homography_solve_batch(const std::vector<const std::vector<Point> *> &pts1,
                            const std::vector<const std::vector<Point> *> &pts2, std::vector<Mat3> &H,
                            int refine_iterations = 0, int device = 0, void *stream = nullptr) {
    const size_t n = pts1.size();
    H.resize(n);
    if (n == 0) return;
    std::vector<int> off;
    cvx_homography *hh = detail::homography_handle(pts1, pts2, off, device, stream);
    std::vector<double> h(9 * n, 0.0);
    int st = cvx_homography_dlt(hh, h.data());
    if (!st && refine_iterations > 0) {
        st = cvx_homography_run(hh, refine_iterations, nullptr);
        if (!st) st = cvx_homography_get(hh, h.data(), nullptr, nullptr);
    }
    cvx_homography_destroy(hh);
    check(st, "cvx_homography_dlt");
    for (size_t i = 0; i < n; ++i)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) H[i](r, c) = h[9 * i + 3 * r + c];
}

template <class Point, class Mat3>
void homography_solve(const std::vector<Point> &pts1, const std::vector<Point> &pts2, Mat3 &H, int device = 0,
                      void *stream = nullptr) {
    std::vector<Mat3> Hs;
    homography_solve_batch<Point, Mat3>({&pts1}, {&pts2}, Hs, 0, device, stream);
    H = Hs[0];
}

// res[i] receives view i's residuals (the reference appends to `res`: one vector per view here)
template <class Point, class Mat3>
void homography_residuals_batch(const std::vector<const std::vector<Point> *> &pts1,
                                const std::vector<const std::vector<Point> *> &pts2, const std::vector<Mat3> &H,
                                std::vector<std::vector<double>> &res, int device = 0, void *stream = nullptr) {
    const size_t n = pts1.size();
    if (H.size() != n) throw std::invalid_argument("homography batch: size mismatch");
    res.assign(n, std::vector<double>());
    if (n == 0) return;
    std::vector<int> off;
    cvx_homography *hh = detail::homography_handle(pts1, pts2, off, device, stream);
    std::vector<double> h(9 * n), r(off[n] > 0 ? off[n] : 1);
    for (size_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) h[9 * i + 3 * a + c] = H[i](a, c);
    const int st = cvx_homography_residuals(hh, h.data(), r.data());
    cvx_homography_destroy(hh);
    check(st, "cvx_homography_residuals");
    for (size_t i = 0; i < n; ++i) res[i].assign(r.begin() + off[i], r.begin() + off[i + 1]);
}

// ------------------------------------------------------------------------------------------
// bool HomographyEstimator::solveRansac(pts1, pts2, H, inliers)   homography.cpp:151-156 on RANSAC::estimate, ransac.hpp:28-78
// for a batch of views.  The subsets are drawn HERE, with the reference's own generator type and calls (RNG::sample,
// rng.hpp:62-75: std::mt19937_64, std::uniform_int_distribution<>(0, max_idx), the swap), trial after trial and view after
// view, so a generator seeded like the reference's picks what the reference's picks; rng_seed = nullptr seeds it from
// std::random_device as RNG::RNG() does (rng.cpp:6-9).  Unlike the reference's loop, every view draws all
// max_num_trials subsets (the device scores them at once; the loop's early exit is taken on the scores), so with one
// generator for several views a view's draws begin where the previous view's hundred end.  refine_iterations > 0 goes on
// to solveLM over each view's inliers (test_homography.cpp:101-110) without the estimates leaving the device.
// ok[i] = what estimate() returns for view i; H[i] is written only then (resized to the batch otherwise untouched).
// ------------------------------------------------------------------------------------------
struct ransac_parameters {        // cvx::util::RANSAC::Parameters (ransac.hpp:12-23)
    double max_error_ = 0.1, outlier_probability_ = 0.1;
    size_t max_num_trials_ = 100, min_inliers_ = 5;
};

template <class Point, class Mat3>
void homography_solve_ransac_batch(const std::vector<const std::vector<Point> *> &pts1,
                                   const std::vector<const std::vector<Point> *> &pts2, std::vector<Mat3> &H,
                                   std::vector<std::vector<size_t>> &inliers, std::vector<char> &ok,
                                   const ransac_parameters &params = ransac_parameters(), const uint64_t *rng_seed = nullptr,
                                   int refine_iterations = 0, int device = 0, void *stream = nullptr,
                                   bool draw_on_device = false) {
    // draw_on_device: the subsets come from the library's own counter-based generator on the device (seeded with
    // *rng_seed, or from std::random_device) instead of std::mt19937_64 here -- for large batches, where four million
    // draws on one host core and their upload cost twenty times the device's whole run; RNG::sample's law, another stream
    const size_t n = pts1.size();
    H.resize(n); inliers.assign(n, std::vector<size_t>()); ok.assign(n, 0);
    if (n == 0) return;
    std::vector<int> off;
    cvx_homography *hh = detail::homography_handle(pts1, pts2, off, device, stream);
    const size_t T = params.max_num_trials_;
    std::vector<int> subsets(draw_on_device ? 0 : n * T * 4, 0);
    uint64_t device_seed = 0;
    if (draw_on_device) {
        if (rng_seed) device_seed = *rng_seed;
        else { std::random_device rd; device_seed = ((uint64_t)rd() << 32) ^ rd(); }
    } else {
        std::mt19937_64 generator;
        if (rng_seed) generator.seed(*rng_seed);
        else { std::random_device rd; generator.seed(rd()); }
        std::vector<int> vidx;
        for (size_t i = 0; i < n; ++i) {
            const int N = off[i + 1] - off[i];
            if (N < 4) continue;                                   // assert(N <= n), ransac.hpp:32
            vidx.resize((size_t)N);
            for (int k = 0; k < N; ++k) vidx[(size_t)k] = k;           // rng.hpp:66-67, made once: the swaps are undone below
            for (size_t t = 0; t < T; ++t) {
                int max_idx = N - 1, drawn[4];
                for (int k = 0; k < 4; ++k) {
                    std::uniform_int_distribution<> ud(0, max_idx);
                    const int index = ud(generator);
                    std::swap(vidx[(size_t)index], vidx[(size_t)max_idx]);
                    subsets[(i * T + t) * 4 + (size_t)k] = vidx[(size_t)max_idx];
                    drawn[k] = index;
                    --max_idx;
                }
                for (int k = 3; k >= 0; --k) std::swap(vidx[(size_t)drawn[k]], vidx[(size_t)(N - 1 - k)]);   // back to 0 .. N-1
            }
        }
    }
    cvx_ransac_params p;
    p.max_error = params.max_error_; p.outlier_probability = params.outlier_probability_;
    p.max_num_trials = (int)params.max_num_trials_; p.min_inliers = (int)params.min_inliers_;
    std::vector<double> h(9 * n, 0.0);
    std::vector<int> cnt(n, 0);
    std::vector<unsigned char> mask((size_t)(off[n] > 0 ? off[n] : 1), 0);
    cvx_homography *nb = nullptr;
    int st = cvx_homography_ransac(hh, &p, draw_on_device ? nullptr : subsets.data(), device_seed, h.data(), cnt.data(), mask.data(),
                                   nullptr, nullptr, refine_iterations > 0 ? &nb : nullptr);
    if (!st && nb) {
        st = cvx_homography_run(nb, refine_iterations, nullptr);
        if (!st) st = cvx_homography_get(nb, h.data(), nullptr, nullptr);
    }
    if (nb) cvx_homography_destroy(nb);
    cvx_homography_destroy(hh);
    check(st, "cvx_homography_ransac");
    for (size_t i = 0; i < n; ++i) {
        if (cnt[i] == 0) continue;                                 // best_subset.empty(): estimate() returns false, H untouched
        ok[i] = 1;
        for (int k = off[i]; k < off[i + 1]; ++k)
            if (mask[(size_t)k]) inliers[i].push_back((size_t)(k - off[i]));
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) H[i](r, c) = h[9 * i + 3 * (size_t)r + (size_t)c];
    }
}

// the reference's single-view signature (plus the parameters it leaves at their defaults, and the seed)
template <class Point, class Mat3>
bool homography_solve_ransac(const std::vector<Point> &pts1, const std::vector<Point> &pts2, Mat3 &H,
                             std::vector<size_t> &inliers, const ransac_parameters &params = ransac_parameters(),
                             const uint64_t *rng_seed = nullptr, int device = 0, void *stream = nullptr) {
    std::vector<Mat3> Hs(1, H);
    std::vector<std::vector<size_t>> in;
    std::vector<char> ok;
    homography_solve_ransac_batch<Point, Mat3>({&pts1}, {&pts2}, Hs, in, ok, params, rng_seed, 0, device, stream);
    if (!ok[0]) return false;
    H = Hs[0];
    inliers = in[0];                                               // best_subset = inliers (ransac.hpp:58)
    return true;
}

// ------------------------------------------------------------------------------------------
// void bundleAdjustment(cam, data, gripper_to_base, pose, off, coords)
//     tools/hand_eye_registration.cpp:144-238 (= tools/camera_ba.cpp:146-240)
// pose: in, the initial sensor pose as the tool passes it (:431-432, 439); out, the refined pose INVERTED as :229-230
// does.  coords: out, the refined board points (x, y, z per point, :161-166).  off: the reference only ever writes
// zeros to it (:159, 232-234), so it is not a parameter here.  Camera goes through `Traits` like refine().
// ------------------------------------------------------------------------------------------
template <class Data, class Camera, class Affine, class Traits = camera_traits<Camera>>
void bundle_adjustment(const Camera &cam, const Data &data, const std::vector<Affine> &gripper_to_base, Affine &pose,
                       std::vector<double> &coords, int device = 0, void *stream = nullptr,
                       cvx_lm_summary *summary = nullptr) {
    const int V = (int)data.markers_.size(), P = (int)data.coords_.size();
    if ((int)gripper_to_base.size() < V) throw std::invalid_argument("bundle_adjustment: one gripper pose per view needed");
    std::vector<int> view_off(V + 1, 0), point_id;
    std::vector<double> ox, oy, trans(12 * (size_t)(V > 0 ? V : 1));
    for (int v = 0; v < V; ++v) {
        for (const auto &m : data.markers_[v]) {                 // :171-191
            point_id.push_back(m.id_); ox.push_back((double)m.pt_.x); oy.push_back((double)m.pt_.y);
        }
        view_off[v + 1] = (int)point_id.size();
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) trans[12 * (size_t)v + 4 * r + c] = gripper_to_base[v].matrix()(r, c);
    }
    coords.clear();
    for (int i = 0; i < P; ++i) {
        coords.push_back(data.coords_[i].x); coords.push_back(data.coords_[i].y); coords.push_back(data.coords_[i].z);
    }
    double M[16], q[4], t[3];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) M[4 * c + r] = pose.matrix()(r, c);
    cvx_affine_to_quat_trans(M, q, t);                           // Quaterniond q(pose.rotation()), :153-158
    double p[7] = {q[0], q[1], q[2], q[3], t[0], t[1], t[2]};
    cvx_toolba_desc d;
    d.n_views = V; d.n_points = P; d.n_obs = (int64_t)point_id.size();
    d.view_off = view_off.data(); d.point_id = point_id.data(); d.obs_x = ox.data(); d.obs_y = oy.data();
    d.trans = trans.data();
    Traits::get(cam, d.intrinsics);                              // fx fy cx cy k1 k2 k3 p1 p2 from :67-77
    d.huber = 1.0;                                               // :168
    d.device = device; d.stream = stream;
    cvx_toolba *h = nullptr;
    check(cvx_toolba_create(&d, &h), "cvx_toolba_create");
    struct Guard { cvx_toolba *h; ~Guard() { cvx_toolba_destroy(h); } } guard{h};
    cvx_lm_options opt;
    cvx_lm_options_toolba(&opt);
    cvx_lm_summary local;
    check(cvx_toolba_solve(h, p, coords.data(), &opt, summary ? summary : &local), "cvx_toolba_solve");
    cvx_quat_trans_to_affine(p, M);                              // Translation(tr) * qx.toRotationMatrix(), :225-229
    // pose = pose.inverse() (:230): general affine inverse, since q is not renormalised
    const double a = M[0], b = M[4], c = M[8], e = M[1], f = M[5], g = M[9], k = M[2], l = M[6], m = M[10];
    const double det = a * (f * m - g * l) - b * (e * m - g * k) + c * (e * l - f * k);
    const double I[9] = {(f * m - g * l) / det, (c * l - b * m) / det, (b * g - c * f) / det,
                         (g * k - e * m) / det, (a * m - c * k) / det, (c * e - a * g) / det,
                         (e * l - f * k) / det, (b * k - a * l) / det, (a * f - b * e) / det};
    for (int r = 0; r < 3; ++r) {
        for (int cc = 0; cc < 3; ++cc) pose.matrix()(r, cc) = I[3 * r + cc];
        pose.matrix()(r, 3) = -(I[3 * r] * M[12] + I[3 * r + 1] * M[13] + I[3 * r + 2] * M[14]);
    }
    pose.matrix()(3, 0) = 0; pose.matrix()(3, 1) = 0; pose.matrix()(3, 2) = 0; pose.matrix()(3, 3) = 1;
}

}  // namespace cvx_amd
