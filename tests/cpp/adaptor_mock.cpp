// Exercises include/cvx_amd_adaptor.hpp with stand-in structs that have the shape of the reference's
// boundary types (CameraCalibration::Data calibration.hpp:18-26, CalibrationPattern::Marker
// pattern.hpp:18-24, Eigen::Matrix4d / Affine3d): BASELINE config 1 through cvx_amd::refine and a
// 30-pose hand-eye problem through cvx_amd::solve_handeye, one view through cvx_amd::homography_solve_lm.  Prints one JSON line; tests/test_adaptor.py
// compares it with the oracle.  Built by cvx_amd/csrc/Makefile (target adaptor_mock).
#include <cstdio>
#include <vector>

#include "../../include/cvx_amd_adaptor.hpp"

struct Pt2f { float x, y; };
struct Pt3f { float x, y, z; };
struct Marker { Pt2f pt_; int id_; };
struct Data {
    std::vector<Pt3f> coords_;
    std::vector<std::vector<Marker>> markers_;
};
struct Mat4 {
    double m[16];   // column-major like Eigen
    double &operator()(int r, int c) { return m[4 * c + r]; }
    double operator()(int r, int c) const { return m[4 * c + r]; }
    Mat4 &matrix() { return *this; }
    const Mat4 &matrix() const { return *this; }
};
struct Mat3 {
    double m[9];    // row-major here; only operator() is used
    double &operator()(int r, int c) { return m[3 * r + c]; }
    double operator()(int r, int c) const { return m[3 * r + c]; }
};
struct Camera {
    double fx_, fy_, cx_, cy_, d[5];
    double fx() const { return fx_; } double fy() const { return fy_; }
    double cx() const { return cx_; } double cy() const { return cy_; }
    double dist(int i) const { return d[i]; }
    void set_dist(int i, double v) { d[i] = v; }
    void set_pinhole(double a, double b, double c, double e) { fx_ = a; fy_ = b; cx_ = c; cy_ = e; }
};

int main() {
    cvx_synth_calib_spec s = {1, 20, 9, 6, 25.0, 20260001ull, 0.1, 0};
    const int V = 20, P = 54;
    const long long N = cvx_synth_calib_count(&s);
    std::vector<int> cov(V), off(V + 1), pid(N);
    std::vector<double> ox(N), oy(N), ig(9), pg(6 * V), xg(3 * P), ii(9), pi(6 * V), xi(3 * P);
    if (cvx_synth_calib(&s, cov.data(), off.data(), pid.data(), ox.data(), oy.data(), ig.data(), pg.data(), xg.data(),
                        ii.data(), pi.data(), xi.data()))
        return 2;
    Data data;
    for (int p = 0; p < P; ++p) data.coords_.push_back({(float)xi[3 * p], (float)xi[3 * p + 1], (float)xi[3 * p + 2]});
    data.markers_.resize(V);
    for (int v = 0; v < V; ++v)
        for (int o = off[v]; o < off[v + 1]; ++o) data.markers_[v].push_back({{(float)ox[o], (float)oy[o]}, pid[o]});
    Camera cam = {ii[0], ii[1], ii[2], ii[3], {ii[4], ii[5], ii[7], ii[8], ii[6]}};
    std::vector<Mat4> ext(V);
    for (int v = 0; v < V; ++v) {
        double R[9];
        cvx_angle_axis_to_rotation(&pi[6 * v], R);
        Mat4 &M = ext[v];
        for (int i = 0; i < 16; ++i) M.m[i] = 0;
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) M(r, c) = R[3 * c + r];
        for (int r = 0; r < 3; ++r) M(r, 3) = pi[6 * v + 3 + r];
        M(3, 3) = 1;
    }
    cvx_amd::RefineReport rep;
    double rms;
    try {
        rms = cvx_amd::refine(data, cam, ext, true, 0, nullptr, &rep);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 3;
    }
    std::printf("{\"rms\": %.17g, \"iterations\": %d, \"final_cost\": %.17g, \"intr\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g], ",
                rms, rep.summary.iterations, rep.summary.final_cost, cam.fx_, cam.fy_, cam.cx_, cam.cy_, cam.d[0], cam.d[1],
                cam.d[4], cam.d[2], cam.d[3]);
    std::printf("\"ext0\": [");
    for (int i = 0; i < 16; ++i) std::printf("%s%.17g", i ? ", " : "", ext[0].m[i]);
    std::printf("], ");

    const int n = 30;
    std::vector<double> G(16 * n), T(16 * n);
    double xgt[7], x0[7];
    if (cvx_synth_handeye(n, 20260002ull, 1e-3, 1e-4, 0, G.data(), T.data(), xgt, x0)) return 4;
    std::vector<Mat4> Gv(n), Tv(n);
    for (int i = 0; i < n; ++i) for (int k = 0; k < 16; ++k) { Gv[i].m[k] = G[16 * i + k]; Tv[i].m[k] = T[16 * i + k]; }
    Mat4 X;
    for (int k = 0; k < 16; ++k) X.m[k] = 0;
    bool ok;
    try {
        ok = cvx_amd::solve_handeye(Gv, Tv, X, false);      // dual-quaternion estimate + GPU refinement
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 5;
    }
    std::printf("\"handeye_ok\": %d, \"X\": [", ok ? 1 : 0);
    for (int i = 0; i < 16; ++i) std::printf("%s%.17g", i ? ", " : "", X.m[i]);
    std::printf("], ");

    // HomographyEstimator::solveLM on view 0: board (x, y) -> image points, inliers = all but every 5th point,
    // start = K [r1 r2 t] of the perturbed initial pose, normalised to H(2,2) = 1
    std::vector<Pt2f> b1, b2;
    std::vector<size_t> inl;
    for (int o = off[0]; o < off[1]; ++o) {
        b1.push_back({(float)xi[3 * pid[o]], (float)xi[3 * pid[o] + 1]});
        b2.push_back({(float)ox[o], (float)oy[o]});
        if ((o - off[0]) % 5) inl.push_back((size_t)(o - off[0]));
    }
    double R0[9];
    cvx_angle_axis_to_rotation(&pi[0], R0);
    Mat3 H;
    const double K[9] = {ii[0], 0, ii[2], 0, ii[1], ii[3], 0, 0, 1};
    const double M[9] = {R0[0], R0[3], pi[3], R0[1], R0[4], pi[4], R0[2], R0[5], pi[5]};   // [r1 r2 t] row-major
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) H(r, c) = K[3 * r] * M[c] + K[3 * r + 1] * M[3 + c] + K[3 * r + 2] * M[6 + c];
    const double h22 = H(2, 2);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H(r, c) /= h22;
    std::printf("\"hom_start\": [");
    for (int i = 0; i < 9; ++i) std::printf("%s%.17g", i ? ", " : "", H.m[i]);
    try {
        ok = cvx_amd::homography_solve_lm(b1, b2, inl, H);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 6;
    }
    std::printf("], \"hom_ok\": %d, \"hom\": [", ok ? 1 : 0);
    for (int i = 0; i < 9; ++i) std::printf("%s%.17g", i ? ", " : "", H.m[i]);
    std::printf("], ");
    // HomographyEstimator::solve (the DLT) and ::residuals on the same view, all pairs
    Mat3 Hd;
    std::vector<std::vector<double>> hres;
    try {
        cvx_amd::homography_solve(b1, b2, Hd);
        std::vector<Mat3> Hs(1, Hd);
        cvx_amd::homography_residuals_batch<Pt2f, Mat3>({&b1}, {&b2}, Hs, hres);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 8;
    }
    std::printf("\"hom_dlt\": [");
    for (int i = 0; i < 9; ++i) std::printf("%s%.17g", i ? ", " : "", Hd.m[i]);
    std::printf("], \"hom_dlt_res\": [");
    for (size_t i = 0; i < hres[0].size(); ++i) std::printf("%s%.17g", i ? ", " : "", hres[0][i]);
    std::printf("], ");

    // HomographyEstimator::solveRansac on the same view with every 7th pair pushed away; twice from one seed
    {
        std::vector<Pt2f> c2 = b2;
        for (size_t i = 0; i < c2.size(); i += 7) { c2[i].x += 40.f; c2[i].y -= 25.f; }
        cvx_amd::ransac_parameters rp;
        rp.max_error_ = 2.0;
        const uint64_t seed = 20260007;
        for (int rep = 0; rep < 2; ++rep) {
            Mat3 Hr;
            for (int i = 0; i < 9; ++i) Hr.m[i] = -1.0;
            std::vector<size_t> rin;
            bool rok = false;
            try {
                rok = cvx_amd::homography_solve_ransac(b1, c2, Hr, rin, rp, &seed);
            } catch (const std::exception &e) {
                std::fprintf(stderr, "%s\n", e.what());
                return 9;
            }
            std::printf("\"ransac_ok%d\": %d, \"ransac_H%d\": [", rep, rok ? 1 : 0, rep);
            for (int i = 0; i < 9; ++i) std::printf("%s%.17g", i ? ", " : "", Hr.m[i]);
            std::printf("], \"ransac_inliers%d\": [", rep);
            for (size_t i = 0; i < rin.size(); ++i) std::printf("%s%zu", i ? ", " : "", rin[i]);
            std::printf("], ");
        }
        // the batch form, going on to solveLM over the inliers (two views: this one and the clean one)
        std::vector<Mat3> Hb;
        std::vector<std::vector<size_t>> inb;
        std::vector<char> okb;
        try {
            cvx_amd::homography_solve_ransac_batch<Pt2f, Mat3>({&b1, &b1}, {&c2, &b2}, Hb, inb, okb, rp, &seed, 100);
        } catch (const std::exception &e) {
            std::fprintf(stderr, "%s\n", e.what());
            return 10;
        }
        // and with the subsets drawn on the device
        std::vector<Mat3> Hd2;
        std::vector<std::vector<size_t>> ind2;
        std::vector<char> okd2;
        try {
            cvx_amd::homography_solve_ransac_batch<Pt2f, Mat3>({&b1}, {&c2}, Hd2, ind2, okd2, rp, &seed, 0, 0, nullptr, true);
        } catch (const std::exception &e) {
            std::fprintf(stderr, "%s\n", e.what());
            return 11;
        }
        std::printf("\"ransac_dev_ok\": %d, \"ransac_dev_inliers\": [", (int)okd2[0]);
        for (size_t i = 0; i < ind2[0].size(); ++i) std::printf("%s%zu", i ? ", " : "", ind2[0][i]);
        std::printf("], ");
        std::printf("\"ransac_lm_ok\": [%d, %d], \"ransac_lm_H\": [", (int)okb[0], (int)okb[1]);
        for (int i = 0; i < 9; ++i) std::printf("%s%.17g", i ? ", " : "", Hb[0].m[i]);
        std::printf("], \"ransac_lm_inliers\": [");
        for (size_t i = 0; i < inb[0].size(); ++i) std::printf("%s%zu", i ? ", " : "", inb[0][i]);
        std::printf("], \"ransac_lm_inliers_clean\": %zu, ", inb[1].size());
    }

    // bundleAdjustment of tools/hand_eye_registration.cpp on the same detections: the "gripper" transforms are
    // chosen so that pose * trans_v equals each view's ground-truth board pose; start = a perturbed pose, inverted
    // on entry like the tool does (:431-432)
    std::vector<Mat4> g2b(V);
    const double wg[3] = {0.1, -0.2, 0.15};
    double Rg[9];
    cvx_angle_axis_to_rotation(wg, Rg);                       // column-major
    const double tgv[3] = {10.0, -20.0, 30.0};
    for (int v = 0; v < V; ++v) {
        double Rv[9];
        cvx_angle_axis_to_rotation(&pg[6 * v], Rv);
        Mat4 &M = g2b[v];
        for (int i = 0; i < 16; ++i) M.m[i] = 0;
        // trans_v = Tg^-1 * Tv:  R = Rg^T Rv, t = Rg^T (tv - tg)
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += Rg[3 * r + k] * Rv[3 * c + k];     // Rg^T(r,k) = Rg_colmajor[3 r + k]
                M(r, c) = s;
            }
            double s = 0;
            for (int k = 0; k < 3; ++k) s += Rg[3 * r + k] * (pg[6 * v + 3 + k] - tgv[k]);
            M(r, 3) = s;
        }
        M(3, 3) = 1;
    }
    Mat4 bpose;                                               // start: rotation slightly off, translation off
    {
        const double w0[3] = {0.11, -0.19, 0.16};
        double R0[9];
        cvx_angle_axis_to_rotation(w0, R0);
        for (int i = 0; i < 16; ++i) bpose.m[i] = 0;
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) bpose(r, c) = R0[3 * c + r];
        bpose(0, 3) = 11.0; bpose(1, 3) = -19.0; bpose(2, 3) = 31.0; bpose(3, 3) = 1;
    }
    std::printf("\"ba_start\": [");
    for (int i = 0; i < 16; ++i) std::printf("%s%.17g", i ? ", " : "", bpose.m[i]);
    Camera cam_gt = {ig[0], ig[1], ig[2], ig[3], {ig[4], ig[5], ig[7], ig[8], ig[6]}};
    Data data_gt = data;
    for (int p = 0; p < P; ++p) data_gt.coords_[p] = {(float)xg[3 * p], (float)xg[3 * p + 1], (float)xg[3 * p + 2]};
    std::vector<double> coords;
    cvx_lm_summary bsum;
    try {
        cvx_amd::bundle_adjustment(cam_gt, data_gt, g2b, bpose, coords, 0, nullptr, &bsum);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 7;
    }
    std::printf("], \"ba_pose_inv\": [");
    for (int i = 0; i < 16; ++i) std::printf("%s%.17g", i ? ", " : "", bpose.m[i]);
    std::printf("], \"ba_cost\": [%.17g, %.17g], \"ba_iterations\": %d, \"ba_coords0\": [%.17g, %.17g, %.17g]}\n",
                bsum.initial_cost, bsum.final_cost, bsum.iterations, coords[0], coords[1], coords[2]);
    return 0;
}
