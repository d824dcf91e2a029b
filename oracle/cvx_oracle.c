/*
 * cvx_oracle.c -- CPU restatement of the cvx_calibration NLLS inner loop.
 * TEST INFRASTRUCTURE ONLY (see cvx_oracle.h for scope, citations and the
 * parity-pinning status).  Plain C99, no dependencies beyond libm.
 */
#include "cvx_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================== */
/* forward-mode dual numbers (what a ceres::Jet<double,N> carries)           */
/* ======================================================================== */
/* constants of the reference's levmar build, lm.c:35-36,71 */
#define LM_EPSILON 1E-12
#define LM_ONE_THIRD 0.3333333334
#define ND 18
typedef struct { double v; double d[ND]; } dual;

static dual d_const(double c) { dual r; r.v = c; memset(r.d, 0, sizeof r.d); return r; }
static dual d_var(double c, int k) { dual r = d_const(c); r.d[k] = 1.0; return r; }
static dual d_add(dual a, dual b) { dual r; r.v = a.v + b.v; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
static dual d_sub(dual a, dual b) { dual r; r.v = a.v - b.v; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
static dual d_mul(dual a, dual b) { dual r; r.v = a.v * b.v; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
static dual d_div(dual a, dual b) {
    dual r; double inv = 1.0 / b.v; r.v = a.v * inv;
    for (int i = 0; i < ND; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
static dual d_sqrt(dual a) { dual r; r.v = sqrt(a.v); double k = 0.5 / r.v; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * k; return r; }
static dual d_sin(dual a) { dual r; r.v = sin(a.v); double k = cos(a.v); for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * k; return r; }
static dual d_cos(dual a) { dual r; r.v = cos(a.v); double k = -sin(a.v); for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * k; return r; }

/* ceres::AngleAxisRotatePoint (Ceres rotation.h; called at calibration.cpp:286) on duals */
static void d_angle_axis_rotate_point(const dual w[3], const dual pt[3], dual out[3]) {
    dual theta2 = d_add(d_add(d_mul(w[0], w[0]), d_mul(w[1], w[1])), d_mul(w[2], w[2]));
    if (theta2.v > DBL_EPSILON) {
        dual theta = d_sqrt(theta2);
        dual c = d_cos(theta), s = d_sin(theta);
        dual ti = d_div(d_const(1.0), theta);
        dual a[3] = { d_mul(w[0], ti), d_mul(w[1], ti), d_mul(w[2], ti) };
        dual wx[3] = { d_sub(d_mul(a[1], pt[2]), d_mul(a[2], pt[1])),
                       d_sub(d_mul(a[2], pt[0]), d_mul(a[0], pt[2])),
                       d_sub(d_mul(a[0], pt[1]), d_mul(a[1], pt[0])) };
        dual tmp = d_mul(d_add(d_add(d_mul(a[0], pt[0]), d_mul(a[1], pt[1])), d_mul(a[2], pt[2])),
                         d_sub(d_const(1.0), c));
        for (int i = 0; i < 3; ++i)
            out[i] = d_add(d_add(d_mul(pt[i], c), d_mul(wx[i], s)), d_mul(a[i], tmp));
    } else {
        dual wx[3] = { d_sub(d_mul(w[1], pt[2]), d_mul(w[2], pt[1])),
                       d_sub(d_mul(w[2], pt[0]), d_mul(w[0], pt[2])),
                       d_sub(d_mul(w[0], pt[1]), d_mul(w[1], pt[0])) };
        for (int i = 0; i < 3; ++i) out[i] = d_add(pt[i], wx[i]);
    }
}

/* calibration.cpp:231-257 + 266-311, evaluated on duals seeded over the 18 inputs */
void orc_reproj_autodiff(const double intr[9], const double pose[6], const double X[3],
                         double ox, double oy, double r[2],
                         double Jc[18], double Jp[12], double Jx[6]) {
    dual in[9], rt[6], pt[3];
    for (int i = 0; i < 9; ++i) in[i] = d_var(intr[i], i);
    for (int i = 0; i < 6; ++i) rt[i] = d_var(pose[i], 9 + i);
    for (int i = 0; i < 3; ++i) pt[i] = d_var(X[i], 15 + i);
    dual x[3];
    d_angle_axis_rotate_point(rt, pt, x);
    x[0] = d_add(x[0], rt[3]); x[1] = d_add(x[1], rt[4]); x[2] = d_add(x[2], rt[5]);
    dual xn = d_div(x[0], x[2]), yn = d_div(x[1], x[2]);
    dual two = d_const(2.0);
    dual r2 = d_add(d_mul(xn, xn), d_mul(yn, yn));
    dual r4 = d_mul(r2, r2);
    dual r6 = d_mul(r4, r2);
    dual rc = d_add(d_add(d_add(d_const(1.0), d_mul(in[ORC_K1], r2)), d_mul(in[ORC_K2], r4)), d_mul(in[ORC_K3], r6));
    dual xd = d_add(d_add(d_mul(xn, rc), d_mul(d_mul(d_mul(two, in[ORC_P1]), xn), yn)),
                    d_mul(in[ORC_P2], d_add(r2, d_mul(d_mul(two, xn), xn))));
    dual yd = d_add(d_add(d_mul(yn, rc), d_mul(d_mul(d_mul(two, in[ORC_P2]), xn), yn)),
                    d_mul(in[ORC_P1], d_add(r2, d_mul(d_mul(two, yn), yn))));
    dual u = d_add(d_mul(in[ORC_FX], xd), in[ORC_CX]);
    dual v = d_add(d_mul(in[ORC_FY], yd), in[ORC_CY]);
    dual e[2] = { d_sub(u, d_const(ox)), d_sub(v, d_const(oy)) };
    for (int k = 0; k < 2; ++k) {
        r[k] = e[k].v;
        if (Jc) for (int i = 0; i < 9; ++i) Jc[k * 9 + i] = e[k].d[i];
        if (Jp) for (int i = 0; i < 6; ++i) Jp[k * 6 + i] = e[k].d[9 + i];
        if (Jx) for (int i = 0; i < 3; ++i) Jx[k * 3 + i] = e[k].d[15 + i];
    }
}

/* ======================================================================== */
/* analytic reprojection residual + Jacobian                                 */
/* ======================================================================== */
/* x = AngleAxisRotatePoint(w, X) and G[i][j] = d x_i / d w_j (global derivative of the
 * Rodrigues expression, both branches), Rm = d x / d X  (row-major 3x3) */
static void rotate_point_jac(const double w[3], const double X[3], double x[3],
                             double G[9], double Rm[9]) {
    double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (theta2 > DBL_EPSILON) {
        double theta = sqrt(theta2), c = cos(theta), s = sin(theta), ti = 1.0 / theta;
        double a[3] = { w[0] * ti, w[1] * ti, w[2] * ti };
        double ax[3] = { a[1] * X[2] - a[2] * X[1], a[2] * X[0] - a[0] * X[2], a[0] * X[1] - a[1] * X[0] };
        double d = a[0] * X[0] + a[1] * X[1] + a[2] * X[2];
        double omc = 1.0 - c;
        for (int i = 0; i < 3; ++i) x[i] = X[i] * c + ax[i] * s + a[i] * (d * omc);
        /* R = c I + s [a]x + (1-c) a a^T */
        Rm[0] = c + omc * a[0] * a[0];        Rm[1] = -s * a[2] + omc * a[0] * a[1]; Rm[2] = s * a[1] + omc * a[0] * a[2];
        Rm[3] = s * a[2] + omc * a[1] * a[0]; Rm[4] = c + omc * a[1] * a[1];         Rm[5] = -s * a[0] + omc * a[1] * a[2];
        Rm[6] = -s * a[1] + omc * a[2] * a[0]; Rm[7] = s * a[0] + omc * a[2] * a[1]; Rm[8] = c + omc * a[2] * a[2];
        for (int j = 0; j < 3; ++j) {
            /* e_j x X */
            double ex[3];
            if (j == 0) { ex[0] = 0; ex[1] = -X[2]; ex[2] = X[1]; }
            else if (j == 1) { ex[0] = X[2]; ex[1] = 0; ex[2] = -X[0]; }
            else { ex[0] = -X[1]; ex[1] = X[0]; ex[2] = 0; }
            for (int i = 0; i < 3; ++i) {
                double t1 = a[j] * (-s * X[i] + c * ax[i] + s * d * a[i]);
                double t2 = (s * ti) * (ex[i] - a[j] * ax[i]);
                double ej_i = (i == j) ? 1.0 : 0.0;
                double t3 = (omc * ti) * ((X[j] - a[j] * d) * a[i] + d * (ej_i - a[j] * a[i]));
                G[i * 3 + j] = t1 + t2 + t3;
            }
        }
    } else {
        x[0] = X[0] + (w[1] * X[2] - w[2] * X[1]);
        x[1] = X[1] + (w[2] * X[0] - w[0] * X[2]);
        x[2] = X[2] + (w[0] * X[1] - w[1] * X[0]);
        Rm[0] = 1; Rm[1] = -w[2]; Rm[2] = w[1];
        Rm[3] = w[2]; Rm[4] = 1; Rm[5] = -w[0];
        Rm[6] = -w[1]; Rm[7] = w[0]; Rm[8] = 1;
        /* d(w x X)/dw_j = e_j x X */
        G[0] = 0;     G[1] = X[2];  G[2] = -X[1];
        G[3] = -X[2]; G[4] = 0;     G[5] = X[0];
        G[6] = X[1];  G[7] = -X[0]; G[8] = 0;
    }
}

void orc_reproj_analytic(const double intr[9], const double pose[6], const double X[3],
                         double ox, double oy, double r[2],
                         double Jc[18], double Jp[12], double Jx[6]) {
    double xc[3], G[9], Rm[9];
    rotate_point_jac(pose, X, xc, G, Rm);
    xc[0] += pose[3]; xc[1] += pose[4]; xc[2] += pose[5];
    const double fx = intr[ORC_FX], fy = intr[ORC_FY], cx = intr[ORC_CX], cy = intr[ORC_CY];
    const double k1 = intr[ORC_K1], k2 = intr[ORC_K2], k3 = intr[ORC_K3], p1 = intr[ORC_P1], p2 = intr[ORC_P2];
    double xn = xc[0] / xc[2], yn = xc[1] / xc[2];
    double r2 = xn * xn + yn * yn, r4 = r2 * r2, r6 = r4 * r2;
    double rc = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
    double xd = xn * rc + 2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn);
    double yd = yn * rc + 2.0 * p2 * xn * yn + p1 * (r2 + 2.0 * yn * yn);
    r[0] = fx * xd + cx - ox;
    r[1] = fy * yd + cy - oy;
    if (!Jc && !Jp && !Jx) return;
    if (Jc) {
        double *a = Jc, *b = Jc + 9;
        a[ORC_FX] = xd;  b[ORC_FX] = 0;
        a[ORC_FY] = 0;   b[ORC_FY] = yd;
        a[ORC_CX] = 1;   b[ORC_CX] = 0;
        a[ORC_CY] = 0;   b[ORC_CY] = 1;
        a[ORC_K1] = fx * xn * r2; b[ORC_K1] = fy * yn * r2;
        a[ORC_K2] = fx * xn * r4; b[ORC_K2] = fy * yn * r4;
        a[ORC_K3] = fx * xn * r6; b[ORC_K3] = fy * yn * r6;
        a[ORC_P1] = fx * 2.0 * xn * yn;           b[ORC_P1] = fy * (r2 + 2.0 * yn * yn);
        a[ORC_P2] = fx * (r2 + 2.0 * xn * xn);    b[ORC_P2] = fy * 2.0 * xn * yn;
    }
    /* d(xd,yd)/d(xn,yn) */
    double rcp = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4;
    double dxx = rc + 2.0 * xn * xn * rcp + 2.0 * p1 * yn + 6.0 * p2 * xn;
    double dxy = 2.0 * xn * yn * rcp + 2.0 * p1 * xn + 2.0 * p2 * yn;
    double dyy = rc + 2.0 * yn * yn * rcp + 2.0 * p2 * xn + 6.0 * p1 * yn;
    double iz = 1.0 / xc[2];
    double a11 = fx * dxx, a12 = fx * dxy, a21 = fy * dxy, a22 = fy * dyy;
    double D[6] = { a11 * iz, a12 * iz, -(a11 * xn + a12 * yn) * iz,
                    a21 * iz, a22 * iz, -(a21 * xn + a22 * yn) * iz };
    if (Jp) for (int k = 0; k < 2; ++k) {
        for (int j = 0; j < 3; ++j)
            Jp[k * 6 + j] = D[k * 3 + 0] * G[0 * 3 + j] + D[k * 3 + 1] * G[1 * 3 + j] + D[k * 3 + 2] * G[2 * 3 + j];
        for (int j = 0; j < 3; ++j) Jp[k * 6 + 3 + j] = D[k * 3 + j];
    }
    if (Jx) for (int k = 0; k < 2; ++k)
        for (int j = 0; j < 3; ++j)
            Jx[k * 3 + j] = D[k * 3 + 0] * Rm[0 * 3 + j] + D[k * 3 + 1] * Rm[1 * 3 + j] + D[k * 3 + 2] * Rm[2 * 3 + j];
}

/* ======================================================================== */
/* Ceres rotation helpers used by pack/unpack (calibration.cpp:318-350)      */
/* ======================================================================== */
void orc_angle_axis_to_rotation(const double w[3], double R[9]) {
#define RR(i, j) R[(j) * 3 + (i)]
    double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (theta2 > DBL_EPSILON) {
        double theta = sqrt(theta2);
        double wx = w[0] / theta, wy = w[1] / theta, wz = w[2] / theta;
        double c = cos(theta), s = sin(theta);
        RR(0, 0) = c + wx * wx * (1.0 - c);
        RR(1, 0) = wz * s + wx * wy * (1.0 - c);
        RR(2, 0) = -wy * s + wx * wz * (1.0 - c);
        RR(0, 1) = wx * wy * (1.0 - c) - wz * s;
        RR(1, 1) = c + wy * wy * (1.0 - c);
        RR(2, 1) = wx * s + wy * wz * (1.0 - c);
        RR(0, 2) = wy * s + wx * wz * (1.0 - c);
        RR(1, 2) = -wx * s + wy * wz * (1.0 - c);
        RR(2, 2) = c + wz * wz * (1.0 - c);
    } else {
        RR(0, 0) = 1; RR(1, 0) = w[2]; RR(2, 0) = -w[1];
        RR(0, 1) = -w[2]; RR(1, 1) = 1; RR(2, 1) = w[0];
        RR(0, 2) = w[1]; RR(1, 2) = -w[0]; RR(2, 2) = 1;
    }
}

void orc_rotation_to_angle_axis(const double R[9], double w[3]) {
    /* RotationMatrixToQuaternion followed by QuaternionToAngleAxis */
    double q[4];
    double trace = RR(0, 0) + RR(1, 1) + RR(2, 2);
    if (trace >= 0.0) {
        double t = sqrt(trace + 1.0);
        q[0] = 0.5 * t; t = 0.5 / t;
        q[1] = (RR(2, 1) - RR(1, 2)) * t;
        q[2] = (RR(0, 2) - RR(2, 0)) * t;
        q[3] = (RR(1, 0) - RR(0, 1)) * t;
    } else {
        int i = 0;
        if (RR(1, 1) > RR(0, 0)) i = 1;
        if (RR(2, 2) > RR(i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        double t = sqrt(RR(i, i) - RR(j, j) - RR(k, k) + 1.0);
        q[i + 1] = 0.5 * t; t = 0.5 / t;
        q[0] = (RR(k, j) - RR(j, k)) * t;
        q[j + 1] = (RR(j, i) + RR(i, j)) * t;
        q[k + 1] = (RR(k, i) + RR(i, k)) * t;
    }
    double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (s2 > 0.0) {
        double st = sqrt(s2), ct = q[0];
        double two_theta = 2.0 * ((ct < 0.0) ? atan2(-st, -ct) : atan2(st, ct));
        double k = two_theta / st;
        w[0] = q[1] * k; w[1] = q[2] * k; w[2] = q[3] * k;
    } else {
        w[0] = q[1] * 2.0; w[1] = q[2] * 2.0; w[2] = q[3] * 2.0;
    }
#undef RR
}

void orc_quaternion_plus(const double x[4], const double delta[3], double out[4]) {
    double n = sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    if (n > 0.0) {
        double sn = sin(n) / n;
        double q[4] = { cos(n), sn * delta[0], sn * delta[1], sn * delta[2] };
        /* q (x) x, Hamilton product */
        out[0] = q[0] * x[0] - q[1] * x[1] - q[2] * x[2] - q[3] * x[3];
        out[1] = q[0] * x[1] + q[1] * x[0] + q[2] * x[3] - q[3] * x[2];
        out[2] = q[0] * x[2] - q[1] * x[3] + q[2] * x[0] + q[3] * x[1];
        out[3] = q[0] * x[3] + q[1] * x[2] - q[2] * x[1] + q[3] * x[0];
    } else {
        for (int i = 0; i < 4; ++i) out[i] = x[i];
    }
}

/* ======================================================================== */
/* hand-eye PoseError2                                                       */
/* ======================================================================== */
/* Eigen QuaternionBase::toRotationMatrix, no normalisation, row-major out */
static void quat_to_R(const double q[4], double R[9]) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
/* dR/dq_k, k = w,x,y,z, row-major */
static void quat_dR(const double q[4], double dR[4][9]) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double dw[9] = { 0, -2 * z, 2 * y,   2 * z, 0, -2 * x,   -2 * y, 2 * x, 0 };
    double dx[9] = { 0, 2 * y, 2 * z,    2 * y, -4 * x, -2 * w,   2 * z, 2 * w, -4 * x };
    double dy[9] = { -4 * y, 2 * x, 2 * w,   2 * x, 0, 2 * z,   -2 * w, 2 * z, -4 * y };
    double dz[9] = { -4 * z, -2 * w, 2 * x,   2 * w, -4 * z, 2 * y,   2 * x, 2 * y, 0 };
    memcpy(dR[0], dw, sizeof dw); memcpy(dR[1], dx, sizeof dx);
    memcpy(dR[2], dy, sizeof dy); memcpy(dR[3], dz, sizeof dz);
}
static void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
        C[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
}

void orc_handeye_analytic(const double q[4], const double t[3],
                          const double qa[4], const double ta[3],
                          const double qb[4], const double tb[3],
                          double r[12], double Jq[48], double Jt[36]) {
    double Rx[9], Ra[9], Rb[9], A[9], B[9];
    quat_to_R(q, Rx); quat_to_R(qa, Ra); quat_to_R(qb, Rb);
    mat3_mul(Ra, Rx, A); mat3_mul(Rx, Rb, B);
    for (int i = 0; i < 9; ++i) r[i] = A[i] - B[i];
    for (int i = 0; i < 3; ++i) {
        double rat = Ra[i * 3] * t[0] + Ra[i * 3 + 1] * t[1] + Ra[i * 3 + 2] * t[2];
        double rxt = Rx[i * 3] * tb[0] + Rx[i * 3 + 1] * tb[1] + Rx[i * 3 + 2] * tb[2];
        r[9 + i] = rat + ta[i] - rxt - t[i];
    }
    if (Jq) {
        double dR[4][9];
        quat_dR(q, dR);
        for (int k = 0; k < 4; ++k) {
            mat3_mul(Ra, dR[k], A); mat3_mul(dR[k], Rb, B);
            for (int i = 0; i < 9; ++i) Jq[i * 4 + k] = A[i] - B[i];
            for (int i = 0; i < 3; ++i)
                Jq[(9 + i) * 4 + k] = -(dR[k][i * 3] * tb[0] + dR[k][i * 3 + 1] * tb[1] + dR[k][i * 3 + 2] * tb[2]);
        }
    }
    if (Jt) {
        memset(Jt, 0, 36 * sizeof(double));
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
            Jt[(9 + i) * 3 + j] = Ra[i * 3 + j] - (i == j ? 1.0 : 0.0);
    }
}

static void d_quat_to_R(const dual q[4], dual R[9]) {
    dual w = q[0], x = q[1], y = q[2], z = q[3], two = d_const(2.0), one = d_const(1.0);
    dual tx = d_mul(two, x), ty = d_mul(two, y), tz = d_mul(two, z);
    dual twx = d_mul(tx, w), twy = d_mul(ty, w), twz = d_mul(tz, w);
    dual txx = d_mul(tx, x), txy = d_mul(ty, x), txz = d_mul(tz, x);
    dual tyy = d_mul(ty, y), tyz = d_mul(tz, y), tzz = d_mul(tz, z);
    R[0] = d_sub(one, d_add(tyy, tzz)); R[1] = d_sub(txy, twz); R[2] = d_add(txz, twy);
    R[3] = d_add(txy, twz); R[4] = d_sub(one, d_add(txx, tzz)); R[5] = d_sub(tyz, twx);
    R[6] = d_sub(txz, twy); R[7] = d_add(tyz, twx); R[8] = d_sub(one, d_add(txx, tyy));
}

void orc_handeye_autodiff(const double q[4], const double t[3],
                          const double qa[4], const double ta[3],
                          const double qb[4], const double tb[3],
                          double r[12], double Jq[48], double Jt[36]) {
    dual dq[4], dt[3], Rx[9];
    for (int i = 0; i < 4; ++i) dq[i] = d_var(q[i], i);
    for (int i = 0; i < 3; ++i) dt[i] = d_var(t[i], 4 + i);
    d_quat_to_R(dq, Rx);
    double Ra[9], Rb[9];
    quat_to_R(qa, Ra); quat_to_R(qb, Rb);
    dual res[12];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        dual a = d_const(0.0), b = d_const(0.0);
        for (int k = 0; k < 3; ++k) {
            a = d_add(a, d_mul(d_const(Ra[i * 3 + k]), Rx[k * 3 + j]));
            b = d_add(b, d_mul(Rx[i * 3 + k], d_const(Rb[k * 3 + j])));
        }
        res[i * 3 + j] = d_sub(a, b);
    }
    for (int i = 0; i < 3; ++i) {
        dual a = d_const(ta[i]), b = d_const(0.0);
        for (int k = 0; k < 3; ++k) {
            a = d_add(a, d_mul(d_const(Ra[i * 3 + k]), dt[k]));
            b = d_add(b, d_mul(Rx[i * 3 + k], d_const(tb[k])));
        }
        res[9 + i] = d_sub(d_sub(a, b), dt[i]);
    }
    for (int i = 0; i < 12; ++i) {
        r[i] = res[i].v;
        if (Jq) for (int k = 0; k < 4; ++k) Jq[i * 4 + k] = res[i].d[k];
        if (Jt) for (int k = 0; k < 3; ++k) Jt[i * 3 + k] = res[i].d[4 + k];
    }
}

/* ======================================================================== */
/* dense SPD solve                                                           */
/* ======================================================================== */
int orc_cholesky_solve(int n, double *A, double *b) {
    /* A = L L^T in the lower triangle, row-major */
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (int k = 0; k < j; ++k) s -= A[j * n + k] * A[j * n + k];
        if (!(s > 0.0) || !isfinite(s)) return 0;
        double l = sqrt(s);
        A[j * n + j] = l;
        double inv = 1.0 / l;
        for (int i = j + 1; i < n; ++i) {
            double t = A[i * n + j];
            const double *ai = A + i * n, *aj = A + j * n;
            for (int k = 0; k < j; ++k) t -= ai[k] * aj[k];
            A[i * n + j] = t * inv;
        }
    }
    for (int i = 0; i < n; ++i) {
        double t = b[i];
        for (int k = 0; k < i; ++k) t -= A[i * n + k] * b[k];
        b[i] = t / A[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double t = b[i];
        for (int k = i + 1; k < n; ++k) t -= A[k * n + i] * b[k];
        b[i] = t / A[i * n + i];
    }
    return 1;
}


/* Dense solve the no-LAPACK reference build uses inside dlevmar_der: Crout LU with implicitly scaled
 * partial pivoting (Axb_core.c:992-1139), restated.  A (n x n row-major) and b are preserved; returns 0
 * only when a row of A is entirely zero, exactly as the reference does (a zero pivot is replaced by
 * machine epsilon instead of failing). */
int orc_lu_solve(int n, const double *A, const double *b, double *x) {
    double *lu = malloc(sizeof(double) * (size_t)n * n), *scale = malloc(sizeof(double) * n);
    int *perm = malloc(sizeof(int) * n), ok = 1;
    memcpy(lu, A, sizeof(double) * (size_t)n * n);
    memcpy(x, b, sizeof(double) * n);
    for (int r = 0; r < n && ok; ++r) {
        double big = 0.0;
        for (int c = 0; c < n; ++c) { double a = fabs(lu[(size_t)r * n + c]); if (a > big) big = a; }
        if (big == 0.0) ok = 0; else scale[r] = 1.0 / big;
    }
    for (int c = 0; c < n && ok; ++c) {
        int piv = -1;
        double best = 0.0;
        for (int r = 0; r < n; ++r) {
            /* upper part (r < c): k < r; lower part (r >= c): k < c */
            const int lim = r < c ? r : c;
            double acc = lu[(size_t)r * n + c];
            for (int k = 0; k < lim; ++k) acc -= lu[(size_t)r * n + k] * lu[(size_t)k * n + c];
            lu[(size_t)r * n + c] = acc;
            if (r >= c) { double t = scale[r] * fabs(acc); if (t >= best) { best = t; piv = r; } }
        }
        if (piv != c) {
            for (int k = 0; k < n; ++k) { double t = lu[(size_t)piv * n + k]; lu[(size_t)piv * n + k] = lu[(size_t)c * n + k]; lu[(size_t)c * n + k] = t; }
            scale[piv] = scale[c];
        }
        perm[c] = piv;
        if (lu[(size_t)c * n + c] == 0.0) lu[(size_t)c * n + c] = DBL_EPSILON;
        if (c != n - 1) {
            const double inv = 1.0 / lu[(size_t)c * n + c];
            for (int r = c + 1; r < n; ++r) lu[(size_t)r * n + c] *= inv;
        }
    }
    if (ok) {
        int first = 0;   /* 1 + index of the first nonzero of the permuted rhs (skips leading zeros) */
        for (int r = 0; r < n; ++r) {
            const int pr = perm[r];
            double acc = x[pr];
            x[pr] = x[r];
            if (first) { for (int k = first - 1; k < r; ++k) acc -= lu[(size_t)r * n + k] * x[k]; }
            else if (acc != 0.0) first = r + 1;
            x[r] = acc;
        }
        for (int r = n - 1; r >= 0; --r) {
            double acc = x[r];
            for (int k = r + 1; k < n; ++k) acc -= lu[(size_t)r * n + k] * x[k];
            x[r] = acc / lu[(size_t)r * n + r];
        }
    }
    free(lu); free(scale); free(perm);
    return ok;
}

/* ======================================================================== */
/* LM options                                                                */
/* ======================================================================== */
static void lm_defaults(orc_lm_options *o) {
    memset(o, 0, sizeof *o);
    o->strategy = 0;
    o->max_iterations = 50;             /* Ceres default */
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
    o->tau = 1e-3; o->eps1 = o->eps2 = o->eps3 = 1e-17;   /* levmar.h:90-91 */
    o->inner_iterations = 0;
    o->inner_iteration_tolerance = 1e-3;
}
void orc_lm_options_refine(orc_lm_options *o) {   /* calibration.cpp:420-428 */
    lm_defaults(o);
    o->nonmonotonic = 1;
    o->max_iterations = 100;
    o->function_tolerance = 1e-10;
    o->gradient_tolerance = 1e-10;
    o->inner_iterations = 1;            /* calibration.cpp:424 */
}
void orc_lm_options_handeye(orc_lm_options *o) {  /* handeye.cpp:310-319 */
    lm_defaults(o);
    o->max_iterations = 100;
}
void orc_lm_options_levmar(orc_lm_options *o) {   /* lm.hpp:43-50,61 */
    lm_defaults(o);
    o->strategy = 1;
    o->max_iterations = 100;
    o->tau = 1e-3;
    o->eps1 = DBL_EPSILON; o->eps2 = DBL_EPSILON; o->eps3 = DBL_EPSILON;
}

/* ======================================================================== */
/* generic LM drivers over a linearisable model                              */
/* ======================================================================== */
typedef struct {
    void *ctx;
    int nfree, nglobal;
    /* linearise at x: diag of J^T J and gradient J^T r over the tangent space; returns cost */
    double (*linearize)(void *ctx, const double *x, double *diag, double *grad);
    /* solve (J^T J + diag(D)) delta = -grad for the last linearisation */
    int (*solve)(void *ctx, const double *D, double *delta);
    void (*plus)(void *ctx, const double *x, const double *delta, double *xout);
    double (*cost)(void *ctx, const double *x);
    double (*free_norm2)(void *ctx, const double *x);   /* Nielsen's ||p||^2 */
    /* inner iterations on a candidate point, in place; returns its new cost (NULL: the model has none) */
    double (*inner)(void *ctx, double *x);
} lm_model;

static void trace_push(orc_lm_summary *s, double cost, int acc, double damp, double grad) {
    if (s->trace_len < ORC_TRACE_MAX) {
        s->trace_cost[s->trace_len] = cost;
        s->trace_accepted[s->trace_len] = acc;
        s->trace_damping[s->trace_len] = damp;
        s->trace_gradient[s->trace_len] = grad;
        s->trace_len++;
    }
}

static double vec_norm(const double *a, int n) { double s = 0; for (int i = 0; i < n; ++i) s += a[i] * a[i]; return sqrt(s); }

/* max-norm of x - Plus(x, -g): Ceres' projected gradient norm */
static double gradient_max_norm(const lm_model *m, const double *x, const double *g, double *tmpd, double *tmpx) {
    for (int i = 0; i < m->nfree; ++i) tmpd[i] = -g[i];
    m->plus(m->ctx, x, tmpd, tmpx);
    double mx = 0;
    for (int i = 0; i < m->nglobal; ++i) { double a = fabs(x[i] - tmpx[i]); if (a > mx) mx = a; }
    return mx;
}

/* Ceres-style Levenberg-Marquardt trust region (TrustRegionMinimizer +
 * LevenbergMarquardtStrategy + TrustRegionStepEvaluator), restated from the
 * published algorithm; every linear solve is exact.  With non-monotonic steps
 * the lowest-cost point visited is what the run returns (documented behaviour of
 * use_nonmonotonic_steps).  Inner iterations (options.inner_iterations, models
 * that have an `inner` callback) as TrustRegionMinimizer::DoInnerIterationsIfNeeded
 * and IsStepSuccessful use them: the candidate goes through one coordinate-descent
 * pass, the cost that pass gains is added to the model's predicted change, a
 * step whose pass ends below the current cost is accepted whatever its ratio,
 * and the passes stop for good once one gains less than
 * inner_iteration_tolerance of the candidate's cost. */
static int lm_ceres(const lm_model *m, double *x, const orc_lm_options *o, orc_lm_summary *sum) {
    const int nf = m->nfree, ng = m->nglobal;
    double *diag = malloc(sizeof(double) * nf), *grad = malloc(sizeof(double) * nf);
    double *s2 = malloc(sizeof(double) * nf), *D = malloc(sizeof(double) * nf);
    double *delta = malloc(sizeof(double) * nf), *tmpd = malloc(sizeof(double) * nf);
    double *xc = malloc(sizeof(double) * ng), *tmpx = malloc(sizeof(double) * ng);
    double *xbest = malloc(sizeof(double) * ng);
    memset(sum, 0, sizeof *sum);

    double cost = m->linearize(m->ctx, x, diag, grad);
    sum->initial_cost = cost;
    double best_cost = cost;
    memcpy(xbest, x, sizeof(double) * ng);
    for (int i = 0; i < nf; ++i) {
        if (o->jacobi_scaling) { double s = 1.0 / (1.0 + sqrt(diag[i])); s2[i] = s * s; }
        else s2[i] = 1.0;
    }
    double gmax = gradient_max_norm(m, x, grad, tmpd, tmpx);
    double radius = o->initial_radius, decrease_factor = 2.0;
    /* TrustRegionStepEvaluator state */
    double ev_min = cost, ev_cur = cost, ev_ref = cost, ev_cand = cost;
    double ev_acc_ref = 0, ev_acc_cand = 0; int ev_nonmono = 0;
    const int max_nonmono = o->nonmonotonic ? o->max_consecutive_nonmonotonic_steps : 0;
    int term = 0, invalid_run = 0;
    int inner_enabled = o->inner_iterations && m->inner != NULL;
    if (gmax <= o->gradient_tolerance) term = 1;

    int it = 0;
    while (!term) {
        if (it >= o->max_iterations) { term = 4; break; }
        ++it;
        for (int i = 0; i < nf; ++i) {
            double d = s2[i] * diag[i];
            if (d < o->min_lm_diagonal) d = o->min_lm_diagonal;
            if (d > o->max_lm_diagonal) d = o->max_lm_diagonal;
            D[i] = d / (s2[i] * radius);
        }
        int ok = m->solve(m->ctx, D, delta);
        sum->num_linear_solves++;
        double model_change = 0;
        if (ok) {
            for (int i = 0; i < nf; ++i) model_change += delta[i] * (D[i] * delta[i] - grad[i]);
            model_change *= 0.5;
        }
        if (!ok || !(model_change > 0.0)) {
            /* HandleInvalidStep + LevenbergMarquardtStrategy::StepIsInvalid */
            trace_push(sum, NAN, 0, radius, gmax);
            if (++invalid_run >= 5) { term = 6; break; }
            radius *= 0.5;
            continue;
        }
        invalid_run = 0;
        m->plus(m->ctx, x, delta, xc);
        double cand = m->cost(m->ctx, xc);
        int inner_useful = 0;
        if (inner_enabled && isfinite(cand)) {
            const double ic = m->inner(m->ctx, xc);
            sum->num_inner_iteration_steps++;
            model_change += cand - ic;
            inner_useful = ic < cost;
            inner_enabled = (1.0 - ic / cand) > o->inner_iteration_tolerance;
            cand = ic;
        }
        /* step / x norms in the global parameter space */
        double sn = 0; for (int i = 0; i < ng; ++i) { double d = x[i] - xc[i]; sn += d * d; } sn = sqrt(sn);
        double xn = vec_norm(x, ng);
        if (sn <= o->parameter_tolerance * (xn + o->parameter_tolerance)) {
            trace_push(sum, cand, 0, radius, gmax);
            term = 2; break;
        }
        if (fabs(cost - cand) <= o->function_tolerance * cost) {
            trace_push(sum, cand, 0, radius, gmax);
            term = 3; break;
        }
        /* StepQuality */
        double rel = (ev_cur - cand) / model_change;
        double hist = (ev_ref - cand) / (ev_acc_ref + model_change);
        double rho = (max_nonmono > 0 && hist > rel) ? hist : rel;
        if (inner_useful || rho > o->min_relative_decrease) {
            trace_push(sum, cand, 1, radius, gmax);
            memcpy(x, xc, sizeof(double) * ng);
            if (cand < best_cost) { best_cost = cand; memcpy(xbest, x, sizeof(double) * ng); }
            /* LevenbergMarquardtStrategy::StepAccepted */
            double t = 2.0 * rho - 1.0;
            double f = 1.0 - t * t * t; if (f < 1.0 / 3.0) f = 1.0 / 3.0;
            radius = radius / f; if (radius > o->max_radius) radius = o->max_radius;
            decrease_factor = 2.0;
            /* TrustRegionStepEvaluator::StepAccepted */
            ev_cur = cand; ev_acc_cand += model_change; ev_acc_ref += model_change;
            if (ev_cur < ev_min) { ev_min = ev_cur; ev_nonmono = 0; ev_cand = ev_cur; ev_acc_cand = 0; }
            else { ++ev_nonmono; if (ev_cur > ev_cand) { ev_cand = ev_cur; ev_acc_cand = 0; } }
            if (ev_nonmono == max_nonmono) { ev_ref = ev_cand; ev_acc_ref = ev_acc_cand; }
            sum->num_successful++;
            cost = m->linearize(m->ctx, x, diag, grad);
            gmax = gradient_max_norm(m, x, grad, tmpd, tmpx);
            if (gmax <= o->gradient_tolerance) { term = 1; break; }
        } else {
            trace_push(sum, cand, 0, radius, gmax);
            radius = radius / decrease_factor; decrease_factor *= 2.0;
            if (radius < o->min_radius) { term = 5; break; }
        }
    }
    if (best_cost < cost) { memcpy(x, xbest, sizeof(double) * ng); cost = best_cost; }
    sum->iterations = it; sum->termination = term; sum->final_cost = cost;
    free(diag); free(grad); free(s2); free(D); free(delta); free(tmpd); free(xc); free(tmpx); free(xbest);
    return term;
}

/* levmar dlevmar_der control flow, lm_core.c:163-409, over the same model
 * interface (the linear algebra is the model's; damping is mu * I). */
static int lm_nielsen(const lm_model *m, double *x, const orc_lm_options *o, orc_lm_summary *sum) {
    const int nf = m->nfree, ng = m->nglobal;
    double *diag = malloc(sizeof(double) * nf), *grad = malloc(sizeof(double) * nf);
    double *D = malloc(sizeof(double) * nf), *delta = malloc(sizeof(double) * nf);
    double *xc = malloc(sizeof(double) * ng);
    memset(sum, 0, sizeof *sum);
    double mu = 0, eps2_sq = o->eps2 * o->eps2;
    int nu = 2, stop = 0, k;
    double p_eL2 = 2.0 * m->cost(m->ctx, x);
    sum->initial_cost = 0.5 * p_eL2;
    if (!isfinite(p_eL2)) stop = 7;
    for (k = 0; k < o->max_iterations && !stop; ++k) {
        if (p_eL2 <= o->eps3) { stop = 6; break; }
        m->linearize(m->ctx, x, diag, grad);
        double ginf = 0; for (int i = 0; i < nf; ++i) { double a = fabs(grad[i]); if (a > ginf) ginf = a; }
        double p_L2 = m->free_norm2(m->ctx, x);
        if (ginf <= o->eps1) { stop = 1; break; }
        if (k == 0) {
            double mx = -DBL_MAX; for (int i = 0; i < nf; ++i) if (diag[i] > mx) mx = diag[i];
            mu = o->tau * mx;
        }
        for (;;) {
            for (int i = 0; i < nf; ++i) D[i] = mu;
            int ok = m->solve(m->ctx, D, delta);
            sum->num_linear_solves++;
            if (ok) {
                double Dp_L2 = 0; for (int i = 0; i < nf; ++i) Dp_L2 += delta[i] * delta[i];
                if (Dp_L2 <= eps2_sq * p_L2) { stop = 2; break; }
                if (Dp_L2 >= (p_L2 + o->eps2) / (LM_EPSILON * LM_EPSILON)) { stop = 4; break; }
                m->plus(m->ctx, x, delta, xc);
                double pDp_eL2 = 2.0 * m->cost(m->ctx, xc);
                if (!isfinite(pDp_eL2)) { stop = 7; break; }
                double dL = 0; for (int i = 0; i < nf; ++i) dL += delta[i] * (mu * delta[i] - grad[i]);
                double dF = p_eL2 - pDp_eL2;
                if (dL > 0.0 && dF > 0.0) {
                    trace_push(sum, 0.5 * pDp_eL2, 1, mu, ginf);
                    double t = 2.0 * dF / dL - 1.0; t = 1.0 - t * t * t;
                    mu = mu * ((t >= LM_ONE_THIRD) ? t : LM_ONE_THIRD);
                    nu = 2;
                    memcpy(x, xc, sizeof(double) * ng);
                    p_eL2 = pDp_eL2;
                    sum->num_successful++;
                    break;
                }
                trace_push(sum, 0.5 * pDp_eL2, 0, mu, ginf);
            } else {
                trace_push(sum, NAN, 0, mu, ginf);
            }
            mu *= nu;
            int nu2 = nu << 1;
            if (nu2 <= nu) { stop = 5; break; }
            nu = nu2;
        }
    }
    if (k >= o->max_iterations && !stop) stop = 3;
    sum->iterations = k; sum->termination = stop; sum->final_cost = 0.5 * p_eL2;
    free(diag); free(grad); free(D); free(delta); free(xc);
    return stop;
}

/* ======================================================================== */
/* calibration problem                                                       */
/* ======================================================================== */
static int n_cam_free(const orc_calib_problem *pb) { return pb->fix_intrinsics ? 0 : (pb->fix_k3 ? 8 : 9); }
/* free column k (0..nci-1) -> intrinsics index */
static int cam_col(const orc_calib_problem *pb, int k) { return (pb->fix_k3 && k >= ORC_K3) ? k + 1 : k; }

int orc_calib_num_shared(const orc_calib_problem *pb) {
    return pb->C * n_cam_free(pb) + (pb->fix_points ? 0 : 3 * pb->P);
}
int orc_calib_num_free(const orc_calib_problem *pb) { return orc_calib_num_shared(pb) + 6 * pb->V; }

static int orc_threads;     /* defined with the linearisation below; 1 unless bench.py's baseline raises it */

static double calib_cost_range(const orc_calib_problem *pb, const double *intr, const double *poses,
                               const double *points, int v0, int v1) {
    double s = 0;
    for (int v = v0; v < v1; ++v) {
        const double *in = intr + 9 * pb->cam_of_view[v];
        for (int o = pb->view_off[v]; o < pb->view_off[v + 1]; ++o) {
            double r[2];
            orc_reproj_analytic(in, poses + 6 * v, points + 3 * pb->point_id[o], pb->obs_x[o], pb->obs_y[o], r, 0, 0, 0);
            s += r[0] * r[0] + r[1] * r[1];
        }
    }
    return s;
}

double orc_calib_cost(const orc_calib_problem *pb, const double *intr, const double *poses,
                      const double *points) {
    const int T = orc_threads;
    if (T <= 1) return 0.5 * calib_cost_range(pb, intr, poses, points, 0, pb->V);
    double part[256], s = 0;
    const int Tn = T > 256 ? 256 : T;
#pragma omp parallel for num_threads(Tn) schedule(static, 1)
    for (int t = 0; t < Tn; ++t)
        part[t] = calib_cost_range(pb, intr, poses, points, (int)((long long)pb->V * t / Tn), (int)((long long)pb->V * (t + 1) / Tn));
    for (int t = 0; t < Tn; ++t) s += part[t];
    return 0.5 * s;
}

void orc_calib_eval(const orc_calib_problem *pb, const double *intr, const double *poses,
                    const double *points, double *r, double *Jc, double *Jp, double *Jx) {
    for (int v = 0; v < pb->V; ++v) {
        const double *in = intr + 9 * pb->cam_of_view[v];
        for (int o = pb->view_off[v]; o < pb->view_off[v + 1]; ++o) {
            double rr[2];
            orc_reproj_analytic(in, poses + 6 * v, points + 3 * pb->point_id[o], pb->obs_x[o], pb->obs_y[o], rr,
                                Jc ? Jc + 18 * (size_t)o : 0, Jp ? Jp + 12 * (size_t)o : 0, Jx ? Jx + 6 * (size_t)o : 0);
            if (r) { r[2 * (size_t)o] = rr[0]; r[2 * (size_t)o + 1] = rr[1]; }
        }
    }
}

void orc_calib_dense_jacobian(const orc_calib_problem *pb, const double *intr,
                              const double *poses, const double *points, double *J) {
    const int nci = n_cam_free(pb), m = orc_calib_num_free(pb);
    const int off_pose = pb->C * nci, off_pt = off_pose + 6 * pb->V;
    memset(J, 0, sizeof(double) * (size_t)2 * pb->N * m);
    for (int v = 0; v < pb->V; ++v) {
        int c = pb->cam_of_view[v];
        for (int o = pb->view_off[v]; o < pb->view_off[v + 1]; ++o) {
            double r[2], Jc[18], Jp[12], Jx[6];
            int pid = pb->point_id[o];
            orc_reproj_analytic(intr + 9 * c, poses + 6 * v, points + 3 * pid, pb->obs_x[o], pb->obs_y[o], r, Jc, Jp, Jx);
            for (int k = 0; k < 2; ++k) {
                double *row = J + (size_t)(2 * o + k) * m;
                for (int i = 0; i < nci; ++i) row[c * nci + i] = Jc[k * 9 + cam_col(pb, i)];
                for (int i = 0; i < 6; ++i) row[off_pose + 6 * v + i] = Jp[k * 6 + i];
                if (!pb->fix_points) for (int i = 0; i < 3; ++i) row[off_pt + 3 * pid + i] = Jx[k * 3 + i];
            }
        }
    }
}

/* compact Schur storage for one linearisation */
typedef struct {
    const orc_calib_problem *pb;
    int nci, ns, off_pt;      /* shared ordering: [cams | points] */
    double *Vv, *gv;          /* [V*36], [V*6] */
    double *Wc;               /* [V*6*nci] pose x own-camera block */
    double *Wx;               /* [N*18]   pose x point block per observation (6x3) */
    double *U, *gs;           /* [ns*ns], [ns] */
    double *S, *rhs, *T;      /* scratch: reduced system, rhs, per-view V^-1 W */
    double *Dcur;             /* damping used by the last solve */
    double cost;
} calib_lin;

static void calib_lin_alloc(calib_lin *L, const orc_calib_problem *pb) {
    L->pb = pb; L->nci = n_cam_free(pb); L->ns = orc_calib_num_shared(pb); L->off_pt = pb->C * L->nci;
    L->Vv = malloc(sizeof(double) * 36 * pb->V); L->gv = malloc(sizeof(double) * 6 * pb->V);
    L->Wc = malloc(sizeof(double) * 6 * (L->nci ? L->nci : 1) * pb->V);
    L->Wx = malloc(sizeof(double) * 18 * (size_t)pb->N);
    int ns = L->ns ? L->ns : 1;
    L->U = malloc(sizeof(double) * (size_t)ns * ns); L->gs = malloc(sizeof(double) * ns);
    L->S = malloc(sizeof(double) * (size_t)ns * ns); L->rhs = malloc(sizeof(double) * ns);
    L->T = malloc(sizeof(double) * 6 * ns);
    L->Dcur = malloc(sizeof(double) * (L->ns + 6 * pb->V));
}
static void calib_lin_free(calib_lin *L) {
    free(L->Vv); free(L->gv); free(L->Wc); free(L->Wx); free(L->U); free(L->gs);
    free(L->S); free(L->rhs); free(L->T); free(L->Dcur);
}

/* CPU baseline only (bench.py, SURVEY.md 8d "(ii) all cores"): with orc_threads > 1 the view loops below run on
 * that many OpenMP threads over contiguous view ranges, each into private U / g / S accumulators that are added
 * in thread order.  The sums are then associated differently from the one-thread path, so the CHECKER always
 * runs with orc_threads == 1 (the default), which is also what the reference does (Ceres num_threads = 1,
 * calibration.cpp:420-433 never sets it). */
static int orc_threads = 1;
void orc_set_threads(int n) { orc_threads = n < 1 ? 1 : n; }
int orc_get_threads(void) { return orc_threads; }

static double calib_linearize_range(calib_lin *L, const double *intr, const double *poses, const double *points,
                                    int v0, int v1, double *U, double *gs) {
    const orc_calib_problem *pb = L->pb;
    const int nci = L->nci, ns = L->ns, fixp = pb->fix_points;
    double cost = 0;
    for (int v = v0; v < v1; ++v) {
        int c = pb->cam_of_view[v];
        double *Vv = L->Vv + 36 * v, *gv = L->gv + 6 * v, *Wc = L->Wc + (size_t)6 * nci * v;
        memset(Vv, 0, 36 * sizeof(double)); memset(gv, 0, 6 * sizeof(double));
        memset(Wc, 0, sizeof(double) * 6 * nci);
        for (int o = pb->view_off[v]; o < pb->view_off[v + 1]; ++o) {
            double r[2], Jc9[18], Jp[12], Jx[6], Jc[18];
            int pid = pb->point_id[o];
            orc_reproj_analytic(intr + 9 * c, poses + 6 * v, points + 3 * pid, pb->obs_x[o], pb->obs_y[o], r, Jc9, Jp, Jx);
            for (int k = 0; k < 2; ++k) for (int i = 0; i < nci; ++i) Jc[k * nci + i] = Jc9[k * 9 + cam_col(pb, i)];
            cost += r[0] * r[0] + r[1] * r[1];
            double *Wx = L->Wx + 18 * (size_t)o;
            memset(Wx, 0, 18 * sizeof(double));
            for (int k = 0; k < 2; ++k) {
                const double *jp = Jp + 6 * k, *jc = Jc + nci * k, *jx = Jx + 3 * k;
                for (int a = 0; a < 6; ++a) {
                    for (int b = 0; b < 6; ++b) Vv[a * 6 + b] += jp[a] * jp[b];
                    gv[a] += jp[a] * r[k];
                    for (int b = 0; b < nci; ++b) Wc[a * nci + b] += jp[a] * jc[b];
                    if (!fixp) for (int b = 0; b < 3; ++b) Wx[a * 3 + b] += jp[a] * jx[b];
                }
                for (int a = 0; a < nci; ++a) {
                    int ia = c * nci + a;
                    for (int b = 0; b < nci; ++b) U[(size_t)ia * ns + c * nci + b] += jc[a] * jc[b];
                    gs[ia] += jc[a] * r[k];
                    if (!fixp) for (int b = 0; b < 3; ++b) {
                        int ib = L->off_pt + 3 * pid + b;
                        U[(size_t)ia * ns + ib] += jc[a] * jx[b];
                        U[(size_t)ib * ns + ia] += jc[a] * jx[b];
                    }
                }
                if (!fixp) for (int a = 0; a < 3; ++a) {
                    int ia = L->off_pt + 3 * pid + a;
                    for (int b = 0; b < 3; ++b) U[(size_t)ia * ns + L->off_pt + 3 * pid + b] += jx[a] * jx[b];
                    gs[ia] += jx[a] * r[k];
                }
            }
        }
    }
    return cost;
}

static void calib_linearize_blocks(calib_lin *L, const double *intr, const double *poses, const double *points) {
    const orc_calib_problem *pb = L->pb;
    const int ns = L->ns, T = orc_threads;
    memset(L->U, 0, sizeof(double) * (size_t)ns * ns); memset(L->gs, 0, sizeof(double) * ns);
    if (T <= 1) { L->cost = 0.5 * calib_linearize_range(L, intr, poses, points, 0, pb->V, L->U, L->gs); return; }
    const size_t per = (size_t)ns * ns + ns + 1;
    double *acc = calloc(per * T, sizeof(double));
#pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int t = 0; t < T; ++t) {
        double *a = acc + per * t;
        a[per - 1] = calib_linearize_range(L, intr, poses, points, (int)((long long)pb->V * t / T),
                                           (int)((long long)pb->V * (t + 1) / T), a, a + (size_t)ns * ns);
    }
    double cost = 0;
    for (int t = 0; t < T; ++t) {
        const double *a = acc + per * t;
        for (size_t i = 0; i < (size_t)ns * ns; ++i) L->U[i] += a[i];
        for (int i = 0; i < ns; ++i) L->gs[i] += a[(size_t)ns * ns + i];
        cost += a[per - 1];
    }
    free(acc);
    L->cost = 0.5 * cost;
}

/* 6x6 SPD inverse-apply helper: Cholesky factor in place (lower), returns 0 if not PD */
static int chol6(double A[36]) {
    for (int j = 0; j < 6; ++j) {
        double s = A[j * 6 + j];
        for (int k = 0; k < j; ++k) s -= A[j * 6 + k] * A[j * 6 + k];
        if (!(s > 0.0) || !isfinite(s)) return 0;
        double l = sqrt(s); A[j * 6 + j] = l;
        for (int i = j + 1; i < 6; ++i) {
            double t = A[i * 6 + j];
            for (int k = 0; k < j; ++k) t -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = t / l;
        }
    }
    return 1;
}
static void chol6_solve(const double L[36], double *b, int stride) {
    for (int i = 0; i < 6; ++i) {
        double t = b[i * stride];
        for (int k = 0; k < i; ++k) t -= L[i * 6 + k] * b[k * stride];
        b[i * stride] = t / L[i * 6 + i];
    }
    for (int i = 5; i >= 0; --i) {
        double t = b[i * stride];
        for (int k = i + 1; k < 6; ++k) t -= L[k * 6 + i] * b[k * stride];
        b[i * stride] = t / L[i * 6 + i];
    }
}

/* Builds S and rhs for damping D = [Ds (ns) | Dp (6V)] ordering of the FREE vector
 * [shared | poses] used internally; returns 0 if some pose block is not PD. */
static int calib_reduce_range(calib_lin *L, const double *Dp, double *S, double *gred, int v0, int v1) {
    const orc_calib_problem *pb = L->pb;
    const int nci = L->nci, ns = L->ns;
    int *cols = malloc(sizeof(int) * (nci + 3 * (size_t)pb->P + 3));
    double *Wrow = malloc(sizeof(double) * 6 * (nci + 3 * (size_t)pb->P + 3));
    double *Trow = malloc(sizeof(double) * 6 * (nci + 3 * (size_t)pb->P + 3));
    int ok = 1;
    for (int v = v0; v < v1 && ok; ++v) {
        int c = pb->cam_of_view[v];
        int nobs = pb->view_off[v + 1] - pb->view_off[v];
        int k = nci + (pb->fix_points ? 0 : 3 * nobs);
        if (k == 0) continue;
        /* gather this view's nonzero columns */
        for (int b = 0; b < nci; ++b) cols[b] = c * nci + b;
        for (int a = 0; a < 6; ++a) for (int b = 0; b < nci; ++b) Wrow[a * k + b] = L->Wc[((size_t)v * 6 + a) * nci + b];
        if (!pb->fix_points) for (int j = 0; j < nobs; ++j) {
            int o = pb->view_off[v] + j;
            for (int b = 0; b < 3; ++b) {
                cols[nci + 3 * j + b] = L->off_pt + 3 * pb->point_id[o] + b;
                for (int a = 0; a < 6; ++a) Wrow[a * k + nci + 3 * j + b] = L->Wx[18 * (size_t)o + a * 3 + b];
            }
        }
        double Lc[36];
        memcpy(Lc, L->Vv + 36 * v, sizeof Lc);
        for (int a = 0; a < 6; ++a) Lc[a * 6 + a] += Dp[6 * v + a];
        if (!chol6(Lc)) { ok = 0; break; }
        memcpy(Trow, Wrow, sizeof(double) * 6 * k);
        for (int j = 0; j < k; ++j) chol6_solve(Lc, Trow + j, k);
        double y[6]; memcpy(y, L->gv + 6 * v, sizeof y);
        chol6_solve(Lc, y, 1);
        for (int i = 0; i < k; ++i) {
            double *Srow = S + (size_t)cols[i] * ns;
            double w0 = Wrow[0 * k + i], w1 = Wrow[1 * k + i], w2 = Wrow[2 * k + i];
            double w3 = Wrow[3 * k + i], w4 = Wrow[4 * k + i], w5 = Wrow[5 * k + i];
            for (int j = 0; j < k; ++j)
                Srow[cols[j]] -= w0 * Trow[j] + w1 * Trow[k + j] + w2 * Trow[2 * k + j] +
                                 w3 * Trow[3 * k + j] + w4 * Trow[4 * k + j] + w5 * Trow[5 * k + j];
            gred[cols[i]] -= w0 * y[0] + w1 * y[1] + w2 * y[2] + w3 * y[3] + w4 * y[4] + w5 * y[5];
        }
    }
    free(cols); free(Wrow); free(Trow);
    return ok;
}

static int calib_build_reduced(calib_lin *L, const double *Ds, const double *Dp, double *S, double *gred) {
    const orc_calib_problem *pb = L->pb;
    const int ns = L->ns, T = orc_threads;
    memcpy(S, L->U, sizeof(double) * (size_t)ns * ns);
    for (int i = 0; i < ns; ++i) { S[(size_t)i * ns + i] += Ds[i]; gred[i] = L->gs[i]; }
    if (T <= 1) return calib_reduce_range(L, Dp, S, gred, 0, pb->V);
    const size_t per = (size_t)ns * ns + ns;
    double *acc = calloc(per * T, sizeof(double));
    int *okt = calloc(T, sizeof(int));
#pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int t = 0; t < T; ++t)
        okt[t] = calib_reduce_range(L, Dp, acc + per * t, acc + per * t + (size_t)ns * ns,
                                    (int)((long long)pb->V * t / T), (int)((long long)pb->V * (t + 1) / T));
    int ok = 1;
    for (int t = 0; t < T; ++t) {
        const double *a = acc + per * t;
        ok = ok && okt[t];
        for (size_t i = 0; i < (size_t)ns * ns; ++i) S[i] += a[i];
        for (int i = 0; i < ns; ++i) gred[i] += a[(size_t)ns * ns + i];
    }
    free(acc); free(okt);
    return ok;
}

/* model callbacks: global x = [intr 9C | poses 6V | points 3P]; tangent = [shared ns | poses 6V] */
typedef struct { calib_lin L; const orc_calib_problem *pb; } calib_ctx;

static double calib_m_linearize(void *vc, const double *x, double *diag, double *grad) {
    calib_ctx *c = vc; const orc_calib_problem *pb = c->pb; calib_lin *L = &c->L;
    calib_linearize_blocks(L, x, x + 9 * pb->C, x + 9 * pb->C + 6 * pb->V);
    for (int i = 0; i < L->ns; ++i) { diag[i] = L->U[(size_t)i * L->ns + i]; grad[i] = L->gs[i]; }
    for (int v = 0; v < pb->V; ++v) for (int a = 0; a < 6; ++a) {
        diag[L->ns + 6 * v + a] = L->Vv[36 * v + a * 6 + a];
        grad[L->ns + 6 * v + a] = L->gv[6 * v + a];
    }
    return L->cost;
}
static int calib_m_solve(void *vc, const double *D, double *delta) {
    calib_ctx *c = vc; const orc_calib_problem *pb = c->pb; calib_lin *L = &c->L;
    const int ns = L->ns, nci = L->nci;
    if (!calib_build_reduced(L, D, D + ns, L->S, L->rhs)) return 0;
    for (int i = 0; i < ns; ++i) L->rhs[i] = -L->rhs[i];
    if (ns > 0 && !orc_cholesky_solve(ns, L->S, L->rhs)) return 0;
    memcpy(delta, L->rhs, sizeof(double) * ns);
    /* back-substitute poses: dp = -(V + Dp)^-1 (g + W ds) */
    for (int v = 0; v < pb->V; ++v) {
        int cc = pb->cam_of_view[v];
        double b[6];
        for (int a = 0; a < 6; ++a) {
            double t = L->gv[6 * v + a];
            for (int k = 0; k < nci; ++k) t += L->Wc[((size_t)v * 6 + a) * nci + k] * delta[cc * nci + k];
            b[a] = t;
        }
        if (!pb->fix_points) for (int o = pb->view_off[v]; o < pb->view_off[v + 1]; ++o) {
            const double *Wx = L->Wx + 18 * (size_t)o;
            const double *dx = delta + L->off_pt + 3 * pb->point_id[o];
            for (int a = 0; a < 6; ++a) b[a] += Wx[a * 3] * dx[0] + Wx[a * 3 + 1] * dx[1] + Wx[a * 3 + 2] * dx[2];
        }
        double Lc[36]; memcpy(Lc, L->Vv + 36 * v, sizeof Lc);
        for (int a = 0; a < 6; ++a) Lc[a * 6 + a] += D[ns + 6 * v + a];
        if (!chol6(Lc)) return 0;
        chol6_solve(Lc, b, 1);
        for (int a = 0; a < 6; ++a) delta[ns + 6 * v + a] = -b[a];
    }
    return 1;
}
static void calib_m_plus(void *vc, const double *x, const double *delta, double *xo) {
    calib_ctx *c = vc; const orc_calib_problem *pb = c->pb; calib_lin *L = &c->L;
    const int ng = 9 * pb->C + 6 * pb->V + 3 * pb->P;
    memcpy(xo, x, sizeof(double) * ng);
    for (int cc = 0; cc < pb->C; ++cc) for (int k = 0; k < L->nci; ++k) xo[9 * cc + cam_col(pb, k)] += delta[cc * L->nci + k];
    for (int i = 0; i < 6 * pb->V; ++i) xo[9 * pb->C + i] += delta[L->ns + i];
    if (!pb->fix_points) for (int i = 0; i < 3 * pb->P; ++i) xo[9 * pb->C + 6 * pb->V + i] += delta[L->off_pt + i];
}
static double calib_m_cost(void *vc, const double *x) {
    calib_ctx *c = vc; const orc_calib_problem *pb = c->pb;
    return orc_calib_cost(pb, x, x + 9 * pb->C, x + 9 * pb->C + 6 * pb->V);
}
static double calib_m_free_norm2(void *vc, const double *x) {
    calib_ctx *c = vc; const orc_calib_problem *pb = c->pb; calib_lin *L = &c->L;
    double s = 0;
    for (int cc = 0; cc < pb->C; ++cc) for (int k = 0; k < L->nci; ++k) { double a = x[9 * cc + cam_col(pb, k)]; s += a * a; }
    for (int i = 0; i < 6 * pb->V; ++i) { double a = x[9 * pb->C + i]; s += a * a; }
    if (!pb->fix_points) for (int i = 0; i < 3 * pb->P; ++i) { double a = x[9 * pb->C + 6 * pb->V + i]; s += a * a; }
    return s;
}

void orc_calib_normal_blocks(const orc_calib_problem *pb, const double *intr, const double *poses,
                             const double *points, double *Vv, double *gv, double *U, double *gs,
                             double *W, double *cost) {
    calib_lin L; calib_lin_alloc(&L, pb);
    calib_linearize_blocks(&L, intr, poses, points);
    const int ns = L.ns, nci = L.nci;
    if (Vv) memcpy(Vv, L.Vv, sizeof(double) * 36 * pb->V);
    if (gv) memcpy(gv, L.gv, sizeof(double) * 6 * pb->V);
    if (U) memcpy(U, L.U, sizeof(double) * (size_t)ns * ns);
    if (gs) memcpy(gs, L.gs, sizeof(double) * ns);
    if (W) {
        memset(W, 0, sizeof(double) * (size_t)pb->V * 6 * ns);
        for (int v = 0; v < pb->V; ++v) {
            int c = pb->cam_of_view[v];
            for (int a = 0; a < 6; ++a) {
                double *row = W + ((size_t)v * 6 + a) * ns;
                for (int k = 0; k < nci; ++k) row[c * nci + k] = L.Wc[((size_t)v * 6 + a) * nci + k];
                if (!pb->fix_points) for (int o = pb->view_off[v]; o < pb->view_off[v + 1]; ++o)
                    for (int b = 0; b < 3; ++b) row[L.off_pt + 3 * pb->point_id[o] + b] += L.Wx[18 * (size_t)o + a * 3 + b];
            }
        }
    }
    if (cost) *cost = L.cost;
    calib_lin_free(&L);
}

void orc_calib_reduced_system(const orc_calib_problem *pb, const double *intr, const double *poses,
                              const double *points, const double *Dp, const double *Ds,
                              double *S, double *gred, double *cost) {
    calib_lin L; calib_lin_alloc(&L, pb);
    calib_linearize_blocks(&L, intr, poses, points);
    calib_build_reduced(&L, Ds, Dp, S, gred);
    if (cost) *cost = L.cost;
    calib_lin_free(&L);
}

/* ---- inner iterations (coordinate descent over independent sets of parameter blocks) ---- */
enum { IB_CAM = 0, IB_POINT = 1, IB_POSE = 2 };

typedef struct { long long key; } ib_edge;
static int ib_cmp_ll(const void *a, const void *b) {
    long long x = *(const long long *)a, y = *(const long long *)b;
    return x < y ? -1 : x > y;
}
typedef struct { int deg, rank, id; } ib_vert;
static int ib_cmp_vert(const void *a, const void *b) {
    const ib_vert *x = a, *y = b;
    if (x->deg != y->deg) return x->deg < y->deg ? -1 : 1;
    return x->rank < y->rank ? -1 : x->rank > y->rank;
}

int orc_calib_inner_ordering(const orc_calib_problem *pb, int *group) {
    const int C = pb->C, P = pb->P, V = pb->V, NB = C + P + V;
    /* vertices: free blocks that occur in a residual, ranked by first appearance in refine's loop
     * (calibration.cpp:384-399: for every view, for every marker: intrinsics, pose, point) */
    int *rank = malloc(sizeof(int) * NB), *present = calloc(NB, sizeof(int)), nr = 0;
    for (int i = 0; i < NB; ++i) rank[i] = -1;
    for (int v = 0; v < V; ++v)
        for (int o = pb->view_off[v]; o < pb->view_off[v + 1]; ++o) {
            const int ids[3] = { pb->cam_of_view[v], C + P + v, C + pb->point_id[o] };
            const int free_[3] = { !pb->fix_intrinsics, 1, !pb->fix_points };
            for (int k = 0; k < 3; ++k)
                if (free_[k] && rank[ids[k]] < 0) { rank[ids[k]] = nr++; present[ids[k]] = 1; }
        }
    /* edges: every pair of free blocks of one residual block, both directions, unique */
    size_t ne = 0, cap = 6 * (size_t)pb->N + 1;
    long long *e = malloc(sizeof(long long) * cap);
    for (int v = 0; v < V; ++v)
        for (int o = pb->view_off[v]; o < pb->view_off[v + 1]; ++o) {
            const int ids[3] = { pb->cam_of_view[v], C + P + v, C + pb->point_id[o] };
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b)
                if (a != b && present[ids[a]] && present[ids[b]]) e[ne++] = (long long)ids[a] * NB + ids[b];
        }
    qsort(e, ne, sizeof(long long), ib_cmp_ll);
    size_t nu = 0;
    for (size_t i = 0; i < ne; ++i) if (i == 0 || e[i] != e[i - 1]) e[nu++] = e[i];
    size_t *adj_off = calloc((size_t)NB + 1, sizeof(size_t));
    for (size_t i = 0; i < nu; ++i) adj_off[e[i] / NB + 1]++;
    for (int i = 0; i < NB; ++i) adj_off[i + 1] += adj_off[i];
    /* ComputeRecursiveIndependentSetOrdering: greedy independent set by ascending degree on what is left */
    int *round = malloc(sizeof(int) * NB), left = 0, rounds = 0;
    char *color = malloc(NB);
    ib_vert *q = malloc(sizeof(ib_vert) * (NB ? NB : 1));
    for (int i = 0; i < NB; ++i) { round[i] = -1; left += present[i]; }
    while (left > 0) {
        int nq = 0;
        for (int i = 0; i < NB; ++i) {
            if (!present[i]) continue;
            int d = 0;
            for (size_t k = adj_off[i]; k < adj_off[i + 1]; ++k) d += present[e[k] % NB];
            q[nq].deg = d; q[nq].rank = rank[i]; q[nq].id = i; ++nq;
            color[i] = 0;
        }
        qsort(q, nq, sizeof(ib_vert), ib_cmp_vert);
        for (int j = 0; j < nq; ++j) {
            const int i = q[j].id;
            if (color[i]) continue;
            color[i] = 2; round[i] = rounds;
            for (size_t k = adj_off[i]; k < adj_off[i + 1]; ++k) { int n = (int)(e[k] % NB); if (present[n] && !color[n]) color[n] = 1; }
        }
        for (int j = 0; j < nq; ++j) if (color[q[j].id] == 2) { present[q[j].id] = 0; --left; }
        ++rounds;
    }
    /* CoordinateDescentMinimizer::CreateOrdering reverses the sets */
    for (int i = 0; i < NB; ++i) group[i] = round[i] < 0 ? -1 : rounds - 1 - round[i];
    free(rank); free(present); free(e); free(adj_off); free(round); free(color); free(q);
    return rounds;
}

/* one parameter block as an lm_model: its own residual blocks only, everything else read from the full vectors */
typedef struct {
    const orc_calib_problem *pb;
    double *intr, *poses, *points;      /* the full point being improved */
    int type, index, nfree, nglob;
    const int *obs; int nobs;           /* IB_POINT: this point's observations */
    const int *view_of_obs;
    double H[81], g[9];
} ib_ctx;

static void ib_visit(ib_ctx *c, const double *xb, double *cost, double *H, double *g) {
    const orc_calib_problem *pb = c->pb;
    const int nf = c->nfree;
    double s = 0;
    if (H) { memset(H, 0, sizeof(double) * nf * nf); memset(g, 0, sizeof(double) * nf); }
    int v0 = 0, v1 = 0, n = 0;
    if (c->type == IB_POSE) { v0 = c->index; v1 = v0 + 1; }
    else if (c->type == IB_CAM) { v0 = 0; v1 = pb->V; }
    else n = c->nobs;
    for (int v = v0; v < v1 || n > 0; ++v) {
        int o0, o1;
        if (c->type == IB_POINT) { o0 = 0; o1 = n; n = 0; }
        else { if (c->type == IB_CAM && pb->cam_of_view[v] != c->index) continue; o0 = pb->view_off[v]; o1 = pb->view_off[v + 1]; }
        for (int oo = o0; oo < o1; ++oo) {
            const int o = c->type == IB_POINT ? c->obs[oo] : oo;
            const int vv = c->type == IB_POINT ? c->view_of_obs[o] : v;
            const double *in = c->type == IB_CAM ? xb : c->intr + 9 * pb->cam_of_view[vv];
            const double *po = c->type == IB_POSE ? xb : c->poses + 6 * vv;
            const double *X = c->type == IB_POINT ? xb : c->points + 3 * pb->point_id[o];
            double r[2], Jc[18], Jp[12], Jx[6];
            orc_reproj_analytic(in, po, X, pb->obs_x[o], pb->obs_y[o], r, H ? Jc : 0, H ? Jp : 0, H ? Jx : 0);
            s += r[0] * r[0] + r[1] * r[1];
            if (!H) continue;
            for (int k = 0; k < 2; ++k) {
                double row[9];
                if (c->type == IB_CAM) for (int i = 0; i < nf; ++i) row[i] = Jc[9 * k + cam_col(pb, i)];
                else if (c->type == IB_POSE) for (int i = 0; i < 6; ++i) row[i] = Jp[6 * k + i];
                else for (int i = 0; i < 3; ++i) row[i] = Jx[3 * k + i];
                for (int a = 0; a < nf; ++a) {
                    for (int b = 0; b < nf; ++b) H[a * nf + b] += row[a] * row[b];
                    g[a] += row[a] * r[k];
                }
            }
        }
        if (c->type == IB_POINT) break;
    }
    *cost = 0.5 * s;
}
static double ib_linearize(void *vc, const double *x, double *diag, double *grad) {
    ib_ctx *c = vc; double cost;
    ib_visit(c, x, &cost, c->H, c->g);
    for (int i = 0; i < c->nfree; ++i) { diag[i] = c->H[i * c->nfree + i]; grad[i] = c->g[i]; }
    return cost;
}
static int ib_solve(void *vc, const double *D, double *delta) {
    ib_ctx *c = vc; const int nf = c->nfree;
    double A[81];
    memcpy(A, c->H, sizeof(double) * nf * nf);
    for (int i = 0; i < nf; ++i) { A[i * nf + i] += D[i]; delta[i] = -c->g[i]; }
    return orc_cholesky_solve(nf, A, delta);
}
static void ib_plus(void *vc, const double *x, const double *delta, double *xo) {
    ib_ctx *c = vc;
    memcpy(xo, x, sizeof(double) * c->nglob);
    if (c->type == IB_CAM) for (int k = 0; k < c->nfree; ++k) xo[cam_col(c->pb, k)] += delta[k];
    else for (int k = 0; k < c->nfree; ++k) xo[k] += delta[k];
}
static double ib_cost(void *vc, const double *x) { ib_ctx *c = vc; double cost; ib_visit(c, x, &cost, 0, 0); return cost; }
static double ib_free_norm2(void *vc, const double *x) { ib_ctx *c = vc; double s = 0; for (int i = 0; i < c->nglob; ++i) s += x[i] * x[i]; return s; }

static int lm_ceres(const lm_model *m, double *x, const orc_lm_options *o, orc_lm_summary *sum);

double orc_calib_inner_pass(const orc_calib_problem *pb, double *intr, double *poses, double *points) {
    const int C = pb->C, P = pb->P, V = pb->V, NB = C + P + V;
    int *group = malloc(sizeof(int) * (NB ? NB : 1));
    const int ng = orc_calib_inner_ordering(pb, group);
    /* observations by point */
    int *pt_off = calloc((size_t)P + 1, sizeof(int)), *pt_obs = malloc(sizeof(int) * (pb->N ? pb->N : 1));
    int *view_of_obs = malloc(sizeof(int) * (pb->N ? pb->N : 1)), *fill = calloc((size_t)P + 1, sizeof(int));
    for (int o = 0; o < pb->N; ++o) pt_off[pb->point_id[o] + 1]++;
    for (int p = 0; p < P; ++p) pt_off[p + 1] += pt_off[p];
    for (int v = 0; v < V; ++v)
        for (int o = pb->view_off[v]; o < pb->view_off[v + 1]; ++o) {
            const int p = pb->point_id[o];
            pt_obs[pt_off[p] + fill[p]++] = o; view_of_obs[o] = v;
        }
    orc_lm_options io; lm_defaults(&io);        /* CoordinateDescentMinimizer::Solve: a default Minimizer::Options */
    io.max_radius = 1e32;                       /* ... and a default TrustRegionStrategy::Options (max_radius 1e32) */
    orc_lm_summary *isum = malloc(sizeof *isum);
    for (int gi = 0; gi < ng; ++gi)
        for (int b = 0; b < NB; ++b) {
            if (group[b] != gi) continue;
            ib_ctx c; memset(&c, 0, sizeof c);
            c.pb = pb; c.intr = intr; c.poses = poses; c.points = points; c.view_of_obs = view_of_obs;
            double *xb;
            if (b < C) { c.type = IB_CAM; c.index = b; c.nfree = n_cam_free(pb); c.nglob = 9; xb = intr + 9 * b; }
            else if (b < C + P) { c.type = IB_POINT; c.index = b - C; c.nfree = c.nglob = 3; xb = points + 3 * (b - C);
                                  c.obs = pt_obs + pt_off[b - C]; c.nobs = pt_off[b - C + 1] - pt_off[b - C]; }
            else { c.type = IB_POSE; c.index = b - C - P; c.nfree = c.nglob = 6; xb = poses + 6 * (b - C - P); }
            double xloc[9];
            memcpy(xloc, xb, sizeof(double) * c.nglob);
            lm_model m = { &c, c.nfree, c.nglob, ib_linearize, ib_solve, ib_plus, ib_cost, ib_free_norm2, NULL };
            lm_ceres(&m, xloc, &io, isum);
            memcpy(xb, xloc, sizeof(double) * c.nglob);
        }
    free(group); free(pt_off); free(pt_obs); free(view_of_obs); free(fill); free(isum);
    return orc_calib_cost(pb, intr, poses, points);
}

static double calib_m_inner(void *vc, double *x) {
    calib_ctx *c = vc; const orc_calib_problem *pb = c->pb;
    return orc_calib_inner_pass(pb, x, x + 9 * pb->C, x + 9 * pb->C + 6 * pb->V);
}

int orc_calib_solve(const orc_calib_problem *pb, double *intr, double *poses, double *points,
                    const orc_lm_options *opt, orc_lm_summary *sum) {
    calib_ctx c; c.pb = pb; calib_lin_alloc(&c.L, pb);
    const int ng = 9 * pb->C + 6 * pb->V + 3 * pb->P;
    double *x = malloc(sizeof(double) * ng);
    memcpy(x, intr, sizeof(double) * 9 * pb->C);
    memcpy(x + 9 * pb->C, poses, sizeof(double) * 6 * pb->V);
    memcpy(x + 9 * pb->C + 6 * pb->V, points, sizeof(double) * 3 * pb->P);
    lm_model m = { &c, c.L.ns + 6 * pb->V, ng, calib_m_linearize, calib_m_solve, calib_m_plus, calib_m_cost, calib_m_free_norm2, calib_m_inner };
    int term = (opt->strategy == 1) ? lm_nielsen(&m, x, opt, sum) : lm_ceres(&m, x, opt, sum);
    memcpy(intr, x, sizeof(double) * 9 * pb->C);
    memcpy(poses, x + 9 * pb->C, sizeof(double) * 6 * pb->V);
    memcpy(points, x + 9 * pb->C + 6 * pb->V, sizeof(double) * 3 * pb->P);
    free(x); calib_lin_free(&c.L);
    return term;
}

double orc_calib_refine_return(double total_cost, int N) { return sqrt(total_cost / (2.0 * N)); }

/* ======================================================================== */
/* hand-eye problem                                                          */
/* ======================================================================== */
typedef struct { const orc_handeye_problem *pb; double H[36], g[6]; } he_ctx;

/* QuaternionParameterization::ComputeJacobian, 4x3 row-major */
static void quat_plus_jac(const double x[4], double P[12]) {
    P[0] = -x[1]; P[1] = -x[2]; P[2] = -x[3];
    P[3] = x[0];  P[4] = x[3];  P[5] = -x[2];
    P[6] = -x[3]; P[7] = x[0];  P[8] = x[1];
    P[9] = x[2];  P[10] = -x[1]; P[11] = x[0];
}

double orc_handeye_cost(const orc_handeye_problem *pb, const double p[7]) {
    double s = 0;
    for (int i = 0; i < pb->n; ++i) {
        double r[12];
        orc_handeye_analytic(p, p + 4, pb->qa + 4 * i, pb->ta + 3 * i, pb->qb + 4 * i, pb->tb + 3 * i, r, 0, 0);
        for (int k = 0; k < 12; ++k) s += r[k] * r[k];
    }
    return 0.5 * s;
}

void orc_handeye_normal_eq(const orc_handeye_problem *pb, const double p[7],
                           double H[36], double g[6], double *cost) {
    double P[12]; quat_plus_jac(p, P);
    memset(H, 0, 36 * sizeof(double)); memset(g, 0, 6 * sizeof(double));
    double s = 0;
    for (int i = 0; i < pb->n; ++i) {
        double r[12], Jq[48], Jt[36], J[72];
        orc_handeye_analytic(p, p + 4, pb->qa + 4 * i, pb->ta + 3 * i, pb->qb + 4 * i, pb->tb + 3 * i, r, Jq, Jt);
        for (int k = 0; k < 12; ++k) {
            for (int a = 0; a < 3; ++a)
                J[k * 6 + a] = Jq[k * 4 + 0] * P[0 + a] + Jq[k * 4 + 1] * P[3 + a] + Jq[k * 4 + 2] * P[6 + a] + Jq[k * 4 + 3] * P[9 + a];
            for (int a = 0; a < 3; ++a) J[k * 6 + 3 + a] = Jt[k * 3 + a];
            s += r[k] * r[k];
            for (int a = 0; a < 6; ++a) {
                for (int b = 0; b < 6; ++b) H[a * 6 + b] += J[k * 6 + a] * J[k * 6 + b];
                g[a] += J[k * 6 + a] * r[k];
            }
        }
    }
    if (cost) *cost = 0.5 * s;
}

static double he_m_linearize(void *vc, const double *x, double *diag, double *grad) {
    he_ctx *c = vc; double cost;
    orc_handeye_normal_eq(c->pb, x, c->H, c->g, &cost);
    for (int a = 0; a < 6; ++a) { diag[a] = c->H[a * 6 + a]; grad[a] = c->g[a]; }
    return cost;
}
static int he_m_solve(void *vc, const double *D, double *delta) {
    he_ctx *c = vc; double A[36];
    memcpy(A, c->H, sizeof A);
    for (int a = 0; a < 6; ++a) { A[a * 6 + a] += D[a]; delta[a] = -c->g[a]; }
    return orc_cholesky_solve(6, A, delta);
}
static void he_m_plus(void *vc, const double *x, const double *delta, double *xo) {
    (void)vc;
    orc_quaternion_plus(x, delta, xo);
    for (int a = 0; a < 3; ++a) xo[4 + a] = x[4 + a] + delta[3 + a];
}
static double he_m_cost(void *vc, const double *x) { he_ctx *c = vc; return orc_handeye_cost(c->pb, x); }
static double he_m_free_norm2(void *vc, const double *x) { (void)vc; double s = 0; for (int i = 0; i < 7; ++i) s += x[i] * x[i]; return s; }

int orc_handeye_solve(const orc_handeye_problem *pb, double p[7],
                      const orc_lm_options *opt, orc_lm_summary *sum) {
    he_ctx c; c.pb = pb;
    lm_model m = { &c, 6, 7, he_m_linearize, he_m_solve, he_m_plus, he_m_cost, he_m_free_norm2 };
    return (opt->strategy == 1) ? lm_nielsen(&m, p, opt, sum) : lm_ceres(&m, p, opt, sum);
}

/* ======================================================================== */
/* dense levmar restatement: lm_core.c:60-423                                */
/* ======================================================================== */
/* e = x - y (x may be NULL = 0) and ||e||^2 with the summation order of LEVMAR_L2NRMXMY
 * (misc_core.c:712-798): four interleaved partial sums, blocks of 8 from the top down, then the tail */
static double lm_l2_xmy(double *e, const double *x, const double *y, int n) {
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    const int blockn = (n >> 3) << 3;
    for (int i = blockn - 1; i >= 0; --i) {
        e[i] = (x ? x[i] : 0.0) - y[i];
        s[(blockn - 1 - i) & 3] += e[i] * e[i];
    }
    for (int i = blockn; i < n; ++i) {
        e[i] = (x ? x[i] : 0.0) - y[i];
        s[(7 - (n - i)) & 3] += e[i] * e[i];
    }
    return s[0] + s[1] + s[2] + s[3];
}

int orc_lm_nielsen_der(orc_lm_func func, orc_lm_func jacf, double *p, const double *x,
                       int m, int n, int itmax, const double opts[4], double info[10],
                       void *adata) {
    double tau = 1e-3, eps1 = 1e-17, eps2 = 1e-17, eps3 = 1e-17;   /* levmar.h:90-91 */
    if (opts) { tau = opts[0]; eps1 = opts[1]; eps2 = opts[2]; eps3 = opts[3]; }
    const double eps2_sq = eps2 * eps2;
    double *e = malloc(sizeof(double) * n), *hx = malloc(sizeof(double) * n);
    double *jac = malloc(sizeof(double) * (size_t)n * m), *JtJ = malloc(sizeof(double) * (size_t)m * m);
    double *A = malloc(sizeof(double) * (size_t)m * m);
    double *Jte = malloc(sizeof(double) * m), *Dp = malloc(sizeof(double) * m), *pDp = malloc(sizeof(double) * m);
    double mu = 0, jacTe_inf = 0, p_eL2, pDp_eL2, p_L2, Dp_L2 = DBL_MAX, init_p_eL2;
    int nu = 2, stop = 0, nfev, njev = 0, nlss = 0, k;

    func(p, hx, m, n, adata); nfev = 1;
    p_eL2 = lm_l2_xmy(e, x, hx, n);
    init_p_eL2 = p_eL2;
    if (!isfinite(p_eL2)) stop = 7;
    for (k = 0; k < itmax && !stop; ++k) {
        if (p_eL2 <= eps3) { stop = 6; break; }
        jacf(p, jac, m, n, adata); ++njev;
        memset(JtJ, 0, sizeof(double) * (size_t)m * m); memset(Jte, 0, sizeof(double) * m);
        if ((long)n * m < 32 * 32) {
            /* small problems, lm_core.c:217-237: rows visited from the last to the first */
            for (int l = n - 1; l >= 0; --l) {
                const double *row = jac + (size_t)l * m;
                for (int i = 0; i < m; ++i) {
                    const double alpha = row[i];
                    double *dst = JtJ + (size_t)i * m;
                    for (int j = 0; j <= i; ++j) dst[j] += row[j] * alpha;
                    Jte[i] += alpha * e[l];
                }
            }
            for (int i = 0; i < m; ++i) for (int j = i + 1; j < m; ++j) JtJ[(size_t)i * m + j] = JtJ[(size_t)j * m + i];
        } else {
            /* large problems, misc_core.c:80-132: every entry is a sum of 32-row partial sums */
            for (int i = 0; i < m; ++i) for (int j = i; j < m; ++j) {
                double tot = 0.0;
                for (int kk = 0; kk < n; kk += 32) {
                    const int ke = kk + 32 < n ? kk + 32 : n;
                    double part = 0.0;
                    for (int r = kk; r < ke; ++r) part += jac[(size_t)r * m + i] * jac[(size_t)r * m + j];
                    tot += part;
                }
                JtJ[(size_t)i * m + j] = tot; JtJ[(size_t)j * m + i] = tot;
            }
            for (int r = 0; r < n; ++r) { const double er = e[r]; const double *row = jac + (size_t)r * m; for (int l = 0; l < m; ++l) Jte[l] += row[l] * er; }
        }
        p_L2 = 0; jacTe_inf = 0;
        for (int i = 0; i < m; ++i) { double a = fabs(Jte[i]); if (a > jacTe_inf) jacTe_inf = a; p_L2 += p[i] * p[i]; }
        if (jacTe_inf <= eps1) { Dp_L2 = 0; stop = 1; break; }
        if (k == 0) {
            double mx = -DBL_MAX; for (int i = 0; i < m; ++i) if (JtJ[(size_t)i * m + i] > mx) mx = JtJ[(size_t)i * m + i];
            mu = tau * mx;
        }
        for (;;) {
            memcpy(A, JtJ, sizeof(double) * (size_t)m * m);
            for (int i = 0; i < m; ++i) A[(size_t)i * m + i] += mu;
            int ok = orc_lu_solve(m, A, Jte, Dp); ++nlss;   /* AX_EQ_B_LU, lm_core.c:302 without LAPACK */
            if (ok) {
                Dp_L2 = 0;
                for (int i = 0; i < m; ++i) { pDp[i] = p[i] + Dp[i]; Dp_L2 += Dp[i] * Dp[i]; }
                if (Dp_L2 <= eps2_sq * p_L2) { stop = 2; break; }
                if (Dp_L2 >= (p_L2 + eps2) / (LM_EPSILON * LM_EPSILON)) { stop = 4; break; }
                func(pDp, hx, m, n, adata); ++nfev;
                pDp_eL2 = lm_l2_xmy(hx, x, hx, n);
                if (!isfinite(pDp_eL2)) { stop = 7; break; }
                double dL = 0; for (int i = 0; i < m; ++i) dL += Dp[i] * (mu * Dp[i] + Jte[i]);
                double dF = p_eL2 - pDp_eL2;
                if (dL > 0.0 && dF > 0.0) {
                    double t = 2.0 * dF / dL - 1.0; t = 1.0 - t * t * t;
                    mu = mu * ((t >= LM_ONE_THIRD) ? t : LM_ONE_THIRD);
                    nu = 2;
                    memcpy(p, pDp, sizeof(double) * m);
                    memcpy(e, hx, sizeof(double) * n);
                    p_eL2 = pDp_eL2;
                    break;
                }
            }
            mu *= nu;
            int nu2 = nu << 1;
            if (nu2 <= nu) { stop = 5; break; }
            nu = nu2;
        }
    }
    if (k >= itmax && !stop) stop = 3;
    if (info) {
        info[0] = init_p_eL2; info[1] = p_eL2; info[2] = jacTe_inf; info[3] = Dp_L2;
        double mx = DBL_MIN; for (int i = 0; i < m; ++i) if (JtJ[(size_t)i * m + i] > mx) mx = JtJ[(size_t)i * m + i];
        info[4] = mu / mx; info[5] = k; info[6] = stop; info[7] = nfev; info[8] = njev; info[9] = nlss;
    }
    free(e); free(hx); free(jac); free(JtJ); free(A); free(Jte); free(Dp); free(pDp);
    return (stop != 4 && stop != 7) ? k : -1;
}

/* ======================================================================== */
/* homography LM: homography.cpp:159-236 on top of the levmar engine          */
/* ======================================================================== */
typedef struct { int n; const float *x1, *y1, *x2, *y2; } hom_ctx;

void orc_homography_errors(int n, const float *px1, const float *py1, const float *px2, const float *py2,
                           const double h[8], double *e) {
    for (int i = 0; i < n; ++i) {
        const float x1 = px1[i], y1 = py1[i], x2 = px2[i], y2 = py2[i];
        double w = h[6] * x1 + h[7] * y1 + 1.0;
        w = (w > DBL_EPSILON) ? 1.0 / w : 0.0;
        const double xi = (h[0] * x1 + h[1] * y1 + h[2]) * w;
        const double yi = (h[3] * x1 + h[4] * y1 + h[5]) * w;
        e[2 * i] = xi - x2;
        e[2 * i + 1] = yi - y2;
    }
}

void orc_homography_jacobian(int n, const float *px1, const float *py1, const double h[8], double *J) {
    for (int i = 0; i < n; ++i) {
        const float x1 = px1[i], y1 = py1[i];
        double w = h[6] * x1 + h[7] * y1 + 1.0;
        w = (w > DBL_EPSILON) ? 1.0 / w : 0.0;
        const double xi = (h[0] * x1 + h[1] * y1 + h[2]) * w;
        const double yi = (h[3] * x1 + h[4] * y1 + h[5]) * w;
        double *r0 = J + (size_t)(2 * i) * 8, *r1 = r0 + 8;
        r0[0] = x1 * w; r0[1] = y1 * w; r0[2] = w; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -x1 * xi * w; r0[7] = -y1 * xi * w;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x1 * w; r1[4] = y1 * w; r1[5] = w; r1[6] = -x1 * yi * w; r1[7] = -y1 * yi * w;
    }
}

static void hom_func(double *p, double *hx, int m, int n, void *adata) {
    const hom_ctx *c = adata; (void)m; (void)n;
    orc_homography_errors(c->n, c->x1, c->y1, c->x2, c->y2, p, hx);
}
static void hom_jac(double *p, double *j, int m, int n, void *adata) {
    const hom_ctx *c = adata; (void)m; (void)n;
    orc_homography_jacobian(c->n, c->x1, c->y1, p, j);
}

typedef int (*dlevmar_der_fn)(orc_lm_func, orc_lm_func, double *, double *, int, int, int, double *, double *,
                              double *, double *, void *);

int orc_homography_lm(void *engine, int n, const float *x1, const float *y1, const float *x2, const float *y2,
                      double h[8], int itmax, const double opts[4], double info[10]) {
    hom_ctx c = { n, x1, y1, x2, y2 };
    if (2 * n < 8) return -1;
    if (engine) {
        double o[5] = { opts[0], opts[1], opts[2], opts[3], 0.0 };
        return ((dlevmar_der_fn)engine)(hom_func, hom_jac, h, NULL, 8, 2 * n, itmax, o, info, NULL, NULL, &c);
    }
    return orc_lm_nielsen_der(hom_func, hom_jac, h, NULL, 8, 2 * n, itmax, opts, info, &c);
}

/* ======================================================================== */
/* homography DLT and transfer residuals: homography.cpp:17-110                */
/* ======================================================================== */
/* normalize_points (homography.cpp:17-42).  The reference sums in cv::Point2f arithmetic: the centre and the mean
 * squared distance are float sums in point order (Point_::dot is a float expression), `sqrt(nrm)` with a float argument
 * under `using namespace std` is the float overload, the scale is a double, and `(p - center) * scale` is
 * Point2f * double: the double product rounded back to float (OpenCV's operator*). */
static void hom_normalize(int n, const float *x, const float *y, float *nx, float *ny, double T[9]) {
    float cx = 0.0f, cy = 0.0f;
    for (int i = 0; i < n; ++i) { cx += x[i]; cy += y[i]; }
    cx /= (float)n; cy /= (float)n;
    float nrm = 0.0f;
    for (int i = 0; i < n; ++i) {
        const float dx = x[i] - cx, dy = y[i] - cy;
        const float d = dx * dx + dy * dy;
        nrm += d;
    }
    nrm /= (float)n;
    const double scale = sqrt(2.0) / (double)sqrtf(nrm);
    T[0] = scale; T[1] = 0.0; T[2] = -scale * cx;
    T[3] = 0.0; T[4] = scale; T[5] = -scale * cy;
    T[6] = 0.0; T[7] = 0.0; T[8] = 1.0;
    for (int i = 0; i < n; ++i) {
        nx[i] = (float)((double)(x[i] - cx) * scale);
        ny[i] = (float)((double)(y[i] - cy) * scale);
    }
}

/* Matrix3d::inverse(): cofactors times the reciprocal of the determinant (Eigen's fixed-size 3x3 path, un-vendored:
 * from its published source; the order of the additions inside a cofactor is not pinned) */
static void hom_inverse3(const double M[9], double R[9]) {
    const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const double det = M[0] * c00 + M[1] * c01 + M[2] * c02, id = 1.0 / det;
    R[0] = c00 * id; R[1] = (M[2] * M[7] - M[1] * M[8]) * id; R[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    R[3] = c01 * id; R[4] = (M[0] * M[8] - M[2] * M[6]) * id; R[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    R[6] = c02 * id; R[7] = (M[1] * M[6] - M[0] * M[7]) * id; R[8] = (M[0] * M[4] - M[1] * M[3]) * id;
}

static void hom_mul3(const double A[9], const double B[9], double C[9]) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}

/* The right singular vector of the smallest singular value of A (m x 9, row-major, destroyed): what
 * JacobiSVD(A, ComputeFullV).matrixV().col(8) is up to sign and rounding (homography.cpp:86-88; Eigen un-vendored).
 * One-sided Jacobi (Hestenes): plane rotations from the right make A's columns orthogonal, cyclic by rows, until a
 * sweep rotates nothing (60 at most); V collects the rotations; the column of smallest norm wins.  Returns the number of sweeps. */
int orc_null_vector9(int m, double *A, double v[9]) {
    double V[81];
    for (int i = 0; i < 81; ++i) V[i] = (i % 10 == 0) ? 1.0 : 0.0;
    /* a column whose norm has fallen below 1e-14 |A|_F is the null direction of a rank-deficient A (four point pairs:
     * eight rows): it is left alone, its inner products are rounding noise */
    double fro = 0.0;
    for (int i = 0; i < 9 * m; ++i) fro += A[i] * A[i];
    const double tiny = 1e-28 * fro;
    int sweeps = 0;
    for (; sweeps < 60; ++sweeps) {
        int rotated = 0;
        for (int p = 0; p < 8; ++p)
            for (int q = p + 1; q < 9; ++q) {
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int i = 0; i < m; ++i) {
                    const double ap = A[9 * i + p], aq = A[9 * i + q];
                    al += ap * ap; be += aq * aq; ga += ap * aq;
                }
                if (!(fabs(ga) > 1e-15 * sqrt(al * be)) || al <= tiny || be <= tiny) continue;
                rotated = 1;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < m; ++i) {
                    const double ap = A[9 * i + p], aq = A[9 * i + q];
                    A[9 * i + p] = c * ap - sn * aq; A[9 * i + q] = sn * ap + c * aq;
                }
                for (int i = 0; i < 9; ++i) {
                    const double vp = V[9 * i + p], vq = V[9 * i + q];
                    V[9 * i + p] = c * vp - sn * vq; V[9 * i + q] = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    int best = 0;
    double least = 0.0;
    for (int j = 0; j < 9; ++j) {
        double s = 0.0;
        for (int i = 0; i < m; ++i) s += A[9 * i + j] * A[9 * i + j];
        if (j == 0 || s < least) { least = s; best = j; }
    }
    for (int i = 0; i < 9; ++i) v[i] = V[9 * i + best];
    return sweeps;
}

/* the constraint matrix of homography.cpp:58-83 over the normalised points: 2n x 9 row-major */
void orc_homography_dlt_matrix(int n, const float *x1, const float *y1, const float *x2, const float *y2,
                               double *A, double T1[9], double T2[9]) {
    float *buf = malloc(sizeof(float) * 4 * (size_t)(n > 0 ? n : 1));
    float *a = buf, *b = buf + n, *c = buf + 2 * n, *d = buf + 3 * n;
    hom_normalize(n, x1, y1, a, b, T1);
    hom_normalize(n, x2, y2, c, d, T2);
    for (int i = 0; i < n; ++i) {
        const double u1 = a[i], v1 = b[i], u2 = c[i], v2 = d[i];
        double *r0 = A + (size_t)18 * i, *r1 = r0 + 9;
        r0[0] = u1; r0[1] = v1; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -u2 * u1; r0[7] = -u2 * v1; r0[8] = -u2;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = -u1; r1[4] = -v1; r1[5] = -1; r1[6] = v2 * u1; r1[7] = v2 * v1; r1[8] = v2;
    }
    free(buf);
}

/* HomographyEstimator::solve (homography.cpp:46-93): H row-major, H(2,2) = 1.  Returns the sweeps of the SVD, -1 and
 * H untouched when n < 4 (the reference asserts nothing there; eight equations do not determine H). */
int orc_homography_dlt(int n, const float *x1, const float *y1, const float *x2, const float *y2, double H[9]) {
    if (n < 4) return -1;
    double *A = malloc(sizeof(double) * 18 * (size_t)n), T1[9], T2[9], v[9], T2i[9], M[9];
    orc_homography_dlt_matrix(n, x1, y1, x2, y2, A, T1, T2);
    const int sweeps = orc_null_vector9(2 * n, A, v);
    free(A);
    /* H_t.transpose() of the column-major map of the null vector = its row-major reading (:89-91) */
    hom_inverse3(T2, T2i);
    hom_mul3(T2i, v, M);
    hom_mul3(M, T1, H);
    const double h22 = H[8];
    for (int i = 0; i < 9; ++i) H[i] /= h22;
    return sweeps;
}

/* HomographyEstimator::residuals (homography.cpp:95-110): per pair |hn(H p1) - p2| + |hn(H^-1 p2) - p1| */
void orc_homography_residuals(int n, const float *x1, const float *y1, const float *x2, const float *y2,
                              const double H[9], double *res) {
    double Hi[9];
    hom_inverse3(H, Hi);
    for (int i = 0; i < n; ++i) {
        const double a = x1[i], b = y1[i], c = x2[i], d = y2[i];
        const double w1 = H[6] * a + H[7] * b + H[8], w2 = Hi[6] * c + Hi[7] * d + Hi[8];
        const double e0 = (H[0] * a + H[1] * b + H[2]) / w1 - c, e1 = (H[3] * a + H[4] * b + H[5]) / w1 - d;
        const double f0 = (Hi[0] * c + Hi[1] * d + Hi[2]) / w2 - a, f1 = (Hi[3] * c + Hi[4] * d + Hi[5]) / w2 - b;
        res[i] = sqrt(e0 * e0 + e1 * e1) + sqrt(f0 * f0 + f1 * f1);
    }
}

/* ---- HomographyEstimator::solveRansac (homography.cpp:112-156) on RANSAC::estimate (ransac.hpp:28-78) ---- */
void orc_ransac_defaults(orc_ransac_params *p) {
    p->max_error = 0.1; p->outlier_probability = 0.1; p->max_num_trials = 100; p->min_inliers = 5;   /* ransac.hpp:14-22 */
}

/* rng.hpp:62-75: vidx = 0 .. N-1; per draw: swap(vidx[index], vidx[max_idx]), take vidx[max_idx], --max_idx */
void orc_rng_sample4(int N, const int draw[4], int subset[4]) {
    int *vidx = malloc(sizeof(int) * (size_t)(N > 0 ? N : 1));
    for (int i = 0; i < N; ++i) vidx[i] = i;
    int max_idx = N - 1;
    for (int k = 0; k < 4; ++k) {
        const int index = draw[k];
        const int t = vidx[index]; vidx[index] = vidx[max_idx]; vidx[max_idx] = t;
        subset[k] = vidx[max_idx];
        --max_idx;
    }
    free(vidx);
}

int orc_homography_ransac(int n, const float *x1, const float *y1, const float *x2, const float *y2,
                          const orc_ransac_params *p, const int *subsets, double H[9], int *inliers, int *n_inliers,
                          int *trials_run) {
    *n_inliers = 0; *trials_run = 0;
    if (n < 4) return 0;                                   /* assert(N <= n), ransac.hpp:32 */
    const double log_probability = log(1.0 - p->outlier_probability);            /* :34 */
    const double one_over_indices = 1.0 / (double)n;                             /* :35 */
    size_t trial_count = 0, max_trials = (size_t)p->max_num_trials;
    const size_t max_num = (size_t)p->max_num_trials;
    double min_residual = DBL_MAX;
    double *res = malloc(sizeof(double) * (size_t)n);
    int *cur = malloc(sizeof(int) * (size_t)n), nbest = 0;
    float *g = malloc(sizeof(float) * 4 * (size_t)n);     /* gathered subsets: the four points of a trial, the inliers at the end */
    while (trial_count < (max_num < max_trials ? max_num : max_trials)) {        /* :44 */
        const int *sub = subsets + 4 * trial_count;
        for (int k = 0; k < 4; ++k) { g[k] = x1[sub[k]]; g[n + k] = y1[sub[k]]; g[2 * n + k] = x2[sub[k]]; g[3 * n + k] = y2[sub[k]]; }
        double Ht[9];
        orc_homography_dlt(4, g, g + n, g + 2 * n, g + 3 * n, Ht);               /* fit(): always true (homography.cpp:119-129) */
        /* findInliers (homography.cpp:131-144) */
        orc_homography_residuals(n, x1, y1, x2, y2, Ht, res);
        int cnt = 0;
        double total = 0.0;
        for (int i = 0; i < n; ++i)
            if (res[i] < p->max_error) { cur[cnt++] = i; total += res[i]; }
        total /= (double)cnt;
        if ((size_t)cnt > (size_t)p->min_inliers) {                              /* :55 */
            if (total < min_residual) {
                min_residual = total;
                memcpy(inliers, cur, sizeof(int) * (size_t)cnt);
                nbest = cnt;
                const double w = (double)cnt * one_over_indices;
                double p_no_outliers = 1.0 - pow(w, 4.0);
                p_no_outliers = p_no_outliers > DBL_EPSILON ? p_no_outliers : DBL_EPSILON;
                p_no_outliers = p_no_outliers < 1.0 - DBL_EPSILON ? p_no_outliers : 1.0 - DBL_EPSILON;
                max_trials = (size_t)(log_probability / log(p_no_outliers));      /* :66, a double stored in a size_t */
            }
        }
        ++trial_count;
    }
    *trials_run = (int)trial_count;
    *n_inliers = nbest;
    int ok = 0;
    if (nbest > 0) {                                       /* :75-77: the model of ALL inliers of the best trial */
        for (int k = 0; k < nbest; ++k) {
            const int i = inliers[k];
            g[k] = x1[i]; g[n + k] = y1[i]; g[2 * n + k] = x2[i]; g[3 * n + k] = y2[i];
        }
        orc_homography_dlt(nbest, g, g + n, g + 2 * n, g + 3 * n, H);
        ok = 1;
    }
    free(g); free(cur); free(res);
    return ok;
}

/* ======================================================================== */
/* tool-level bundle adjustment: hand_eye_registration.cpp:48-238             */
/* ======================================================================== */
/* ceres::QuaternionRotatePoint (Ceres rotation.h, un-vendored: restated from the published 1.x source): the
 * quaternion is normalised, then UnitQuaternionRotatePoint's expanded polynomial form. */
static void d_quaternion_rotate_point(const dual q[4], const dual pt[3], dual out[3]) {
    dual n2 = d_add(d_add(d_mul(q[0], q[0]), d_mul(q[1], q[1])), d_add(d_mul(q[2], q[2]), d_mul(q[3], q[3])));
    dual sc = d_div(d_const(1.0), d_sqrt(n2));
    dual u[4] = { d_mul(q[0], sc), d_mul(q[1], sc), d_mul(q[2], sc), d_mul(q[3], sc) };
    dual zero = d_const(0.0), two = d_const(2.0);
    dual t2 = d_mul(u[0], u[1]), t3 = d_mul(u[0], u[2]), t4 = d_mul(u[0], u[3]);
    dual t5 = d_sub(zero, d_mul(u[1], u[1])), t6 = d_mul(u[1], u[2]), t7 = d_mul(u[1], u[3]);
    dual t8 = d_sub(zero, d_mul(u[2], u[2])), t9 = d_mul(u[2], u[3]), t1 = d_sub(zero, d_mul(u[3], u[3]));
    out[0] = d_add(d_mul(two, d_add(d_add(d_mul(d_add(t8, t1), pt[0]), d_mul(d_sub(t6, t4), pt[1])), d_mul(d_add(t3, t7), pt[2]))), pt[0]);
    out[1] = d_add(d_mul(two, d_add(d_add(d_mul(d_add(t4, t6), pt[0]), d_mul(d_add(t5, t1), pt[1])), d_mul(d_sub(t9, t2), pt[2]))), pt[1]);
    out[2] = d_add(d_mul(two, d_add(d_add(d_mul(d_sub(t7, t3), pt[0]), d_mul(d_add(t2, t9), pt[1])), d_mul(d_add(t5, t8), pt[2]))), pt[2]);
}

void orc_toolba_autodiff(const double intr[9], const double trans[12], const double q[4], const double t[3],
                         const double X[3], double ox, double oy, double r[2], double Jq[8], double Jt[6], double Jx[6]) {
    dual dq[4], dt[3], dX[3], Xt[3], x[3];
    for (int i = 0; i < 4; ++i) dq[i] = d_var(q[i], i);
    for (int i = 0; i < 3; ++i) dt[i] = d_var(t[i], 4 + i);
    for (int i = 0; i < 3; ++i) dX[i] = d_var(X[i], 7 + i);
    for (int i = 0; i < 3; ++i)      /* C = trans_ * (X, 1), hand_eye_registration.cpp:82-92 */
        Xt[i] = d_add(d_add(d_add(d_mul(d_const(trans[4 * i]), dX[0]), d_mul(d_const(trans[4 * i + 1]), dX[1])),
                            d_mul(d_const(trans[4 * i + 2]), dX[2])), d_const(trans[4 * i + 3]));
    d_quaternion_rotate_point(dq, Xt, x);
    for (int i = 0; i < 3; ++i) x[i] = d_add(x[i], dt[i]);
    dual xn = d_div(x[0], x[2]), yn = d_div(x[1], x[2]);
    dual two = d_const(2.0);
    dual k1 = d_const(intr[ORC_K1]), k2 = d_const(intr[ORC_K2]), k3 = d_const(intr[ORC_K3]);
    dual p1 = d_const(intr[ORC_P1]), p2 = d_const(intr[ORC_P2]);
    dual r2 = d_add(d_mul(xn, xn), d_mul(yn, yn));
    dual r4 = d_mul(r2, r2);
    dual r6 = d_mul(r4, r2);
    dual rc = d_add(d_add(d_add(d_const(1.0), d_mul(k1, r2)), d_mul(k2, r4)), d_mul(k3, r6));
    dual xd = d_add(d_add(d_mul(xn, rc), d_mul(d_mul(d_mul(two, p1), xn), yn)), d_mul(p2, d_add(r2, d_mul(d_mul(two, xn), xn))));
    dual yd = d_add(d_add(d_mul(yn, rc), d_mul(d_mul(d_mul(two, p2), xn), yn)), d_mul(p1, d_add(r2, d_mul(d_mul(two, yn), yn))));
    dual e[2] = { d_sub(d_add(d_mul(d_const(intr[ORC_FX]), xd), d_const(intr[ORC_CX])), d_const(ox)),
                  d_sub(d_add(d_mul(d_const(intr[ORC_FY]), yd), d_const(intr[ORC_CY])), d_const(oy)) };
    for (int k = 0; k < 2; ++k) {
        r[k] = e[k].v;
        if (Jq) for (int i = 0; i < 4; ++i) Jq[k * 4 + i] = e[k].d[i];
        if (Jt) for (int i = 0; i < 3; ++i) Jt[k * 3 + i] = e[k].d[4 + i];
        if (Jx) for (int i = 0; i < 3; ++i) Jx[k * 3 + i] = e[k].d[7 + i];
    }
}

void orc_toolba_analytic(const double intr[9], const double trans[12], const double q[4], const double t[3],
                         const double X[3], double ox, double oy, double r[2], double Jq[8], double Jt[6], double Jx[6]) {
    double p[3];
    for (int i = 0; i < 3; ++i) p[i] = trans[4 * i] * X[0] + trans[4 * i + 1] * X[1] + trans[4 * i + 2] * X[2] + trans[4 * i + 3];
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const double sc = 1.0 / sqrt(n2);
    const double u0 = q[0] * sc, u1 = q[1] * sc, u2 = q[2] * sc, u3 = q[3] * sc;
    const double t2 = u0 * u1, t3 = u0 * u2, t4 = u0 * u3, t5 = -u1 * u1, t6 = u1 * u2, t7 = u1 * u3;
    const double t8 = -u2 * u2, t9 = u2 * u3, t1 = -u3 * u3;
    /* R - I, halved (the published polynomial form) */
    const double M[9] = { t8 + t1, t6 - t4, t3 + t7,  t4 + t6, t5 + t1, t9 - t2,  t7 - t3, t2 + t9, t5 + t8 };
    double xc[3];
    for (int i = 0; i < 3; ++i) xc[i] = 2.0 * (M[3 * i] * p[0] + M[3 * i + 1] * p[1] + M[3 * i + 2] * p[2]) + p[i] + t[i];
    const double fx = intr[ORC_FX], fy = intr[ORC_FY], cx = intr[ORC_CX], cy = intr[ORC_CY];
    const double k1 = intr[ORC_K1], k2 = intr[ORC_K2], k3 = intr[ORC_K3], p1 = intr[ORC_P1], p2 = intr[ORC_P2];
    const double xn = xc[0] / xc[2], yn = xc[1] / xc[2];
    const double r2 = xn * xn + yn * yn, r4 = r2 * r2, r6 = r4 * r2;
    const double rc = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
    const double xd = xn * rc + 2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn);
    const double yd = yn * rc + 2.0 * p2 * xn * yn + p1 * (r2 + 2.0 * yn * yn);
    r[0] = fx * xd + cx - ox;
    r[1] = fy * yd + cy - oy;
    if (!Jq && !Jt && !Jx) return;
    const double rcp = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4;
    const double dxx = rc + 2.0 * xn * xn * rcp + 2.0 * p1 * yn + 6.0 * p2 * xn;
    const double dxy = 2.0 * xn * yn * rcp + 2.0 * p1 * xn + 2.0 * p2 * yn;
    const double dyy = rc + 2.0 * yn * yn * rcp + 2.0 * p2 * xn + 6.0 * p1 * yn;
    const double iz = 1.0 / xc[2];
    const double a11 = fx * dxx, a12 = fx * dxy, a21 = fy * dxy, a22 = fy * dyy;
    const double D[6] = { a11 * iz, a12 * iz, -(a11 * xn + a12 * yn) * iz,  a21 * iz, a22 * iz, -(a21 * xn + a22 * yn) * iz };
    if (Jt) memcpy(Jt, D, sizeof D);
    if (Jx) {      /* d x / d X = R(q) * trans[:, :3] */
        double R[9];
        for (int i = 0; i < 9; ++i) R[i] = 2.0 * M[i] + ((i % 4 == 0) ? 1.0 : 0.0);
        for (int k = 0; k < 2; ++k) for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int i = 0; i < 3; ++i) {
                const double rt = R[3 * i] * trans[j] + R[3 * i + 1] * trans[4 + j] + R[3 * i + 2] * trans[8 + j];
                s += D[k * 3 + i] * rt;
            }
            Jx[k * 3 + j] = s;
        }
    }
    if (Jq) {
        /* d x / d u for the unit quaternion u, then through u = q / |q|: (I - u u^T) / |q| */
        const double G[12] = {
            2 * (-u3 * p[1] + u2 * p[2]), 2 * (u2 * p[1] + u3 * p[2]), 2 * (-2 * u2 * p[0] + u1 * p[1] + u0 * p[2]), 2 * (-2 * u3 * p[0] - u0 * p[1] + u1 * p[2]),
            2 * (u3 * p[0] - u1 * p[2]), 2 * (u2 * p[0] - 2 * u1 * p[1] - u0 * p[2]), 2 * (u1 * p[0] + u3 * p[2]), 2 * (u0 * p[0] - 2 * u3 * p[1] + u2 * p[2]),
            2 * (-u2 * p[0] + u1 * p[1]), 2 * (u3 * p[0] + u0 * p[1] - 2 * u1 * p[2]), 2 * (-u0 * p[0] + u3 * p[1] - 2 * u2 * p[2]), 2 * (u1 * p[0] + u2 * p[1]) };
        const double u[4] = { u0, u1, u2, u3 };
        double Gq[12];
        for (int i = 0; i < 3; ++i) {
            const double gu = G[4 * i] * u0 + G[4 * i + 1] * u1 + G[4 * i + 2] * u2 + G[4 * i + 3] * u3;
            for (int j = 0; j < 4; ++j) Gq[4 * i + j] = (G[4 * i + j] - gu * u[j]) * sc;
        }
        for (int k = 0; k < 2; ++k) for (int j = 0; j < 4; ++j)
            Jq[k * 4 + j] = D[k * 3] * Gq[j] + D[k * 3 + 1] * Gq[4 + j] + D[k * 3 + 2] * Gq[8 + j];
    }
}

void orc_huber(double a, double s, double rho[3]) {     /* ceres::HuberLoss::Evaluate, b = a^2 */
    const double b = a * a;
    if (a > 0.0 && s > b) {
        const double r = sqrt(s);
        rho[0] = 2.0 * a * r - b;
        rho[1] = a / r; if (rho[1] < DBL_MIN) rho[1] = DBL_MIN;
        rho[2] = -rho[1] / (2.0 * s);
    } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
}

double orc_toolba_cost(const orc_toolba_problem *pb, const double p[7], const double *points) {
    double c = 0;
    for (int v = 0; v < pb->V; ++v)
        for (int o = pb->view_off[v]; o < pb->view_off[v + 1]; ++o) {
            double r[2], rho[3];
            orc_toolba_analytic(pb->intr, pb->trans + 12 * v, p, p + 4, points + 3 * pb->point_id[o], pb->obs_x[o], pb->obs_y[o], r, 0, 0, 0);
            orc_huber(pb->huber, r[0] * r[0] + r[1] * r[1], rho);
            c += rho[0];
        }
    return 0.5 * c;
}

void orc_toolba_normal_eq(const orc_toolba_problem *pb, const double p[7], const double *points, double *H, double *g,
                          double *cost) {
    const int nf = 6 + 3 * pb->P;
    double Pq[12]; quat_plus_jac(p, Pq);
    memset(H, 0, sizeof(double) * (size_t)nf * nf); memset(g, 0, sizeof(double) * nf);
    double c = 0;
    for (int v = 0; v < pb->V; ++v)
        for (int o = pb->view_off[v]; o < pb->view_off[v + 1]; ++o) {
            const int pid = pb->point_id[o];
            double r[2], Jq[8], Jt[6], Jx[6], rho[3], J[2][9];
            orc_toolba_analytic(pb->intr, pb->trans + 12 * v, p, p + 4, points + 3 * pid, pb->obs_x[o], pb->obs_y[o], r, Jq, Jt, Jx);
            orc_huber(pb->huber, r[0] * r[0] + r[1] * r[1], rho);
            c += rho[0];
            /* ceres::Corrector with rho'' <= 0: residuals and Jacobians scaled by sqrt(rho') */
            const double w = sqrt(rho[1]);
            for (int k = 0; k < 2; ++k) {
                for (int a = 0; a < 3; ++a)
                    J[k][a] = w * (Jq[k * 4] * Pq[a] + Jq[k * 4 + 1] * Pq[3 + a] + Jq[k * 4 + 2] * Pq[6 + a] + Jq[k * 4 + 3] * Pq[9 + a]);
                for (int a = 0; a < 3; ++a) { J[k][3 + a] = w * Jt[k * 3 + a]; J[k][6 + a] = w * Jx[k * 3 + a]; }
            }
            int idx[9]; for (int a = 0; a < 6; ++a) idx[a] = a; for (int a = 0; a < 3; ++a) idx[6 + a] = 6 + 3 * pid + a;
            for (int k = 0; k < 2; ++k) {
                const double rk = w * r[k];
                for (int a = 0; a < 9; ++a) {
                    g[idx[a]] += J[k][a] * rk;
                    for (int b = 0; b < 9; ++b) H[(size_t)idx[a] * nf + idx[b]] += J[k][a] * J[k][b];
                }
            }
        }
    if (cost) *cost = 0.5 * c;
}

void orc_lm_options_toolba(orc_lm_options *o) {   /* hand_eye_registration.cpp:201-210 */
    lm_defaults(o);
    o->nonmonotonic = 0;
    o->max_iterations = 100;
    o->function_tolerance = 1e-20; o->gradient_tolerance = 1e-20; o->parameter_tolerance = 1e-20;
    o->inner_iterations = 1;            /* :205 */
}

typedef struct { const orc_toolba_problem *pb; int nf; double *H, *g, *A; } tba_ctx;
static double tba_m_linearize(void *vc, const double *x, double *diag, double *grad) {
    tba_ctx *c = vc; double cost;
    orc_toolba_normal_eq(c->pb, x, x + 7, c->H, c->g, &cost);
    for (int a = 0; a < c->nf; ++a) { diag[a] = c->H[(size_t)a * c->nf + a]; grad[a] = c->g[a]; }
    return cost;
}
static int tba_m_solve(void *vc, const double *D, double *delta) {
    tba_ctx *c = vc;
    memcpy(c->A, c->H, sizeof(double) * (size_t)c->nf * c->nf);
    for (int a = 0; a < c->nf; ++a) { c->A[(size_t)a * c->nf + a] += D[a]; delta[a] = -c->g[a]; }
    return orc_cholesky_solve(c->nf, c->A, delta);
}
static void tba_m_plus(void *vc, const double *x, const double *delta, double *xo) {
    tba_ctx *c = vc;
    orc_quaternion_plus(x, delta, xo);
    for (int a = 0; a < 3; ++a) xo[4 + a] = x[4 + a] + delta[3 + a];
    for (int a = 0; a < 3 * c->pb->P; ++a) xo[7 + a] = x[7 + a] + delta[6 + a];
}
static double tba_m_cost(void *vc, const double *x) { tba_ctx *c = vc; return orc_toolba_cost(c->pb, x, x + 7); }
static double tba_m_free_norm2(void *vc, const double *x) { tba_ctx *c = vc; double s = 0; for (int i = 0; i < 7 + 3 * c->pb->P; ++i) s += x[i] * x[i]; return s; }

/* ---- inner iterations of the tool-level BA (use_inner_iterations = true, hand_eye_registration.cpp:205).  The
 * parameter blocks are q (4, QuaternionParameterization), t (3) and the points (:176-181); every residual block
 * touches q, t and one point, so Ceres' automatic ordering (see orc_calib_inner_ordering) is: with two or more
 * points in use the points have the lowest degree (2 < 1 + points) and go first, then q (named before t), then t;
 * reversed: t, q, points.  With a single point all three degrees tie and the naming order decides: q, then t, then
 * the point; reversed: the point, t, q. */
typedef struct { const orc_toolba_problem *pb; const double *x; int type, index; double H[9], g[3]; } tbi_ctx;   /* type 0 t, 1 q, 2 point */

static void tbi_visit(tbi_ctx *c, const double *xb, double *cost, double *H, double *g) {
    const orc_toolba_problem *pb = c->pb;
    double p[7], Pq[12], s = 0;
    memcpy(p, c->x, sizeof p);
    if (c->type == 0) memcpy(p + 4, xb, 3 * sizeof(double));
    if (c->type == 1) memcpy(p, xb, 4 * sizeof(double));
    quat_plus_jac(p, Pq);
    if (H) { memset(H, 0, 9 * sizeof(double)); memset(g, 0, 3 * sizeof(double)); }
    for (int v = 0; v < pb->V; ++v)
        for (int o = pb->view_off[v]; o < pb->view_off[v + 1]; ++o) {
            const int pid = pb->point_id[o];
            if (c->type == 2 && pid != c->index) continue;
            const double *X = c->type == 2 ? xb : c->x + 7 + 3 * pid;
            double r[2], Jq[8], Jt[6], Jx[6], rho[3];
            orc_toolba_analytic(pb->intr, pb->trans + 12 * v, p, p + 4, X, pb->obs_x[o], pb->obs_y[o], r, Jq, Jt, Jx);
            orc_huber(pb->huber, r[0] * r[0] + r[1] * r[1], rho);
            s += rho[0];
            if (!H) continue;
            const double w = sqrt(rho[1]);
            for (int k = 0; k < 2; ++k) {
                double row[3];
                for (int a = 0; a < 3; ++a)
                    row[a] = c->type == 0 ? w * Jt[k * 3 + a] : c->type == 2 ? w * Jx[k * 3 + a]
                           : w * (Jq[k * 4] * Pq[a] + Jq[k * 4 + 1] * Pq[3 + a] + Jq[k * 4 + 2] * Pq[6 + a] + Jq[k * 4 + 3] * Pq[9 + a]);
                for (int a = 0; a < 3; ++a) {
                    g[a] += row[a] * w * r[k];
                    for (int b = 0; b < 3; ++b) H[a * 3 + b] += row[a] * row[b];
                }
            }
        }
    *cost = 0.5 * s;
}
static double tbi_linearize(void *vc, const double *x, double *diag, double *grad) {
    tbi_ctx *c = vc; double cost;
    tbi_visit(c, x, &cost, c->H, c->g);
    for (int i = 0; i < 3; ++i) { diag[i] = c->H[i * 3 + i]; grad[i] = c->g[i]; }
    return cost;
}
static int tbi_solve(void *vc, const double *D, double *delta) {
    tbi_ctx *c = vc; double A[9];
    memcpy(A, c->H, sizeof A);
    for (int i = 0; i < 3; ++i) { A[i * 3 + i] += D[i]; delta[i] = -c->g[i]; }
    return orc_cholesky_solve(3, A, delta);
}
static void tbi_plus(void *vc, const double *x, const double *delta, double *xo) {
    tbi_ctx *c = vc;
    if (c->type == 1) orc_quaternion_plus(x, delta, xo);
    else for (int i = 0; i < 3; ++i) xo[i] = x[i] + delta[i];
}
static double tbi_cost(void *vc, const double *x) { tbi_ctx *c = vc; double cost; tbi_visit(c, x, &cost, 0, 0); return cost; }
static double tbi_norm2(void *vc, const double *x) { tbi_ctx *c = vc; double s = 0; for (int i = 0; i < (c->type == 1 ? 4 : 3); ++i) s += x[i] * x[i]; return s; }

double orc_toolba_inner_pass(const orc_toolba_problem *pb, double p[7], double *points) {
    const int ng = 7 + 3 * pb->P;
    double *x = malloc(sizeof(double) * ng);
    memcpy(x, p, 7 * sizeof(double)); memcpy(x + 7, points, sizeof(double) * 3 * pb->P);
    char *used = calloc(pb->P ? pb->P : 1, 1);
    int n_used = 0;
    for (int o = 0; o < pb->N; ++o) if (!used[pb->point_id[o]]) { used[pb->point_id[o]] = 1; ++n_used; }
    orc_lm_options io; lm_defaults(&io);
    io.max_radius = 1e32;                       /* default TrustRegionStrategy::Options, as in orc_calib_inner_pass */
    orc_lm_summary *isum = malloc(sizeof *isum);
    /* the order of the sets, as block types: 0 t, 1 q, 2 points */
    int order[3] = { 0, 1, 2 };
    if (n_used == 1) { order[0] = 2; order[1] = 0; order[2] = 1; }
    for (int k = 0; k < 3 && n_used > 0; ++k) {
        const int type = order[k];
        for (int idx = 0; idx < (type == 2 ? pb->P : 1); ++idx) {
            if (type == 2 && !used[idx]) continue;
            tbi_ctx c; memset(&c, 0, sizeof c);
            c.pb = pb; c.x = x; c.type = type; c.index = idx;
            double *xb = type == 0 ? x + 4 : type == 1 ? x : x + 7 + 3 * idx;
            const int nglob = type == 1 ? 4 : 3;
            double xloc[4];
            memcpy(xloc, xb, sizeof(double) * nglob);
            lm_model m = { &c, 3, nglob, tbi_linearize, tbi_solve, tbi_plus, tbi_cost, tbi_norm2, NULL };
            lm_ceres(&m, xloc, &io, isum);
            memcpy(xb, xloc, sizeof(double) * nglob);
        }
    }
    memcpy(p, x, 7 * sizeof(double)); memcpy(points, x + 7, sizeof(double) * 3 * pb->P);
    const double cost = orc_toolba_cost(pb, p, points);
    free(x); free(used); free(isum);
    return cost;
}

static double tba_m_inner(void *vc, double *x) { tba_ctx *c = vc; return orc_toolba_inner_pass(c->pb, x, x + 7); }

int orc_toolba_solve(const orc_toolba_problem *pb, double p[7], double *points, const orc_lm_options *opt,
                     orc_lm_summary *sum) {
    const int nf = 6 + 3 * pb->P, ng = 7 + 3 * pb->P;
    tba_ctx c = { pb, nf, malloc(sizeof(double) * (size_t)nf * nf), malloc(sizeof(double) * nf), malloc(sizeof(double) * (size_t)nf * nf) };
    double *x = malloc(sizeof(double) * ng);
    memcpy(x, p, 7 * sizeof(double)); memcpy(x + 7, points, sizeof(double) * 3 * pb->P);
    lm_model m = { &c, nf, ng, tba_m_linearize, tba_m_solve, tba_m_plus, tba_m_cost, tba_m_free_norm2, tba_m_inner };
    const int term = (opt->strategy == 1) ? lm_nielsen(&m, x, opt, sum) : lm_ceres(&m, x, opt, sum);
    memcpy(p, x, 7 * sizeof(double)); memcpy(points, x + 7, sizeof(double) * 3 * pb->P);
    free(c.H); free(c.g); free(c.A); free(x);
    return term;
}
