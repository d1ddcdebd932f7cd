/*
 * ilqr_oracle.c -- CPU restatement of the idiap/ilqr_planner hot path (see ilqr_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked into, imported by or called from the product.
 *
 * Reference paths are relative to /root/reference/ilqr_planner/ilqr_planner.
 * All matrices are row-major.  Quirks of the reference are reproduced on purpose.
 */
#include "ilqr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ small dense helpers */

/* Test aid (see below, orc_set_variant): bit 2 makes the three helpers accumulate with fused multiply-adds -- what a GPU (or a
   -mfma build of Eigen) does; equal in exact arithmetic, another rounding. */
static int g_variant = 0;
#define ACC(s, a, b) (((g_variant) & 4) ? fma((a), (b), (s)) : ((s) + (a) * (b)))

/* C(m x n) = A(m x k) * B(k x n) */
static void mm(double* C, const double* A, const double* B, int m, int k, int n) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int l = 0; l < k; l++) s = ACC(s, A[i * k + l], B[l * n + j]);
            C[i * n + j] = s;
        }
}
/* C(m x n) = A^T * B, A is (k x m), B is (k x n) */
static void mtm(double* C, const double* A, const double* B, int k, int m, int n) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int l = 0; l < k; l++) s = ACC(s, A[l * m + i], B[l * n + j]);
            C[i * n + j] = s;
        }
}
static double dot(const double* a, const double* b, int n) {
    double s = 0;
    for (int i = 0; i < n; i++) s = ACC(s, a[i], b[i]);
    return s;
}
static double norm(const double* a, int n) { return sqrt(dot(a, a, n)); }
/* Eigen DenseBase::isZero(prec = 1e-12): every |coeff| <= prec */
static int is_zero(const double* a, int n) {
    for (int i = 0; i < n; i++)
        if (!(fabs(a[i]) <= 1e-12)) return 0;
    return 1;
}

/* Eigen MatrixXd::inverse() on a dynamic-size matrix = PartialPivLU, then solve against identity. */
/* Test aid, not part of the restated algorithm: bit 0 replaces Qxu = A'PB by Qux' in the backward sweep.  The two are equal in
   exact arithmetic; tests use the switch to show that an instance's result depends on rounding-level reassociation. */
void orc_set_variant(int v) { g_variant = v; }
/* Test aids (see ilqr_oracle.h): decision-margin probe and resume-from-state.  They record / seed, they change no arithmetic. */
static __thread orc_probe_rec* g_probe = NULL;
static __thread int g_probe_cap = 0;
static __thread int g_it0 = 0;
static __thread double g_init_penalty = 0;
static __thread const double* g_lambda_mask = NULL;
static __thread int g_probe_all = 0;            /* keep evaluating the step sizes below the accepted one, for the record only */
static __thread const double* g_resume_x = NULL; /* the trajectory the handed-over controls belong to (the caller's own rollout of them) */
static __thread double g_resume_x_dev = 0;       /* largest |x_given - x_rolled| / max(1, |x|) seen when that trajectory was taken over */
void orc_set_probe(orc_probe_rec* buf, int cap) { g_probe = buf; g_probe_cap = buf ? cap : 0; }
void orc_set_probe_all(int on) { g_probe_all = on; }
void orc_set_resume_x(const double* X) { g_resume_x = X; g_resume_x_dev = 0; }
double orc_get_resume_x_dev(void) { return g_resume_x_dev; }
void orc_set_resume(int it0, double init_penalty, const double* lambda_mask) { g_it0 = it0; g_init_penalty = init_penalty; g_lambda_mask = lambda_mask; }

/* Test aid (variant bit 1): the inverse of a symmetric positive definite matrix by the symmetric sweep operator without pivoting --
   another backward-stable algorithm for the same quantity.  Used only to measure how far an iteration's result depends on HOW Quu is
   inverted (cond(Quu) * eps), never in the restated algorithm. */
static int inverse_sweep_nopivot(int n, const double* A, double* Ainv) {
    memcpy(Ainv, A, sizeof(double) * n * n);
    for (int c = 0; c < n; c++) {
        double d = Ainv[c * n + c];
        if (d == 0.0) return 1;
        for (int i = 0; i < n; i++) {
            if (i == c) continue;
            double f = Ainv[i * n + c] / d;
            for (int j = 0; j < n; j++)
                if (j != c) Ainv[i * n + j] -= f * Ainv[c * n + j];
        }
        for (int i = 0; i < n; i++)
            if (i != c) { Ainv[i * n + c] = Ainv[i * n + c] / d; Ainv[c * n + i] = Ainv[c * n + i] / d; }
        Ainv[c * n + c] = -1.0 / d;
    }
    for (int i = 0; i < n * n; i++) Ainv[i] = -Ainv[i];
    return 0;
}

int orc_inverse(int n, const double* A, double* Ainv) {
    if (g_variant & 2) return inverse_sweep_nopivot(n, A, Ainv);
    double* lu = (double*)malloc(sizeof(double) * n * n);
    int* piv = (int*)malloc(sizeof(int) * n);
    memcpy(lu, A, sizeof(double) * n * n);
    for (int i = 0; i < n; i++) piv[i] = i;
    int rc = 0;
    for (int k = 0; k < n; k++) {
        int r = k;
        double best = fabs(lu[k * n + k]);
        for (int i = k + 1; i < n; i++) {
            double v = fabs(lu[i * n + k]);
            if (v > best) { best = v; r = i; }
        }
        if (r != k) {
            for (int j = 0; j < n; j++) { double t = lu[k * n + j]; lu[k * n + j] = lu[r * n + j]; lu[r * n + j] = t; }
            int t = piv[k]; piv[k] = piv[r]; piv[r] = t;
        }
        double pv = lu[k * n + k];
        if (pv == 0.0) rc = 1;
        for (int i = k + 1; i < n; i++) {
            lu[i * n + k] /= pv;
            double f = lu[i * n + k];
            for (int j = k + 1; j < n; j++) lu[i * n + j] -= f * lu[k * n + j];
        }
    }
    for (int c = 0; c < n; c++) {
        /* solve L U x = P e_c */
        double* x = Ainv; /* column c written strided */
        for (int i = 0; i < n; i++) {
            double s = (piv[i] == c) ? 1.0 : 0.0;
            for (int j = 0; j < i; j++) s -= lu[i * n + j] * x[j * n + c];
            x[i * n + c] = s;
        }
        for (int i = n - 1; i >= 0; i--) {
            double s = x[i * n + c];
            for (int j = i + 1; j < n; j++) s -= lu[i * n + j] * x[j * n + c];
            x[i * n + c] = s / lu[i * n + i];
        }
    }
    free(lu);
    free(piv);
    return rc;
}

/* ------------------------------------------------------------------ Sd manifold utils (utils/sd.h) */

void orc_sd_H(const double q[4], double H[12]) { /* sd.h:23-27 dQuatToDxJac */
    H[0] = -q[1]; H[1] = q[0];  H[2] = -q[3];  H[3] = q[2];
    H[4] = -q[2]; H[5] = q[3];  H[6] = q[0];   H[7] = -q[1];
    H[8] = -q[3]; H[9] = -q[2]; H[10] = q[1];  H[11] = q[0];
}

double orc_sd_distance(const double x[4], const double y[4]) { /* sd.h:48-62 */
    double dist = dot(x, y, 4);
    if (dist > 1) dist = 1;
    else if (dist < -1) dist = -1;
    double ac = acos(dist);
    if (dist < 0) ac -= M_PI;
    return ac;
}

void orc_sd_logmap(const double base_in[4], const double y_in[4], double out[4]) { /* sd.h:67-82 */
    if (is_zero(base_in, 4) || is_zero(y_in, 4)) { memset(out, 0, 4 * sizeof(double)); return; }
    double b[4], y[4], tmp[4];
    double nb = norm(base_in, 4), ny = norm(y_in, 4);
    for (int i = 0; i < 4; i++) { b[i] = base_in[i] / nb; y[i] = y_in[i] / ny; }
    double by = dot(b, y, 4);
    for (int i = 0; i < 4; i++) tmp[i] = y[i] - by * b[i];
    double nt = norm(tmp, 4);
    if (nt == 0) { memset(out, 0, 4 * sizeof(double)); return; }
    double d = orc_sd_distance(b, y);
    for (int i = 0; i < 4; i++) out[i] = d * tmp[i] / nt;
}

void orc_sd_expmap(const double base_in[4], const double u[4], double out[4]) { /* sd.h:32-43 */
    double b[4];
    double nb = norm(base_in, 4);
    for (int i = 0; i < 4; i++) b[i] = base_in[i] / nb;
    double nu = norm(u, 4);
    if (nu == 0) { memcpy(out, b, sizeof(b)); return; }
    double r[4];
    for (int i = 0; i < 4; i++) r[i] = b[i] * cos(nu) + u[i] / nu * sin(nu);
    double nr = norm(r, 4);
    for (int i = 0; i < 4; i++) out[i] = r[i] / nr;
}

void orc_sd_transport(const double v[4], const double b1[4], const double b2[4], double out[4]) { /* sd.h:87-99 */
    if (is_zero(b1, 4) || is_zero(b2, 4)) { memcpy(out, v, 4 * sizeof(double)); return; }
    double dsq = pow(orc_sd_distance(b1, b2), 2);
    if (dsq == 0) { memcpy(out, v, 4 * sizeof(double)); return; }
    double l12[4], l21[4];
    orc_sd_logmap(b1, b2, l12);
    orc_sd_logmap(b2, b1, l21);
    double f = dot(l12, v, 4) / dsq;
    for (int i = 0; i < 4; i++) out[i] = v[i] - f * (l12[i] + l21[i]);
}

/* ------------------------------------------------------------------ FK + geometric Jacobian */

static void rot_axis(const double a[3], double th, double R[9]) { /* KDL Rotation::Rot2 (Rodrigues) */
    double ct = cos(th), st = sin(th), vt = 1 - ct;
    double x = a[0], y = a[1], z = a[2];
    R[0] = ct + vt * x * x;     R[1] = -z * st + vt * x * y; R[2] = y * st + vt * x * z;
    R[3] = z * st + vt * x * y; R[4] = ct + vt * y * y;      R[5] = -x * st + vt * y * z;
    R[6] = -y * st + vt * x * z; R[7] = x * st + vt * y * z; R[8] = ct + vt * z * z;
}

/* KDL Rotation::GetQuaternion (orocos_kdl frames.cpp; un-vendored dependency, version unpinned,
 * restated from the published algorithm; pinned by the negative-w FK literal of POS_ORN_MULTI_SYS.ipynb
 * cell 8).  Output order (w,x,y,z) as KDLRobot.cpp:103 stores it. */
static void kdl_quat(const double R[9], double q[4]) {
    double tr = R[0] + R[4] + R[8];
    double w, x, y, z;
    if (tr > 1e-12) {
        double s = 0.5 / sqrt(tr + 1.0);
        w = 0.25 / s;
        x = (R[7] - R[5]) * s;
        y = (R[2] - R[6]) * s;
        z = (R[3] - R[1]) * s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        double s = 2.0 * sqrt(1.0 + R[0] - R[4] - R[8]);
        w = (R[7] - R[5]) / s;
        x = 0.25 * s;
        y = (R[1] + R[3]) / s;
        z = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        double s = 2.0 * sqrt(1.0 + R[4] - R[0] - R[8]);
        w = (R[2] - R[6]) / s;
        x = (R[1] + R[3]) / s;
        y = 0.25 * s;
        z = (R[5] + R[7]) / s;
    } else {
        double s = 2.0 * sqrt(1.0 + R[8] - R[0] - R[4]);
        w = (R[3] - R[1]) / s;
        x = (R[2] + R[6]) / s;
        y = (R[5] + R[7]) / s;
        z = 0.25 * s;
    }
    q[0] = w; q[1] = x; q[2] = y; q[3] = z;
}

/* src/sim/KDLRobot.cpp:83-115: JntToJac + JntToCart + GetQuaternion, dx = Jt dq, w = Jr dq.
 * (dJac, :112, is never read by any System and is not restated.) */
void orc_fk(const orc_chain* c, const double* q, const double* dq,
            double p[3], double quat[4], double J[6 * ORC_MAX_DOF], double dx[3], double w[3]) {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pos[3] = {0, 0, 0};
    double org[ORC_MAX_DOF][3], ax[ORC_MAX_DOF][3];
    int dof = c->dof;
    for (int s = 0; s < c->n_seg; s++) {
        double t[3], Rn[9];
        for (int i = 0; i < 3; i++) t[i] = R[3 * i] * c->seg_xyz[s][0] + R[3 * i + 1] * c->seg_xyz[s][1] + R[3 * i + 2] * c->seg_xyz[s][2];
        for (int i = 0; i < 3; i++) pos[i] += t[i];
        mm(Rn, R, c->seg_R[s], 3, 3, 3);
        memcpy(R, Rn, sizeof(R));
        int j = c->seg_joint[s];
        if (j >= 0) {
            double Rq[9];
            for (int i = 0; i < 3; i++) {
                org[j][i] = pos[i];
                ax[j][i] = R[3 * i] * c->seg_axis[s][0] + R[3 * i + 1] * c->seg_axis[s][1] + R[3 * i + 2] * c->seg_axis[s][2];
            }
            rot_axis(c->seg_axis[s], q[j], Rq);
            mm(Rn, R, Rq, 3, 3, 3);
            memcpy(R, Rn, sizeof(R));
        }
    }
    for (int i = 0; i < 3; i++) p[i] = pos[i];
    kdl_quat(R, quat);
    for (int j = 0; j < dof; j++) {
        double r[3] = {pos[0] - org[j][0], pos[1] - org[j][1], pos[2] - org[j][2]};
        const double* z = ax[j];
        J[0 * dof + j] = z[1] * r[2] - z[2] * r[1];
        J[1 * dof + j] = z[2] * r[0] - z[0] * r[2];
        J[2 * dof + j] = z[0] * r[1] - z[1] * r[0];
        J[3 * dof + j] = z[0];
        J[4 * dof + j] = z[1];
        J[5 * dof + j] = z[2];
    }
    for (int i = 0; i < 3; i++) {
        dx[i] = 0;
        w[i] = 0;
        if (dq)
            for (int j = 0; j < dof; j++) {
                dx[i] += J[i * dof + j] * dq[j];
                w[i] += J[(3 + i) * dof + j] * dq[j];
            }
    }
}

/* ------------------------------------------------------------------ System */

void orc_system_finalize(orc_system* s) {
    s->dof = s->chain.dof;
    int tm = ORC_IS_TM(s->kind) ? 1 : 0;
    s->n_x = s->nb_deriv * s->dof + tm;  /* PosOrnPlannerSys.cpp:74 / PosOrnTimePlannerSys.cpp:67 */
    s->n_u = s->dof + tm;                /* :75 / :68 */
    s->n_f = 7 * s->nb_deriv + tm;       /* :76 / :69 */
    s->n_Q = s->n_f - s->nb_deriv;       /* :77 / :70 */
    if (ORC_IS_JOINT(s->kind)) {         /* JointSpacePlannerSys.cpp:71-74, JointSpaceTimePlannerSys.cpp:62-65: target space = state space */
        s->n_f = s->n_x;
        s->n_Q = s->n_x;
    }
}

/* getFxJac(xk): System.cpp:163-179 + PosOrnPlannerSys.cpp:80-102 ; PosOrnTimePlannerSys.cpp:85-137.
 * fx = [p; quat (; dp; dquat) (; t)], J = Jac | blkdiag(Jac,Jac) | bordered with 1 for the time state. */
/* Eigen::Quaterniond::toRotationMatrix (Eigen/src/Geometry/Quaternion.h), q = (w,x,y,z), no normalisation; row-major out */
static void eig_quat_to_mat(const double q[4], double m[9]) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    m[0] = 1 - (tyy + tzz); m[1] = txy - twz; m[2] = txz + twy;
    m[3] = txy + twz; m[4] = 1 - (txx + tzz); m[5] = tyz - twx;
    m[6] = txz - twy; m[7] = tyz + twx; m[8] = 1 - (txx + tyy);
}
/* Eigen::Quaterniond(Matrix3d) (quaternionbase_assign_impl<Other,3,3>): trace > 0 branch, else largest diagonal */
static void eig_mat_to_quat(const double m[9], double q[4]) {
    double t = m[0] + m[4] + m[8];
    double c[3]; /* x, y, z */
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        c[0] = (m[7] - m[5]) * t; c[1] = (m[2] - m[6]) * t; c[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        c[i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        c[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        c[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
    q[1] = c[0]; q[2] = c[1]; q[3] = c[2];
}

/* getFxJac (PosOrnPlannerSys.cpp:80-102, PosOrnTimePlannerSys.cpp:85-112) seen through the keypoint's frame when it has one
 * (TransformedSimulationInterface.cpp:53-103): p' = R'(p - t), R_ee' = R' R_ee -> Eigen quaternion, J' = blkdiag(R,R)' J,
 * dx' = R' dx, w' = R' w. */
static void fx_jac_frame(const orc_system* s, const orc_keypoint* kp, const double* x, double* fx, double* J) {
    int dof = s->dof, nd = s->nb_deriv, tm = ORC_IS_TM(s->kind);
    if (ORC_IS_JOINT(s->kind)) { /* JointSpace(Time)PlannerSys::getFxJac: f(x) = x (time state included), J = I */
        if (fx) memcpy(fx, x, sizeof(double) * s->n_f);
        if (J) {
            memset(J, 0, sizeof(double) * s->n_Q * s->n_x);
            for (int i = 0; i < s->n_Q && i < s->n_x; i++) J[i * s->n_x + i] = 1;
        }
        return;
    }
    if (kp && kp->joint) { /* keypoint of a joint-space sub-system of a hybrid SequentialSystem: that sub-system's f(x) = x, J = I */
        if (fx) memcpy(fx, x, sizeof(double) * s->n_x);
        if (J) {
            memset(J, 0, sizeof(double) * s->n_x * s->n_x);
            for (int i = 0; i < s->n_x; i++) J[i * s->n_x + i] = 1;
        }
        return;
    }
    double p[3], quat[4], Jac[6 * ORC_MAX_DOF], dx[3], w[3], dq0[ORC_MAX_DOF] = {0};
    const double* dq = (nd == 2) ? x + dof : dq0;
    orc_fk(&s->chain, x, dq, p, quat, Jac, dx, w);
    if (kp && kp->has_frame) {
        const double* R = kp->fR;
        double pp[3], ree[9], mm_[9], dxp[3], wp[3], Jn[6 * ORC_MAX_DOF];
        for (int i = 0; i < 3; i++) {
            pp[i] = 0; dxp[i] = 0; wp[i] = 0;
            for (int j = 0; j < 3; j++) {  /* R^T v */
                pp[i] += R[j * 3 + i] * (p[j] - kp->fp[j]);
                dxp[i] += R[j * 3 + i] * dx[j];
                wp[i] += R[j * 3 + i] * w[j];
            }
        }
        eig_quat_to_mat(quat, ree);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double a = 0;
                for (int l = 0; l < 3; l++) a += R[l * 3 + i] * ree[l * 3 + j];
                mm_[i * 3 + j] = a;
            }
        eig_mat_to_quat(mm_, quat);
        for (int blk = 0; blk < 2; blk++)
            for (int i = 0; i < 3; i++)
                for (int c = 0; c < dof; c++) {
                    double a = 0;
                    for (int l = 0; l < 3; l++) a += R[l * 3 + i] * Jac[(3 * blk + l) * dof + c];
                    Jn[(3 * blk + i) * dof + c] = a;
                }
        memcpy(Jac, Jn, sizeof(double) * 6 * dof);
        memcpy(p, pp, sizeof(pp)); memcpy(dx, dxp, sizeof(dxp)); memcpy(w, wp, sizeof(wp));
    }
    if (fx) {
        memset(fx, 0, sizeof(double) * s->n_f);
        memcpy(fx, p, 3 * sizeof(double));
        memcpy(fx + 3, quat, 4 * sizeof(double));
        if (nd == 2) {
            double H[12];
            orc_sd_H(quat, H);
            memcpy(fx + 7, dx, 3 * sizeof(double));
            for (int i = 0; i < 4; i++) /* SimulationInterface.cpp:69-73: .5 * H(quat)^T w */
                fx[10 + i] = .5 * (H[i] * w[0] + H[4 + i] * w[1] + H[8 + i] * w[2]);
        }
        if (tm) fx[s->n_f - 1] = x[s->n_x - 1];
    }
    if (J) {
        int nx = s->n_x;
        memset(J, 0, sizeof(double) * s->n_Q * nx);
        for (int r = 0; r < 6; r++)
            for (int c = 0; c < dof; c++) {
                J[r * nx + c] = Jac[r * dof + c];
                if (nd == 2) J[(6 + r) * nx + dof + c] = Jac[r * dof + c];
            }
        if (tm) J[(s->n_Q - 1) * nx + nx - 1] = 1;
    }
}
void orc_get_fx_jac(const orc_system* s, const double* x, double* fx, double* J) { fx_jac_frame(s, NULL, x, fx, J); }

/* PosOrnKeypoint::diff (PosOrnKeypoint.cpp:24-45), SpacetimeKeypoint::diff (SpacetimeKeypoint.cpp:19-25) */
void orc_kp_diff(const orc_system* s, const orc_keypoint* kp, const double* fx, double* e) {
    int nd = s->nb_deriv, tm = ORC_IS_TM(s->kind);
    if (ORC_IS_JOINT(s->kind) || kp->joint) { /* AngularKeypoint::diff (AngularKeypoint.cpp:24-27), AngularTimeKeypoint::diff (:22-27): target - state, t* - t last */
        const int n = kp->joint ? s->n_x : s->n_Q;
        for (int i = 0; i < n; i++) e[i] = kp->jt[i] - fx[i];
        return;
    }
    int nst = 7 * nd;
    memset(e, 0, sizeof(double) * s->n_Q);
    if (!is_zero(fx, nst)) { /* :29 */
        double lm[4], H[12];
        orc_sd_H(kp->orn, H);
        for (int i = 0; i < 3; i++) e[i] = kp->pos[i] - fx[i];
        orc_sd_logmap(kp->orn, fx + 3, lm);
        for (int i = 0; i < 3; i++) e[3 + i] = -2 * dot(H + 4 * i, lm, 4);
        if (nd == 2) {
            double tr[4], dv[4];
            for (int i = 0; i < 3; i++) e[6 + i] = kp->dpos[i] - fx[7 + i];
            orc_sd_transport(fx + 10, fx + 3, kp->orn, tr);
            for (int i = 0; i < 4; i++) dv[i] = kp->dorn[i] - tr[i];
            for (int i = 0; i < 3; i++) e[9 + i] = -2 * dot(H + 4 * i, dv, 4);
        }
    }
    if (tm) e[s->n_Q - 1] = kp->ctime - fx[s->n_f - 1];
    if (kp->dist) { /* PosOrnKeypointDistFunct::diff, PosOrnKeypointDistFunct.cpp:13-35 (no trace exercises it: pinned by reading only) */
        double n = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        if (n <= kp->pos_radius) { /* :18-19 */
            e[0] = e[1] = e[2] = 0;
        } else { /* :21 normalized() * (norm - radius) */
            double f = n - kp->pos_radius;
            for (int i = 0; i < 3; i++) e[i] = e[i] / n * f;
        }
        for (int i = 0; i < 3; i++) { /* :25-32; `abs` taken as the floating-point overload */
            double v = e[3 + i];
            if (fabs(v) <= kp->orn_thresh[i]) e[3 + i] = 0;
            else e[3 + i] = v - (v < 0 ? -1 : 1) * kp->orn_thresh[i];
        }
    }
}

/* residual size of a keypoint, and the largest one of the system (row stride of per-step J / e buffers) */
static int kp_nq(const orc_system* s, const orc_keypoint* kp) { return (kp && kp->joint) ? s->n_x : s->n_Q; }
static int nq_stride(const orc_system* s) {
    int n = s->n_Q;
    for (int i = 0; i < s->n_kp; i++)
        if (s->kp[i].joint && s->n_x > n) n = s->n_x;
    return n;
}

static const orc_keypoint* find_kp(const orc_system* s, int k) { /* System.cpp:96-101; later duplicates win in the map */
    const orc_keypoint* r = NULL;
    for (int i = 0; i < s->n_kp; i++)
        if (s->kp[i].timestep == k) r = &s->kp[i];
    return r;
}

/* inspectJointLimit, System.cpp:121-142: returns diag(L) and q */
static void limits(const orc_system* s, const double* x, double* Ld, double* q) {
    for (int i = 0; i < s->n_x; i++) { Ld[i] = 0; q[i] = 0; }
    if (!s->limits_set) return;
    for (int i = 0; i < s->n_x; i++) {
        if (s->limit_weight[i] != 0) {
            if (x[i] > s->state_max[i]) { q[i] = s->state_max[i] - x[i]; Ld[i] = s->penalty; }
            else if (x[i] < s->state_min[i]) { q[i] = s->state_min[i] - x[i]; Ld[i] = s->penalty; }
        }
    }
}

/* the second group of sub-systems of a sequence: same inspectJointLimit on its own bounds (penalty 1 like every limited System) */
static void limits2(const orc_system* s, const double* x, double* Ld, double* q) {
    for (int i = 0; i < s->n_x; i++) { Ld[i] = 0; q[i] = 0; }
    if (!s->limits2_set) return;
    for (int i = 0; i < s->n_x; i++) {
        if (s->limit_weight2[i] != 0) {
            if (x[i] > s->state_max2[i]) { q[i] = s->state_max2[i] - x[i]; Ld[i] = s->penalty; }
            else if (x[i] < s->state_min2[i]) { q[i] = s->state_min2[i] - x[i]; Ld[i] = s->penalty; }
        }
    }
}

/* System::cost, System.cpp:213-234 (control cost only at keypoint steps: quirk D-2) */
double orc_cost(const orc_system* s, const double* x, const double* u, int k) {
    double c = 0;
    const orc_keypoint* kp = find_kp(s, k);
    if (kp) {
        double fx[ORC_MAX_NF], e[ORC_MAX_NQ], Qe[ORC_MAX_NQ];
        fx_jac_frame(s, kp, x, fx, NULL);
        orc_kp_diff(s, kp, fx, e);
        int nq = kp_nq(s, kp);
        for (int i = 0; i < nq; i++) Qe[i] = dot(kp->Q + i * nq, e, nq);
        double ru = 0;
        const double* Rk = kp->has_Ru ? kp->Ru : s->R_diag;
        for (int i = 0; i < s->n_u; i++) ru += u[i] * Rk[i] * u[i];
        c += dot(e, Qe, nq) + ru;
    }
    if (s->limits_set) {
        double Ld[ORC_MAX_NX], q[ORC_MAX_NX], a = 0;
        limits(s, x, Ld, q);
        for (int i = 0; i < s->n_x; i++) a += q[i] * Ld[i] * q[i];
        c += a * (s->lim_mult > 1 ? s->lim_mult : 1);
    }
    if (s->limits2_set) {
        double Ld[ORC_MAX_NX], q[ORC_MAX_NX], a = 0;
        limits2(s, x, Ld, q);
        for (int i = 0; i < s->n_x; i++) a += q[i] * Ld[i] * q[i];
        c += a * (s->lim2_mult > 1 ? s->lim2_mult : 1);
    }
    return c;
}

/* System::cost_x, System.cpp:248-272:  -J^T Q e - L^T q */
void orc_cost_x(const orc_system* s, const double* x, int k, double* lx) {
    int nx = s->n_x;
    memset(lx, 0, sizeof(double) * nx);
    const orc_keypoint* kp = find_kp(s, k);
    const int nq = kp_nq(s, kp);
    if (kp) {
        double fx[ORC_MAX_NF], J[ORC_MAX_NQ * ORC_MAX_NX], e[ORC_MAX_NQ], Qe[ORC_MAX_NQ];
        fx_jac_frame(s, kp, x, fx, J);
        orc_kp_diff(s, kp, fx, e);
        for (int i = 0; i < nq; i++) Qe[i] = dot(kp->Q + i * nq, e, nq);
        for (int c = 0; c < nx; c++) {
            double a = 0;
            for (int r = 0; r < nq; r++) a += J[r * nx + c] * Qe[r];
            lx[c] += -1 * a;
        }
    }
    if (s->limits_set) {
        double Ld[ORC_MAX_NX], q[ORC_MAX_NX];
        limits(s, x, Ld, q);
        for (int i = 0; i < nx; i++) lx[i] += -Ld[i] * q[i] * (s->lim_mult > 1 ? s->lim_mult : 1);
    }
    if (s->limits2_set) {
        double Ld[ORC_MAX_NX], q[ORC_MAX_NX];
        limits2(s, x, Ld, q);
        for (int i = 0; i < nx; i++) lx[i] += -Ld[i] * q[i] * (s->lim2_mult > 1 ? s->lim2_mult : 1);
    }
}

/* System::cost_xx, System.cpp:286-308:  J^T Q J + L^T L */
void orc_cost_xx(const orc_system* s, const double* x, int k, double* lxx) {
    int nx = s->n_x;
    memset(lxx, 0, sizeof(double) * nx * nx);
    const orc_keypoint* kp = find_kp(s, k);
    const int nq = kp_nq(s, kp);
    if (kp) {
        double J[ORC_MAX_NQ * ORC_MAX_NX], JtQ[ORC_MAX_NX * ORC_MAX_NQ], JtQJ[ORC_MAX_NX * ORC_MAX_NX];
        fx_jac_frame(s, kp, x, NULL, J);
        mtm(JtQ, J, kp->Q, nq, nx, nq);
        mm(JtQJ, JtQ, J, nx, nq, nx);
        for (int i = 0; i < nx * nx; i++) lxx[i] += JtQJ[i];
    }
    if (s->limits_set) {
        double Ld[ORC_MAX_NX], q[ORC_MAX_NX];
        limits(s, x, Ld, q);
        for (int i = 0; i < nx; i++) lxx[i * nx + i] += Ld[i] * Ld[i] * (s->lim_mult > 1 ? s->lim_mult : 1);
    }
    if (s->limits2_set) {
        double Ld[ORC_MAX_NX], q[ORC_MAX_NX];
        limits2(s, x, Ld, q);
        for (int i = 0; i < nx; i++) lxx[i * nx + i] += Ld[i] * Ld[i] * (s->lim2_mult > 1 ? s->lim2_mult : 1);
    }
}

/* forwardPass: PosOrnPlannerSys.cpp:114-138, PosOrnTimePlannerSys.cpp:149-184, with the integrator of
 * SimulationInterface.cpp:19-31.  Functional form of the stateful simulator (the solvers always call it
 * as reset(); step; step; ... so x is the simulator's own state: quirk D-7). */
void orc_step(const orc_system* s, const double* x, const double* u,
              double* x_next, double* fx_next, double* A, double* B, double* J) {
    int dof = s->dof, nd = s->nb_deriv, nx = s->n_x, nu = s->n_u, tm = ORC_IS_TM(s->kind);
    double xn[ORC_MAX_NX];
    double dtSqrt = tm ? u[nu - 1] : 0.0;
    double dt = tm ? dtSqrt * dtSqrt : s->dt;
    if (nd == 1) { /* sendVel: dq = u; q += dt*dq + dt*dt/2*0 */
        for (int i = 0; i < dof; i++) xn[i] = x[i] + (dt * u[i] + dt * dt / 2 * 0.0);
    } else { /* sendAcc */
        for (int i = 0; i < dof; i++) {
            xn[i] = x[i] + (dt * x[dof + i] + dt * dt / 2 * u[i]);
            xn[dof + i] = x[dof + i] + dt * u[i];
        }
    }
    if (tm) xn[nx - 1] = x[nx - 1] + dt;
    if (A) {
        memset(A, 0, sizeof(double) * nx * nx);
        for (int i = 0; i < nx; i++) A[i * nx + i] = 1;
        if (nd == 2)
            for (int i = 0; i < dof; i++) A[i * nx + dof + i] = dt;
    }
    if (B) {
        memset(B, 0, sizeof(double) * nx * nu);
        if (nd == 1) {
            for (int i = 0; i < dof; i++) B[i * nu + i] = dt;
            if (tm) {
                for (int i = 0; i < dof; i++) B[i * nu + nu - 1] = 2 * dtSqrt * u[i];
                B[(nx - 1) * nu + nu - 1] = 2 * dtSqrt;
            }
        } else {
            for (int i = 0; i < dof; i++) {
                B[i * nu + i] = dt * dt / 2;
                B[(dof + i) * nu + i] = dt;
            }
            if (tm) { /* PosOrnTimePlannerSys.cpp:176: uses the velocity AFTER the step */
                for (int i = 0; i < dof; i++) {
                    B[i * nu + nu - 1] = 2 * dtSqrt * xn[dof + i] + 2 * dtSqrt * dtSqrt * dtSqrt * u[i];
                    B[(dof + i) * nu + nu - 1] = 2 * dtSqrt * u[i];
                }
                B[(nx - 1) * nu + nu - 1] = 2 * dtSqrt;
            }
        }
    }
    if (fx_next || J) orc_get_fx_jac(s, xn, fx_next, J);
    if (x_next) memcpy(x_next, xn, sizeof(double) * nx);
}

static void init_state(const orc_system* s, double* x0) {
    int dof = s->dof;
    memset(x0, 0, sizeof(double) * s->n_x);
    memcpy(x0, s->q0, sizeof(double) * dof);
    if (s->nb_deriv == 2) memcpy(x0 + dof, s->dq0, sizeof(double) * dof);
}

/* probe only: smallest distance of a weighted state coordinate of a trajectory to one of its bounds (inspectJointLimit's tests) */
static double traj_limit_margin(const orc_system* s, const double* X) {
    double mg = INFINITY;
    if (!s->limits_set) return mg;
    for (int k = 0; k < s->T; k++)
        for (int i = 0; i < s->n_x; i++)
            if (s->limit_weight[i] != 0) {
                double xv = X[(size_t)k * s->n_x + i];
                double a1 = fabs(xv - s->state_max[i]), a2 = fabs(xv - s->state_min[i]);
                if (a1 < mg) mg = a1;
                if (a2 < mg) mg = a2;
            }
    if (s->limits2_set)
        for (int k = 0; k < s->T; k++)
            for (int i = 0; i < s->n_x; i++)
                if (s->limit_weight2[i] != 0) {
                    double xv = X[(size_t)k * s->n_x + i];
                    double a1 = fabs(xv - s->state_max2[i]), a2 = fabs(xv - s->state_min2[i]);
                    if (a1 < mg) mg = a1;
                    if (a2 < mg) mg = a2;
                }
    return mg;
}

/* ------------------------------------------------------------------ ILQRRecursive::solve (solver/ILQRRecursive.cpp:21-181)
 * and AL_ILQR::solve (solver/AL-ILQR.cpp:50-232) share this body; c == NULL -> plain recursive. */

static int solve_riccati(const orc_system* s, const orc_constraints* c, double* lambda,
                         const double* U0, int nb_iter, int lag_update_step, double penalty, double scaling,
                         int line_search, int early_stop,
                         double* Xo, double* fXo, double* Uo, double* Ko, double* dout, double* cost_out,
                         double* trace_cost, double* trace_alpha) {
    const int T = s->T, nx = s->n_x, nu = s->n_u, nf = s->n_f;
    const int m = c ? c->m : 0, ns = nx + nu;
    double* X = (double*)calloc((size_t)T * nx, 8);
    double* fX = (double*)calloc((size_t)T * nf, 8);
    double* U = (double*)calloc((size_t)(T - 1) * nu, 8);
    double* nX = (double*)calloc((size_t)T * nx, 8);
    double* nfX = (double*)calloc((size_t)T * nf, 8);
    double* nU = (double*)calloc((size_t)(T - 1) * nu, 8);
    double* As = (double*)calloc((size_t)(T - 1) * nx * nx, 8);
    double* Bs = (double*)calloc((size_t)(T - 1) * nx * nu, 8);
    double* Ks = (double*)calloc((size_t)(T - 1) * nu * nx, 8); /* indexed by timestep k */
    double* ds = (double*)calloc((size_t)(T - 1) * nu, 8);
    double* Is = m ? (double*)calloc((size_t)(T - 1) * m, 8) : NULL; /* penalty * diag(I_k) */
    double* Cs = m ? (double*)calloc((size_t)(T - 1) * m, 8) : NULL; /* g_k */
    double zero_u[ORC_MAX_NU] = {0};
    double alpha = 1;
    int it_done = 0;

    init_state(s, X);
    orc_get_fx_jac(s, X, fX, NULL);
    memcpy(U, U0, sizeof(double) * (T - 1) * nu);

#define CON_A(k) (c->A + (c->per_step ? (size_t)(k) * m * ns : 0))
#define CON_B(k) (c->b + (c->per_step ? (size_t)(k) * m : 0))
    /* AL_ILQR::constraints, AL-ILQR.cpp:21-44 (stored pre-multiplied by the current penalty, :72,190) */
#define CONSTRAINTS(k, xk, uk, pen_, lam_)                                                       \
    do {                                                                                         \
        double sk[ORC_MAX_NX + ORC_MAX_NU];                                                      \
        memcpy(sk, (xk), sizeof(double) * nx);                                                   \
        memcpy(sk + nx, (uk), sizeof(double) * nu);                                              \
        for (int r_ = 0; r_ < m; r_++) {                                                         \
            double g_ = dot(CON_A(k) + (size_t)r_ * ns, sk, ns) - CON_B(k)[r_];                  \
            double i_ = 1;                                                                       \
            if (g_ < 0 && (lam_)[(size_t)(k) * m + r_] == 0) i_ = 0;                              \
            if ((lam_)[(size_t)(k) * m + r_] == 0 && fabs(g_) < mask_margin) mask_margin = fabs(g_); /* probe only */ \
            Is[(size_t)(k) * m + r_] = (pen_) * i_;                                              \
            Cs[(size_t)(k) * m + r_] = g_;                                                       \
        }                                                                                        \
    } while (0)
    double mask_margin = INFINITY; /* probe: distance of the active-set test g < 0 (rows with lambda == 0) to its threshold */
    const int resume = (g_it0 > 0);

    /* initial rollout: ILQRRecursive.cpp:41-56 / AL-ILQR.cpp:68-85 */
    double cost0 = 0;
    for (int i = 0; i < T - 1; i++) {
        const double* xk = X + (size_t)i * nx;
        const double* uk = U + (size_t)i * nu;
        if (m) CONSTRAINTS(i, xk, uk, (resume ? g_init_penalty : penalty), ((resume && g_lambda_mask) ? g_lambda_mask : lambda));
        cost0 += orc_cost(s, xk, uk, i);
        orc_step(s, xk, uk, X + (size_t)(i + 1) * nx, fX + (size_t)(i + 1) * nf,
                 As + (size_t)i * nx * nx, Bs + (size_t)i * nx * nu, NULL);
        if (g_resume_x) { /* test aid: the caller's own rollout of these controls IS the incoming trajectory (its active-set and limit
                             tests were taken on it); it must be this rollout up to rounding -- the deviation is reported */
            double* xn_ = X + (size_t)(i + 1) * nx;
            const double* xg_ = g_resume_x + (size_t)(i + 1) * nx;
            for (int j_ = 0; j_ < nx; j_++) {
                double dv_ = fabs(xg_[j_] - xn_[j_]) / fmax(1.0, fabs(xn_[j_]));
                if (!(dv_ <= g_resume_x_dev)) g_resume_x_dev = dv_; /* a NaN sticks */
                xn_[j_] = xg_[j_];
            }
            orc_get_fx_jac(s, xn_, fX + (size_t)(i + 1) * nf, NULL);
        }
    }
    cost0 += orc_cost(s, X + (size_t)(T - 1) * nx, zero_u, T - 1);

    for (int it = 0; it < nb_iter; it++) {
        /* ---------------- backward pass: ILQRRecursive.cpp:68-97 / AL-ILQR.cpp:94-145 */
        double P[ORC_MAX_NX * ORC_MAX_NX], p[ORC_MAX_NX];
        orc_cost_xx(s, X + (size_t)(T - 1) * nx, T - 1, P);
        orc_cost_x(s, X + (size_t)(T - 1) * nx, T - 1, p);
        for (int k = T - 2; k >= 0; k--) {
            const double* xk = X + (size_t)k * nx;
            const double* uk = U + (size_t)k * nu;
            const double* A = As + (size_t)k * nx * nx;
            const double* B = Bs + (size_t)k * nx * nu;
            double BtP[ORC_MAX_NU * ORC_MAX_NX], AtP[ORC_MAX_NX * ORC_MAX_NX];
            double Qux[ORC_MAX_NU * ORC_MAX_NX], Quu[ORC_MAX_NU * ORC_MAX_NU], Qxx[ORC_MAX_NX * ORC_MAX_NX];
            double Qxu[ORC_MAX_NX * ORC_MAX_NU], Qu[ORC_MAX_NU], Qx[ORC_MAX_NX];
            double lxx[ORC_MAX_NX * ORC_MAX_NX], lx[ORC_MAX_NX], tmp[ORC_MAX_NX * ORC_MAX_NX];
            mtm(BtP, B, P, nx, nu, nx);
            mtm(AtP, A, P, nx, nx, nx);
            mm(Qux, BtP, A, nu, nx, nx);                 /* cost_ux = 0 */
            mm(Quu, BtP, B, nu, nx, nu);
            for (int i = 0; i < nu; i++) Quu[i * nu + i] = s->R_diag[i] + Quu[i * nu + i];
            orc_cost_xx(s, xk, k, lxx);
            mm(tmp, AtP, A, nx, nx, nx);
            for (int i = 0; i < nx * nx; i++) Qxx[i] = lxx[i] + tmp[i];
            mm(Qxu, AtP, B, nx, nx, nu);                 /* cost_xu = 0 */
            if (g_variant & 1) /* test aid: Qxu := Qux^T, identical in exact arithmetic (P symmetric) */
                for (int i = 0; i < nx; i++) for (int j = 0; j < nu; j++) Qxu[i * nu + j] = Qux[j * nx + i];
            mtm(Qu, B, p, nx, nu, 1);
            for (int i = 0; i < nu; i++) Qu[i] = s->R_diag[i] * uk[i] + Qu[i]; /* cost_u = R u for ALL k */
            orc_cost_x(s, xk, k, lx);
            mtm(Qx, A, p, nx, nx, 1);
            for (int i = 0; i < nx; i++) Qx[i] = lx[i] + Qx[i];
            if (m) { /* AL-ILQR.cpp:110-134 */
                const double* Ak = CON_A(k);
                const double* Ik = Is + (size_t)k * m;
                const double* ck = Cs + (size_t)k * m;
                const double* lam = lambda + (size_t)k * m;
                for (int r = 0; r < m; r++) {
                    const double* ax = Ak + (size_t)r * ns;
                    const double* au = ax + nx;
                    double wv = lam[r] + Ik[r] * ck[r];
                    for (int i = 0; i < nu; i++) {
                        for (int j = 0; j < nx; j++) Qux[i * nx + j] += au[i] * Ik[r] * ax[j];
                        for (int j = 0; j < nu; j++) Quu[i * nu + j] += au[i] * Ik[r] * au[j];
                        Qu[i] += au[i] * wv;
                    }
                    for (int i = 0; i < nx; i++) {
                        for (int j = 0; j < nx; j++) Qxx[i * nx + j] += ax[i] * Ik[r] * ax[j];
                        for (int j = 0; j < nu; j++) Qxu[i * nu + j] += ax[i] * Ik[r] * au[j];
                        Qx[i] += ax[i] * wv;
                    }
                }
            }
            /* Quu_inv = -(Quu + 1e-6 I)^-1 ; regularisation only inside the inverse (quirk D-3) */
            double Qr[ORC_MAX_NU * ORC_MAX_NU], Qi[ORC_MAX_NU * ORC_MAX_NU];
            memcpy(Qr, Quu, sizeof(double) * nu * nu);
            for (int i = 0; i < nu; i++) Qr[i * nu + i] += 1e-6;
            orc_inverse(nu, Qr, Qi);
            for (int i = 0; i < nu * nu; i++) Qi[i] = -1 * Qi[i];
            double* Kk = Ks + (size_t)k * nu * nx;
            double* dk = ds + (size_t)k * nu;
            mm(Kk, Qi, Qux, nu, nu, nx);
            mm(dk, Qi, Qu, nu, nu, 1);
            /* P = Qxx + K'QuuK + K'Qux + QxuK ; p = Qx + K'Quu d + K'Qu + Qxu d */
            double KtQuu[ORC_MAX_NX * ORC_MAX_NU], t1[ORC_MAX_NX * ORC_MAX_NX], t2[ORC_MAX_NX * ORC_MAX_NX], t3[ORC_MAX_NX * ORC_MAX_NX];
            double v1[ORC_MAX_NX], v2[ORC_MAX_NX], v3[ORC_MAX_NX];
            mtm(KtQuu, Kk, Quu, nu, nx, nu);
            mm(t1, KtQuu, Kk, nx, nu, nx);
            mtm(t2, Kk, Qux, nu, nx, nx);
            mm(t3, Qxu, Kk, nx, nu, nx);
            for (int i = 0; i < nx * nx; i++) P[i] = ((Qxx[i] + t1[i]) + t2[i]) + t3[i];
            mm(v1, KtQuu, dk, nx, nu, 1);
            mtm(v2, Kk, Qu, nu, nx, 1);
            mm(v3, Qxu, dk, nx, nu, 1);
            for (int i = 0; i < nx; i++) p[i] = ((Qx[i] + v1[i]) + v2[i]) + v3[i];
        }

        /* ---------------- forward pass with step-halving line search: ILQRRecursive.cpp:101-155 */
        double newCost = 0, dun = 0;
        orc_probe_rec* pr = (g_probe && it < g_probe_cap) ? &g_probe[it] : NULL;
        if (pr) {
            memset(pr, 0, sizeof(*pr));
            pr->cost0 = cost0; pr->clamp_margin = INFINITY; pr->limit_margin = INFINITY; pr->mask_margin_in = mask_margin;
            pr->limit_margin_in = traj_limit_margin(s, X); /* l_xx of the sweep jumps where a coordinate of the incoming trajectory sits on a bound */
        }
        /* the do/while of ILQRRecursive.cpp:101-155.  With the probe's "all trials" aid the loop goes on below the accepted step size
           for the record only: the accepted rollout is put aside and restored (nothing of the extra trials survives). */
        int accepted = 0;
        double alpha_t = 2, acc_alpha = 1, acc_cost = 0, acc_dun = 0, acc_mask = INFINITY;
        double *sv_X = NULL, *sv_fX = NULL, *sv_U = NULL, *sv_A = NULL, *sv_B = NULL, *sv_I = NULL, *sv_C = NULL;
        do {
            alpha_t /= 2.0;
            const double alpha = alpha_t; /* the trial's step size (shadows the solve's: set on acceptance) */
            mask_margin = INFINITY;
            init_state(s, nX);
            orc_get_fx_jac(s, nX, nfX, NULL);
            dun = 0;
            newCost = 0;
            for (int k = 0; k < T - 1; k++) {
                const double* Kk = Ks + (size_t)k * nu * nx;
                const double* dk = ds + (size_t)k * nu;
                double dxv[ORC_MAX_NX], du[ORC_MAX_NU];
                double* nuk = nU + (size_t)k * nu;
                const double* nxk = nX + (size_t)k * nx;
                for (int i = 0; i < nx; i++) dxv[i] = nxk[i] - X[(size_t)k * nx + i];
                for (int i = 0; i < nu; i++) du[i] = dot(Kk + i * nx, dxv, nx) + alpha * dk[i];
                dun += norm(du, nu); /* quirk D-5: accumulates ||du||, not ||du||^2 */
                for (int i = 0; i < nu; i++) nuk[i] = U[(size_t)k * nu + i] + du[i];
                orc_step(s, nxk, nuk, nX + (size_t)(k + 1) * nx, nfX + (size_t)(k + 1) * nf,
                         As + (size_t)k * nx * nx, Bs + (size_t)k * nx * nu, NULL); /* last trial's A,B survive (D-7) */
                if (m) CONSTRAINTS(k, nxk, nuk, penalty, lambda);
                newCost += orc_cost(s, nxk, nuk, k);
            }
            newCost += orc_cost(s, nX + (size_t)(T - 1) * nx, zero_u, T - 1);
            if (pr && pr->n_trials < ORC_MAX_TRIALS) { pr->trial_alpha[pr->n_trials] = alpha; pr->trial_cost[pr->n_trials] = newCost; pr->n_trials++; }
            if (!accepted && !(((newCost >= cost0) || isnan(newCost)) && alpha > 1e-3 && line_search)) {
                accepted = 1;
                acc_alpha = alpha; acc_cost = newCost; acc_dun = dun; acc_mask = mask_margin;
                if (pr && g_probe_all && alpha > 1e-3 && line_search) { /* put the accepted rollout aside */
#define SAVE_(dst, src, n) do { dst = (double*)malloc(sizeof(double) * (n)); memcpy(dst, src, sizeof(double) * (n)); } while (0)
                    SAVE_(sv_X, nX, (size_t)T * nx); SAVE_(sv_fX, nfX, (size_t)T * nf); SAVE_(sv_U, nU, (size_t)(T - 1) * nu);
                    SAVE_(sv_A, As, (size_t)(T - 1) * nx * nx); SAVE_(sv_B, Bs, (size_t)(T - 1) * nx * nu);
                    if (m) { SAVE_(sv_I, Is, (size_t)(T - 1) * m); SAVE_(sv_C, Cs, (size_t)(T - 1) * m); }
#undef SAVE_
                }
            }
        } while (!accepted || (sv_X && alpha_t > 1e-3));
        alpha = acc_alpha;
        if (sv_X) { /* back to the accepted rollout; alpha_t ran on */
            memcpy(nX, sv_X, sizeof(double) * T * nx); memcpy(nfX, sv_fX, sizeof(double) * T * nf); memcpy(nU, sv_U, sizeof(double) * (T - 1) * nu);
            memcpy(As, sv_A, sizeof(double) * (T - 1) * nx * nx); memcpy(Bs, sv_B, sizeof(double) * (T - 1) * nx * nu);
            if (m) { memcpy(Is, sv_I, sizeof(double) * (T - 1) * m); memcpy(Cs, sv_C, sizeof(double) * (T - 1) * m); }
            free(sv_X); free(sv_fX); free(sv_U); free(sv_A); free(sv_B); free(sv_I); free(sv_C);
        }
        newCost = acc_cost; dun = acc_dun; mask_margin = acc_mask;
        if (pr) { /* probe: margins of the accepted rollout */
            pr->dun = dun;
            pr->mask_margin = mask_margin;
            pr->limit_margin = traj_limit_margin(s, nX);
        }

        /* multiplier update: AL-ILQR.cpp:202-208 (uses the UPDATED penalty) */
        if (m && ((g_it0 + it + 1) % lag_update_step == 0)) {
            penalty *= scaling;
            for (size_t i = 0; i < (size_t)(T - 1) * m; i++) {
                double v = lambda[i] + penalty * Cs[i];
                if (pr && fabs(v) < pr->clamp_margin) pr->clamp_margin = fabs(v);
                lambda[i] = v > 0 ? v : 0; /* cwiseMax(0) */
            }
        }

        /* accept unconditionally (quirk D-4) */
        cost0 = newCost;
        memcpy(X, nX, sizeof(double) * T * nx);
        memcpy(fX, nfX, sizeof(double) * T * nf);
        memcpy(U, nU, sizeof(double) * (T - 1) * nu);
        if (trace_cost) trace_cost[it] = cost0;
        if (trace_alpha) trace_alpha[it] = alpha;
        it_done = it + 1;
        if (c) {
            if (early_stop && alpha * sqrt(dun) < 1e-3) break; /* AL-ILQR.cpp:225 */
        } else {
            if (early_stop && alpha * sqrt(dun) < 1e-3 && cost0 < 1e-3) break; /* ILQRRecursive.cpp:174 */
        }
    }
#undef CONSTRAINTS
#undef CON_A
#undef CON_B

    if (Xo) memcpy(Xo, X, sizeof(double) * T * nx);
    if (fXo) memcpy(fXo, fX, sizeof(double) * T * nf);
    if (Uo) memcpy(Uo, U, sizeof(double) * (T - 1) * nu);
    if (Ko) memcpy(Ko, Ks, sizeof(double) * (T - 1) * nu * nx);
    if (dout) /* returned ds are scaled by the accepted alpha (ILQRRecursive.cpp:128,144,162) */
        for (size_t i = 0; i < (size_t)(T - 1) * nu; i++) dout[i] = (it_done > 0 ? alpha : 1.0) * ds[i];
    if (cost_out) *cost_out = cost0;
    free(X); free(fX); free(U); free(nX); free(nfX); free(nU); free(As); free(Bs); free(Ks); free(ds);
    free(Is); free(Cs);
    return it_done;
}

int orc_solve_recursive(const orc_system* s, const double* U0, int nb_iter, int line_search, int early_stop,
                        double* X, double* fX, double* U, double* K, double* d, double* cost,
                        double* trace_cost, double* trace_alpha) {
    return solve_riccati(s, NULL, NULL, U0, nb_iter, 1, 0.0, 1.0, line_search, early_stop, X, fX, U, K, d, cost,
                         trace_cost, trace_alpha);
}

int orc_solve_al(const orc_system* s, const orc_constraints* c, double* lambda, const double* U0, int nb_iter,
                 int lag_update_step, double penalty, double scaling, int line_search, int early_stop,
                 double* X, double* fX, double* U, double* cost, double* trace_cost, double* trace_alpha) {
    return solve_riccati(s, c, lambda, U0, nb_iter, lag_update_step, penalty, scaling, line_search, early_stop,
                         X, fX, U, NULL, NULL, cost, trace_cost, trace_alpha);
}

/* ------------------------------------------------------------------ BatchILQRCP (solver/BatchILQRCP.cpp) */

typedef struct {
    double *fX, *qL, *A, *B, *J, *L; /* per step: fX[T][nf], qL[T][nx], A[T][nx*nx], B[T][nx*nu], J[T][nq*nx], L[T][nx] (diag) */
} fp_batch;

/* System::fpBatch, System.cpp:181-211 with forwardPassWithLimits :144-161 (limits on the PRE-step state) */
static void fp_batch_run(const orc_system* s, const double* u, fp_batch* f) {
    const int T = s->T, nx = s->n_x, nu = s->n_u, nf = s->n_f, nq = nq_stride(s);
    double x[ORC_MAX_NX], xn[ORC_MAX_NX];
    init_state(s, x);
    memset(f->qL, 0, sizeof(double) * T * nx);
    memset(f->L, 0, sizeof(double) * T * nx);
    fx_jac_frame(s, find_kp(s, 0), x, f->fX, f->J);
    memset(f->A, 0, sizeof(double) * nx * nx);
    for (int i = 0; i < nx; i++) f->A[i * nx + i] = 1;
    memset(f->B, 0, sizeof(double) * nx * nu);
    for (int i = 0; i < T - 1; i++) {
        orc_step(s, x, u + (size_t)i * nu, xn, f->fX + (size_t)(i + 1) * nf, f->A + (size_t)(i + 1) * nx * nx,
                 f->B + (size_t)(i + 1) * nx * nu, f->J + (size_t)(i + 1) * nq * nx);
        /* a SequentialSystem does not override fpBatch / forwardPassWithLimits (SequentialSystem.h:31-41): they run on the sequence
         * object itself, which is built by the constructor WITHOUT limits (SequentialSystem.cpp:12-18: limits_set_ = false), so the batch
         * solvers see no limit terms at all -- unlike ILQRRecursive, whose cost, cost_x, cost_xx are the sums over the sub-systems */
        if (!(s->lim_mult > 1 || s->sequence)) limits(s, x, f->L + (size_t)(i + 1) * nx, f->qL + (size_t)(i + 1) * nx);
        {   /* a keypoint whose system works in an object frame (TransformedSimulationInterface): f(x) and J in that frame */
            const orc_keypoint* kf = find_kp(s, i + 1);
            if (kf && (kf->has_frame || kf->joint)) fx_jac_frame(s, kf, xn, f->fX + (size_t)(i + 1) * nf, f->J + (size_t)(i + 1) * nq * nx);
        }
        memcpy(x, xn, sizeof(x));
    }
}

static double cp_cost(const orc_system* s, const fp_batch* f, const double* u, double* e_out) {
    /* e'Qe + u'Ru + ql'L ql over keypoint rows: BatchILQRCP.cpp:135,150 */
    const int T = s->T, nx = s->n_x, nu = s->n_u, nf = s->n_f, nqs = nq_stride(s);
    double c_e = 0, c_u = 0, c_l = 0;
    for (int t = 0; t < s->n_kp; t++) {
        const orc_keypoint* kp = &s->kp[t];
        int ts = kp->timestep;
        const int nq = kp_nq(s, kp);
        double e[ORC_MAX_NQ], Qe[ORC_MAX_NQ];
        orc_kp_diff(s, find_kp(s, ts), f->fX + (size_t)ts * nf, e); /* System::diff looks the keypoint up in the map */
        for (int i = 0; i < nq; i++) Qe[i] = dot(kp->Q + i * nq, e, nq);
        c_e += dot(e, Qe, nq);
        if (e_out) memcpy(e_out + (size_t)t * nqs, e, sizeof(double) * nq);
        for (int i = 0; i < nx; i++) c_l += f->qL[(size_t)ts * nx + i] * f->L[(size_t)ts * nx + i] * f->qL[(size_t)ts * nx + i];
    }
    for (int k = 0; k < T - 1; k++)
        for (int i = 0; i < nu; i++) c_u += u[(size_t)k * nu + i] * s->R_diag[i] * u[(size_t)k * nu + i];
    return c_e + c_u + c_l;
}

int orc_solve_batch_cp(const orc_system* s, const double* psi, int Kw, double* u, int nb_iter, int early_stop,
                       double* trace_cost, double* trace_alpha) {
    const int T = s->T, nx = s->n_x, nu = s->n_u, nf = s->n_f, nq = nq_stride(s), nkp = s->n_kp;
    const int NU = (T - 1) * nu;
    fp_batch f, ft;
    fp_batch* fs[2] = {&f, &ft};
    for (int i = 0; i < 2; i++) {
        fs[i]->fX = (double*)calloc((size_t)T * nf, 8);
        fs[i]->qL = (double*)calloc((size_t)T * nx, 8);
        fs[i]->A = (double*)calloc((size_t)T * nx * nx, 8);
        fs[i]->B = (double*)calloc((size_t)T * nx * nu, 8);
        fs[i]->J = (double*)calloc((size_t)T * nq * nx, 8);
        fs[i]->L = (double*)calloc((size_t)T * nx, 8);
    }
    double* Su = (double*)calloc((size_t)nkp * nx * NU, 8);
    double* M = (double*)calloc((size_t)nx * (NU + nu), 8);
    double* Mn = (double*)calloc((size_t)nx * (NU + nu), 8);
    double* SP = (double*)calloc((size_t)nkp * nx * Kw, 8);
    double* H = (double*)calloc((size_t)Kw * Kw, 8);
    double* Hi = (double*)calloc((size_t)Kw * Kw, 8);
    double* g = (double*)calloc((size_t)Kw, 8);
    double* dw = (double*)calloc((size_t)Kw, 8);
    double* du = (double*)calloc((size_t)NU, 8);
    double* ut = (double*)calloc((size_t)NU, 8);
    double* e = (double*)calloc((size_t)nkp * nq, 8);
    int it_done = 0;

    for (int it = 0; it < nb_iter; it++) {
        fp_batch_run(s, u, &f);
        /* buildSuJL, BatchILQRCP.cpp:61-97: M seeded with B_0 = 0 and sampled BEFORE the step-i update (quirk D-1) */
        memset(Su, 0, sizeof(double) * nkp * nx * NU);
        int mc = nu; /* current number of columns of M */
        memset(M, 0, sizeof(double) * nx * (NU + nu));
        for (int i = 0; i < T; i++) {
            for (int t = 0; t < nkp; t++)
                if (s->kp[t].timestep == i && i > 0)
                    for (int r = 0; r < nx; r++) memcpy(Su + ((size_t)t * nx + r) * NU, M + (size_t)r * (NU + nu), sizeof(double) * mc);
            if (i > 0) {
                const double* At = f.A + (size_t)i * nx * nx;
                const double* Bt = f.B + (size_t)i * nx * nu;
                for (int r = 0; r < nx; r++) {
                    for (int cc = 0; cc < mc; cc++) {
                        double a = 0;
                        for (int l = 0; l < nx; l++) a += At[r * nx + l] * M[(size_t)l * (NU + nu) + cc];
                        Mn[(size_t)r * (NU + nu) + cc] = a;
                    }
                    for (int cc = 0; cc < nu; cc++) Mn[(size_t)r * (NU + nu) + mc + cc] = Bt[r * nu + cc];
                }
                mc += nu;
                double* t_ = M; M = Mn; Mn = t_;
            }
        }
        double cost0 = cp_cost(s, &f, u, e);
        /* H = Psi'Su'(J'QJ + L)Su Psi + Psi'R Psi ; g = Psi'Su'(J'Q e + L ql) - Psi'R u   (:129-130) */
        mm(SP, Su, psi, nkp * nx, NU, Kw);
        memset(H, 0, sizeof(double) * Kw * Kw);
        memset(g, 0, sizeof(double) * Kw);
        for (int t = 0; t < nkp; t++) {
            const orc_keypoint* kp = &s->kp[t];
            int ts = kp->timestep;
            const double* Jt = f.J + (size_t)ts * nq * nx;
            const int nqk = kp_nq(s, kp);
            double JtQ[ORC_MAX_NX * ORC_MAX_NQ], W[ORC_MAX_NX * ORC_MAX_NX], r[ORC_MAX_NX];
            mtm(JtQ, Jt, kp->Q, nqk, nx, nqk);
            mm(W, JtQ, Jt, nx, nqk, nx);
            mm(r, JtQ, e + (size_t)t * nq, nx, nqk, 1);
            for (int i = 0; i < nx; i++) {
                W[i * nx + i] += f.L[(size_t)ts * nx + i];
                r[i] += f.L[(size_t)ts * nx + i] * f.qL[(size_t)ts * nx + i];
            }
            const double* SPt = SP + (size_t)t * nx * Kw;
            double* WS = (double*)malloc(sizeof(double) * nx * Kw);
            mm(WS, W, SPt, nx, nx, Kw);
            for (int a = 0; a < Kw; a++) {
                for (int b = 0; b < Kw; b++) {
                    double acc = 0;
                    for (int l = 0; l < nx; l++) acc += SPt[l * Kw + a] * WS[l * Kw + b];
                    H[a * Kw + b] += acc;
                }
                double acc = 0;
                for (int l = 0; l < nx; l++) acc += SPt[l * Kw + a] * r[l];
                g[a] += acc;
            }
            free(WS);
        }
        for (int a = 0; a < Kw; a++) {
            for (int b = 0; b < Kw; b++) {
                double acc = 0;
                for (int k = 0; k < NU; k++) acc += psi[(size_t)k * Kw + a] * s->R_diag[k % nu] * psi[(size_t)k * Kw + b];
                H[a * Kw + b] += acc;
            }
            double acc = 0;
            for (int k = 0; k < NU; k++) acc += psi[(size_t)k * Kw + a] * s->R_diag[k % nu] * u[k];
            g[a] -= acc;
        }
        orc_inverse(Kw, H, Hi); /* explicit .inverse(), :131 */
        mm(dw, Hi, g, Kw, Kw, 1);
        mm(du, psi, dw, NU, Kw, 1);

        orc_probe_rec* pr = (g_probe && it < g_probe_cap) ? &g_probe[it] : NULL; /* test aid: what the backtracking decided on */
        if (pr) {
            memset(pr, 0, sizeof(*pr));
            pr->cost0 = cost0; pr->dun = norm(du, NU);
            pr->mask_margin_in = pr->mask_margin = pr->clamp_margin = pr->limit_margin = INFINITY;
            pr->limit_margin_in = INFINITY; /* (the limit tests of forwardPassWithLimits are not probed: a tie there is never excused) */
        }
        double alpha = 1.0, alpha_acc = 0;
        int accepted = 0;
        double* u_acc = NULL;
        while (1) { /* :138-158 */
            for (int k = 0; k < NU; k++) ut[k] = u[k] + alpha * du[k];
            fp_batch_run(s, ut, &ft);
            double cost = cp_cost(s, &ft, ut, NULL);
            if (pr && pr->n_trials < ORC_MAX_TRIALS) { pr->trial_alpha[pr->n_trials] = alpha; pr->trial_cost[pr->n_trials] = cost; pr->n_trials++; }
            if (!accepted && ((cost < cost0) || (alpha < 1e-3))) {
                accepted = 1;
                alpha_acc = alpha;
                if (pr && g_probe_all && !(alpha < 1e-3)) { /* go on for the record only; the accepted controls are put aside */
                    u_acc = (double*)malloc(sizeof(double) * NU);
                    memcpy(u_acc, ut, sizeof(double) * NU);
                } else {
                    memcpy(u, ut, sizeof(double) * NU);
                    break;
                }
            }
            if (accepted && alpha < 1e-3) break;
            alpha /= 2;
        }
        if (u_acc) { memcpy(u, u_acc, sizeof(double) * NU); free(u_acc); }
        alpha = alpha_acc;
        if (trace_cost) trace_cost[it] = cost0; /* printed cost is the PRE-step cost (:160) */
        if (trace_alpha) trace_alpha[it] = alpha;
        it_done = it + 1;
        if (early_stop && alpha * norm(du, NU) < 1e-3) break; /* :167 */
    }
    for (int i = 0; i < 2; i++) { free(fs[i]->fX); free(fs[i]->qL); free(fs[i]->A); free(fs[i]->B); free(fs[i]->J); free(fs[i]->L); }
    free(Su); free(M); free(Mn); free(SP); free(H); free(Hi); free(g); free(dw); free(du); free(ut); free(e);
    return it_done;
}

/* ------------------------------------------------------------------ primitives (utils/primitives.cpp) */

static int binom(int n, int k) { /* :13-17 */
    if (k == 0 || k == n) return 1;
    return binom(n - 1, k - 1) + binom(n - 1, k);
}

void orc_psi_rbf(int dim, int K, double* out) { /* :19-33 */
    double bw = ((double)dim) / K, avg = bw / 2, sig = bw;
    for (int i = 0; i < K; i++) {
        for (int j = 0; j < dim; j++) {
            double t = (double)j;
            out[j * K + i] = 1 / (2 * M_PI * sig) * exp(-1 * (t - avg) * (t - avg) / (2 * sig * sig));
        }
        avg += bw;
    }
}

void orc_psi_bernstein(int dim, int K, double* out) { /* :35-50 */
    int order = K - 1;
    for (int i = 0; i < K; i++) {
        int b = binom(order, i);
        for (int j = 0; j < dim; j++) {
            double t = ((double)j) / (double)(dim - 1);
            out[j * K + i] = b * pow(t, i) * pow(1 - t, order - i);
        }
    }
}

void orc_psi_unitstep(int dim, int K, double* out) { /* :52-68 */
    int bw = (int)round(((double)dim) / K);
    int lo = 0, hi = bw;
    for (int i = 0; i < K; i++) {
        for (int j = 0; j < dim; j++) out[j * K + i] = (j >= lo && j < hi) ? 1.0 / bw : 0;
        lo += bw;
        hi += bw;
    }
}

void orc_psi_sawtooth(int dim, int K, double* out) { /* :70-86 */
    int bw = (int)ceil(((double)dim) / K);
    double lo = 0, hi = bw;
    for (int i = 0; i < K; i++) {
        for (int j = 0; j < dim; j++) out[j * K + i] = (j >= lo && j < hi) ? ((j - lo) / (bw - 1) - 0.5) : 0;
        lo += bw;
        hi += bw;
    }
}

void orc_psi_linear(int dim, int K, double* out) { /* :88-94 */
    double* a = (double*)malloc(sizeof(double) * dim * K);
    double* b = (double*)malloc(sizeof(double) * dim * K);
    orc_psi_unitstep(dim, K, a);
    orc_psi_sawtooth(dim, K, b);
    for (int j = 0; j < dim; j++)
        for (int i = 0; i < K; i++) {
            out[j * 2 * K + i] = a[j * K + i];
            out[j * 2 * K + K + i] = b[j * K + i];
        }
    free(a);
    free(b);
}
