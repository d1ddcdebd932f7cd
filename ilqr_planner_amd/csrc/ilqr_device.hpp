// ilqr_device.hpp -- per-instance device math of the batched iLQR hot path (gfx950, fp64).
//
// Everything here is what ONE problem instance needs at ONE timestep: forward kinematics + geometric Jacobian of
// the chain, the S^3 log/transport maps, keypoint residuals, stage cost and its derivatives, and one dynamics
// step.  The kernels in ilqr_kernels.hip decide how instances are mapped onto lanes.
// Reference behaviour restated (paths relative to ilqr_planner/ilqr_planner in the reference tree):
//   FK/Jacobian      src/sim/KDLRobot.cpp:83-115 (orocos_kdl JntToJac/JntToCart/GetQuaternion)
//   integrator       src/sim/SimulationInterface.cpp:19-31
//   Sd utils         include/ilqr_planner/utils/sd.h:23-99
//   residuals        src/system/PosOrnKeypoint.cpp:24-45, src/system/SpacetimeKeypoint.cpp:19-25
//   cost + derivs    src/system/System.cpp:121-142,213-312
//   dynamics         src/system/PosOrnPlannerSys.cpp:80-138, src/system/PosOrnTimePlannerSys.cpp:85-184
#pragma once
#include <hip/hip_runtime.h>

namespace ilqr {

constexpr int DOF = 7;
constexpr int MAX_KP = 8;
constexpr int MAX_NQ = 13;
constexpr int MAX_NX = 15;
constexpr int MAX_NU = 8;

// Chain with consecutive fixed segments folded into the following joint's pre-transform (host side, csrc/ilqr_capi.cpp).
struct DevChain {
    double Rpre[DOF][9];   // fixed rotation applied before joint j's rotation (row-major)
    double ppre[DOF][3];   // translation (in the frame before Rpre) to joint j's origin
    double axis[DOF][3];   // joint axis in the joint frame
    double Rtail[9];       // everything after the last joint, incl. the user tool frame
    double ptail[3];
};

struct DevDesc {
    DevChain chain;
    int kind, nd, T, B, Bp;
    double dt;
    double R_diag[MAX_NU];
    int limits_set;
    double penalty;
    double smax[MAX_NX + 1], smin[MAX_NX + 1];
    int lw[MAX_NX + 1];
    int n_kp;
    int kp_t[MAX_KP];
    double reg, alpha_floor, stop_tol;
    double kp_Q[MAX_KP][MAX_NQ * MAX_NQ];  // leading dimension n_Q
    int kp_dist[MAX_KP];               // PosOrnKeypointDistFunct dead zones (0 = plain keypoint)
    double kp_pos_radius[MAX_KP], kp_orn_thresh[MAX_KP][3];
    int kp_frame[MAX_KP];              // keypoint seen through a TransformedSimulationInterface: frame [R | p]
    double kp_fR[MAX_KP][9], kp_fp[MAX_KP][3];
    int kp_has_Ru[MAX_KP];             // control penalty of the keypoint's own sub-system (SequentialSystem)
    double kp_Ru[MAX_KP][MAX_NU];
    int kp_joint[MAX_KP];              // Angular(Time)Keypoint of a joint-space sub-system inside a PosOrn(Time) system (hybrid SequentialSystem,
                                       // nb_deriv = 1): residual target - x, J = I, precision n_x x n_x (leading dimension n_x)
    int lim2;                          // second limit set (sub-systems of a sequence with other bounds); generic kernels only
    double smax2[MAX_NX + 1], smin2[MAX_NX + 1], penalty2, pen_xx2;  // penalty2 = penalty x multiplicity, pen_xx2 = penalty^2 x multiplicity
    int lw2[MAX_NX + 1];
    int batch_limits;                  // 0: the batch solvers see no limit terms (sequence of sub-systems: SequentialSystem does not override
                                       // fpBatch, and the sequence object itself has no limits); 1: plain system
    double pen_xx;                     // penalty^2 x limit multiplicity (l_xx of a violated limit); `penalty` holds penalty x multiplicity
};

template <int KIND_, int ND_>
struct Sys {
    // KIND 0 = PosOrnPlannerSys, 1 = PosOrnTimePlannerSys, 2 = JointSpacePlannerSys (target space = state space, J = I;
    // JointSpacePlannerSys.cpp:71-81, nb_deriv = 1 only: the reference's 2nd-order variant is dimensionally inconsistent)
    // KIND 3 = JointSpaceTimePlannerSys (joint space + time state, dt = u_last^2; nb_deriv = 1)
    static constexpr int KIND = KIND_, ND = ND_, TM = (KIND_ == 1 || KIND_ == 3) ? 1 : 0;
    static constexpr bool JOINT = (KIND_ == 2 || KIND_ == 3);
    static constexpr int NX = ND_ * DOF + TM;
    static constexpr int NU = DOF + TM;
    static constexpr int NF = JOINT ? NX : 7 * ND_ + TM;
    static constexpr int NQ = JOINT ? NX : NF - ND_;
};

#define ILQR_DEV __device__ __forceinline__
// empty statement the optimiser cannot see through (device code; tests/tools/hostsim builds the same source with g++, which has no "v" constraint)
#if defined(__HIP_DEVICE_COMPILE__)
#define ILQR_OPAQUE_VGPR(x) asm volatile("" : "+v"(x))
#else
#define ILQR_OPAQUE_VGPR(x) ((void)0)
#endif

// ------------------------------------------------------------------------------------------------ Sd (sd.h)
ILQR_DEV double dot4(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
ILQR_DEV bool is_zero4(const double* a) {  // Eigen isZero(1e-12)
    return fabs(a[0]) <= 1e-12 && fabs(a[1]) <= 1e-12 && fabs(a[2]) <= 1e-12 && fabs(a[3]) <= 1e-12;
}
// row r of H(q) = dQuatToDxJac (sd.h:23-27) dotted with v
ILQR_DEV void H_mul(const double* q, const double* v, double out[3]) {
    out[0] = -q[1] * v[0] + q[0] * v[1] - q[3] * v[2] + q[2] * v[3];
    out[1] = -q[2] * v[0] + q[3] * v[1] + q[0] * v[2] - q[1] * v[3];
    out[2] = -q[3] * v[0] - q[2] * v[1] + q[1] * v[2] + q[0] * v[3];
}
ILQR_DEV double sd_distance(const double* x, const double* y) {  // sd.h:48-62
    double d = dot4(x, y);
    if (d > 1) d = 1;
    else if (d < -1) d = -1;
    double ac = acos(d);
    if (d < 0) ac -= 3.14159265358979323846;
    return ac;
}
ILQR_DEV void sd_logmap(const double* base_in, const double* y_in, double out[4]) {  // sd.h:67-82
    out[0] = out[1] = out[2] = out[3] = 0;
    if (is_zero4(base_in) || is_zero4(y_in)) return;
    double nb = sqrt(dot4(base_in, base_in)), ny = sqrt(dot4(y_in, y_in));
    double b[4], y[4], t[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { b[i] = base_in[i] / nb; y[i] = y_in[i] / ny; }
    double by = dot4(b, y);
#pragma unroll
    for (int i = 0; i < 4; i++) t[i] = y[i] - by * b[i];
    double nt = sqrt(dot4(t, t));
    if (nt == 0) return;
    double d = sd_distance(b, y);
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = d * t[i] / nt;
}
ILQR_DEV void sd_transport(const double* v, const double* b1, const double* b2, double out[4]) {  // sd.h:87-99
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = v[i];
    if (is_zero4(b1) || is_zero4(b2)) return;
    double dist = sd_distance(b1, b2);
    double dsq = dist * dist;
    if (dsq == 0) return;
    double l12[4], l21[4];
    sd_logmap(b1, b2, l12);
    sd_logmap(b2, b1, l21);
    double f = dot4(l12, v) / dsq;
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = v[i] - f * (l12[i] + l21[i]);
}

// ------------------------------------------------------------------------------------------------ FK
ILQR_DEV void mat3_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// orocos_kdl Rotation::GetQuaternion, output (w,x,y,z) as KDLRobot.cpp:103 stores it
ILQR_DEV void kdl_quat(const double* R, double q[4]) {
    double tr = R[0] + R[4] + R[8];
    if (tr > 1e-12) {
        double s = 0.5 / sqrt(tr + 1.0);
        q[0] = 0.25 / s;
        q[1] = (R[7] - R[5]) * s;
        q[2] = (R[2] - R[6]) * s;
        q[3] = (R[3] - R[1]) * s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        double s = 2.0 * sqrt(1.0 + R[0] - R[4] - R[8]);
        q[0] = (R[7] - R[5]) / s;
        q[1] = 0.25 * s;
        q[2] = (R[1] + R[3]) / s;
        q[3] = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        double s = 2.0 * sqrt(1.0 + R[4] - R[0] - R[8]);
        q[0] = (R[2] - R[6]) / s;
        q[1] = (R[1] + R[3]) / s;
        q[2] = 0.25 * s;
        q[3] = (R[5] + R[7]) / s;
    } else {
        double s = 2.0 * sqrt(1.0 + R[8] - R[0] - R[4]);
        q[0] = (R[3] - R[1]) / s;
        q[1] = (R[2] + R[6]) / s;
        q[2] = (R[5] + R[7]) / s;
        q[3] = 0.25 * s;
    }
}

// p, quat and (optionally) the 6x7 geometric Jacobian, columns [z_j x (p - o_j); z_j] in the base frame.
// ROLL > 0 (with WANT_J): the joints' origins and axes go to `lj` -- LDS, entry e of this lane at lj[e * ROLL] -- instead of
// register arrays, so that the joint loop can stay rolled (see below); the Jacobian is formed from there with static indices.
template <bool WANT_J, int ROLL = 0>
ILQR_DEV void fk(const DevChain& c, const double* q, double p[3], double quat[4], double (*J)[DOF], double* lj = nullptr) {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pos[3] = {0, 0, 0};
    double org[DOF][3], ax[DOF][3];
    auto joint = [&](const int j, const double qj) {
        double Rn[9];
#pragma unroll
        for (int i = 0; i < 3; i++) pos[i] += R[3 * i] * c.ppre[j][0] + R[3 * i + 1] * c.ppre[j][1] + R[3 * i + 2] * c.ppre[j][2];
        mat3_mul(R, c.Rpre[j], Rn);
        const double x = c.axis[j][0], y = c.axis[j][1], z = c.axis[j][2];
        if constexpr (WANT_J) {
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const double axi = Rn[3 * i] * x + Rn[3 * i + 1] * y + Rn[3 * i + 2] * z;
                if constexpr (ROLL) { lj[(j * 6 + i) * ROLL] = pos[i]; lj[(j * 6 + 3 + i) * ROLL] = axi; }
                else { org[j][i] = pos[i]; ax[j][i] = axi; }
            }
        }
        double st, ct;
        sincos(qj, &st, &ct);
        double vt = 1 - ct;
        double Rq[9] = {ct + vt * x * x,      -z * st + vt * x * y, y * st + vt * x * z,
                        z * st + vt * x * y,  ct + vt * y * y,      -x * st + vt * y * z,
                        -y * st + vt * x * z, x * st + vt * y * z,  ct + vt * z * z};
        mat3_mul(Rn, Rq, R);
    };
    if constexpr (WANT_J && !ROLL) {  // the Jacobian keeps every joint's origin and axis: static indices, so the joints are unrolled
#pragma unroll
        for (int j = 0; j < DOF; j++) joint(j, q[j]);
    } else {
        // Pose only: ONE copy of the joint step (sincos + two rotation products, ~400 instructions) in a rolled loop instead of seven.  The
        // kernels that hold this code are short launches between the streaming kernels of an iteration: they start with the instruction
        // cache and the L2 cold, and fetching their code -- not executing it -- set their time.  The joint angle is picked
        // by selects (q[] stays in registers); the chain constants are uniform loads.  Same operations in the same order: same bits.
        if constexpr (ROLL) {  // (with the Jacobian's register pressure the select chain became an indexed stack object: the angles go through LDS too)
#pragma unroll
            for (int i = 0; i < DOF; i++) lj[(6 * DOF + i) * ROLL] = q[i];
        }
#pragma unroll 1
        for (int j = 0; j < DOF; j++) {
            double qj;
            if constexpr (ROLL) {
                qj = lj[(6 * DOF + j) * ROLL];
            } else {
                qj = q[0];
#pragma unroll
                for (int i = 1; i < DOF; i++) {
                    qj = (j == i) ? q[i] : qj;
                    ILQR_OPAQUE_VGPR(qj);  // one select at a time: the whole chain is recognised as q[j] and q[] becomes a stack object (k_select_x: 36 B)
                }
            }
            joint(j, qj);
        }
    }
    {
        double Rn[9];
#pragma unroll
        for (int i = 0; i < 3; i++) pos[i] += R[3 * i] * c.ptail[0] + R[3 * i + 1] * c.ptail[1] + R[3 * i + 2] * c.ptail[2];
        mat3_mul(R, c.Rtail, Rn);
        kdl_quat(Rn, quat);
    }
    p[0] = pos[0]; p[1] = pos[1]; p[2] = pos[2];
    if (WANT_J) {
        if constexpr (ROLL) {
#pragma unroll
            for (int j = 0; j < DOF; j++)
#pragma unroll
                for (int i = 0; i < 3; i++) { org[j][i] = lj[(j * 6 + i) * ROLL]; ax[j][i] = lj[(j * 6 + 3 + i) * ROLL]; }
        }
#pragma unroll
        for (int j = 0; j < DOF; j++) {
            double r0 = pos[0] - org[j][0], r1 = pos[1] - org[j][1], r2 = pos[2] - org[j][2];
            J[0][j] = ax[j][1] * r2 - ax[j][2] * r1;
            J[1][j] = ax[j][2] * r0 - ax[j][0] * r2;
            J[2][j] = ax[j][0] * r1 - ax[j][1] * r0;
            J[3][j] = ax[j][0];
            J[4][j] = ax[j][1];
            J[5][j] = ax[j][2];
        }
    }
}

// Eigen::Quaterniond::toRotationMatrix / Quaterniond(Matrix3d) as TransformedSimulationInterface::getEEOrnQuat uses them
// (q = (w,x,y,z), row-major matrix; trace > 0 branch, else the largest diagonal entry)
ILQR_DEV void eig_quat_to_mat(const double* q, double* m) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    m[0] = 1 - (tyy + tzz); m[1] = txy - twz; m[2] = txz + twy;
    m[3] = txy + twz; m[4] = 1 - (txx + tzz); m[5] = tyz - twx;
    m[6] = txz - twy; m[7] = tyz + twx; m[8] = 1 - (txx + tyy);
}
ILQR_DEV void eig_mat_to_quat(const double* m, double* q) {
    // Eigen's Quaternion(Matrix3) (the four cases by the trace and the largest diagonal entry; ties resolve as its strict '>' comparisons do),
    // written without branches: ONE square root and ONE division on a selected radicand, the components by selects.  Same operations on
    // the same operands as the four-way branch -- whose merged tails the compiler turned into a dynamically indexed q[], i.e. scratch (or,
    // promoted, hidden LDS) in every kernel that holds FK.
    const double tr = m[0] + m[4] + m[8];
    const bool c0 = tr > 0;
    const bool c1 = !c0 && (m[0] >= m[4] && m[0] >= m[8]);
    const bool c2 = !c0 && !c1 && (m[4] > m[0] && m[4] >= m[8]);
    const double r0 = tr + 1.0, r1 = m[0] - m[4] - m[8] + 1.0, r2 = m[4] - m[8] - m[0] + 1.0, r3 = m[8] - m[0] - m[4] + 1.0;
    const double t = sqrt(c0 ? r0 : (c1 ? r1 : (c2 ? r2 : r3)));
    const double hf = 0.5 * t, ti = 0.5 / t;
    const double d75 = m[7] - m[5], d26 = m[2] - m[6], d31 = m[3] - m[1], s31 = m[3] + m[1], s62 = m[6] + m[2], s75 = m[7] + m[5];
    // case:    0: q = (hf, d75 ti, d26 ti, d31 ti)   1: (d75 ti, hf, s31 ti, s62 ti)   2: (d26 ti, (m1+m3) ti, hf, s75 ti)   3: (d31 ti, (m2+m6) ti, (m5+m7) ti, hf)
    const double s13 = m[1] + m[3], s26 = m[2] + m[6], s57 = m[5] + m[7];
    q[0] = c0 ? hf : ((c1 ? d75 : (c2 ? d26 : d31)) * ti);
    q[1] = c1 ? hf : ((c0 ? d75 : (c2 ? s13 : s26)) * ti);
    q[2] = c2 ? hf : ((c0 ? d26 : (c1 ? s31 : s57)) * ti);
    q[3] = (c0 || c1 || c2) ? ((c0 ? d31 : (c1 ? s62 : s75)) * ti) : hf;
}

// f(x) of getFxJac: [p; quat (; dp; dquat) (; t)]  and the 6x7 Jacobian block (the full J is blkdiag(J,J) bordered by 1)
template <class S, bool WANT_J, int ROLL = 0>
ILQR_DEV void fx_of(const DevDesc& d, const double* x, double* fxv, double (*J)[DOF], int kpi = -1, double* lj = nullptr) {
    if (S::JOINT) {  // JointSpacePlannerSys::getFxJac: f(x) = x (J = I is applied by the callers)
#pragma unroll
        for (int i = 0; i < S::NF; i++) fxv[i] = x[i];
        return;
    }
    double Jl[6][DOF];
    double (*Jp)[DOF] = nullptr;
    if constexpr (WANT_J) Jp = J;
    else if constexpr (S::ND == 2) Jp = Jl;
    if constexpr (WANT_J || S::ND == 2) fk<true, ROLL>(d.chain, x, fxv, fxv + 3, Jp, lj);
    else fk<false>(d.chain, x, fxv, fxv + 3, nullptr);
    if (kpi >= 0 && d.kp_frame[kpi]) {  // TransformedSimulationInterface.cpp:53-103
        const double* R = d.kp_fR[kpi];
        double pp[3], ree[9], mm[9];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double a = 0;
#pragma unroll
            for (int j = 0; j < 3; j++) a += R[j * 3 + i] * (fxv[j] - d.kp_fp[kpi][j]);
            pp[i] = a;
        }
        eig_quat_to_mat(fxv + 3, ree);
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                double a = 0;
#pragma unroll
                for (int l = 0; l < 3; l++) a += R[l * 3 + i] * ree[l * 3 + j];
                mm[i * 3 + j] = a;
            }
        eig_mat_to_quat(mm, fxv + 3);
#pragma unroll
        for (int i = 0; i < 3; i++) fxv[i] = pp[i];
        if constexpr (WANT_J || S::ND == 2) {  // (not `if (Jp)`: the null test of a private pointer is not folded and keeps the array in scratch)
#pragma unroll
            for (int blk = 0; blk < 2; blk++)
#pragma unroll
                for (int c = 0; c < DOF; c++) {
                    const double v0 = Jp[3 * blk][c], v1 = Jp[3 * blk + 1][c], v2 = Jp[3 * blk + 2][c];
#pragma unroll
                    for (int i = 0; i < 3; i++) Jp[3 * blk + i][c] = R[i] * v0 + R[3 + i] * v1 + R[6 + i] * v2;
                }
        }
    }
    if (S::ND == 2) {
        double w[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double a = 0, b = 0;
#pragma unroll
            for (int j = 0; j < DOF; j++) { a += Jp[i][j] * x[DOF + j]; b += Jp[3 + i][j] * x[DOF + j]; }
            fxv[7 + i] = a;
            w[i] = b;
        }
        const double* qt = fxv + 3;  // .5 * H(quat)^T w   (SimulationInterface.cpp:69-73)
        fxv[10] = .5 * (-qt[1] * w[0] - qt[2] * w[1] - qt[3] * w[2]);
        fxv[11] = .5 * (qt[0] * w[0] + qt[3] * w[1] - qt[2] * w[2]);
        fxv[12] = .5 * (-qt[3] * w[0] + qt[0] * w[1] + qt[1] * w[2]);
        fxv[13] = .5 * (qt[2] * w[0] - qt[1] * w[1] + qt[0] * w[2]);
    }
    if (S::TM) fxv[S::NF - 1] = x[S::NX - 1];
}

// Keypoint::diff: tg is the target in f(x) layout
template <class S>
ILQR_DEV void kp_diff(const double* tg, const double* fxv, double* e) {
    if (S::JOINT) {  // AngularKeypoint::diff (AngularKeypoint.cpp:24-27): target - state
#pragma unroll
        for (int i = 0; i < S::NQ; i++) e[i] = tg[i] - fxv[i];
        return;
    }
#pragma unroll
    for (int i = 0; i < S::NQ; i++) e[i] = 0;
    bool allz = true;
#pragma unroll
    for (int i = 0; i < 7 * S::ND; i++) allz = allz && (fabs(fxv[i]) <= 1e-12);
    if (!allz) {  // PosOrnKeypoint.cpp:29
        double lm[4];
#pragma unroll
        for (int i = 0; i < 3; i++) e[i] = tg[i] - fxv[i];
        sd_logmap(tg + 3, fxv + 3, lm);
        double h[3];
        H_mul(tg + 3, lm, h);
#pragma unroll
        for (int i = 0; i < 3; i++) e[3 + i] = -2 * h[i];
        if (S::ND == 2) {
            double tr[4], dv[4];
#pragma unroll
            for (int i = 0; i < 3; i++) e[6 + i] = tg[7 + i] - fxv[7 + i];
            sd_transport(fxv + 10, fxv + 3, tg + 3, tr);
#pragma unroll
            for (int i = 0; i < 4; i++) dv[i] = tg[10 + i] - tr[i];
            H_mul(tg + 3, dv, h);
#pragma unroll
            for (int i = 0; i < 3; i++) e[9 + i] = -2 * h[i];
        }
    }
    if (S::TM) e[S::NQ - 1] = tg[S::NF - 1] - fxv[S::NF - 1];
}

// PosOrnKeypointDistFunct::diff (PosOrnKeypointDistFunct.cpp:13-35) applied to the residual of PosOrnKeypoint::diff: the
// position part is shrunk by the radius along its direction (zero inside the ball), every orientation component by its
// threshold towards zero.  The Jacobian used by cost_x / cost_xx is the plain one (the reference does not differentiate the
// dead zone either).
ILQR_DEV void kp_deadzone(const DevDesc& d, int kpi, double* e) {
    if (!d.kp_dist[kpi]) return;
    const double rad = d.kp_pos_radius[kpi];
    const double n = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    if (n <= rad) {
        e[0] = e[1] = e[2] = 0;
    } else {
        const double f = n - rad;
#pragma unroll
        for (int i = 0; i < 3; i++) e[i] = e[i] / n * f;  // normalized() * (norm - radius)
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double th = d.kp_orn_thresh[kpi][i], v = e[3 + i];
        if (fabs(v) <= th) e[3 + i] = 0;
        else e[3 + i] = v - (v < 0 ? -1.0 : 1.0) * th;
    }
}

// limits (inspectJointLimit, System.cpp:121-142): returns sum q_i L_ii q_i ; optionally adds -L q to lx and L^2 to diag(lxx)
template <class S>
ILQR_DEV double limit_cost(const DevDesc& d, const double* x) {
    double a = 0;
    if (d.limits_set) {
#pragma unroll
        for (int i = 0; i < S::NX; i++) {
            if (d.lw[i] != 0) {
                double qv = 0, L = 0;
                if (x[i] > d.smax[i]) { qv = d.smax[i] - x[i]; L = d.penalty; }
                else if (x[i] < d.smin[i]) { qv = d.smin[i] - x[i]; L = d.penalty; }
                a += qv * L * qv;
            }
        }
    }
    return a;
}

// System::cost at a keypoint step (System.cpp:213-234) -- task part only
// EXT = false: plain PosOrn / Spacetime keypoints in the base frame (no dead zone, no object frame, the system's R): the lean
// instantiation the hot kernels use when the descriptor has none of those (FwdArgs::kp_ext)
template <class S, bool EXT = true>
ILQR_DEV double kp_cost(const DevDesc& d, int kpi, const double* tg, const double* x, const double* u) {
    if (EXT && !S::JOINT && S::ND == 1 && d.kp_joint[kpi]) {  // joint-space keypoint of a hybrid sequence: e = target - x
        constexpr int NJ = S::NX;
        const double* Q = d.kp_Q[kpi];
        double c = 0;
#pragma unroll
        for (int i = 0; i < NJ; i++) {
            double qe = 0;
#pragma unroll
            for (int j = 0; j < NJ; j++) qe += Q[i * NJ + j] * (tg[j] - x[j]);
            c += (tg[i] - x[i]) * qe;
        }
        double ru = 0;
        if (u) {
#pragma unroll
            for (int i = 0; i < S::NU; i++) ru += u[i] * (d.kp_has_Ru[kpi] ? d.kp_Ru[kpi][i] : d.R_diag[i]) * u[i];
        }
        return c + ru;
    }
    double fxv[S::NF], e[S::NQ];
    fx_of<S, false>(d, x, fxv, nullptr, EXT ? kpi : -1);
    kp_diff<S>(tg, fxv, e);
    if (EXT) kp_deadzone(d, kpi, e);
    const double* Q = d.kp_Q[kpi];
    double c = 0;
#pragma unroll
    for (int i = 0; i < S::NQ; i++) {
        double qe = 0;
#pragma unroll
        for (int j = 0; j < S::NQ; j++) qe += Q[i * S::NQ + j] * e[j];
        c += e[i] * qe;
    }
    double ru = 0;
    if (u) {
#pragma unroll
        for (int i = 0; i < S::NU; i++) ru += u[i] * ((EXT && d.kp_has_Ru[kpi]) ? d.kp_Ru[kpi][i] : d.R_diag[i]) * u[i];
    }
    return c + ru;
}

// one dynamics step, functional form of reset()+sendVel/sendAcc (SimulationInterface.cpp:19-31)
template <class S>
ILQR_DEV void dyn_step(const DevDesc& d, const double* x, const double* u, double* xn) {
    const double dts = S::TM ? u[S::NU - 1] : 0.0;
    const double dt = S::TM ? dts * dts : d.dt;
    if (S::ND == 1) {
#pragma unroll
        for (int i = 0; i < DOF; i++) xn[i] = x[i] + (dt * u[i] + dt * dt / 2 * 0.0);
    } else {
#pragma unroll
        for (int i = 0; i < DOF; i++) {
            xn[i] = x[i] + (dt * x[DOF + i] + dt * dt / 2 * u[i]);
            xn[DOF + i] = x[DOF + i] + dt * u[i];
        }
    }
    if (S::TM) xn[S::NX - 1] = x[S::NX - 1] + dt;
}

}  // namespace ilqr
