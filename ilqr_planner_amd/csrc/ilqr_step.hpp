// ilqr_step.hpp -- per-instance building blocks shared by the kernel files: constraint rows, stage cost and its
// derivatives, the partial-pivot LU inverse, initial state.  See ilqr_kernels.hip for the reference citations.
#pragma once
#include "ilqr_kernels.hpp"

namespace ilqr {

#define UNR _Pragma("unroll")

// XCD-aware tile index.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an XCD and its L2), so
// with tile = blockIdx consecutive tiles -- whose 16..64-byte row segments share 128-byte lines -- would land on 8 different
// L2s and every line would be fetched (or partially written back) once per XCD.  Grids are launched with a multiple of 8
// blocks; this maps the blocks of one XCD onto a contiguous range of tiles.  Placement affects speed only.
__device__ __forceinline__ int xcd_tile() {
    const int per = (int)gridDim.x >> 3;
    return ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
}
static inline unsigned grid_x8(unsigned tiles) { return (tiles + 7u) / 8u * 8u; }
#define AT(buf, row, b) (buf)[(size_t)(row) * (size_t)Bp + (size_t)(b)]

// ------------------------------------------------------------------------------------------------ helpers

template <class S>
ILQR_DEV void load_vec(const double* buf, int row0, int n, int Bp, int b, double* out) {
    for (int i = 0; i < n; i++) out[i] = AT(buf, row0 + i, b);
}

// AL_ILQR::constraints (AL-ILQR.cpp:21-44): g = A [x;u] - b ; I = 0 iff (g<0 && lambda==0) ; stored as penalty*I
template <class S>
ILQR_DEV double con_g(const Bufs& a, int k, int r, const double* x, const double* u) {
    const int ns = S::NX + S::NU;
    const double* Ar = a.conA + ((size_t)(a.per_step ? k : 0) * a.m + r) * ns;
    double g = 0;
    UNR for (int i = 0; i < S::NX; i++) g += Ar[i] * x[i];
    UNR for (int i = 0; i < S::NU; i++) g += Ar[S::NX + i] * u[i];
    return g - a.conb[(size_t)(a.per_step ? k : 0) * a.m + r];
}

// lx, lxx of a stage (System::cost_x / cost_xx, System.cpp:248-308).  P <- lxx, p <- lx.
template <class S, bool WITH_LIMITS = true, bool EXT = true, int ROLL = 0>
ILQR_DEV void stage_derivs(const DevDesc& d, const Bufs& a, int b, const double* x, int kpi, double (*lxx)[S::NX], double* lx, double* lj = nullptr) {
    constexpr int NX = S::NX, NQ = S::NQ, NF = S::NF;
    const int Bp = d.Bp;
    UNR for (int i = 0; i < NX; i++) {
        lx[i] = 0;
        UNR for (int j = 0; j < NX; j++) lxx[i][j] = 0;
    }
    if (kpi >= 0 && !S::JOINT && EXT && S::ND == 1 && d.kp_joint[kpi]) {  // joint-space keypoint of a hybrid sequence (J = I, n_x x n_x precision)
        const double* Q = d.kp_Q[kpi];
        UNR for (int i = 0; i < NX; i++) {
            double s = 0;
            UNR for (int j = 0; j < NX; j++) s += Q[i * NX + j] * (AT(a.kp_tg, kpi * NF + j, b) - x[j]);
            lx[i] += -1 * s;
            UNR for (int j = 0; j < NX; j++) lxx[i][j] += Q[i * NX + j];
        }
    } else if (kpi >= 0 && S::JOINT) {  // J = I: l_x = -Q e, l_xx = Q
        double e[NQ], tg[NF];
        UNR for (int i = 0; i < NF; i++) tg[i] = AT(a.kp_tg, kpi * NF + i, b);
        kp_diff<S>(tg, x, e);
        const double* Q = d.kp_Q[kpi];
        UNR for (int i = 0; i < NQ; i++) {
            double s = 0;
            UNR for (int j = 0; j < NQ; j++) s += Q[i * NQ + j] * e[j];
            lx[i] += -1 * s;
            UNR for (int j = 0; j < NQ; j++) lxx[i][j] += Q[i * NQ + j];
        }
    } else if (kpi >= 0) {
        double fxv[NF], J[6][DOF], e[NQ], tg[NF], Qe[NQ];
        fx_of<S, true, ROLL>(d, x, fxv, J, EXT ? kpi : -1, lj);
        UNR for (int i = 0; i < NF; i++) tg[i] = AT(a.kp_tg, kpi * NF + i, b);
        kp_diff<S>(tg, fxv, e);
        if (EXT) kp_deadzone(d, kpi, e);
        const double* Q = d.kp_Q[kpi];
        UNR for (int i = 0; i < NQ; i++) {
            double s = 0;
            UNR for (int j = 0; j < NQ; j++) s += Q[i * NQ + j] * e[j];
            Qe[i] = s;
        }
        // Jf (NQ x NX) = blkdiag(J, J) bordered by 1 for the time state; lx = -Jf^T Q e ; lxx = Jf^T Q Jf
        UNR for (int blk = 0; blk < S::ND; blk++) {
            UNR for (int c = 0; c < DOF; c++) {
                double s = 0;
                UNR for (int r = 0; r < 6; r++) s += J[r][c] * Qe[6 * blk + r];
                lx[DOF * blk + c] += -1 * s;
            }
        }
        if (S::TM) lx[NX - 1] += -1 * Qe[NQ - 1];
        // JtQ[c][r'] for c in block blk: sum_r J[r][c%7] Q[6 blk + r][r']
        UNR for (int blk = 0; blk < S::ND; blk++) {
            UNR for (int c = 0; c < DOF; c++) {
                double jq[NQ];
                UNR for (int rp = 0; rp < NQ; rp++) {
                    double s = 0;
                    UNR for (int r = 0; r < 6; r++) s += J[r][c] * Q[(6 * blk + r) * NQ + rp];
                    jq[rp] = s;
                }
                UNR for (int blk2 = 0; blk2 < S::ND; blk2++) {
                    UNR for (int c2 = 0; c2 < DOF; c2++) {
                        double s = 0;
                        UNR for (int r = 0; r < 6; r++) s += jq[6 * blk2 + r] * J[r][c2];
                        lxx[DOF * blk + c][DOF * blk2 + c2] += s;
                    }
                }
                if (S::TM) lxx[DOF * blk + c][NX - 1] += jq[NQ - 1];
            }
        }
        if (S::TM) {
            UNR for (int blk2 = 0; blk2 < S::ND; blk2++) {
                UNR for (int c2 = 0; c2 < DOF; c2++) {
                    double s = 0;
                    UNR for (int r = 0; r < 6; r++) s += Q[(NQ - 1) * NQ + 6 * blk2 + r] * J[r][c2];
                    lxx[NX - 1][DOF * blk2 + c2] += s;
                }
            }
            lxx[NX - 1][NX - 1] += Q[(NQ - 1) * NQ + NQ - 1];
        }
    }
    if (WITH_LIMITS && d.limits_set) {
        UNR for (int i = 0; i < NX; i++) {
            if (d.lw[i] != 0) {
                double qv = 0, L = 0;
                if (x[i] > d.smax[i]) { qv = d.smax[i] - x[i]; L = d.penalty; }
                else if (x[i] < d.smin[i]) { qv = d.smin[i] - x[i]; L = d.penalty; }
                lx[i] += -L * qv;
                lxx[i][i] += (L != 0.0) ? d.pen_xx : 0.0;  // L^2, once per sub-system of a SequentialSystem
            }
        }
    }
}

// The same derivatives of a KEYPOINT stage handed out row by row -- sink(i, row_i of l_xx, l_x[i]) -- for k_kp_derivs: a lane that holds the
// whole n_x x n_x matrix (225 doubles for n_x = 15) spills; a row at a time does not.  Expression for expression stage_derivs (same bits).
template <class S, bool EXT, int ROLL, class Sink>
ILQR_DEV void stage_derivs_rows(const DevDesc& d, const Bufs& a, int b, const double* x, int kpi, double* lj, Sink&& sink) {
    constexpr int NX = S::NX, NQ = S::NQ, NF = S::NF;
    const int Bp = d.Bp;
    auto finish = [&](int i, double* row, double lxi) {  // limit terms of coordinate i (System.cpp:121-142), then out
        if (d.limits_set && d.lw[i] != 0) {
            double qv = 0, L = 0;
            if (x[i] > d.smax[i]) { qv = d.smax[i] - x[i]; L = d.penalty; }
            else if (x[i] < d.smin[i]) { qv = d.smin[i] - x[i]; L = d.penalty; }
            lxi += -L * qv;
            row[i] += (L != 0.0) ? d.pen_xx : 0.0;
        }
        sink(i, row, lxi);
    };
    if (!S::JOINT && EXT && S::ND == 1 && d.kp_joint[kpi]) {  // joint-space keypoint of a hybrid sequence (J = I, n_x x n_x precision)
        const double* Q = d.kp_Q[kpi];
        UNR for (int i = 0; i < NX; i++) {
            double s = 0, row[NX];
            UNR for (int j = 0; j < NX; j++) s += Q[i * NX + j] * (AT(a.kp_tg, kpi * NF + j, b) - x[j]);
            UNR for (int j = 0; j < NX; j++) row[j] = 0.0 + Q[i * NX + j];
            finish(i, row, 0.0 + -1 * s);
        }
    } else if (S::JOINT) {  // J = I: l_x = -Q e, l_xx = Q
        double e[NQ], tg[NF];
        UNR for (int i = 0; i < NF; i++) tg[i] = AT(a.kp_tg, kpi * NF + i, b);
        kp_diff<S>(tg, x, e);
        const double* Q = d.kp_Q[kpi];
        UNR for (int i = 0; i < NQ; i++) {
            double s = 0, row[NX];
            UNR for (int j = 0; j < NQ; j++) s += Q[i * NQ + j] * e[j];
            UNR for (int j = 0; j < NX; j++) row[j] = (j < NQ) ? 0.0 + Q[i * NQ + j] : 0.0;
            finish(i, row, 0.0 + -1 * s);
        }
    } else {
        double fxv[NF], J[6][DOF], e[NQ], tg[NF], Qe[NQ];
        fx_of<S, true, ROLL>(d, x, fxv, J, EXT ? kpi : -1, lj);
        UNR for (int i = 0; i < NF; i++) tg[i] = AT(a.kp_tg, kpi * NF + i, b);
        kp_diff<S>(tg, fxv, e);
        if (EXT) kp_deadzone(d, kpi, e);
        const double* Q = d.kp_Q[kpi];
        UNR for (int i = 0; i < NQ; i++) {
            double s = 0;
            UNR for (int j = 0; j < NQ; j++) s += Q[i * NQ + j] * e[j];
            Qe[i] = s;
        }
        // Jf (NQ x NX) = blkdiag(J, J) bordered by 1 for the time state; lx = -Jf^T Q e ; lxx = Jf^T Q Jf
        UNR for (int blk = 0; blk < S::ND; blk++) {
            UNR for (int c = 0; c < DOF; c++) {
                double sl = 0;
                UNR for (int r = 0; r < 6; r++) sl += J[r][c] * Qe[6 * blk + r];
                double jq[NQ], row[NX];
                UNR for (int rp = 0; rp < NQ; rp++) {
                    double s = 0;
                    UNR for (int r = 0; r < 6; r++) s += J[r][c] * Q[(6 * blk + r) * NQ + rp];
                    jq[rp] = s;
                }
                UNR for (int blk2 = 0; blk2 < S::ND; blk2++) {
                    UNR for (int c2 = 0; c2 < DOF; c2++) {
                        double s = 0;
                        UNR for (int r = 0; r < 6; r++) s += jq[6 * blk2 + r] * J[r][c2];
                        row[DOF * blk2 + c2] = 0.0 + s;
                    }
                }
                if (S::TM) row[NX - 1] = 0.0 + jq[NQ - 1];
                finish(DOF * blk + c, row, 0.0 + -1 * sl);
            }
        }
        if (S::TM) {
            double row[NX];
            UNR for (int blk2 = 0; blk2 < S::ND; blk2++) {
                UNR for (int c2 = 0; c2 < DOF; c2++) {
                    double s = 0;
                    UNR for (int r = 0; r < 6; r++) s += Q[(NQ - 1) * NQ + 6 * blk2 + r] * J[r][c2];
                    row[DOF * blk2 + c2] = 0.0 + s;
                }
            }
            row[NX - 1] = 0.0 + Q[(NQ - 1) * NQ + NQ - 1];
            finish(NX - 1, row, 0.0 + -1 * Qe[NQ - 1]);
        }
    }
}

// Eigen MatrixXd::inverse() (PartialPivLU + solve against identity), fully unrolled, no dynamic indexing.
template <int N>
ILQR_DEV void inverse_lu(double (*M)[N], double (*Inv)[N]) {
    int piv[N];
    UNR for (int i = 0; i < N; i++) piv[i] = i;
    UNR for (int k = 0; k < N; k++) {
        // pivot search: first row with the largest |M[r][k]|, r >= k
        double best = fabs(M[k][k]);
        int r = k;
        UNR for (int i = k + 1; i < N; i++) {
            double v = fabs(M[i][k]);
            if (v > best) { best = v; r = i; }
        }
        UNR for (int i = k + 1; i < N; i++) {
            bool sw = (r == i);
            UNR for (int j = 0; j < N; j++) {
                double t0 = M[k][j], t1 = M[i][j];
                M[k][j] = sw ? t1 : t0;
                M[i][j] = sw ? t0 : t1;
            }
            int p0 = piv[k], p1 = piv[i];
            piv[k] = sw ? p1 : p0;
            piv[i] = sw ? p0 : p1;
        }
        double pv = M[k][k];
        UNR for (int i = k + 1; i < N; i++) {
            M[i][k] /= pv;
            double f = M[i][k];
            UNR for (int j = k + 1; j < N; j++) M[i][j] -= f * M[k][j];
        }
    }
    UNR for (int c = 0; c < N; c++) {
        UNR for (int i = 0; i < N; i++) {
            double s = (piv[i] == c) ? 1.0 : 0.0;
            UNR for (int j = 0; j < i; j++) s -= M[i][j] * Inv[j][c];
            Inv[i][c] = s;
        }
        UNR for (int i = N - 1; i >= 0; i--) {
            double s = Inv[i][c];
            UNR for (int j = i + 1; j < N; j++) s -= M[i][j] * Inv[j][c];
            Inv[i][c] = s / M[i][i];
        }
    }
}

// ------------------------------------------------------------------------------------------------ rollout pieces

// Second limit set of a sequence whose sub-systems have different bounds (ilqr_problem_desc::limits2_set): the limit terms of that
// group of sub-systems, same form as the first set (inspectJointLimit, System.cpp:121-142).  Only the generic kernels call these.
template <class S>
ILQR_DEV double lim2_cost(const DevDesc& d, const double* x) {
    double a = 0;
    UNR for (int i = 0; i < S::NX; i++) {
        if (d.lw2[i] != 0) {
            double qv = 0, L = 0;
            if (x[i] > d.smax2[i]) { qv = d.smax2[i] - x[i]; L = d.penalty2; }
            else if (x[i] < d.smin2[i]) { qv = d.smin2[i] - x[i]; L = d.penalty2; }
            a += qv * L * qv;
        }
    }
    return a;
}
template <class S>
ILQR_DEV void lim2_derivs(const DevDesc& d, const double* x, double (*lxx)[S::NX], double* lx) {
    UNR for (int i = 0; i < S::NX; i++) {
        if (d.lw2[i] != 0) {
            double qv = 0, L = 0;
            if (x[i] > d.smax2[i]) { qv = d.smax2[i] - x[i]; L = d.penalty2; }
            else if (x[i] < d.smin2[i]) { qv = d.smin2[i] - x[i]; L = d.penalty2; }
            lx[i] += -L * qv;
            lxx[i][i] += (L != 0.0) ? d.pen_xx2 : 0.0;
        }
    }
}

// stage cost l(x,u,k) (System::cost): task part only at keypoint steps, limits always
template <class S>
ILQR_DEV double stage_cost(const DevDesc& d, const Bufs& a, int b, int kpi, const double* x, const double* u) {
    const int Bp = d.Bp;
    double c = 0;
    if (kpi >= 0) {
        double tg[S::NF];
        UNR for (int i = 0; i < S::NF; i++) tg[i] = AT(a.kp_tg, kpi * S::NF + i, b);
        c += kp_cost<S>(d, kpi, tg, x, u);
    }
    if (d.limits_set) c += limit_cost<S>(d, x);
    if (d.lim2) c += lim2_cost<S>(d, x);
    return c;
}

// Out-of-line variants of the keypoint-step work for the hot loops of the v2 kernels: a keypoint occurs at ~2 of the
// ~200 timesteps, so its code (FK with fp64 sincos, quaternion log map, J'QJ) is kept out of the loop body -- smaller
// loop, lower register pressure.  Arguments are copies in private memory; only the rare path touches them.
template <class S>
__device__ __noinline__ double kp_cost_call(const DevDesc* d, const double* kp_tg, int Bp, int b, int kpi, const double* xt, const double* ut) {
    double tg[S::NF];
    UNR for (int i = 0; i < S::NF; i++) tg[i] = AT(kp_tg, kpi * S::NF + i, b);
    return kp_cost<S>(*d, kpi, tg, xt, ut);
}

template <class S>
ILQR_DEV double stage_cost_ool(const DevDesc& d, const Bufs& a, int b, int kpi, const double* x, const double* u) {
    double c = 0;
    if (kpi >= 0) {
        double xt[S::NX], ut[S::NU];
        UNR for (int i = 0; i < S::NX; i++) xt[i] = x[i];
        UNR for (int i = 0; i < S::NU; i++) ut[i] = u[i];
        c += kp_cost_call<S>(&d, a.kp_tg, d.Bp, b, kpi, xt, ut);
    }
    if (d.limits_set) c += limit_cost<S>(d, x);
    return c;
}

// l_x, l_xx (dense, row-major NX x NX) of a keypoint step, out of line
template <class S>
__device__ __noinline__ void stage_derivs_call(const DevDesc* d, const Bufs* a, int b, int kpi, const double* xt, double* lxx_out, double* lx_out) {
    double lxx[S::NX][S::NX], lx[S::NX];
    stage_derivs<S>(*d, *a, b, xt, kpi, lxx, lx);
    UNR for (int i = 0; i < S::NX; i++) {
        lx_out[i] = lx[i];
        UNR for (int j = 0; j < S::NX; j++) lxx_out[i * S::NX + j] = lxx[i][j];
    }
}

template <class S>
ILQR_DEV void init_state(const DevDesc& d, const Bufs& a, int b, double* x) {
    const int Bp = d.Bp;
    UNR for (int i = 0; i < DOF; i++) x[i] = AT(a.q0, i, b);
    if (S::ND == 2) { UNR for (int i = 0; i < DOF; i++) x[DOF + i] = AT(a.dq0, i, b); }
    if (S::TM) x[S::NX - 1] = 0;
}


}  // namespace ilqr
