// ilqr_kernels.hip -- batched iLQR kernels for gfx950 (MI355X), fp64.
//
// v1 mapping: ONE LANE PER PROBLEM INSTANCE.  Trajectory/gain buffers are structure-of-arrays with the batch
// innermost ([t][component][Bp]) so every load/store of a wave is one contiguous 512-byte segment; all the small
// dense algebra of a timestep lives in that lane's registers.  The time recursion (rollout, Riccati sweep) is
// sequential inside the lane; parallelism is across instances only.
//
// Reference behaviour restated (paths relative to ilqr_planner/ilqr_planner in the reference tree):
//   initial rollout            src/solver/ILQRRecursive.cpp:27-56   (AL: src/solver/AL-ILQR.cpp:53-85)
//   backward Riccati sweep     src/solver/ILQRRecursive.cpp:68-97   (AL terms: src/solver/AL-ILQR.cpp:110-134)
//   forward pass + line search src/solver/ILQRRecursive.cpp:101-176 (AL: src/solver/AL-ILQR.cpp:149-227)
// A_k, B_k are never stored: they are rebuilt from (x_k, u_k) exactly as forwardPass builds them.
#include "ilqr_kernels.hpp"
#include "ilqr_step.hpp"

namespace ilqr {

// ------------------------------------------------------------------------------------------------ kernels

// Initial rollout from U0 (ILQRRecursive.cpp:27-56): X, cost0; resets the per-instance solve state.
template <class S, bool AL>
__global__ __launch_bounds__(64) void k_init_rollout(Bufs a, double penalty) {
    constexpr int NX = S::NX, NU = S::NU;
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= d.B) return;
    const int Bp = d.Bp, T = d.T;
    double x[NX], u[NU], xn[NX];
    init_state<S>(d, a, b, x);
    double cost = 0;
    int kpi = 0;
    double* X = a.X[0];
    double* U = a.U[0];
    for (int k = 0; k < T - 1; k++) {
        UNR for (int i = 0; i < NX; i++) AT(X, k * NX + i, b) = x[i];
        UNR for (int i = 0; i < NU; i++) { u[i] = AT(a.U0, k * NU + i, b); AT(U, k * NU + i, b) = u[i]; }
        if (AL) {
            for (int r = 0; r < a.m; r++) {
                double g = con_g<S>(a, k, r, x, u);
                double lam = AT(a.lambda, k * a.m + r, b);
                AT(a.Is, k * a.m + r, b) = penalty * ((g < 0 && lam == 0) ? 0.0 : 1.0);
            }
        }
        const bool iskp = (kpi < d.n_kp && d.kp_t[kpi] == k);
        cost += stage_cost<S>(d, a, b, iskp ? kpi : -1, x, u);
        if (iskp) kpi++;
        dyn_step<S>(d, x, u, xn);
        UNR for (int i = 0; i < NX; i++) x[i] = xn[i];
    }
    UNR for (int i = 0; i < NX; i++) AT(X, (T - 1) * NX + i, b) = x[i];
    {
        const bool iskp = (kpi < d.n_kp && d.kp_t[kpi] == T - 1);
        double zu[NU];
        UNR for (int i = 0; i < NU; i++) zu[i] = 0;
        cost += stage_cost<S>(d, a, b, iskp ? kpi : -1, x, zu);
    }
    a.cost[b] = cost;
    a.alpha[b] = 1.0;
    a.cur[b] = 0;
    a.active[b] = 1;
    a.iters[b] = 0;
    a.pend[b] = 0;
    a.pred[b] = 0;
    a.status[b] = isfinite(cost) ? 0 : 1;
}

// l_x, l_xx of the keypoint steps (System::cost_x / cost_xx incl. the limit terms) for the current trajectory, one lane per
// (instance, keypoint).  Keeps FK, the quaternion log map and J'QJ out of the sequential sweep: the sweep only loads
// NX + NX*NX doubles at the (two) keypoint steps.
template <class S, bool EXT>
__global__ __launch_bounds__(64) void k_kp_derivs(Bufs a, int fused) {
    constexpr int NX = S::NX;
    // origins and axes of the joints and the joint angles (7 x DOF per lane) for the rolled FK loop: a short launch between the streaming kernels is bound by
    // fetching its code, and the unrolled joints were most of it (ilqr_device.hpp: fk)
    constexpr int ROLLN = (!S::JOINT && S::ND == 1) ? 64 : 0;  // (2nd order: the rolled form spills more)
    __shared__ double sj[ROLLN ? 7 * DOF : 1][64];  // (nothing reserved where the loop is not rolled)
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * 64 + threadIdx.x;
    const int kpi = blockIdx.y;
    if (b >= d.B) return;
    if (!a.active[b]) return;
    const int Bp = d.Bp;
    const int k = d.kp_t[kpi];
    const int cur = a.cur[b];
    const double* X = a.X[cur];
    double x[NX];
    UNR for (int i = 0; i < NX; i++) x[i] = AT(X, k * NX + i, b);
    const int w = fused ? a.pend[b] - 1 : -1;
    if (w >= 0) {  // fused acceptance (FwdArgs::fused): the accepted state is still spread over the two buffers -- k_apply's expression
        const double aa = ldexp(1.0, -w);
        const double* X1 = a.X[1 - cur];
        UNR for (int i = 0; i < NX; i++) {
            const double x1 = AT(X1, k * NX + i, b);
            x[i] = (w == 0) ? x1 : fma(aa, x1 - x[i], x[i]);
        }
    }
    // row by row straight to memory: the lane never holds the n_x x n_x matrix (it spilled for n_x = 14, 15)
    double* out = a.kpd + (size_t)kpi * (NX + NX * NX) * Bp;
    stage_derivs_rows<S, EXT, ROLLN>(d, a, b, x, kpi, &sj[0][threadIdx.x], [&](int i, const double* row, double lxi) {
        AT(out, i, b) = lxi;
        UNR for (int j = 0; j < NX; j++) AT(out, NX + i * NX + j, b) = row[j];
    });
}

// Backward Riccati sweep (ILQRRecursive.cpp:68-97): writes K_k, d_k for k = T-2..0.  Needs k_kp_derivs first.
//
// One lane per instance, the reference's operation order (dense products, partial-pivot LU as Eigen's inverse()).  The matrices of a step
// -- 1500 doubles for n_x = 15 against the 256 a lane's whole register file holds -- live in an EXPLICIT workspace in global memory
// (a.ws, entry e of instance b at ws[e Bp + b]: coalesced across the lanes of a wave), walked by plain loops: 60 VGPRs, no scratch, no
// compiler-managed private segment.  (As local arrays of a fully unrolled kernel they made 512-VGPR code objects with up to 7 KB of scratch per
// lane, and in round 1 one of them -- then still with the keypoint code inlined -- ended in a memory aperture violation: DESIGN.md 5.5.)
// The stage derivatives come from k_kp_derivs at the keypoint steps (FK, the quaternion log map, frames, dead zones and J'QJ stay out of
// the sweep) and are the limit terms elsewhere; the second limit set on top of either.
#define NOUNR _Pragma("unroll 1")
template <class S>
constexpr int backward_ws_doubles() {
    return 3 * S::NX * S::NX + 5 * S::NU * S::NX + 3 * S::NU * S::NU + 4 * S::NX + 2 * S::NU;
}
int backward_ws_entries(int kind, int nd) {
    if (kind == 2) return backward_ws_doubles<Sys<2, 1>>();
    if (kind == 3) return backward_ws_doubles<Sys<3, 1>>();
    if (kind == 0) return nd == 1 ? backward_ws_doubles<Sys<0, 1>>() : backward_ws_doubles<Sys<0, 2>>();
    return nd == 1 ? backward_ws_doubles<Sys<1, 1>>() : backward_ws_doubles<Sys<1, 2>>();
}

template <class S, bool AL>
__global__ __launch_bounds__(64) void k_backward(Bufs a) {
    constexpr int NX = S::NX, NU = S::NU, ND = S::ND, TM = S::TM;
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= d.B) return;
    if (!a.active[b]) return;
    const int Bp = d.Bp, T = d.T;
    const int cur = a.cur[b];
    const double* X = a.X[cur];
    const double* U = a.U[cur];
    // workspace views
    double* const w = a.ws + b;
    constexpr int oP = 0, op = oP + NX * NX, oQxx = op + NX, oQx = oQxx + NX * NX, oAtP = oQx + NX, oBtP = oAtP + NX * NX, oQux = oBtP + NU * NX,
                  oQxu = oQux + NU * NX, oK = oQxu + NX * NU, oKtQ = oK + NU * NX, oQuu = oKtQ + NX * NU, oMr = oQuu + NU * NU, oQi = oMr + NU * NU,
                  oQu = oQi + NU * NU, odk = oQu + NU, obc = odk + NU, ox = obc + NX, oEnd = ox + NX;
    static_assert(oEnd == backward_ws_doubles<S>(), "workspace layout");
#define WS(o, i, j, C) w[(size_t)((o) + (i) * (C) + (j)) * Bp]
#define WV(o, i) w[(size_t)((o) + (i)) * Bp]
    // l_xx -> matrix at oM, l_x -> vector at oV for the state at ox (statement for statement stage_derivs / lim2_derivs of ilqr_step.hpp)
    auto stage_to_ws = [&](int kpi_, int oM, int oV) {
        if (kpi_ >= 0) {
            const double* src = a.kpd + (size_t)kpi_ * (NX + NX * NX) * Bp;
            NOUNR for (int i = 0; i < NX; i++) {
                WV(oV, i) = AT(src, i, b);
                NOUNR for (int j = 0; j < NX; j++) WS(oM, i, j, NX) = AT(src, NX + i * NX + j, b);
            }
        } else {
            NOUNR for (int i = 0; i < NX; i++) {
                double lxi = 0, lxxi = 0;
                NOUNR for (int j = 0; j < NX; j++) WS(oM, i, j, NX) = 0;
                if (d.limits_set && d.lw[i] != 0) {
                    const double xi = WV(ox, i);
                    double qv = 0, L = 0;
                    if (xi > d.smax[i]) { qv = d.smax[i] - xi; L = d.penalty; }
                    else if (xi < d.smin[i]) { qv = d.smin[i] - xi; L = d.penalty; }
                    lxi += -L * qv;
                    lxxi += (L != 0.0) ? d.pen_xx : 0.0;
                }
                WV(oV, i) = lxi;
                WS(oM, i, i, NX) = lxxi;
            }
        }
        if (d.lim2) {
            NOUNR for (int i = 0; i < NX; i++) {
                if (d.lw2[i] != 0) {
                    const double xi = WV(ox, i);
                    double qv = 0, L = 0;
                    if (xi > d.smax2[i]) { qv = d.smax2[i] - xi; L = d.penalty2; }
                    else if (xi < d.smin2[i]) { qv = d.smin2[i] - xi; L = d.penalty2; }
                    WV(oV, i) += -L * qv;
                    WS(oM, i, i, NX) += (L != 0.0) ? d.pen_xx2 : 0.0;
                }
            }
        }
    };
    auto is_v = [](int i) { return ND == 2 && i >= DOF && i < 2 * DOF; };

    int kpi = d.n_kp - 1;
    NOUNR for (int i = 0; i < NX; i++) WV(ox, i) = AT(X, (T - 1) * NX + i, b);
    {
        const bool iskp = (kpi >= 0 && d.kp_t[kpi] == T - 1);
        stage_to_ws(iskp ? kpi : -1, oP, op);
        if (iskp) kpi--;
    }
    for (int k = T - 2; k >= 0; k--) {
        NOUNR for (int i = 0; i < NX; i++) WV(ox, i) = AT(X, k * NX + i, b);
        const double* u = U + (size_t)k * NU * Bp + b;  // u_i = u[i Bp]
        const double dts = TM ? u[(size_t)(NU - 1) * Bp] : 0.0;
        const double dt = TM ? dts * dts : d.dt;
        const double hdt2 = dt * dt / 2;
        // last column of B for time systems (PosOrnTimePlannerSys.cpp:161-162,176)
        if (TM) {
            if (ND == 1) {
                NOUNR for (int i = 0; i < DOF; i++) WV(obc, i) = 2 * dts * u[(size_t)i * Bp];
            } else {
                NOUNR for (int i = 0; i < DOF; i++) {
                    const double ui = u[(size_t)i * Bp];
                    double dqn = WV(ox, DOF + i) + dt * ui;  // velocity AFTER the step
                    WV(obc, i) = 2 * dts * dqn + 2 * dts * dts * dts * ui;
                    WV(obc, DOF + i) = 2 * dts * ui;
                }
            }
            WV(obc, NX - 1) = 2 * dts;
        }
        {   // l_xx, l_x of the stage go straight into the Qxx / Qx slots (Qxx = l_xx + A'PA below)
            const bool iskp = (kpi >= 0 && d.kp_t[kpi] == k);
            stage_to_ws(iskp ? kpi : -1, oQxx, oQx);
            if (iskp) kpi--;
        }
        // BtP = B^T P (NU x NX), AtP = A^T P (NX x NX)
        NOUNR for (int i = 0; i < DOF; i++)
            NOUNR for (int j = 0; j < NX; j++)
                WS(oBtP, i, j, NX) = (ND == 1) ? dt * WS(oP, i, j, NX) : hdt2 * WS(oP, i, j, NX) + dt * WS(oP, DOF + i, j, NX);
        if (TM) {
            NOUNR for (int j = 0; j < NX; j++) {
                double s = 0;
                NOUNR for (int l = 0; l < NX; l++) s += WV(obc, l) * WS(oP, l, j, NX);
                WS(oBtP, NU - 1, j, NX) = s;
            }
        }
        NOUNR for (int i = 0; i < NX; i++)
            NOUNR for (int j = 0; j < NX; j++)
                WS(oAtP, i, j, NX) = is_v(i) ? dt * WS(oP, i - DOF, j, NX) + WS(oP, i, j, NX) : WS(oP, i, j, NX);
        // Qux = BtP A ; Quu = R + BtP B ; Qxx = lxx + AtP A ; Qxu = AtP B ; Qu = R u + B^T p ; Qx = lx + A^T p
        NOUNR for (int i = 0; i < NU; i++) {
            NOUNR for (int j = 0; j < NX; j++)
                WS(oQux, i, j, NX) = is_v(j) ? WS(oBtP, i, j - DOF, NX) * dt + WS(oBtP, i, j, NX) : WS(oBtP, i, j, NX);
            NOUNR for (int j = 0; j < DOF; j++)
                WS(oQuu, i, j, NU) = (ND == 1) ? WS(oBtP, i, j, NX) * dt : WS(oBtP, i, j, NX) * hdt2 + WS(oBtP, i, DOF + j, NX) * dt;
            if (TM) {
                double s = 0;
                NOUNR for (int l = 0; l < NX; l++) s += WS(oBtP, i, l, NX) * WV(obc, l);
                WS(oQuu, i, NU - 1, NU) = s;
            }
            WS(oQuu, i, i, NU) = d.R_diag[i] + WS(oQuu, i, i, NU);
        }
        NOUNR for (int i = 0; i < NX; i++) {
            NOUNR for (int j = 0; j < NX; j++) {
                double v = is_v(j) ? WS(oAtP, i, j - DOF, NX) * dt + WS(oAtP, i, j, NX) : WS(oAtP, i, j, NX);
                WS(oQxx, i, j, NX) = WS(oQxx, i, j, NX) + v;
            }
            NOUNR for (int j = 0; j < DOF; j++)
                WS(oQxu, i, j, NU) = (ND == 1) ? WS(oAtP, i, j, NX) * dt : WS(oAtP, i, j, NX) * hdt2 + WS(oAtP, i, DOF + j, NX) * dt;
            if (TM) {
                double s = 0;
                NOUNR for (int l = 0; l < NX; l++) s += WS(oAtP, i, l, NX) * WV(obc, l);
                WS(oQxu, i, NU - 1, NU) = s;
            }
        }
        NOUNR for (int i = 0; i < DOF; i++) {
            double v = (ND == 1) ? dt * WV(op, i) : hdt2 * WV(op, i) + dt * WV(op, DOF + i);
            WV(oQu, i) = d.R_diag[i] * u[(size_t)i * Bp] + v;
        }
        if (TM) {
            double s = 0;
            NOUNR for (int l = 0; l < NX; l++) s += WV(obc, l) * WV(op, l);
            WV(oQu, NU - 1) = d.R_diag[NU - 1] * u[(size_t)(NU - 1) * Bp] + s;
        }
        NOUNR for (int i = 0; i < NX; i++) {
            double v = is_v(i) ? dt * WV(op, i - DOF) + WV(op, i) : WV(op, i);
            WV(oQx, i) = WV(oQx, i) + v;
        }
        if (AL) {  // AL-ILQR.cpp:110-134: c_u' I c_x etc., lambda + I c
            const int ns = NX + NU;
            for (int r = 0; r < a.m; r++) {
                const double* Ar = a.conA + ((size_t)(a.per_step ? k : 0) * a.m + r) * ns;
                const double Ik = AT(a.Is, k * a.m + r, b);
                const double lam = AT(a.lambda, k * a.m + r, b);
                double g = 0;  // con_g (ilqr_step.hpp): A [x;u] - b, same order
                NOUNR for (int i = 0; i < NX; i++) g += Ar[i] * WV(ox, i);
                NOUNR for (int i = 0; i < NU; i++) g += Ar[NX + i] * u[(size_t)i * Bp];
                g = g - a.conb[(size_t)(a.per_step ? k : 0) * a.m + r];
                const double wv = lam + Ik * g;
                NOUNR for (int i = 0; i < NU; i++) {
                    const double au = Ar[NX + i];
                    NOUNR for (int j = 0; j < NX; j++) WS(oQux, i, j, NX) += au * Ik * Ar[j];
                    NOUNR for (int j = 0; j < NU; j++) WS(oQuu, i, j, NU) += au * Ik * Ar[NX + j];
                    WV(oQu, i) += au * wv;
                }
                NOUNR for (int i = 0; i < NX; i++) {
                    const double ax = Ar[i];
                    NOUNR for (int j = 0; j < NX; j++) WS(oQxx, i, j, NX) += ax * Ik * Ar[j];
                    NOUNR for (int j = 0; j < NU; j++) WS(oQxu, i, j, NU) += ax * Ik * Ar[NX + j];
                    WV(oQx, i) += ax * wv;
                }
            }
        }
        // Quu_inv = -(Quu + reg I)^-1 ; K = Quu_inv Qux ; d = Quu_inv Qu.  Inverse: Eigen's MatrixXd::inverse() = PartialPivLU + solve against
        // the identity (the algorithm of inverse_lu in ilqr_step.hpp, on the workspace)
        NOUNR for (int i = 0; i < NU; i++)
            NOUNR for (int j = 0; j < NU; j++) WS(oMr, i, j, NU) = WS(oQuu, i, j, NU) + ((i == j) ? d.reg : 0.0);
        {
            unsigned long long piv = 0;  // row permutation, 4 bits per row
            NOUNR for (int i = 0; i < NU; i++) piv |= (unsigned long long)i << (4 * i);
            NOUNR for (int kk = 0; kk < NU; kk++) {
                double best = fabs(WS(oMr, kk, kk, NU));
                int r = kk;
                NOUNR for (int i = kk + 1; i < NU; i++) {
                    const double v = fabs(WS(oMr, i, kk, NU));
                    if (v > best) { best = v; r = i; }
                }
                if (r != kk) {
                    NOUNR for (int j = 0; j < NU; j++) {
                        const double t0 = WS(oMr, kk, j, NU), t1 = WS(oMr, r, j, NU);
                        WS(oMr, kk, j, NU) = t1;
                        WS(oMr, r, j, NU) = t0;
                    }
                    const unsigned long long p0 = (piv >> (4 * kk)) & 15ull, p1 = (piv >> (4 * r)) & 15ull;
                    piv = (piv & ~((15ull << (4 * kk)) | (15ull << (4 * r)))) | (p1 << (4 * kk)) | (p0 << (4 * r));
                }
                const double pv = WS(oMr, kk, kk, NU);
                NOUNR for (int i = kk + 1; i < NU; i++) {
                    WS(oMr, i, kk, NU) /= pv;
                    const double f = WS(oMr, i, kk, NU);
                    NOUNR for (int j = kk + 1; j < NU; j++) WS(oMr, i, j, NU) -= f * WS(oMr, kk, j, NU);
                }
            }
            NOUNR for (int c = 0; c < NU; c++) {
                NOUNR for (int i = 0; i < NU; i++) {
                    double s = ((int)((piv >> (4 * i)) & 15ull) == c) ? 1.0 : 0.0;
                    NOUNR for (int j = 0; j < i; j++) s -= WS(oMr, i, j, NU) * WS(oQi, j, c, NU);
                    WS(oQi, i, c, NU) = s;
                }
                NOUNR for (int i = NU - 1; i >= 0; i--) {
                    double s = WS(oQi, i, c, NU);
                    NOUNR for (int j = i + 1; j < NU; j++) s -= WS(oMr, i, j, NU) * WS(oQi, j, c, NU);
                    WS(oQi, i, c, NU) = s / WS(oMr, i, i, NU);
                }
            }
        }
        double* const Krec = KD_REC(a.KD, Bp, NU * kd_rowp(NX), k, b);
        NOUNR for (int i = 0; i < NU; i++) {
            NOUNR for (int j = 0; j < NX; j++) {
                double s = 0;
                NOUNR for (int l = 0; l < NU; l++) s += (-1 * WS(oQi, i, l, NU)) * WS(oQux, l, j, NX);
                WS(oK, i, j, NX) = s;
                Krec[i * kd_rowp(NX) + j] = s;
            }
            double s = 0;
            NOUNR for (int l = 0; l < NU; l++) s += (-1 * WS(oQi, i, l, NU)) * WV(oQu, l);
            WV(odk, i) = s;
            Krec[i * kd_rowp(NX) + NX] = s;
        }
        // P = Qxx + K'QuuK + K'Qux + QxuK ; p = Qx + K'Quu d + K'Qu + Qxu d   (un-regularised Quu)
        NOUNR for (int i = 0; i < NX; i++)
            NOUNR for (int j = 0; j < NU; j++) {
                double s = 0;
                NOUNR for (int l = 0; l < NU; l++) s += WS(oK, l, i, NX) * WS(oQuu, l, j, NU);
                WS(oKtQ, i, j, NU) = s;
            }
        NOUNR for (int i = 0; i < NX; i++) {
            NOUNR for (int j = 0; j < NX; j++) {
                double t1 = 0, t2 = 0, t3 = 0;
                NOUNR for (int l = 0; l < NU; l++) {
                    t1 += WS(oKtQ, i, l, NU) * WS(oK, l, j, NX);
                    t2 += WS(oK, l, i, NX) * WS(oQux, l, j, NX);
                    t3 += WS(oQxu, i, l, NU) * WS(oK, l, j, NX);
                }
                WS(oP, i, j, NX) = ((WS(oQxx, i, j, NX) + t1) + t2) + t3;
            }
            double v1 = 0, v2 = 0, v3 = 0;
            NOUNR for (int l = 0; l < NU; l++) {
                v1 += WS(oKtQ, i, l, NU) * WV(odk, l);
                v2 += WS(oK, l, i, NX) * WV(oQu, l);
                v3 += WS(oQxu, i, l, NU) * WV(odk, l);
            }
            WV(op, i) = ((WV(oQx, i) + v1) + v2) + v3;
        }
    }
#undef WS
#undef WV
}

// Forward pass with step-halving line search (ILQRRecursive.cpp:101-176).  Each trial re-rolls the whole horizon
// and overwrites the instance's inactive trajectory buffer; the last trial executed is the accepted one.
template <class S, bool AL>
__global__ __launch_bounds__(64) void k_forward(Bufs a, int it, int line_search, int early_stop, double penalty_roll,
                                                double penalty_update, int do_update, int nb_iter) {
    constexpr int NX = S::NX, NU = S::NU;
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= d.B) return;
    if (!a.active[b]) return;
    const int Bp = d.Bp, T = d.T;
    const int cur = a.cur[b];
    const double* X = a.X[cur];
    const double* U = a.U[cur];
    double* Xn = a.X[1 - cur];
    double* Un = a.U[1 - cur];
    const double cost0 = a.cost[b];
    double alpha = 2, newCost = 0, dun = 0;
    do {
        alpha /= 2.0;
        double x[NX], xn[NX], u[NU];
        init_state<S>(d, a, b, x);
        dun = 0;
        newCost = 0;
        int kpi = 0;
        for (int k = 0; k < T - 1; k++) {
            double dx[NX];
            UNR for (int i = 0; i < NX; i++) dx[i] = x[i] - AT(X, k * NX + i, b);
            double n2 = 0;
            UNR for (int i = 0; i < NU; i++) {
                double s = 0;
                UNR for (int j = 0; j < NX; j++) s += KD_REC(a.KD, Bp, NU * kd_rowp(NX), k, b)[i * kd_rowp(NX) + j] * dx[j];
                double du = s + alpha * KD_REC(a.KD, Bp, NU * kd_rowp(NX), k, b)[i * kd_rowp(NX) + NX];
                n2 += du * du;
                u[i] = AT(U, k * NU + i, b) + du;
            }
            dun += sqrt(n2);  // accumulates ||du||, not ||du||^2 (ILQRRecursive.cpp:132)
            UNR for (int i = 0; i < NX; i++) AT(Xn, k * NX + i, b) = x[i];
            UNR for (int i = 0; i < NU; i++) AT(Un, k * NU + i, b) = u[i];
            if (AL) {
                for (int r = 0; r < a.m; r++) {
                    double g = con_g<S>(a, k, r, x, u);
                    double lam = AT(a.lambda, k * a.m + r, b);
                    AT(a.Is, k * a.m + r, b) = penalty_roll * ((g < 0 && lam == 0) ? 0.0 : 1.0);
                }
            }
            const bool iskp = (kpi < d.n_kp && d.kp_t[kpi] == k);
            newCost += stage_cost<S>(d, a, b, iskp ? kpi : -1, x, u);
            if (iskp) kpi++;
            dyn_step<S>(d, x, u, xn);
            UNR for (int i = 0; i < NX; i++) x[i] = xn[i];
        }
        UNR for (int i = 0; i < NX; i++) AT(Xn, (T - 1) * NX + i, b) = x[i];
        {
            const bool iskp = (kpi < d.n_kp && d.kp_t[kpi] == T - 1);
            double zu[NU];
            UNR for (int i = 0; i < NU; i++) zu[i] = 0;
            newCost += stage_cost<S>(d, a, b, iskp ? kpi : -1, x, zu);
        }
    } while (((newCost >= cost0) || isnan(newCost)) && alpha > d.alpha_floor && line_search);

    if (AL && do_update) {  // multiplier update with the UPDATED penalty, on the accepted trajectory (AL-ILQR.cpp:202-208)
        for (int k = 0; k < T - 1; k++) {
            double x[NX], u[NU];
            UNR for (int i = 0; i < NX; i++) x[i] = AT(Xn, k * NX + i, b);
            UNR for (int i = 0; i < NU; i++) u[i] = AT(Un, k * NU + i, b);
            for (int r = 0; r < a.m; r++) {
                double g = con_g<S>(a, k, r, x, u);
                double v = AT(a.lambda, k * a.m + r, b) + penalty_update * g;
                AT(a.lambda, k * a.m + r, b) = v > 0 ? v : 0;
            }
        }
    }
    // accept unconditionally (ILQRRecursive.cpp:157-162)
    a.cost[b] = newCost;
    a.alpha[b] = alpha;
    a.cur[b] = 1 - cur;
    a.iters[b] = it + 1;
    a.status[b] = (isfinite(newCost) ? 0 : 1) | ((alpha <= d.alpha_floor) ? 2 : 0);
    if (a.cost_trace) {
        a.cost_trace[(size_t)it * Bp + b] = newCost;
        a.alpha_trace[(size_t)it * Bp + b] = alpha;
    }
    bool stop = early_stop && (alpha * sqrt(dun) < d.stop_tol);
    if (!AL) stop = stop && (newCost < 1e-3);  // ILQRRecursive.cpp:174 vs AL-ILQR.cpp:225
    if (stop) a.active[b] = 0;
}

// ------------------------------------------------------------------------------------------------ gather / scatter

// natural [B][rows] (host/ABI layout) <-> device [rows][Bp]
__global__ void k_to_soa(const double* __restrict__ src, double* __restrict__ dst, int B, int Bp, int rows) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (b < B) dst[(size_t)r * Bp + b] = src[(size_t)b * rows + r];
}
__global__ void k_from_soa(const double* __restrict__ src, double* __restrict__ dst, int B, int Bp, int rows) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (b < B) dst[(size_t)b * rows + r] = src[(size_t)r * Bp + b];
}
// same, selecting each instance's current trajectory buffer
__global__ void k_from_soa_cur(const double* __restrict__ s0, const double* __restrict__ s1, const int* __restrict__ cur,
                               double* __restrict__ dst, int B, int Bp, int rows) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (b < B) dst[(size_t)b * rows + r] = (cur[b] ? s1 : s0)[(size_t)r * Bp + b];
}
// returned ds are scaled by the accepted alpha (ILQRRecursive.cpp:128,144,162)
__global__ void k_from_soa_scaled(const double* __restrict__ src, const double* __restrict__ alpha, const int* __restrict__ iters,
                                  double* __restrict__ dst, int B, int Bp, int rows) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (b < B) dst[(size_t)b * rows + r] = (iters[b] > 0 ? alpha[b] : 1.0) * src[(size_t)r * Bp + b];
}

// gains in the reference's layout: K[B][T-1][NU][NX], d[B][T-1][NU] scaled by the accepted alpha (ILQRRecursive.cpp:128,144,162)
__global__ void k_get_gains(const double* __restrict__ kd, int sym, const double* __restrict__ alpha, const int* __restrict__ iters,
                            double* __restrict__ K_out, double* __restrict__ d_out, int B, int Bp, int T1, int nu, int nx) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (b >= B) return;
    const int rowp = kd_rowp(nx), rs = kd_rs(sym, nu, rowp);  // (sym: the packed symmetric record of the uniform-R single-integrator sweep)
    const double* rec = KD_REC(kd, Bp, rs, k, b);
    const double sc = (iters[b] > 0) ? alpha[b] : 1.0;
    for (int i = 0; i < nu; i++) {
        if (K_out)
            for (int j = 0; j < nx; j++) K_out[(((size_t)b * T1 + k) * nu + i) * nx + j] = rec[kd_off(sym, rowp, i, j)];
        if (d_out) d_out[((size_t)b * T1 + k) * nu + i] = sc * rec[kd_off(sym, rowp, i, nx)];
    }
}

// Receding-horizon warm start: the next solve starts from the accepted plan shifted by `shift` timesteps --
// U0[k] = U[min(k + shift, T-2)], q0 (dq0) = the joint part of x_{shift}.  One lane per (instance, timestep).
__global__ void k_warm_start(Bufs a, double* __restrict__ U0, double* __restrict__ q0, double* __restrict__ dq0, int shift, int nx, int nu, int nd) {
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (b >= d.B) return;
    const int Bp = d.Bp, T = d.T;
    const int cur = a.cur[b];
    const int ks = (k + shift < T - 1) ? k + shift : T - 2;
    for (int i = 0; i < nu; i++) AT(U0, k * nu + i, b) = AT(a.U[cur], ks * nu + i, b);
    if (k == 0 && shift > 0) {
        const int kx = shift < T ? shift : T - 1;
        for (int i = 0; i < DOF; i++) {
            AT(q0, i, b) = AT(a.X[cur], kx * nx + i, b);
            if (nd == 2) AT(dq0, i, b) = AT(a.X[cur], kx * nx + DOF + i, b);
        }
    }
}

// Tracking law of the tutorials (POS_ORN_SYS.ipynb cell 7): u = ubar_k + K_k (x - xbar_k) [+ alpha d_k] for a measured state
// per instance; natural layouts x_meas[B][NX], u_out[B][NU].  One lane per instance.
__global__ void k_track(Bufs a, const double* __restrict__ x_meas, int k, int with_ff, double* __restrict__ u_out, int nx, int nu) {
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= d.B) return;
    const int Bp = d.Bp;
    const int cur = a.cur[b];
    const int sym = a.kd_sym, rowp = kd_rowp(nx), rs = kd_rs(sym, nu, rowp);
    const double* rec = KD_REC(a.KD, Bp, rs, k, b);
    const double sc = (a.iters[b] > 0) ? a.alpha[b] : 1.0;
    for (int i = 0; i < nu; i++) {
        double s = AT(a.U[cur], k * nu + i, b);
        for (int j = 0; j < nx; j++) s += rec[kd_off(sym, rowp, i, j)] * (x_meas[(size_t)b * nx + j] - AT(a.X[cur], k * nx + j, b));
        if (with_ff) s += sc * rec[kd_off(sym, rowp, i, nx)];
        u_out[(size_t)b * nu + i] = s;
    }
}

// f(X) for every (instance, timestep): one lane per pair (tuple<1> of ILQRRecursive::solve)
template <class S>
__global__ __launch_bounds__(64) void k_fx_all(Bufs a, double* __restrict__ out /* natural [B][T][NF] */) {
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y;
    if (b >= d.B) return;
    const int Bp = d.Bp;
    const double* X = a.X[a.cur[b]];
    double x[S::NX], fxv[S::NF];
    UNR for (int i = 0; i < S::NX; i++) x[i] = AT(X, t * S::NX + i, b);
    fx_of<S, false>(d, x, fxv, nullptr);
    UNR for (int i = 0; i < S::NF; i++) out[((size_t)b * d.T + t) * S::NF + i] = fxv[i];
}

// stand-alone batched FK (KDLRobot::updateKinematics), natural layouts
__global__ __launch_bounds__(64) void k_fk_batch(const DevDesc* dd, int n, const double* __restrict__ q, double* __restrict__ pos, double* __restrict__ quat,
                           double* __restrict__ jac) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double qv[DOF], p[3], qt[4], J[6][DOF];
    UNR for (int j = 0; j < DOF; j++) qv[j] = q[(size_t)i * DOF + j];
    fk<true>(dd->chain, qv, p, qt, J);
    if (pos) { UNR for (int j = 0; j < 3; j++) pos[(size_t)i * 3 + j] = p[j]; }
    if (quat) { UNR for (int j = 0; j < 4; j++) quat[(size_t)i * 4 + j] = qt[j]; }
    if (jac) {
        UNR for (int r = 0; r < 6; r++)
            UNR for (int j = 0; j < DOF; j++) jac[((size_t)i * 6 + r) * DOF + j] = J[r][j];
    }
}

// ------------------------------------------------------------------------------------------------ launchers

template <class S>
static void launch_solver_kernel(int which, bool al, const Bufs& a, int B, hipStream_t st, const FwdArgs& f) {
    const dim3 grid((B + 63) / 64), block(64);
    switch (which) {
        case KER_KP_DERIVS:
            if (f.n_kp > 0 && f.kp_ext) hipLaunchKernelGGL((k_kp_derivs<S, true>), dim3((B + 63) / 64, f.n_kp), dim3(64), 0, st, a, f.fused);
            else if (f.n_kp > 0) hipLaunchKernelGGL((k_kp_derivs<S, false>), dim3((B + 63) / 64, f.n_kp), dim3(64), 0, st, a, f.fused);
            break;
        case KER_INIT:
            if (al) hipLaunchKernelGGL((k_init_rollout<S, true>), grid, block, 0, st, a, f.penalty_roll);
            else hipLaunchKernelGGL((k_init_rollout<S, false>), grid, block, 0, st, a, f.penalty_roll);
            break;
        case KER_BACKWARD:
            if (al) hipLaunchKernelGGL((k_backward<S, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_backward<S, false>), grid, block, 0, st, a);
            break;
        case KER_FORWARD:
            if (al)
                hipLaunchKernelGGL((k_forward<S, true>), grid, block, 0, st, a, f.it, f.line_search, f.early_stop, f.penalty_roll,
                                   f.penalty_update, f.do_update, f.nb_iter);
            else
                hipLaunchKernelGGL((k_forward<S, false>), grid, block, 0, st, a, f.it, f.line_search, f.early_stop, f.penalty_roll,
                                   f.penalty_update, f.do_update, f.nb_iter);
            break;
    }
}

void launch_solver(int kind, int nd, int which, bool al, const Bufs& a, int B, hipStream_t st, const FwdArgs& f) {
    if (kind == 2) launch_solver_kernel<Sys<2, 1>>(which, al, a, B, st, f);
    else if (kind == 3) launch_solver_kernel<Sys<3, 1>>(which, al, a, B, st, f);
    else if (kind == 0 && nd == 1) launch_solver_kernel<Sys<0, 1>>(which, al, a, B, st, f);
    else if (kind == 0 && nd == 2) launch_solver_kernel<Sys<0, 2>>(which, al, a, B, st, f);
    else if (kind == 1 && nd == 1) launch_solver_kernel<Sys<1, 1>>(which, al, a, B, st, f);
    else launch_solver_kernel<Sys<1, 2>>(which, al, a, B, st, f);
}

void launch_fx_all(int kind, int nd, const Bufs& a, int B, int T, double* out, hipStream_t st) {
    const dim3 grid((B + 63) / 64, T), block(64);
    if (kind == 2) hipLaunchKernelGGL((k_fx_all<Sys<2, 1>>), grid, block, 0, st, a, out);
    else if (kind == 3) hipLaunchKernelGGL((k_fx_all<Sys<3, 1>>), grid, block, 0, st, a, out);
    else if (kind == 0 && nd == 1) hipLaunchKernelGGL((k_fx_all<Sys<0, 1>>), grid, block, 0, st, a, out);
    else if (kind == 0 && nd == 2) hipLaunchKernelGGL((k_fx_all<Sys<0, 2>>), grid, block, 0, st, a, out);
    else if (kind == 1 && nd == 1) hipLaunchKernelGGL((k_fx_all<Sys<1, 1>>), grid, block, 0, st, a, out);
    else hipLaunchKernelGGL((k_fx_all<Sys<1, 2>>), grid, block, 0, st, a, out);
}

void launch_to_soa(const double* src, double* dst, int B, int Bp, int rows, hipStream_t st) {
    hipLaunchKernelGGL(k_to_soa, dim3((B + 255) / 256, rows), dim3(256), 0, st, src, dst, B, Bp, rows);
}
void launch_from_soa(const double* src, double* dst, int B, int Bp, int rows, hipStream_t st) {
    hipLaunchKernelGGL(k_from_soa, dim3((B + 255) / 256, rows), dim3(256), 0, st, src, dst, B, Bp, rows);
}
void launch_from_soa_cur(const double* s0, const double* s1, const int* cur, double* dst, int B, int Bp, int rows, hipStream_t st) {
    hipLaunchKernelGGL(k_from_soa_cur, dim3((B + 255) / 256, rows), dim3(256), 0, st, s0, s1, cur, dst, B, Bp, rows);
}
void launch_from_soa_scaled(const double* src, const double* alpha, const int* iters, double* dst, int B, int Bp, int rows, hipStream_t st) {
    hipLaunchKernelGGL(k_from_soa_scaled, dim3((B + 255) / 256, rows), dim3(256), 0, st, src, alpha, iters, dst, B, Bp, rows);
}
void launch_get_gains(const double* kd, int kd_sym, const double* alpha, const int* iters, double* K_out, double* d_out, int B, int Bp, int T1, int nu, int nx,
                       hipStream_t st) {
    hipLaunchKernelGGL(k_get_gains, dim3((B + 63) / 64, T1), dim3(64), 0, st, kd, kd_sym, alpha, iters, K_out, d_out, B, Bp, T1, nu, nx);
}
void launch_warm_start(const Bufs& a, double* U0, double* q0, double* dq0, int shift, int B, int T, int nx, int nu, int nd, hipStream_t st) {
    hipLaunchKernelGGL(k_warm_start, dim3((B + 63) / 64, T - 1), dim3(64), 0, st, a, U0, q0, dq0, shift, nx, nu, nd);
}
void launch_track(const Bufs& a, const double* x_meas, int k, int with_ff, double* u_out, int B, int nx, int nu, hipStream_t st) {
    hipLaunchKernelGGL(k_track, dim3((B + 63) / 64), dim3(64), 0, st, a, x_meas, k, with_ff, u_out, nx, nu);
}
void launch_fk_batch(const DevDesc* dd, int n, const double* q, double* pos, double* quat, double* jac, hipStream_t st) {
    hipLaunchKernelGGL(k_fk_batch, dim3((n + 63) / 64), dim3(64), 0, st, dd, n, q, pos, quat, jac);
}

}  // namespace ilqr
