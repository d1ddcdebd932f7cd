// ilqr_batch_dev.hpp -- device helpers shared by the batch (Gauss-Newton on the whole control sequence) solvers:
// ilqr_batchcp.hip (narrow bases, Kw <= 16) and ilqr_batchwide.hip (wide bases and BatchILQR's identity basis).
#pragma once
#include "ilqr_step.hpp"

namespace ilqr {

// dynamics step + the pieces of A, B the W recurrence needs (forwardPass, PosOrn*PlannerSys.cpp)
template <class S>
struct StepAB {
    double dt, dts, hdt2;
    double bc[S::NX];  // last column of B (time systems)
};

template <class S>
ILQR_DEV void step_ab(const DevDesc& d, const double* x, const double* u, double* xn, StepAB<S>& ab) {
    constexpr int NX = S::NX, NU = S::NU;
    dyn_step<S>(d, x, u, xn);
    ab.dts = S::TM ? u[NU - 1] : 0.0;
    ab.dt = S::TM ? ab.dts * ab.dts : d.dt;
    ab.hdt2 = ab.dt * ab.dt / 2;
    if (S::TM) {
        if (S::ND == 1) {
            UNR for (int i = 0; i < DOF; i++) ab.bc[i] = 2 * ab.dts * u[i];
        } else {
            UNR for (int i = 0; i < DOF; i++) {
                ab.bc[i] = 2 * ab.dts * xn[DOF + i] + 2 * ab.dts * ab.dts * ab.dts * u[i];  // velocity AFTER the step
                ab.bc[DOF + i] = 2 * ab.dts * u[i];
            }
        }
        ab.bc[NX - 1] = 2 * ab.dts;
    }
}

// limits on a state: diag(L) and q (inspectJointLimit)
template <class S>
ILQR_DEV void limit_terms(const DevDesc& d, const double* x, double* Ld, double* q) {
    UNR for (int i = 0; i < S::NX; i++) {
        Ld[i] = 0;
        q[i] = 0;
        if (d.limits_set && d.batch_limits && d.lw[i] != 0) {
            if (x[i] > d.smax[i]) { q[i] = d.smax[i] - x[i]; Ld[i] = d.penalty; }
            else if (x[i] < d.smin[i]) { q[i] = d.smin[i] - x[i]; Ld[i] = d.penalty; }
        }
    }
}

// Out-of-line keypoint evaluations (FK, log map, J'QJ): the iteration kernels call them a few times from different places; one
// copy each keeps their registers and code out of the callers' loops.
template <class S>
__device__ __noinline__ double w_kp_cost(const DevDesc* d, const double* kp_tg, int b, int kpi, const double* xt) {
    const int Bp = d->Bp;
    double tg[S::NF];
    UNR for (int i = 0; i < S::NF; i++) tg[i] = AT(kp_tg, kpi * S::NF + i, b);
    return kp_cost<S>(*d, kpi, tg, xt, nullptr);
}
template <class S>
__device__ __noinline__ void w_kp_derivs(const DevDesc* d, const Bufs* a, int b, int kpi, const double* xt, double* lxx_out, double* lx_out) {
    double lxx[S::NX][S::NX], lx[S::NX];
    stage_derivs<S, false>(*d, *a, b, xt, kpi, lxx, lx);  // lxx = J'QJ, lx = -J'Q e
    UNR for (int i = 0; i < S::NX; i++) {
        lx_out[i] = lx[i];
        UNR for (int j = 0; j < S::NX; j++) lxx_out[i * S::NX + j] = lxx[i][j];
    }
}

struct BTArgs { int it, early_stop; };

// Backtracking with all step sizes at once: 16 lanes per instance, lane l rolls out u + 2^-l du (BatchILQR.cpp:138-158: the first
// alpha whose cost improves wins, alpha < 1e-3 is accepted anyway).  On the time systems the search often ends at its floor, so
// the chain a lane walks is one rollout instead of eleven.
template <class S>
__global__ __launch_bounds__(64) void k_bt_linesearch(Bufs a, BTArgs c) {
    constexpr int NX = S::NX, NU = S::NU;
    const DevDesc& d = *a.desc;
    const int lane = threadIdx.x, l = lane & 15, bq = blockIdx.x * 4 + (lane >> 4);
    const bool ok = bq < d.B && a.active[bq < d.B ? bq : 0];
    const int b = ok ? bq : 0;
    const int Bp = d.Bp, T = d.T;
    double* U = a.U[0];
    const double* DU = a.U[1];
    const double cost0 = a.cost[b];
    const double alpha = ldexp(1.0, -(l < 11 ? l : 10));
    double cost;
    {
        double x[NX], xp[NX], u[NU], xn[NX];
        init_state<S>(d, a, b, x);
        UNR for (int i = 0; i < NX; i++) xp[i] = x[i];
        double cost_e = 0, cost_u = 0, cost_l = 0;
        int kpi = 0;
        auto kp_here = [&](int i) {
            double xt[NX];
            UNR for (int r = 0; r < NX; r++) xt[r] = x[r];
            cost_e += w_kp_cost<S>(&d, a.kp_tg, b, kpi, xt);
            if (i > 0) {
                double Ld[NX], ql[NX];
                limit_terms<S>(d, xp, Ld, ql);
                UNR for (int r = 0; r < NX; r++) cost_l += ql[r] * Ld[r] * ql[r];
            }
            kpi++;
        };
        if (kpi < d.n_kp && d.kp_t[kpi] == 0) kp_here(0);
        for (int s = 0; s < T - 1; s++) {
            UNR for (int i = 0; i < NU; i++) {
                u[i] = AT(U, s * NU + i, b) + alpha * AT(DU, s * NU + i, b);
                cost_u += u[i] * d.R_diag[i] * u[i];
            }
            dyn_step<S>(d, x, u, xn);
            UNR for (int i = 0; i < NX; i++) { xp[i] = x[i]; x[i] = xn[i]; }
            if (kpi < d.n_kp && d.kp_t[kpi] == s + 1) kp_here(s + 1);
        }
        cost = cost_e + cost_u + cost_l;
    }
    const bool take = (l < 11) && ((cost < cost0) || (alpha < 1e-3));
    const unsigned long long mk = __ballot(take ? 1 : 0);
    const int win = __ffs((unsigned)((mk >> (lane & 48)) & 0xffffull)) - 1;
    double dun2 = 0;  // ||du||^2: the 16 lanes sum strided slices
    for (int s = l; s < (T - 1) * NU; s += 16) { const double v = AT(DU, s, b); dun2 += v * v; }
    for (int o = 8; o > 0; o >>= 1) dun2 += __shfl_xor(dun2, o);
    if (!ok) return;
    const double aw = ldexp(1.0, -win);
    for (int s = l; s < (T - 1) * NU; s += 16) AT(U, s, b) += aw * AT(DU, s, b);  // u = utmp of the winner, the 16 lanes share the copy
    if (l != win) return;
    a.alpha[b] = alpha;
    a.iters[b] = c.it + 1;
    a.status[b] = (isfinite(cost) ? 0 : 1) | ((alpha < 1e-3) ? 2 : 0);
    if (a.cost_trace) {
        a.cost_trace[(size_t)c.it * Bp + b] = cost0;
        a.alpha_trace[(size_t)c.it * Bp + b] = alpha;
    }
    a.cost[b] = cost;
    if (c.early_stop && alpha * sqrt(dun2) < 1e-3) a.active[b] = 0;  // :167
}

}  // namespace ilqr
