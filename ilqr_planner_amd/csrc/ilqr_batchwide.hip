// ilqr_batchwide.hip -- the batch Gauss-Newton solvers for WIDE bases: BatchILQR (reference src/solver/BatchILQR.cpp:110-173, which
// is BatchILQRCP with PSI = I, Kw = (T-1) n_u) and BatchILQRCP with Kw > 16 (src/solver/BatchILQRCP.cpp:109-175).
//
// The reference forms H = PSI'Su'(J'QJ + L)Su PSI + PSI'R PSI (Kw x Kw, 693 x 693 for the first tutorial) and inverts it every
// iteration.  H is a positive matrix H0 = PSI'R PSI that does not depend on the instance or the iterate, plus a term of rank
// m = n_kp n_x (14 .. 30): with V = Su PSI at the keypoint rows (m x Kw) and C = blkdiag(J'QJ + L) (m x m)
//     H^-1 = H0^-1 - H0^-1 V' (I + C G)^-1 C V H0^-1,      G = V H0^-1 V'            (push-through identity; C may be singular)
// so the step only needs an m x m solve.  Nothing of size Kw x Kw is built per instance, and no GEMM: the work per iteration is
// the rollout / keypoint evaluation, as for the recursive solver (HBM-bound streaming, not MFMA work).
//
// LTI systems (PosOrn 1st/2nd order, JointSpace: constant A, B) -- V, G and Z = H0^-1 V' are the same for every instance and are
// tabulated once on the host.  The iterate stays in an (m+1)-dimensional affine family
//     u = u0 + PSI w,   w = (beta - 1) y0 + Z c,   y0 = H0^-1 PSI'R u0        (beta = 1, c = 0 at the start)
// because the Gauss-Newton step is dw = Z d - beta y0 with d = (I + C G)^-1 (r - c + beta C V y0), r = stacked J'Qe + L ql:
//     beta <- (1 - alpha) beta,   c <- c + alpha d.
// States at the keypoint steps are affine in (beta, c), the control cost is a quadratic form in them, ||PSI dw|| likewise: an
// iteration touches m + 1 numbers per instance and never walks the horizon (k_wl_linearize, k_wl_solve, k_wl_linesearch).  The horizon is walked twice per solve:
// k_wl_init (rollouts of u0, of its projection u0^ = PSI y0 and of u = 0) and k_wl_controls + k_wide_final (u, X out).
//
// Time systems (dt = u_last^2: A, B depend on the iterate) -- identity basis only (BatchILQR).  V is per instance and iterate but
// has the closed form V_k[:, j] = (I + (tau_{t_k - 1} - tau_j) E) B_j from the stored rollout (tau = the time state, E the
// position-from-velocity block), so with kappa = (I + C G)^-1 (r + C V u):  du = R^-1 V' kappa - u.  One wave per instance
// accumulates G = V R^-1 V' (lanes own entries of G), solves the m x m system in LDS and writes du (k_wt_solve); k_wt_roll and
// k_wt_linesearch are one lane per instance like the narrow-basis kernels.
//
// Reference quirks kept: the shifted sensitivity (block j of Su meets B_j = dx_j/du_{j-1}, block 0 is zero),
// limits on the pre-step state, acceptance of a step once alpha < 1e-3, the PRE-step cost in the message stream.
#include "ilqr_batchcp.hpp"

#include <cmath>
#include <cstring>

#include "ilqr_batch_dev.hpp"

namespace ilqr {

struct WArgs {
    const double *G, *Et, *ZPZ, *PZ;  // shared tables (LTI)
    double *xbk, *av, *v0, *p0, *scal, *cv, *beta, *Ckp, *rkp, *dvb, *sc;
    const double* u0hat;
    int m, it, early_stop;
};

// s <- A s + B u of the constant-dt systems
template <class S>
ILQR_DEV void lin_step(const DevDesc& d, double* s, const double* u) {
    const double dt = d.dt, hdt2 = dt * dt / 2;
    if (S::ND == 1) {
        UNR for (int i = 0; i < DOF; i++) s[i] += dt * u[i];
    } else {
        UNR for (int i = 0; i < DOF; i++) {
            s[i] += dt * s[DOF + i] + hdt2 * u[i];
            s[DOF + i] += dt * u[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------ LTI systems

// MC lanes per instance: lane j accumulates p0[j] = (PSI Z)[:, j] . u0^ from coalesced rows of PZ (with one lane per instance
// the rows are uniform -> scalar loads, one latency per control entry); all lanes walk the three rollouts, lane 0 records.
template <class S, int MC>
__global__ __launch_bounds__(64) void k_wl_init(Bufs a, WArgs c) {
    constexpr int NX = S::NX, NU = S::NU, IPB = 64 / MC;
    const DevDesc& d = *a.desc;
    const int j = threadIdx.x % MC, b = blockIdx.x * IPB + threadIdx.x / MC;
    if (b >= d.B) return;
    const int Bp = d.Bp, T = d.T, m = c.m;
    const int jj = j < m ? j : 0;
    double x[NX], xp[NX], xh[NX], xhp[NX], xz[NX], xzp[NX], sv[NX], u[NU], uh[NU], zero[NU], xn[NX], p0 = 0;
    UNR for (int i = 0; i < NU; i++) zero[i] = 0;
    init_state<S>(d, a, b, x);
    UNR for (int i = 0; i < NX; i++) { xp[i] = xh[i] = xhp[i] = xz[i] = xzp[i] = x[i]; sv[i] = 0; }
    double c00 = 0, gam = 0, pi = 0;
    int kpi = 0;
    auto record = [&]() {
        if (j == 0) {
            double* xb = c.xbk + (size_t)kpi * 2 * NX * Bp;
            double* ab = c.av + (size_t)kpi * 2 * NX * Bp;
            UNR for (int r = 0; r < NX; r++) {
                AT(xb, r, b) = x[r];
                AT(xb, NX + r, b) = xp[r];
                AT(ab, r, b) = xh[r] - xz[r];         // Wt_t y0: response of the trajectory to the projected u0
                AT(ab, NX + r, b) = xhp[r] - xzp[r];  // Wt_{t-1} y0
                AT(c.v0, kpi * NX + r, b) = sv[r];    // V y0 (shifted sensitivity)
            }
        }
        kpi++;
    };
    if (kpi < d.n_kp && d.kp_t[kpi] == 0) record();
#pragma unroll 2
    for (int s = 0; s < T - 1; s++) {
        UNR for (int i = 0; i < NU; i++) {
            u[i] = AT(a.U0, s * NU + i, b);
            uh[i] = AT(c.u0hat, s * NU + i, b);
            c00 += u[i] * d.R_diag[i] * u[i];
            gam += u[i] * d.R_diag[i] * uh[i];
            pi += uh[i] * uh[i];
            p0 += c.PZ[(size_t)(s * NU + i) * m + jj] * uh[i];
        }
        dyn_step<S>(d, x, u, xn);
        UNR for (int i = 0; i < NX; i++) { xp[i] = x[i]; x[i] = xn[i]; }
        dyn_step<S>(d, xh, uh, xn);
        UNR for (int i = 0; i < NX; i++) { xhp[i] = xh[i]; xh[i] = xn[i]; }
        dyn_step<S>(d, xz, zero, xn);
        UNR for (int i = 0; i < NX; i++) { xzp[i] = xz[i]; xz[i] = xn[i]; }
        if (s >= 1) lin_step<S>(d, sv, uh);  // block 0 of the reference's Su is zero
        if (kpi < d.n_kp && d.kp_t[kpi] == s + 1) record();
    }
    if (j < m) { AT(c.p0, j, b) = p0; AT(c.cv, j, b) = 0; }
    if (j != 0) return;
    AT(c.scal, 0, b) = c00;
    AT(c.scal, 1, b) = gam;
    AT(c.scal, 2, b) = pi;
    c.beta[b] = 1.0;
    a.cur[b] = 0;
    a.active[b] = 1;
    a.iters[b] = 0;
    a.status[b] = 0;
    a.alpha[b] = 1.0;
    a.pend[b] = 0;
    a.pred[b] = 0;
}

// states at keypoint kpi and the step before it for the family member (beta, cv + al dv)
template <class S, int MC>
ILQR_DEV void wl_states(const DevDesc& d, const WArgs& c, int b, int kpi, double beta, const double* cv, const double* dv, double al, double* x, double* xp) {
    constexpr int NX = S::NX;
    const int Bp = d.Bp, m = c.m;
    const double* xb = c.xbk + (size_t)kpi * 2 * NX * Bp;
    const double* ab = c.av + (size_t)kpi * 2 * NX * Bp;
    UNR for (int r = 0; r < NX; r++) {
        double s0 = AT(xb, r, b) + (beta - 1) * AT(ab, r, b), s1 = AT(xb, NX + r, b) + (beta - 1) * AT(ab, NX + r, b);
        const double* e0 = c.Et + ((size_t)(kpi * 2 + 0) * NX + r) * m;
        const double* e1 = c.Et + ((size_t)(kpi * 2 + 1) * NX + r) * m;
        UNR for (int j = 0; j < MC; j++)
            if (j < m) {
                const double cj = cv[j] + al * dv[j];
                s0 += e0[j] * cj;
                s1 += e1[j] * cj;
            }
        x[r] = s0;
        xp[r] = s1;
    }
}

// task + limit cost of that family member (the control cost is the caller's quadratic form)
template <class S, int MC>
ILQR_DEV double wl_task_cost(const DevDesc& d, const Bufs& a, const WArgs& c, int b, double beta, const double* cv, const double* dv, double al) {
    constexpr int NX = S::NX;
    const int Bp = d.Bp;
    double cost_e = 0, cost_l = 0;
    for (int kpi = 0; kpi < d.n_kp; kpi++) {
        double x[NX], xp[NX];
        wl_states<S, MC>(d, c, b, kpi, beta, cv, dv, al, x, xp);
        cost_e += w_kp_cost<S>(&d, a.kp_tg, b, kpi, x);
        if (d.kp_t[kpi] > 0) {
            double Ld[NX], ql[NX];
            limit_terms<S>(d, xp, Ld, ql);
            UNR for (int r = 0; r < NX; r++) cost_l += ql[r] * Ld[r] * ql[r];
        }
    }
    return cost_e + cost_l;
}

// control cost of the family member (beta (1 - al), c + al d):  u'Ru = c00 + (bn^2 - 1) gamma + 2 bn v0.(c + al d) + (c + al d)'G(c + al d)
ILQR_DEV double wl_uru(double c00, double gam, double bn, double al, double v0c, double v0d, double cGc, double cGd, double dGd) {
    return c00 + (bn * bn - 1) * gam + 2 * bn * (v0c + al * v0d) + ((cGc + 2 * al * cGd) + al * al * dGd);
}

// Linearisation at the current iterate, one lane per (instance, keypoint): C_k = J'QJ + L, r_k = J'Q e + L ql (System::fpBatch +
// BatchILQRCP.cpp:129-133).  cost0 is formed in the first iteration only; afterwards it is the cost of the accepted trial (a.cost).
template <class S, int MC>
__global__ __launch_bounds__(16) void k_wl_linearize(Bufs a, WArgs c) {
    constexpr int NX = S::NX;
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * 16 + threadIdx.x, kpi = blockIdx.y;
    if (b >= d.B || !a.active[b]) return;
    const int Bp = d.Bp, m = c.m;
    double cv[MC], dv[MC];
    UNR for (int j = 0; j < MC; j++) { cv[j] = (j < m) ? AT(c.cv, j, b) : 0.0; dv[j] = 0; }
    const double beta = c.beta[b];
    {
        double x[NX], xp[NX], lxx[NX * NX], lx[NX], Ld[NX], ql[NX];
        wl_states<S, MC>(d, c, b, kpi, beta, cv, dv, 0.0, x, xp);
        w_kp_derivs<S>(&d, &a, b, kpi, x, lxx, lx);
        if (d.kp_t[kpi] > 0) limit_terms<S>(d, xp, Ld, ql);
        else { UNR for (int r = 0; r < NX; r++) { Ld[r] = 0; ql[r] = 0; } }
        double* Ck = c.Ckp + (size_t)kpi * NX * NX * Bp;
        UNR for (int r = 0; r < NX; r++) {
            UNR for (int s = 0; s < NX; s++) AT(Ck, r * NX + s, b) = lxx[r * NX + s] + ((r == s) ? Ld[r] : 0.0);
            AT(c.rkp, kpi * NX + r, b) = -lx[r] + Ld[r] * ql[r];
        }
    }
    if (kpi != 0 || c.it != 0) return;
    double v0c = 0, cGc = 0;
    for (int i = 0; i < m; i++) {
        double s = 0;
        UNR for (int j = 0; j < MC; j++)
            if (j < m) s += c.G[(size_t)i * m + j] * cv[j];
        cGc += AT(c.cv, i, b) * s;
        v0c += AT(c.v0, i, b) * AT(c.cv, i, b);
    }
    a.cost[b] = wl_task_cost<S, MC>(d, a, c, b, beta, cv, dv, 0.0) + wl_uru(AT(c.scal, 0, b), AT(c.scal, 1, b), beta, 0.0, v0c, 0.0, cGc, 0.0, 0.0);
}

// (I + C G) d = (r - c) + beta C v0 with one wave per instance: the m x (m+1) system in LDS, lane q owns column q during the
// partial-pivot elimination (column m = right-hand side).  Leaves d and the scalars the line search needs:
//   sc = { v0.c, c'Gc, v0.d, c'Gd, d'Gd, ||PSI dw||^2 },  dw = Z d - beta y0
template <class S, int MC>
__global__ __launch_bounds__(64) void k_wl_solve(Bufs a, WArgs c) {
    constexpr int NX = S::NX;
    __shared__ double Ms[MC][MC + 2], xs[MC], cs[MC], t1[MC], t2[MC], t3[MC];
    __shared__ int prS;
    const DevDesc& d = *a.desc;
    const int lane = threadIdx.x, b = blockIdx.x;
    if (!a.active[b]) return;  // uniform
    const int Bp = d.Bp, m = c.m;
    const double beta = c.beta[b];
    for (int e = lane; e < m * (m + 1); e += 64) {
        const int row = e / (m + 1), col = e % (m + 1), kpi = row / NX, i = row % NX;
        const double* Ck = c.Ckp + (size_t)kpi * NX * NX * Bp;
        double s;
        if (col == m) {
            double cv0 = 0;
            UNR for (int j = 0; j < NX; j++) cv0 += AT(Ck, i * NX + j, b) * AT(c.v0, kpi * NX + j, b);
            s = (AT(c.rkp, row, b) - AT(c.cv, row, b)) + beta * cv0;
        } else {
            s = (row == col) ? 1.0 : 0.0;
            UNR for (int j = 0; j < NX; j++) s += AT(Ck, i * NX + j, b) * c.G[(size_t)(kpi * NX + j) * m + col];
        }
        Ms[row][col] = s;
    }
    if (lane < m) cs[lane] = AT(c.cv, lane, b);
    __syncthreads();
    for (int k = 0; k < m; k++) {
        if (lane == 0) {
            int pr = k;
            double pv = fabs(Ms[k][k]);
            for (int i = k + 1; i < m; i++) {
                const double v = fabs(Ms[i][k]);
                if (v > pv) { pv = v; pr = i; }
            }
            prS = pr;
        }
        __syncthreads();
        const int pr = prS;
        if (lane <= m && lane >= k && pr != k) { const double t0 = Ms[k][lane]; Ms[k][lane] = Ms[pr][lane]; Ms[pr][lane] = t0; }
        __syncthreads();
        if (lane <= m && lane > k) {
            const double piv = Ms[k][k], mk = Ms[k][lane];
            for (int i = k + 1; i < m; i++) Ms[i][lane] -= (Ms[i][k] / piv) * mk;
        }
        __syncthreads();
    }
    if (lane == 0) {
        for (int i = m - 1; i >= 0; i--) {
            double s = Ms[i][m];
            for (int q = i + 1; q < m; q++) s -= Ms[i][q] * xs[q];
            xs[i] = s / Ms[i][i];
        }
    }
    __syncthreads();
    if (lane < m) {  // rows of ZPZ d, G d, G c
        double sz = 0, sgd = 0, sgc = 0;
        for (int j = 0; j < m; j++) {
            sz += c.ZPZ[(size_t)lane * m + j] * xs[j];
            sgd += c.G[(size_t)lane * m + j] * xs[j];
            sgc += c.G[(size_t)lane * m + j] * cs[j];
        }
        t1[lane] = sz; t2[lane] = sgd; t3[lane] = sgc;
        AT(c.dvb, lane, b) = xs[lane];
    }
    __syncthreads();
    if (lane == 0) {
        double dZd = 0, dp0 = 0, v0d = 0, cGd = 0, dGd = 0, v0c = 0, cGc = 0;
        for (int i = 0; i < m; i++) {
            const double di = xs[i], ci = cs[i], v0i = AT(c.v0, i, b);
            dZd += di * t1[i];
            dGd += di * t2[i];
            cGd += di * t3[i];
            cGc += ci * t3[i];
            dp0 += di * AT(c.p0, i, b);
            v0d += di * v0i;
            v0c += v0i * ci;
        }
        AT(c.sc, 0, b) = v0c; AT(c.sc, 1, b) = cGc; AT(c.sc, 2, b) = v0d; AT(c.sc, 3, b) = cGd; AT(c.sc, 4, b) = dGd;
        AT(c.sc, 5, b) = dZd - 2 * beta * dp0 + beta * beta * AT(c.scal, 2, b);
    }
}

// Backtracking with all step sizes at once: 16 lanes per instance, lane l tries alpha = 2^-l; the first lane whose cost improves
// (or lane 10, alpha < 1e-3) wins (BatchILQRCP.cpp:138-158).
template <class S, int MC>
__global__ __launch_bounds__(64) void k_wl_linesearch(Bufs a, WArgs c) {
    const DevDesc& d = *a.desc;
    const int lane = threadIdx.x, l = lane & 15, bq = blockIdx.x * 4 + (lane >> 4);
    const bool ok = bq < d.B && a.active[bq < d.B ? bq : 0];
    const int b = ok ? bq : 0;
    const int Bp = d.Bp, m = c.m;
    double cv[MC], dv[MC];
    UNR for (int j = 0; j < MC; j++) { cv[j] = (j < m) ? AT(c.cv, j, b) : 0.0; dv[j] = (j < m) ? AT(c.dvb, j, b) : 0.0; }
    const double beta = c.beta[b], cost0 = a.cost[b];
    const double alpha = ldexp(1.0, -(l < 11 ? l : 10)), bn = (1 - alpha) * beta;
    const double uru = wl_uru(AT(c.scal, 0, b), AT(c.scal, 1, b), bn, alpha, AT(c.sc, 0, b), AT(c.sc, 2, b), AT(c.sc, 1, b), AT(c.sc, 3, b), AT(c.sc, 4, b));
    const double cost = wl_task_cost<S, MC>(d, a, c, b, bn, cv, dv, alpha) + uru;
    const bool take = (l < 11) && ((cost < cost0) || (alpha < 1e-3));
    const unsigned long long mk = __ballot(take ? 1 : 0);
    const int win = __ffs((unsigned)((mk >> (lane & 48)) & 0xffffull)) - 1;
    if (!ok || l != win) return;
    UNR for (int j = 0; j < MC; j++)
        if (j < m) AT(c.cv, j, b) = cv[j] + alpha * dv[j];
    c.beta[b] = bn;
    a.alpha[b] = alpha;
    a.iters[b] = c.it + 1;
    a.status[b] = (isfinite(cost) ? 0 : 1) | ((alpha < 1e-3) ? 2 : 0);
    if (a.cost_trace) {
        a.cost_trace[(size_t)c.it * Bp + b] = cost0;  // the reference prints the PRE-step cost (BatchILQRCP.cpp:160)
        a.alpha_trace[(size_t)c.it * Bp + b] = alpha;
    }
    a.cost[b] = cost;
    const double dun2 = AT(c.sc, 5, b);
    if (c.early_stop && alpha * sqrt(dun2 > 0 ? dun2 : 0.0) < 1e-3) a.active[b] = 0;  // :167
}

// u = u0 + (beta - 1) u0^ + (PSI Z) c
template <class S, int MC>
__global__ __launch_bounds__(64) void k_wl_controls(Bufs a, WArgs c) {
    constexpr int NU = S::NU;
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * 64 + threadIdx.x, s = blockIdx.y;
    if (b >= d.B) return;
    const int Bp = d.Bp, m = c.m;
    double cv[MC];
    UNR for (int j = 0; j < MC; j++) cv[j] = (j < m) ? AT(c.cv, j, b) : 0.0;
    const double bm1 = c.beta[b] - 1;
    UNR for (int i = 0; i < NU; i++) {
        const double* pz = c.PZ + (size_t)(s * NU + i) * m;
        double du = 0;
        UNR for (int j = 0; j < MC; j++)
            if (j < m) du += pz[j] * cv[j];
        AT(a.U[0], s * NU + i, b) = (AT(a.U0, s * NU + i, b) + bm1 * AT(c.u0hat, s * NU + i, b)) + du;
    }
}

// rollout of the solution so that X can be read back
template <class S>
__global__ __launch_bounds__(64) void k_wide_final(Bufs a) {
    constexpr int NX = S::NX, NU = S::NU;
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= d.B) return;
    const int Bp = d.Bp, T = d.T;
    double x[NX], u[NU], xn[NX];
    init_state<S>(d, a, b, x);
    for (int s = 0; s < T - 1; s++) {
        UNR for (int i = 0; i < NX; i++) AT(a.X[0], s * NX + i, b) = x[i];
        UNR for (int i = 0; i < NU; i++) u[i] = AT(a.U[0], s * NU + i, b);
        dyn_step<S>(d, x, u, xn);
        UNR for (int i = 0; i < NX; i++) x[i] = xn[i];
    }
    UNR for (int i = 0; i < NX; i++) AT(a.X[0], (T - 1) * NX + i, b) = x[i];
}

// projection of u0 on the basis, u0^ = PSI H0^-1 PSI'R u0, for a general wide PSI (three thin products with shared matrices)
__global__ void k_w_g0(const double* __restrict__ psi, const double* __restrict__ U0, const DevDesc* dp, double* __restrict__ g0, int N, int Kw, int nu) {
    const DevDesc& d = *dp;
    const int b = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y;
    if (b >= d.B) return;
    const int Bp = d.Bp;
    double s = 0;
    for (int k = 0; k < N; k++) s += psi[(size_t)k * Kw + q] * d.R_diag[k % nu] * AT(U0, k, b);
    AT(g0, q, b) = s;
}
__global__ void k_w_matvec(const double* __restrict__ M, const double* __restrict__ in, const DevDesc* dp, double* __restrict__ out, int rows, int cols) {
    const DevDesc& d = *dp;
    const int b = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (b >= d.B || r >= rows) return;
    const int Bp = d.Bp;
    double s = 0;
    for (int q = 0; q < cols; q++) s += M[(size_t)r * cols + q] * AT(in, q, b);
    AT(out, r, b) = s;
}

// ------------------------------------------------------------------------------------------------ time systems, identity basis

struct WTArgs {
    double *Ckp, *rkp, *dun2;
    int m, it, early_stop;
};

// System::fpBatch of the current controls (System.cpp:181-211): X out, C_k / r_k at the keypoint steps, cost0
template <class S>
__global__ __launch_bounds__(64) void k_wt_roll(Bufs a, WTArgs c) {
    constexpr int NX = S::NX, NU = S::NU;
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= d.B || !a.active[b]) return;
    const int Bp = d.Bp, T = d.T;
    double x[NX], xp[NX], u[NU], xn[NX];
    init_state<S>(d, a, b, x);
    UNR for (int i = 0; i < NX; i++) xp[i] = x[i];
    double cost_e = 0, cost_u = 0, cost_l = 0;
    int kpi = 0;
    auto record = [&](int i) {
        double lxx[NX * NX], lx[NX], Ld[NX], ql[NX], xt[NX];
        UNR for (int r = 0; r < NX; r++) xt[r] = x[r];
        w_kp_derivs<S>(&d, &a, b, kpi, xt, lxx, lx);
        if (i > 0) limit_terms<S>(d, xp, Ld, ql);
        else { UNR for (int r = 0; r < NX; r++) { Ld[r] = 0; ql[r] = 0; } }
        cost_e += w_kp_cost<S>(&d, a.kp_tg, b, kpi, xt);
        UNR for (int r = 0; r < NX; r++) cost_l += ql[r] * Ld[r] * ql[r];
        double* Ck = c.Ckp + (size_t)kpi * NX * NX * Bp;
        UNR for (int r = 0; r < NX; r++) {
            UNR for (int s = 0; s < NX; s++) AT(Ck, r * NX + s, b) = lxx[r * NX + s] + ((r == s) ? Ld[r] : 0.0);
            AT(c.rkp, kpi * NX + r, b) = -lx[r] + Ld[r] * ql[r];
        }
        kpi++;
    };
    if (kpi < d.n_kp && d.kp_t[kpi] == 0) record(0);
    for (int s = 0; s < T - 1; s++) {
        UNR for (int i = 0; i < NX; i++) AT(a.X[0], s * NX + i, b) = x[i];
        UNR for (int i = 0; i < NU; i++) { u[i] = AT(a.U[0], s * NU + i, b); cost_u += u[i] * d.R_diag[i] * u[i]; }
        dyn_step<S>(d, x, u, xn);
        UNR for (int i = 0; i < NX; i++) { xp[i] = x[i]; x[i] = xn[i]; }
        if (kpi < d.n_kp && d.kp_t[kpi] == s + 1) record(s + 1);
    }
    UNR for (int i = 0; i < NX; i++) AT(a.X[0], (T - 1) * NX + i, b) = x[i];
    a.cost[b] = cost_e + cost_u + cost_l;  // cost0 of this iteration (BatchILQR.cpp:135)
}

// entry (r, cc) of B_j, the linearisation of the step x_{j-1}, u_{j-1} -> x_j (PosOrnTimePlannerSys.cpp:149-185; the 2nd-order
// time column uses the velocity AFTER the step)
template <class S>
ILQR_DEV double wt_bj(int r, int cc, const double* up, const double* xj) {
    constexpr int NX = S::NX, NU = S::NU;
    const double dts = up[NU - 1], dt = dts * dts;
    if (r == NX - 1) return cc == NU - 1 ? 2 * dts : 0.0;
    if (S::ND == 1) {
        if (cc == NU - 1) return 2 * dts * up[r];
        return r == cc ? dt : 0.0;
    }
    if (r < DOF) {
        if (cc == NU - 1) return 2 * dts * xj[DOF + r] + 2 * dts * dts * dts * up[r];
        return r == cc ? dt * dt / 2 : 0.0;
    }
    if (cc == NU - 1) return 2 * dts * up[r - DOF];
    return (r - DOF) == cc ? dt : 0.0;
}

// One wave per instance.  G = sum_j V_j R^-1 V_j' and V u over the column blocks of the reference's Su (block j: Phi_{k,j} B_j for
// 1 <= j <= t_k - 1, Phi = I + (tau_{t_k-1} - tau_j) E), then (I + C G) kappa = r + C V u in LDS, then du_j = R^-1 V_j' kappa - u_j.
template <class S, int MC>
__global__ __launch_bounds__(64) void k_wt_solve(Bufs a, WTArgs c) {
    constexpr int NX = S::NX, NU = S::NU, GE = MC * MC / 64;
    __shared__ double Vj[MC][NU], Gs[MC][MC + 1], Ms[MC][MC + 2], vuS[MC], kap[MC];
    __shared__ int prS;
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (!a.active[b]) return;
    const int Bp = d.Bp, T = d.T, m = c.m, nkp = d.n_kp;
    const double* X = a.X[0];
    const double* U = a.U[0];
    double gacc[GE], vu = 0, tk[MAX_KP], rinv[NU];
    UNR for (int i = 0; i < GE; i++) gacc[i] = 0;
    UNR for (int i = 0; i < NU; i++) rinv[i] = 1.0 / d.R_diag[i];
    int jmax = 0;
    for (int k = 0; k < nkp; k++) {
        const int t = d.kp_t[k];
        tk[k] = t >= 1 ? AT(X, (t - 1) * NX + NX - 1, b) : 0.0;
        if (t - 1 > jmax) jmax = t - 1;
    }
    for (int j = 1; j <= jmax && j <= T - 2; j++) {
        double up[NU], uj[NU], xj[NX];
        UNR for (int i = 0; i < NU; i++) { up[i] = AT(U, (j - 1) * NU + i, b); uj[i] = AT(U, j * NU + i, b); }
        UNR for (int i = 0; i < NX; i++) xj[i] = AT(X, j * NX + i, b);
        for (int idx = lane; idx < m * NU; idx += 64) {
            const int row = idx / NU, cc = idx % NU, k = row / NX, r = row % NX;
            double v = 0;
            if (d.kp_t[k] > j) {
                v = wt_bj<S>(r, cc, up, xj);
                if (S::ND == 2 && r < DOF) v += (tk[k] - xj[NX - 1]) * wt_bj<S>(DOF + r, cc, up, xj);
            }
            Vj[row][cc] = v;
        }
        __syncthreads();
        UNR for (int i = 0; i < GE; i++) {
            const int e = lane + 64 * i, row = e / MC, col = e % MC;
            if (row < m && col < m) {
                double s = 0;
                UNR for (int cc = 0; cc < NU; cc++) s += Vj[row][cc] * rinv[cc] * Vj[col][cc];
                gacc[i] += s;
            }
        }
        if (lane < m) { UNR for (int cc = 0; cc < NU; cc++) vu += Vj[lane][cc] * uj[cc]; }
        __syncthreads();
    }
    UNR for (int i = 0; i < GE; i++) {
        const int e = lane + 64 * i, row = e / MC, col = e % MC;
        if (row < m && col < m) Gs[row][col] = gacc[i];
    }
    if (lane < m) vuS[lane] = vu;
    __syncthreads();
    // M = I + C G | rhs = r + C (V u)
    for (int e = lane; e < m * (m + 1); e += 64) {
        const int row = e / (m + 1), col = e % (m + 1), k = row / NX, i = row % NX;
        const double* Ck = c.Ckp + (size_t)k * NX * NX * Bp;
        double s = (col == m) ? AT(c.rkp, row, b) : ((row == col) ? 1.0 : 0.0);
        UNR for (int jj = 0; jj < NX; jj++) s += AT(Ck, i * NX + jj, b) * ((col == m) ? vuS[k * NX + jj] : Gs[k * NX + jj][col]);
        Ms[row][col] = s;
    }
    __syncthreads();
    for (int k = 0; k < m; k++) {  // LU with partial pivoting, lane q owns column q (column m = right-hand side)
        if (lane == 0) {
            int pr = k;
            double pv = fabs(Ms[k][k]);
            for (int i = k + 1; i < m; i++) {
                const double v = fabs(Ms[i][k]);
                if (v > pv) { pv = v; pr = i; }
            }
            prS = pr;
        }
        __syncthreads();
        const int pr = prS;
        if (lane <= m && lane >= k && pr != k) { const double t0 = Ms[k][lane]; Ms[k][lane] = Ms[pr][lane]; Ms[pr][lane] = t0; }
        __syncthreads();
        if (lane <= m && lane > k) {
            const double piv = Ms[k][k], mk = Ms[k][lane];
            for (int i = k + 1; i < m; i++) Ms[i][lane] -= (Ms[i][k] / piv) * mk;
        }
        __syncthreads();
    }
    if (lane == 0) {
        for (int i = m - 1; i >= 0; i--) {
            double s = Ms[i][m];
            for (int q = i + 1; q < m; q++) s -= Ms[i][q] * kap[q];
            kap[i] = s / Ms[i][i];
        }
    }
    __syncthreads();
    // du (a.U[1]) and ||du||^2; lane = column block
    double dn = 0;
    for (int j = lane; j <= T - 2; j += 64) {
        double lam[NX], uj[NU], t[NU];
        UNR for (int r = 0; r < NX; r++) lam[r] = 0;
        UNR for (int i = 0; i < NU; i++) { uj[i] = AT(U, j * NU + i, b); t[i] = 0; }
        if (j >= 1 && j <= jmax) {
            double up[NU], xj[NX];
            UNR for (int i = 0; i < NU; i++) up[i] = AT(U, (j - 1) * NU + i, b);
            UNR for (int i = 0; i < NX; i++) xj[i] = AT(X, j * NX + i, b);
            for (int k = 0; k < nkp; k++) {  // lambda = sum_k Phi_{k,j}' kappa_k
                if (d.kp_t[k] <= j) continue;
                const double dl = tk[k] - xj[NX - 1];
                UNR for (int r = 0; r < NX; r++) lam[r] += kap[k * NX + r];
                if (S::ND == 2) { UNR for (int r = 0; r < DOF; r++) lam[DOF + r] += dl * kap[k * NX + r]; }
            }
            UNR for (int cc = 0; cc < NU; cc++) {  // t = B_j' lambda
                double s = 0;
                UNR for (int r = 0; r < NX; r++) s += wt_bj<S>(r, cc, up, xj) * lam[r];
                t[cc] = s;
            }
        }
        UNR for (int cc = 0; cc < NU; cc++) {
            const double du = t[cc] * rinv[cc] - uj[cc];
            AT(a.U[1], j * NU + cc, b) = du;
            dn += du * du;
        }
    }
    for (int o = 32; o > 0; o >>= 1) dn += __shfl_xor(dn, o);
    if (lane == 0) c.dun2[b] = dn;
}

template <class S>
__global__ void k_wt_init(Bufs a) {
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= d.B) return;
    const int Bp = d.Bp;
    for (int r = 0; r < (d.T - 1) * S::NU; r++) AT(a.U[0], r, b) = AT(a.U0, r, b);
    a.cur[b] = 0;
    a.active[b] = 1;
    a.iters[b] = 0;
    a.status[b] = 0;
    a.alpha[b] = 1.0;
    a.pend[b] = 0;
    a.pred[b] = 0;
}

// ------------------------------------------------------------------------------------------------ host side

template <class T>
static bool w_alloc(BatchWideState& st, T** p, size_t n, hipStream_t s) {
    void* q = nullptr;
    if (hipMalloc(&q, (n ? n : 1) * sizeof(T)) != hipSuccess) return false;
    st.allocs.push_back(q);
    (void)hipMemsetAsync(q, 0, (n ? n : 1) * sizeof(T), s);
    *p = (T*)q;
    return true;
}

void batchwide_free(BatchWideState& st) {
    for (void* q : st.allocs) (void)hipFree(q);
    st = BatchWideState();
}

static bool upload(double* dst, const std::vector<double>& v, hipStream_t s) {
    return hipMemcpyAsync(dst, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, s) == hipSuccess;
}

// Cholesky inverse of a symmetric positive matrix (row-major n x n); false if it is not positive
static bool spd_inverse(std::vector<double>& A, int n) {
    std::vector<double> L((size_t)n * n, 0.0);
    for (int j = 0; j < n; j++) {
        double s = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) s -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
        if (!(s > 0)) return false;
        const double ljj = std::sqrt(s);
        L[(size_t)j * n + j] = ljj;
        for (int i = j + 1; i < n; i++) {
            double t = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) t -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
            L[(size_t)i * n + j] = t / ljj;
        }
    }
    std::vector<double> Li((size_t)n * n, 0.0);  // L^-1 (lower)
    for (int c = 0; c < n; c++) {
        Li[(size_t)c * n + c] = 1.0 / L[(size_t)c * n + c];
        for (int i = c + 1; i < n; i++) {
            double t = 0;
            for (int k = c; k < i; k++) t -= L[(size_t)i * n + k] * Li[(size_t)k * n + c];
            Li[(size_t)i * n + c] = t / L[(size_t)i * n + i];
        }
    }
    for (int i = 0; i < n; i++)  // A^-1 = L^-T L^-1
        for (int j = 0; j <= i; j++) {
            double t = 0;
            for (int k = i; k < n; k++) t += Li[(size_t)k * n + i] * Li[(size_t)k * n + j];
            A[(size_t)i * n + j] = A[(size_t)j * n + i] = t;
        }
    return true;
}

// LTI systems.  psi == nullptr: identity basis.
template <class S, int MC>
static int run_wl(BatchWideState& st, const DevDesc& h, Bufs& bufs, const double* psi, int Kw, int nb_iter, int early_stop, bool u0_zero,
                  hipStream_t stream, std::string& err) {
    constexpr int NX = S::NX, NU = S::NU, ND = S::ND;
    const int B = h.B, Bp = h.Bp, T = h.T, nkp = h.n_kp, m = nkp * NX, N = (T - 1) * NU;
    const bool ident = psi == nullptr;
    const double dt = h.dt, hdt2 = dt * dt / 2;
    for (int i = 0; i < NU; i++)
        if (!(h.R_diag[i] > 0)) { err = "wide-basis batch solve: the control penalty RtDiag must be positive"; return 1; }

    // the shared tables depend on (system shape, dt, keypoint steps, R, PSI) only: keep them on the device across solves
    std::vector<double> sig = {(double)h.kind, (double)ND, (double)T, dt, (double)nkp, (double)Kw, ident ? 1.0 : 0.0};
    for (int t = 0; t < nkp; t++) sig.push_back(h.kp_t[t]);
    for (int i = 0; i < NU; i++) sig.push_back(h.R_diag[i]);
    const long long key = (((long long)m * 100003 + N) * 100003 + Bp) * 8 + h.kind * 2 + (ident ? 1 : 0) + (long long)Kw * 1000000007LL;
    const bool fresh = st.key != key;
    if (fresh) {
        batchwide_free(st);
        bool ok = w_alloc(st, &st.G, (size_t)m * m, stream) && w_alloc(st, &st.ZPZ, (size_t)m * m, stream) && w_alloc(st, &st.Et, (size_t)nkp * 2 * NX * m, stream) &&
                  w_alloc(st, &st.PZ, (size_t)N * m, stream) && w_alloc(st, &st.xbk, (size_t)nkp * 2 * NX * Bp, stream) &&
                  w_alloc(st, &st.av, (size_t)nkp * 2 * NX * Bp, stream) && w_alloc(st, &st.v0, (size_t)m * Bp, stream) && w_alloc(st, &st.p0, (size_t)m * Bp, stream) &&
                  w_alloc(st, &st.scal, (size_t)3 * Bp, stream) && w_alloc(st, &st.cv, (size_t)m * Bp, stream) && w_alloc(st, &st.beta, (size_t)Bp, stream) &&
                  w_alloc(st, &st.Ckp, (size_t)nkp * NX * NX * Bp, stream) && w_alloc(st, &st.rkp, (size_t)m * Bp, stream) &&
                  w_alloc(st, &st.dvb, (size_t)m * Bp, stream) && w_alloc(st, &st.sc, (size_t)6 * Bp, stream);
        if (ok && !ident)
            ok = w_alloc(st, &st.u0hat, (size_t)N * Bp, stream) && w_alloc(st, &st.g0, (size_t)Kw * Bp, stream) && w_alloc(st, &st.y0, (size_t)Kw * Bp, stream) &&
                 w_alloc(st, &st.psi, (size_t)N * Kw, stream) && w_alloc(st, &st.h0inv, (size_t)Kw * Kw, stream);
        if (!ok) { batchwide_free(st); err = "wide-basis batch solve: hipMalloc failed"; return 1; }
        st.key = key;
    }
    const bool same_tables = !fresh && st.tab_sig == sig && (ident || (st.tab_psi.size() == (size_t)N * Kw && !std::memcmp(st.tab_psi.data(), psi, sizeof(double) * N * Kw)));
    if (!same_tables) {
        // sensitivities at the keypoint steps by their recurrences: Wt_{i+1} = A Wt_i + B PSI_i from Wt_0 = 0 (true), and the
        // reference's shifted Wr_{i+1} = A Wr_i + B PSI_i from Wr_1 = 0, i >= 1 (BatchILQRCP.cpp:61-97, quirk D-1).  NX x Kw, row-major.
        auto advance = [&](std::vector<double>& W, int i) {
            if (ND == 2)
                for (int r = 0; r < DOF; r++)
                    for (int q = 0; q < Kw; q++) W[(size_t)r * Kw + q] += dt * W[(size_t)(DOF + r) * Kw + q];
            for (int r = 0; r < DOF; r++) {
                if (ident) {
                    const int q = i * NU + r;
                    if (ND == 1) W[(size_t)r * Kw + q] += dt;
                    else { W[(size_t)r * Kw + q] += hdt2; W[(size_t)(DOF + r) * Kw + q] += dt; }
                } else {
                    const double* pr = psi + (size_t)(i * NU + r) * Kw;
                    for (int q = 0; q < Kw; q++) {
                        if (ND == 1) W[(size_t)r * Kw + q] += dt * pr[q];
                        else { W[(size_t)r * Kw + q] += hdt2 * pr[q]; W[(size_t)(DOF + r) * Kw + q] += dt * pr[q]; }
                    }
                }
            }
        };
        std::vector<std::vector<double>> wt(2 * nkp, std::vector<double>((size_t)NX * Kw, 0.0));  // [kp][which]
        std::vector<double> V((size_t)m * Kw, 0.0);
        {
            std::vector<double> Wt((size_t)NX * Kw, 0.0), Wr((size_t)NX * Kw, 0.0);
            auto sample = [&](int i) {  // Wt = Wt_i, Wr = Wr_i
                for (int t = 0; t < nkp; t++) {
                    if (h.kp_t[t] == i) { wt[2 * t + 0] = Wt; std::copy(Wr.begin(), Wr.end(), V.begin() + (size_t)t * NX * Kw); }
                    if (h.kp_t[t] - 1 == i) wt[2 * t + 1] = Wt;
                }
            };
            sample(0);
            for (int i = 0; i + 1 < T; i++) {
                advance(Wt, i);
                if (i >= 1) advance(Wr, i);
                sample(i + 1);
            }
        }
        // Z = H0^-1 V' (Kw x m)
        std::vector<double> Z((size_t)Kw * m, 0.0), h0inv;
        bool diag = ident;
        if (ident) {
            for (int q = 0; q < Kw; q++)
                for (int j = 0; j < m; j++) Z[(size_t)q * m + j] = V[(size_t)j * Kw + q] / h.R_diag[q % NU];
        } else {
            h0inv.assign((size_t)Kw * Kw, 0.0);
            for (int k = 0; k < N; k++) {
                const double* pr = psi + (size_t)k * Kw;
                const double rk = h.R_diag[k % NU];
                for (int a_ = 0; a_ < Kw; a_++) {
                    if (pr[a_] == 0.0) continue;
                    const double pa = pr[a_] * rk;
                    for (int b_ = 0; b_ < Kw; b_++) h0inv[(size_t)a_ * Kw + b_] += pa * pr[b_];
                }
            }
            if (!spd_inverse(h0inv, Kw)) { err = "wide-basis batch solve: PSI'R PSI is not positive definite (PSI must have full column rank)"; return 1; }
            for (int q = 0; q < Kw; q++)
                for (int j = 0; j < m; j++) {
                    double s = 0;
                    for (int r = 0; r < Kw; r++) s += h0inv[(size_t)q * Kw + r] * V[(size_t)j * Kw + r];
                    Z[(size_t)q * m + j] = s;
                }
        }
        (void)diag;
        std::vector<double> G((size_t)m * m, 0.0), Et((size_t)nkp * 2 * NX * m, 0.0), PZ((size_t)N * m, 0.0), ZPZ((size_t)m * m, 0.0);
        for (int i = 0; i < m; i++)
            for (int q = 0; q < Kw; q++) {
                const double v = V[(size_t)i * Kw + q];
                if (v == 0.0) continue;
                for (int j = 0; j < m; j++) G[(size_t)i * m + j] += v * Z[(size_t)q * m + j];
            }
        for (int i = 0; i < m; i++)  // symmetric in exact arithmetic; the kernel relies on it
            for (int j = 0; j < i; j++) G[(size_t)i * m + j] = G[(size_t)j * m + i] = 0.5 * (G[(size_t)i * m + j] + G[(size_t)j * m + i]);
        for (int t = 0; t < 2 * nkp; t++)
            for (int r = 0; r < NX; r++)
                for (int q = 0; q < Kw; q++) {
                    const double v = wt[t][(size_t)r * Kw + q];
                    if (v == 0.0) continue;
                    for (int j = 0; j < m; j++) Et[((size_t)t * NX + r) * m + j] += v * Z[(size_t)q * m + j];
                }
        if (ident) PZ = Z;
        else
            for (int k = 0; k < N; k++)
                for (int q = 0; q < Kw; q++) {
                    const double v = psi[(size_t)k * Kw + q];
                    if (v == 0.0) continue;
                    for (int j = 0; j < m; j++) PZ[(size_t)k * m + j] += v * Z[(size_t)q * m + j];
                }
        for (int k = 0; k < N; k++)
            for (int i = 0; i < m; i++) {
                const double v = PZ[(size_t)k * m + i];
                if (v == 0.0) continue;
                for (int j = 0; j < m; j++) ZPZ[(size_t)i * m + j] += v * PZ[(size_t)k * m + j];
            }
        bool up = upload(st.G, G, stream) && upload(st.ZPZ, ZPZ, stream) && upload(st.Et, Et, stream) && upload(st.PZ, PZ, stream);
        if (up && !ident)
            up = hipMemcpyAsync(st.psi, psi, (size_t)N * Kw * sizeof(double), hipMemcpyHostToDevice, stream) == hipSuccess && upload(st.h0inv, h0inv, stream);
        if (!up || hipStreamSynchronize(stream) != hipSuccess) { err = "wide-basis batch solve: table upload failed"; return 1; }
        st.tab_sig = sig;
        if (!ident) st.tab_psi.assign(psi, psi + (size_t)N * Kw);
    }

    WArgs c;
    c.G = st.G; c.Et = st.Et; c.ZPZ = st.ZPZ; c.PZ = st.PZ;
    c.xbk = st.xbk; c.av = st.av; c.v0 = st.v0; c.p0 = st.p0; c.scal = st.scal; c.cv = st.cv; c.beta = st.beta;
    c.Ckp = st.Ckp; c.rkp = st.rkp; c.dvb = st.dvb; c.sc = st.sc;
    c.u0hat = ident ? bufs.U0 : st.u0hat;
    c.m = m; c.it = 0; c.early_stop = early_stop;
    const dim3 grid((B + 63) / 64), block(64);
    if (!ident && u0_zero) {  // the projection of zero controls
        if (hipMemsetAsync(st.u0hat, 0, (size_t)N * Bp * sizeof(double), stream) != hipSuccess) { err = "wide-basis batch solve: memset failed"; return 1; }
    } else if (!ident) {  // u0^ = PSI (H0^-1 (PSI'R u0))
        hipLaunchKernelGGL(k_w_g0, dim3((B + 63) / 64, Kw), block, 0, stream, st.psi, bufs.U0, bufs.desc, st.g0, N, Kw, NU);
        hipLaunchKernelGGL(k_w_matvec, dim3((B + 63) / 64, Kw), block, 0, stream, st.h0inv, st.g0, bufs.desc, st.y0, Kw, Kw);
        hipLaunchKernelGGL(k_w_matvec, dim3((B + 63) / 64, N), block, 0, stream, st.psi, st.y0, bufs.desc, st.u0hat, N, Kw);
    }
    hipLaunchKernelGGL((k_wl_init<S, MC>), dim3((B + 64 / MC - 1) / (64 / MC)), dim3(64), 0, stream, bufs, c);
    for (int it = 0; it < nb_iter; it++) {
        c.it = it;
        hipLaunchKernelGGL((k_wl_linearize<S, MC>), dim3((B + 15) / 16, nkp), dim3(16), 0, stream, bufs, c);
        hipLaunchKernelGGL((k_wl_solve<S, MC>), dim3(B), dim3(64), 0, stream, bufs, c);
        hipLaunchKernelGGL((k_wl_linesearch<S, MC>), dim3((B + 3) / 4), dim3(64), 0, stream, bufs, c);
    }
    hipLaunchKernelGGL((k_wl_controls<S, MC>), dim3((B + 63) / 64, T - 1), block, 0, stream, bufs, c);
    hipLaunchKernelGGL((k_wide_final<S>), grid, block, 0, stream, bufs);
    if (hipGetLastError() != hipSuccess) { err = "wide-basis batch solve: kernel launch failed"; return 1; }
    return 0;
}

template <class S>
static int run_wl_m(BatchWideState& st, const DevDesc& h, Bufs& bufs, const double* psi, int Kw, int nb_iter, int early_stop, bool u0_zero,
                    hipStream_t stream, std::string& err) {
    const int m = h.n_kp * S::NX;
    if (m <= 16) return run_wl<S, 16>(st, h, bufs, psi, Kw, nb_iter, early_stop, u0_zero, stream, err);
    if (m <= 32) return run_wl<S, 32>(st, h, bufs, psi, Kw, nb_iter, early_stop, u0_zero, stream, err);
    err = "wide-basis batch solve: n_keypoints * n_x must not exceed 32";
    return 1;
}

// time systems, identity basis
template <class S, int MC>
static int run_wt(BatchWideState& st, const DevDesc& h, Bufs& bufs, int nb_iter, int early_stop, hipStream_t stream, std::string& err) {
    constexpr int NX = S::NX, NU = S::NU;
    const int B = h.B, Bp = h.Bp, nkp = h.n_kp, m = nkp * NX;
    for (int i = 0; i < NU; i++)
        if (!(h.R_diag[i] > 0)) { err = "wide-basis batch solve: the control penalty RtDiag must be positive"; return 1; }
    const long long key = -2 - ((((long long)m * 100003 + Bp) * 8) + h.kind * 2 + h.nd);
    if (st.key != key) {
        batchwide_free(st);
        if (!(w_alloc(st, &st.Ckp, (size_t)nkp * NX * NX * Bp, stream) && w_alloc(st, &st.rkp, (size_t)m * Bp, stream) && w_alloc(st, &st.scal, (size_t)Bp, stream))) {
            batchwide_free(st);
            err = "wide-basis batch solve: hipMalloc failed";
            return 1;
        }
        st.key = key;
    }
    WTArgs c;
    c.Ckp = st.Ckp; c.rkp = st.rkp; c.dun2 = st.scal; c.m = m; c.it = 0; c.early_stop = early_stop;
    const dim3 grid((B + 63) / 64), block(64);
    hipLaunchKernelGGL((k_wt_init<S>), dim3((B + 255) / 256), dim3(256), 0, stream, bufs);
    for (int it = 0; it < nb_iter; it++) {
        c.it = it;
        hipLaunchKernelGGL((k_wt_roll<S>), grid, block, 0, stream, bufs, c);
        hipLaunchKernelGGL((k_wt_solve<S, MC>), dim3(B), block, 0, stream, bufs, c);
        BTArgs bt;
        bt.it = it; bt.early_stop = early_stop;
        hipLaunchKernelGGL((k_bt_linesearch<S>), dim3((B + 3) / 4), dim3(64), 0, stream, bufs, bt);
    }
    hipLaunchKernelGGL((k_wide_final<S>), grid, block, 0, stream, bufs);
    if (hipGetLastError() != hipSuccess) { err = "wide-basis batch solve: kernel launch failed"; return 1; }
    return 0;
}

template <class S>
static int run_wt_m(BatchWideState& st, const DevDesc& h, Bufs& bufs, int nb_iter, int early_stop, hipStream_t stream, std::string& err) {
    const int m = h.n_kp * S::NX;
    if (m <= 16) return run_wt<S, 16>(st, h, bufs, nb_iter, early_stop, stream, err);
    if (m <= 32) return run_wt<S, 32>(st, h, bufs, nb_iter, early_stop, stream, err);
    err = "wide-basis batch solve: n_keypoints * n_x must not exceed 32";
    return 1;
}

int batchwide_solve(BatchWideState& st, const DevDesc& h, Bufs& bufs, int nx, int nu, const double* psi_host, int Kw, int nb_iter, int early_stop,
                    bool u0_zero, hipStream_t stream, std::string& err, const ProfHook& ph) {
    (void)nx;
    ph(ILQR_PROF_BACKWARD);  // the whole wide-basis solve is charged to one category
    const int N = (h.T - 1) * nu;
    if (nb_iter < 0) { err = "nb_iter < 0"; return 1; }
    if (h.n_kp <= 0) { err = "wide-basis batch solve: the system has no keypoint"; return 1; }
    if (!psi_host) Kw = N;
    if (Kw <= 0 || Kw > N) { err = "wide-basis batch solve: Kw must be in 1 .. (T-1) n_u"; return 1; }
    if (h.kind == 0 && h.nd == 1) return run_wl_m<Sys<0, 1>>(st, h, bufs, psi_host, Kw, nb_iter, early_stop, u0_zero, stream, err);
    if (h.kind == 0 && h.nd == 2) return run_wl_m<Sys<0, 2>>(st, h, bufs, psi_host, Kw, nb_iter, early_stop, u0_zero, stream, err);
    if (h.kind == 2) return run_wl_m<Sys<2, 1>>(st, h, bufs, psi_host, Kw, nb_iter, early_stop, u0_zero, stream, err);
    if (psi_host) { err = "wide-basis batch solve: on time systems only the identity basis (ilqr_solve_batch) is supported for Kw > 32"; return 1; }
    if (h.kind == 1 && h.nd == 1) return run_wt_m<Sys<1, 1>>(st, h, bufs, nb_iter, early_stop, stream, err);
    if (h.kind == 1 && h.nd == 2) return run_wt_m<Sys<1, 2>>(st, h, bufs, nb_iter, early_stop, stream, err);
    if (h.kind == 3) return run_wt_m<Sys<3, 1>>(st, h, bufs, nb_iter, early_stop, stream, err);
    err = "wide-basis batch solve: unknown system kind";
    return 1;
}

}  // namespace ilqr
