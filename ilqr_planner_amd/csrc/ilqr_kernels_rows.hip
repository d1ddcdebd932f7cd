// ilqr_kernels_rows.hip -- 8 lanes per instance, lane r owns control row r (gfx950, fp64):
//   k_forward_lin (+ k_blend, k_flip)   linear line search of the 2nd-order PosOrn system (ILQRRecursive.cpp:101-155)
//   k_apply_rows_tm                     time systems: re-roll of the accepted step size where the speculated one lost
// A lane's gain row K_k[r,:], d_k[r], xbar_k[r], ubar_k[r] stream straight from HBM into its registers (4-step prefetch ring, no
// staging); the only exchange per timestep is the all-gather of the state deviation through LDS.
// Workgroup = one wave = 8 instances: all exchange is wave-local (LDS operations of a wave execute in order), no barrier.
#include "ilqr_kernels.hpp"
#include "ilqr_step.hpp"

namespace ilqr {

#define LDS_ORDER() asm volatile("" ::: "memory")

// ------------------------------------------------------------------------------------------------ linear line search
//
// For the PosOrn systems the closed loop  du_k = K_k dx_k + alpha d_k,  dx_{k+1} = A dx_k + B du_k  (dx_0 = 0) is linear
// and homogeneous in alpha:  dx_k(alpha) = alpha dx_k(1),  du_k(alpha) = alpha du_k(1).  Every trial trajectory of the
// reference's step-halving search is therefore  x(alpha) = xbar + alpha (x(1) - xbar),  u(alpha) = ubar + alpha (u(1) - ubar):
// ONE rollout (alpha = 1, in deviation coordinates) serves the whole line search.  Per trial only the costs differ: the task
// cost at the (two) keypoint steps is evaluated for every alpha by lane (alpha mod 8); the limit penalty can only be non-zero
// for some alpha if it is non-zero at an end of the segment [xbar, x(1)], which is checked per step (4 compares) before the
// per-alpha evaluation is entered.  sum_k ||du_k(alpha)|| = alpha sum_k ||du_k(1)||.
// The pass writes x(1), u(1) into the inactive trajectory buffer; if the winner is another alpha, k_blend rewrites that buffer
// as xbar + alpha (x(1) - xbar) -- an elementwise, fully parallel pass with no gain traffic at all.
template <class S, int NA>
__global__ __launch_bounds__(64) void k_forward_lin(Bufs a, FwdArgs f) {
    static_assert(S::KIND == 0, "linear line search: PosOrn systems only");
    constexpr int NX = S::NX, NU = S::NU, ND = S::ND, NK = NU * NX;
    constexpr int NXP = (NX + 1) & ~1;
    constexpr int DXS = NXP + 2;                  // per-instance stride (doubles) of the dx row, +16 B bank spread
    constexpr int ROWP = kd_rowp(NX), RS = NU * ROWP;
    constexpr int NLD = ROWP + ND + 1;
    constexpr int PF = 4;                         // timesteps in flight; (PF+1) * 6 load instructions stay below the 6-bit vmcnt (63)
    __shared__ __attribute__((aligned(16))) double sDX[8 * DXS];
    __shared__ __attribute__((aligned(16))) double sXB[8][NXP + 2];
    __shared__ __attribute__((aligned(16))) double sUB[8][10];
    __shared__ __attribute__((aligned(16))) double sDU[8][10];
    __shared__ double sP[8][8][NA];
    __shared__ double sKc[8][NA];

    const DevDesc& d = *a.desc;
    const int lane = threadIdx.x, g = lane >> 3, r0 = lane & 7;
    const int b = xcd_tile() * 8 + g;
    const int Bp = d.Bp, T = d.T, B = d.B;
    const bool inst_ok = (b < B) && (a.active[b < B ? b : 0] != 0);
    if (__ballot(inst_ok ? 1 : 0) == 0ull) return;  // wave-uniform
    const int bb = (b < B) ? b : 0;
    const bool act = r0 < NU;
    const int r = act ? r0 : NU - 1;
    const int n_alpha = f.n_alpha < NA ? f.n_alpha : NA;

    const double dt = d.dt, hdt2 = dt * dt / 2;
    const int lim_on = d.limits_set;
    const double pen = d.penalty;
    double smax[ND], smin[ND];
    int lw[ND];
    UNR for (int q = 0; q < ND; q++) { smax[q] = d.smax[q * DOF + r]; smin[q] = d.smin[q * DOF + r]; lw[q] = d.lw[q * DOF + r]; }
    const int n_kp = d.n_kp;
    int kpi = 0, kp_next = (n_kp > 0) ? d.kp_t[0] : -1;
    const double my_alpha0 = ldexp(1.0, -r0), my_alpha1 = ldexp(1.0, -(r0 + 8));

    const int cur = a.cur[bb];
    const double* pK = a.KD + (size_t)bb * RS + r * ROWP;  // this lane's gain row {K[r][0..NX-1], d[r]}: 16-byte aligned, contiguous
    const double* pX = a.X[cur] + (size_t)r * Bp + bb;
    const double* pU = a.U[cur] + (size_t)r * Bp + bb;
    const size_t sK_ = (size_t)Bp * RS, sU_ = (size_t)NU * Bp, sX_ = (size_t)NX * Bp;
    double* oX = a.X[1 - cur] + (size_t)r * Bp + bb;
    double* oU = a.U[1 - cur] + (size_t)r * Bp + bb;

    constexpr int NS = PF + 1;  // ring slots: the slot consumed in step k-1 receives step k+PF-1 ... no register copies, distance PF
    double ring[NS][NLD];
    auto fetch = [&](int slot, int k) {  // unconditional (a CFG path that skips a load makes the waitcnt pass fall back to vmcnt(0)); pointers stop at the last timestep
        UNR for (int q = 0; q < ROWP / 2; q++) {  // 16-byte loads
            const double2 v2 = reinterpret_cast<const double2*>(pK)[q];
            ring[slot][2 * q] = v2.x;
            ring[slot][2 * q + 1] = v2.y;
        }
        UNR for (int q = 0; q < ND; q++) ring[slot][ROWP + q] = pX[(size_t)(q * DOF) * Bp];
        ring[slot][ROWP + ND] = *pU;
        if (k < T - 2) { pK += sK_; pX += sX_; pU += sU_; }  // uniform; no load inside the branch
    };
    double xT[ND];  // xbar_{T-1} of this lane's coordinates (terminal step)
    UNR for (int q = 0; q < ND; q++) xT[q] = AT(a.X[cur], (T - 1) * NX + q * DOF + r, bb);
    __builtin_amdgcn_sched_barrier(0);
    UNR for (int q = 0; q < PF; q++) { fetch(q, q); __builtin_amdgcn_sched_barrier(0); }  // in issue order (vmcnt)

    double dxq = 0, dxd = 0, pc[NA];
    UNR for (int al = 0; al < NA; al++) pc[al] = 0;
    double kpc0 = 0, kpc1 = 0, dun = 0;
    double* myDX = sDX + g * DXS;

    // the segment [xbar, x(1)] of a coordinate leaves its limits iff the larger end exceeds the max or the smaller end is below
    // the min; branch-free on purpose (short-circuit evaluation costs an exec-mask branch per term and step)
    double emx[ND], emn[ND];
    UNR for (int q = 0; q < ND; q++) { emx[q] = lw[q] ? smax[q] : INFINITY; emn[q] = lw[q] ? smin[q] : -INFINITY; }
    auto seg_bad = [&](const double* xb_) -> bool {
        const double e0 = xb_[0] + dxq;
        bool bad = (fmax(xb_[0], e0) > emx[0]) | (fmin(xb_[0], e0) < emn[0]);
        if (ND == 2) {
            const double e1 = xb_[1] + dxd;
            bad |= (fmax(xb_[1], e1) > emx[1]) | (fmin(xb_[1], e1) < emn[1]);
        }
        return bad;
    };
    auto limit_cost_of = [&](double v, int q) -> double {
        double qv = 0, L = 0;
        if (lw[q] != 0) {
            if (v > smax[q]) { qv = smax[q] - v; L = pen; }
            else if (v < smin[q]) { qv = smin[q] - v; L = pen; }
        }
        return qv * L * qv;
    };
    // limit penalty of the stage for every alpha; entered only when an end of the segment violates a limit somewhere in the wave
    auto limits_all = [&](const double* xb) {
        UNR for (int al = 0; al < NA; al++) {
            const double aa = ldexp(1.0, -al);
            pc[al] += limit_cost_of(fma(aa, dxq, xb[0]), 0);
            if (ND == 2) pc[al] += limit_cost_of(fma(aa, dxd, xb[1]), 1);
        }
    };
    auto kp_eval = [&](double aa, bool with_u) -> double {
        double xt[NX], ut[NU];
        UNR for (int jx = 0; jx < NX; jx++) xt[jx] = fma(aa, myDX[jx], sXB[g][jx]);
        UNR for (int ju = 0; ju < NU; ju++) ut[ju] = with_u ? fma(aa, sDU[g][ju], sUB[g][ju]) : 0.0;
        return kp_cost_call<S>(&d, a.kp_tg, Bp, bb, kpi, xt, ut);
    };

    const int nsteps = T - 1;
    // No `break` inside the unrolled group: a CFG path that skips a fetch makes the waitcnt pass assume the shortest distance
    // between a slot's loads and its use (vmcnt(8) instead of vmcnt(6 PF)).  The last group runs dummy steps whose fetches
    // re-read the last timestep and whose work is skipped.
    for (int k0 = 0; k0 < nsteps; k0 += NS) {
        UNR for (int jj = 0; jj < NS; jj++) {
            const int k = k0 + jj;
            fetch((jj + PF) % NS, k + PF);  // into the slot freed by the previous step
            if (k >= nsteps) continue;  // uniform; skips the work, never a fetch
            const double* Kr = ring[jj];
            const double dr = ring[jj][NX], ub = ring[jj][ROWP + ND];
            double xb[ND];
            UNR for (int q = 0; q < ND; q++) xb[q] = ring[jj][ROWP + q];
            // ---- all-gather of dx
            if (act) {
                myDX[r] = dxq;
                if (ND == 2) myDX[DOF + r] = dxd;
            }
            LDS_ORDER();
            double s0 = 0, s1 = 0;
            UNR for (int jx = 0; jx < NX; jx += 2) s0 += Kr[jx] * myDX[jx];
            UNR for (int jx = 1; jx < NX; jx += 2) s1 += Kr[jx] * myDX[jx];
            const double du = (s0 + s1) + dr;
            // ---- x(1), u(1) out
            if (inst_ok && act) {
                *oX = xb[0] + dxq;
                if (ND == 2) oX[(size_t)DOF * Bp] = xb[1] + dxd;
                *oU = ub + du;
            }
            oX += sX_;
            oU += sU_;
            if (f.early_stop) {  // ||du_k(1)||
                if (act) sDU[g][r] = du * du;
                LDS_ORDER();
                double n2 = 0;
                UNR for (int ju = 0; ju < NU; ju++) n2 += sDU[g][ju];
                dun += sqrt(n2);
                LDS_ORDER();
            }
            // ---- stage cost: limits (end-of-segment test first), task cost at keypoint steps
            if (lim_on) {
                if (__ballot((seg_bad(xb) & act) ? 1 : 0) != 0ull) limits_all(xb);
            }
            if (k == kp_next) {  // uniform, rare
                if (act) {
                    sXB[g][r] = xb[0];
                    if (ND == 2) sXB[g][DOF + r] = xb[1];
                    sUB[g][r] = ub;
                    sDU[g][r] = du;
                }
                LDS_ORDER();
                if (r0 < n_alpha) kpc0 += kp_eval(my_alpha0, true);
                if (NA > 8 && r0 + 8 < n_alpha) kpc1 += kp_eval(my_alpha1, true);
                LDS_ORDER();
                kpi++;
                kp_next = (kpi < n_kp) ? __builtin_amdgcn_readfirstlane(d.kp_t[kpi]) : -1;  // (scalar register: see k_forward_mfma)
            }
            // ---- deviation dynamics: dx' = A dx + B du
            if (ND == 1) {
                dxq = dxq + dt * du;
            } else {
                const double vv = dxd;
                dxq = dxq + (dt * vv + hdt2 * du);
                dxd = vv + dt * du;
            }
            LDS_ORDER();
        }
    }
    // ---- terminal step
    if (inst_ok && act) {
        *oX = xT[0] + dxq;
        if (ND == 2) oX[(size_t)DOF * Bp] = xT[1] + dxd;
    }
    if (lim_on) {
        if (__ballot((seg_bad(xT) & act) ? 1 : 0) != 0ull) limits_all(xT);
    }
    if (kp_next == T - 1) {
        if (act) {
            myDX[r] = dxq;
            sXB[g][r] = xT[0];
            if (ND == 2) { myDX[DOF + r] = dxd; sXB[g][DOF + r] = xT[1]; }
        }
        LDS_ORDER();
        if (r0 < n_alpha) kpc0 += kp_eval(my_alpha0, false);
        if (NA > 8 && r0 + 8 < n_alpha) kpc1 += kp_eval(my_alpha1, false);
        LDS_ORDER();
    }
    // ---- total cost per alpha, winner, bookkeeping (lane 0 of each instance)
    UNR for (int al = 0; al < NA; al++) sP[g][r0][al] = act ? pc[al] : 0.0;
    if (r0 < NA) sKc[g][r0] = kpc0;
    if (NA > 8 && r0 + 8 < NA) sKc[g][r0 + 8] = kpc1;
    LDS_ORDER();
    if (inst_ok && r0 == 0) {
        const double cost0 = a.cost[bb];
        int w = n_alpha - 1;
        double wcost = 0;
        bool found = false;
        UNR for (int al = 0; al < NA; al++) {
            double c = sKc[g][al];
            UNR for (int q = 0; q < 8; q++) c += sP[g][q][al];
            const bool okc = (al < n_alpha) && !((c >= cost0) || isnan(c));
            if (!found && (okc || al == n_alpha - 1)) { w = al; wcost = c; found = true; }
        }
        const double walpha = ldexp(1.0, -w);
        a.cost[bb] = wcost;
        a.alpha[bb] = walpha;
        a.iters[bb] = f.it + 1;
        a.status[bb] = (isfinite(wcost) ? 0 : 1) | ((walpha <= d.alpha_floor) ? 2 : 0);
        if (a.cost_trace) {
            a.cost_trace[(size_t)f.it * Bp + bb] = wcost;
            a.alpha_trace[(size_t)f.it * Bp + bb] = walpha;
        }
        a.pend[bb] = w + 1;  // k_blend / k_flip finish the acceptance (w == 0: the buffer already holds x(1), u(1))
        a.pred[bb] = w;
        bool stop = f.early_stop && (walpha * sqrt(walpha * dun) < d.stop_tol);  // sum ||du(alpha)|| = alpha sum ||du(1)||
        if (!f.al) stop = stop && (wcost < 1e-3);
        if (stop) a.active[bb] = 0;
    }
}

// x(alpha) = xbar + alpha (x(1) - xbar), u likewise, for the instances whose winner is not alpha = 1; one lane per (b, k)
template <class S>
__global__ __launch_bounds__(256) void k_blend(Bufs a) {
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    if (b >= d.B) return;
    const int w = a.pend[b];
    if (w <= 1) return;  // nothing pending, or alpha = 1
    const double aa = ldexp(1.0, -(w - 1));
    const int Bp = d.Bp, cur = a.cur[b];
    const double* Xb = a.X[cur];
    const double* Ub = a.U[cur];
    double* Xn = a.X[1 - cur];
    double* Un = a.U[1 - cur];
    UNR for (int i = 0; i < S::NX; i++) {
        const double xb = AT(Xb, k * S::NX + i, b);
        AT(Xn, k * S::NX + i, b) = fma(aa, AT(Xn, k * S::NX + i, b) - xb, xb);
    }
    if (k < d.T - 1) {
        UNR for (int i = 0; i < S::NU; i++) {
            const double ub = AT(Ub, k * S::NU + i, b);
            AT(Un, k * S::NU + i, b) = fma(aa, AT(Un, k * S::NU + i, b) - ub, ub);
        }
    }
}
__global__ void k_flip(Bufs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.desc->B) return;
    if (a.pend[b] > 0) {
        a.cur[b] = 1 - a.cur[b];
        a.pend[b] = 0;
    }
}

// Re-roll of the accepted step size for the TIME systems (dt = u_last^2: the closed loop is not linear in alpha, so the winner of
// the alpha-parallel pass -- k_forward_mfma -- has to be rolled out again when it was not the speculated one).  8 lanes share an instance: lane r owns control row r and the joint state (q_r, dq_r),
// lane 7 the time control and the time state.  Per step: LDS all-gather of dx (n_x values), 16-term dot product with the lane's
// gain row (register ring, 16-byte loads), dt = s^2 broadcast from lane 7, dynamics local to the lane.
template <class S>
__global__ __launch_bounds__(64) void k_apply_rows_tm(Bufs a, FwdArgs f) {
    static_assert(S::TM == 1, "time systems");
    constexpr int NX = S::NX, NU = S::NU, ND = S::ND;
    static_assert(NU == 8, "one control row per lane of an 8-lane group");
    constexpr int ROWP = kd_rowp(NX), RS = NU * ROWP;
    constexpr int DXS = ROWP + 2;   // per-instance stride of the dx image (+16 B bank spread)
    constexpr int NLD = ROWP + ND + 1;
    constexpr int PF = 3;
    __shared__ __attribute__((aligned(16))) double sDX[8 * DXS];
    __shared__ double sS[8][2];     // per instance: the time control of the step

    const DevDesc& d = *a.desc;
    const int lane = threadIdx.x, g = lane >> 3, r = lane & 7;
    const int b = xcd_tile() * 8 + g;
    const int Bp = d.Bp, T = d.T, B = d.B;
    const bool inst_ok = (b < B) && (a.pend[b < B ? b : 0] > 0);
    if (__ballot(inst_ok ? 1 : 0) == 0ull) return;  // wave-uniform
    const int bb = (b < B) ? b : 0;
    const bool isT = (r == DOF);    // lane 7: time control / time state
    const int rj = isT ? 0 : r;     // joint index for the state rows of this lane (clamped for lane 7)
    const double alpha = ldexp(1.0, -((inst_ok ? a.pend[bb] : 1) - 1));
    const int cur = a.cur[bb];
    const double* pXq = a.X[cur] + (size_t)(isT ? NX - 1 : rj) * Bp + bb;            // q_r, or t for lane 7
    const double* pXd = a.X[cur] + (size_t)(ND == 2 ? DOF + rj : rj) * Bp + bb;      // dq_r (2nd order)
    const double* pU = a.U[cur] + (size_t)r * Bp + bb;
    const size_t sK2_ = (size_t)Bp * RS / 2, sU_ = (size_t)NU * Bp, sX_ = (size_t)NX * Bp;
    double* oXq = a.X[1 - cur] + (size_t)(isT ? NX - 1 : rj) * Bp + bb;
    double* oXd = a.X[1 - cur] + (size_t)(ND == 2 ? DOF + rj : rj) * Bp + bb;
    double* oU = a.U[1 - cur] + (size_t)r * Bp + bb;

    // Gain records: the eight records of the wave's instances are 8 RS doubles in one run.  Lane (g, r) needs row r of record g -- loaded
    // as such (16-byte pieces of a 128-byte row per lane) every load instruction touches 64 different cache lines, and the address unit of
    // the CU, not the arithmetic, set the pace of a step.  Instead instruction q loads record q with lane l on bytes 16 l .. 16 l + 15 (one
    // contiguous KiB), the ring keeps those pieces, and the rows are picked out of an LDS image of the step (dropped there a step ahead;
    // LDS operations of a wave execute in order: no barrier).  Records of instances with nothing pending stay on their first timestep:
    // cache hits instead of a second pass over their gains in HBM.
    constexpr int KS = ROWP + 2;    // row stride of the image (bank spread, rows stay 16-byte aligned)
    __shared__ __attribute__((aligned(16))) double sKi[8][NU * KS];
    const int b0 = b - g;
    const bool pk = 2 * lane < RS;
    const int prow = pk ? (2 * lane) / ROWP : 0, pcol = pk ? (2 * lane) % ROWP : 0;
    double* const wKi = &sKi[0][prow * KS + pcol];
    const double* const rKi = &sKi[g][r * KS];
    const double2* const K0 = reinterpret_cast<const double2*>(a.KD + (size_t)b0 * RS) + (pk ? lane : 0);
    const double2* Kk = K0;         // records of the step being fetched
    bool okq[8];
    UNR for (int q = 0; q < 8; q++) okq[q] = (b0 + q < B) && (a.pend[b0 + q < B ? b0 + q : 0] > 0);  // wave-uniform

    double rk0[PF][8], rk1[PF][8];  // (two scalar arrays: an array of double2 is not split into registers)
    double ring[PF][ND + 1];
    auto fetch = [&](int slot, int k) {  // unconditional; the pointers stop at the last timestep
        UNR for (int q = 0; q < 8; q++) {
            const double2 v2 = (okq[q] ? Kk : K0)[(size_t)q * (RS / 2)];
            rk0[slot][q] = v2.x;
            rk1[slot][q] = v2.y;
        }
        ring[slot][0] = *pXq;
        if (ND == 2) ring[slot][1] = *pXd;
        ring[slot][ND] = *pU;
        if (k < T - 2) Kk += sK2_;  // uniform
        const size_t adv = (k < T - 2 && inst_ok) ? 1 : 0;
        pXq += adv * sX_; pXd += adv * sX_; pU += adv * sU_;
    };
    auto stage = [&](int slot) {  // pieces of ring slot -> image
        if (pk) { UNR for (int q = 0; q < 8; q++) *reinterpret_cast<double2*>(wKi + q * NU * KS) = make_double2(rk0[slot][q], rk1[slot][q]); }
    };
    UNR for (int q = 0; q < PF; q++) { fetch(q, q); __builtin_amdgcn_sched_barrier(0); }
    stage(0);
    LDS_ORDER();

    // state of this lane: joint r (q, dq) or, lane 7, the time
    double xq = isT ? 0.0 : AT(a.q0, rj, bb);
    double xd = (ND == 2 && !isT) ? AT(a.dq0, rj, bb) : 0.0;
    double* myDX = sDX + g * DXS;

    const int nsteps = T - 1;
    for (int k0 = 0; k0 < nsteps; k0 += PF) {
        UNR for (int jj = 0; jj < PF; jj++) {
            const int k = k0 + jj;
            double Kr[NX];
            UNR for (int jx = 0; jx < NX; jx++) Kr[jx] = rKi[jx];  // this lane's gain row of step k
            const double dr = rKi[NX], xbq = ring[jj][0], xbd = (ND == 2) ? ring[jj][1] : 0.0, ub = ring[jj][ND];
            LDS_ORDER();
            stage((jj + 1) % PF);                                   // step k + 1 into the image
            LDS_ORDER();
            fetch(jj, k + PF);
            if (k >= nsteps) continue;  // uniform; dummy step of the last group
            // ---- all-gather of dx = x - xbar (state order [q(7) | dq(7) | t])
            if (isT) myDX[NX - 1] = xq - xbq;
            else {
                myDX[rj] = xq - xbq;
                if (ND == 2) myDX[DOF + rj] = xd - xbd;
            }
            LDS_ORDER();
            double s0 = 0, s1 = 0;
            UNR for (int jx = 0; jx < NX; jx += 2) s0 += Kr[jx] * myDX[jx];
            UNR for (int jx = 1; jx < NX; jx += 2) s1 += Kr[jx] * myDX[jx];
            const double du = (s0 + s1) + alpha * dr;
            const double u = ub + du;
            if (isT) sS[g][0] = u;
            LDS_ORDER();
            const double dts = sS[g][0];
            const double dt = dts * dts;
            if (inst_ok) {
                *oXq = xq;
                if (ND == 2 && !isT) *oXd = xd;
                *oU = u;
            }
            oXq += sX_; oXd += sX_; oU += sU_;
            // ---- dynamics (SimulationInterface.cpp:19-31 with dt = u_last^2, PosOrnTimePlannerSys.cpp:149-184), local to the lane
            if (isT) {
                xq = xq + dt;
            } else if (ND == 1) {
                xq = xq + (dt * u + dt * dt / 2 * 0.0);
            } else {
                const double vv = xd;
                xq = xq + (dt * vv + dt * dt / 2 * u);
                xd = vv + dt * u;
            }
            LDS_ORDER();
        }
    }
    if (inst_ok) {  // x_{T-1}
        *oXq = xq;
        if (ND == 2 && !isT) *oXd = xd;
    }
    if (inst_ok && r == 0) {  // the early stop was decided by k_select_x
        a.cur[bb] = 1 - cur;
        a.pend[bb] = 0;
    }
}


// The same re-roll with 16 lanes per instance on registers, for SMALL batches (round 3; the design of k_forward_dpp, ilqr_kernels_wave.hip).  k_apply_rows_tm above is a
// chain of ~0.46 us per step at any batch size: 31 LDS reads per lane and step (gain row and the gathered dx), three LDS exchanges.  Here the chain is 0.3 us per step
// (B = 256: 64 against 92 us per launch); at large batches the older kernel wins, because its waves hold 8 adjacent instances -- 64 contiguous bytes of every
// [row][b] line of xbar, ubar, x, u against 32 here (B = 2048: 117 against 126 us, B = 4096: 190 against 240): chosen up to 1024 instances.  Here
//   lane c < n_x holds state c (q_0..q_6 | dq_0..dq_6 | t) -- its deviation from xbar is the `row_newbcast:c` operand of the FMAs;
//   lane (h, r), h = half of the 16-lane row, holds entries 8h .. 8h+7 of row r of the gain record and forms its part of du_r = alpha d_r + sum_c K_rc dx_c;
//   the halves meet by one row_ror:8 move, the time control's step s = u_7 comes out of lane 15, dq_i and u_i reach the lanes that integrate q_i and dq_i by
//   row_shl:7 / row_shl:1 moves.  No LDS, nothing couples the waves; instances with nothing pending keep re-reading their first record (cache hits).
#define AD_LO " row_mask:0xf bank_mask:0x3"
#define AD_HI " row_mask:0xf bank_mask:0xc"
template <int CTRL>
__device__ __forceinline__ double ad_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// this half's part of du_r: half 0 sum_{c<8} K[c] dx_c; half 1 the NHI - 1 remaining columns (broadcast lanes 8 ..) and al * K[NHI-1] (the feed-forward);
// three accumulators in rotation (a DPP instruction reads its accumulator early: no register is touched again within two instructions)
template <int NHI>
__device__ __forceinline__ double ad_half_dot(const double (&K)[8], double x, double al) {
    static_assert(NHI == 8 || NHI == 1, "n_x = 15 or 8");
    double s0, s1, s2;
#define L_(A, J) "v_fmac_f64_dpp %[" A "], %[x], %[k" #J "] row_newbcast:" #J AD_LO "\n\t"
#define H_(A, J, C) "v_fmac_f64_dpp %[" A "], %[x], %[k" #J "] row_newbcast:" #C AD_HI "\n\t"
#define F_(A, J) "v_fmac_f64_dpp %[" A "], %[al], %[k" #J "] row_newbcast:0" AD_HI "\n\t"
#define OPS_ : [s0] "=&v"(s0), [s1] "=&v"(s1), [s2] "=&v"(s2)                                                                              \
             : [x] "v"(x), [al] "v"(al), [k0] "v"(K[0]), [k1] "v"(K[1]), [k2] "v"(K[2]), [k3] "v"(K[3]), [k4] "v"(K[4]), [k5] "v"(K[5]), [k6] "v"(K[6]), [k7] "v"(K[7])
#define ZERO_ "v_mov_b64 %[s0], 0\n\tv_mov_b64 %[s1], 0\n\tv_mov_b64 %[s2], 0\n\ts_nop 1\n\t"
#define LO8_ L_("s0", 0) L_("s1", 1) L_("s2", 2) L_("s0", 3) L_("s1", 4) L_("s2", 5) L_("s0", 6) L_("s1", 7)
    if (NHI == 8)
        asm volatile(ZERO_ LO8_ H_("s2", 0, 8) H_("s0", 1, 9) H_("s1", 2, 10) H_("s2", 3, 11) H_("s0", 4, 12) H_("s1", 5, 13) H_("s2", 6, 14) F_("s0", 7) "s_nop 0" OPS_);
    else
        asm volatile(ZERO_ LO8_ F_("s2", 0) "s_nop 0" OPS_);
#undef L_
#undef H_
#undef F_
#undef OPS_
#undef ZERO_
#undef LO8_
    return (s0 + s1) + s2;
}

template <class S>
__global__ __launch_bounds__(64) void k_apply_dpp_tm(Bufs a, FwdArgs f) {
    static_assert(S::TM == 1 && S::NU == 8, "time systems: seven joint controls and the time control");
    constexpr int NX = S::NX, NU = S::NU, ND = S::ND;
    constexpr int ROWP = kd_rowp(NX), RS = NU * ROWP, PF = 4, IPW = 4;
    constexpr int NHI = ROWP - 8 >= 8 ? 8 : 1;       // entries of a row in the second half: 7 gains + d (n_x = 15) or d alone (n_x = 8)
    static_assert((NX == 15 && ROWP == 16) || (NX == 8 && ROWP == 10), "row halves");
    // Gain records: the four records of the wave's instances are 4 RS doubles in one run.  Loaded by the lanes that use them -- lane (h, r) its half row -- every load
    // instruction touches 64 different 64-byte segments and the address unit sets the pace (measured: 262 us per launch at B = 4096 against 186 for k_apply_rows_tm).
    // So, as there, instruction q loads record q with lane p on its piece p (bytes 16 p ..: one contiguous run), the ring keeps the pieces, and the half rows are picked
    // out of a double-buffered LDS image of the step, dropped there a step ahead (LDS operations of a wave execute in order: no barrier).  Rows 16 bytes apart
    // from a multiple of 128 in the image (bank spread).  Records of instances with nothing pending stay on their first timestep: cache hits.
    constexpr int KS = ROWP + 2, IS = NU * KS, PCS = RS / 2;   // row / record stride of the image (doubles), pieces of a record
    static_assert(PCS <= 64, "one piece per lane");
    __shared__ __attribute__((aligned(16))) double sKi[2][IPW * IS];
    const DevDesc& d = *a.desc;
    const int lane = threadIdx.x, g = lane >> 4, l = lane & 15, h = l >> 3, r = l & 7;
    const int b = xcd_tile() * IPW + g;
    const int Bp = d.Bp, T = d.T, B = d.B;
    const bool inst_ok = (b < B) && (a.pend[b < B ? b : 0] > 0);
    if (__ballot(inst_ok ? 1 : 0) == 0ull) return;  // wave-uniform
    const int bb = (b < B) ? b : 0;
    const bool isS = l < NX;                          // state lane
    const int c = isS ? l : 0;
    const bool isQ = l < DOF, isD = ND == 2 && l >= DOF && l < 2 * DOF, isTm = l == NX - 1;
    const double alpha = ldexp(1.0, -((inst_ok ? a.pend[bb] : 1) - 1));
    const int cur = a.cur[bb];
    const size_t sX_ = (size_t)NX * Bp, sU_ = (size_t)NU * Bp, sK2_ = (size_t)Bp * RS / 2;
    const double* pX = a.X[cur] + (size_t)c * Bp + bb;
    const double* pU = a.U[cur] + (size_t)r * Bp + bb;
    double* oX = a.X[1 - cur] + (size_t)c * Bp + bb;
    double* oU = a.U[1 - cur] + (size_t)r * Bp + bb;

    const int b0 = b - g;
    const bool pk = lane < PCS;                       // this lane carries a piece of each record
    const int prow = pk ? (2 * lane) / ROWP : 0, pcol = pk ? (2 * lane) % ROWP : 0;
    const int wofs = prow * KS + pcol;                // where the piece goes in a record's image
    // this half's entries of row r in the image: 16-byte pieces j = 0 .. 3 at 2 j doubles, as far as the row goes (a piece beyond it re-reads piece 0)
    const int rofs = g * IS + r * KS + 8 * h;
    int jo[4];
    UNR for (int j = 0; j < 4; j++) jo[j] = (8 * h + 2 * j < ROWP) ? 2 * j : 0;
    const double2* const K0 = reinterpret_cast<const double2*>(a.KD + (size_t)b0 * RS) + (pk ? lane : 0);
    const double2* Kk = K0;                           // records of the step being fetched
    bool okq[IPW];
    UNR for (int q = 0; q < IPW; q++) okq[q] = (b0 + q < B) && (a.pend[b0 + q < B ? b0 + q : 0] > 0);  // wave-uniform

    double rk0[PF][IPW], rk1[PF][IPW], xr[PF], ur[PF];  // (two scalar arrays: an array of double2 is not split into registers)
    auto fetch = [&](int slot, int kk) {  // unconditional; the pointers stop at the last control step, and do not move for an instance with nothing pending
        UNR for (int q = 0; q < IPW; q++) {
            const double2 v2 = (okq[q] ? Kk : K0)[(size_t)q * PCS];
            rk0[slot][q] = v2.x;
            rk1[slot][q] = v2.y;
        }
        xr[slot] = *pX;
        ur[slot] = *pU;
        if (kk < T - 2) Kk += sK2_;  // uniform
        const size_t adv = (kk < T - 2 && inst_ok) ? 1 : 0;
        pX += adv * sX_; pU += adv * sU_;
    };
    auto stage = [&](int slot, int buf) {  // pieces of ring slot -> image buf
        if (pk) { UNR for (int q = 0; q < IPW; q++) *reinterpret_cast<double2*>(&sKi[buf][q * IS + wofs]) = make_double2(rk0[slot][q], rk1[slot][q]); }
    };
    UNR for (int q = 0; q < PF; q++) { fetch(q, q); __builtin_amdgcn_sched_barrier(0); }
    stage(0, 0);
    LDS_ORDER();

    // the lane's state: q_c from q0, dq from dq0, the time from 0 (as k_apply_rows_tm)
    double xv = 0.0;
    if (isQ) xv = AT(a.q0, c, bb);
    if (isD) xv = AT(a.dq0, c - DOF, bb);

    const int nsteps = T - 1;
    static_assert(PF % 2 == 0, "image parity = slot parity");
    for (int k0 = 0; k0 < nsteps; k0 += PF) {
        UNR for (int jj = 0; jj < PF; jj++) {
            const int k = k0 + jj;
            double Kr[8];
            UNR for (int j = 0; j < 4; j++) {
                const double2 v2 = *reinterpret_cast<const double2*>(&sKi[jj & 1][rofs + jo[j]]);  // this lane's half row of step k
                Kr[2 * j] = v2.x; Kr[2 * j + 1] = v2.y;
            }
            const double xb = xr[jj], ub = ur[jj];
            LDS_ORDER();
            stage((jj + 1) % PF, (jj + 1) & 1);                     // step k + 1 into the other image
            LDS_ORDER();
            __builtin_amdgcn_sched_barrier(0);                      // the slot is free: only now its next load
            fetch(jj, k + PF);
            if (k >= nsteps) continue;  // uniform; dummy step of the last group
            const double dxv = isS ? xv - xb : 0.0;
            const double part = ad_half_dot<NHI>(Kr, dxv, alpha);
            const double du = part + ad_dpp<0x128>(part);   // row_ror:8: the other half's part; the same two numbers in both halves
            const double u = ub + du;
            const double dts = ad_dpp<0x15F>(u);            // row_newbcast:15: the time control
            const double dt = dts * dts;
            if (inst_ok) {
                if (isS) *oX = xv;
                if (h == 0) *oU = u;
            }
            oX += sX_; oU += sU_;
            // ---- dynamics (SimulationInterface.cpp:19-31 with dt = u_last^2, PosOrnTimePlannerSys.cpp:149-184): k_apply_rows_tm's expressions per coordinate
            if (ND == 2) {
                const double vv = ad_dpp<0x107>(xv);        // row_shl:7: dq_i to the lane of q_i
                const double un = ad_dpp<0x101>(u);         // row_shl:1: u_i (lane 8 + i) to the lane of dq_i (lane 7 + i)
                if (isQ) xv = xv + (dt * vv + dt * dt / 2 * u);
                else if (isD) xv = xv + dt * un;
                else if (isTm) xv = xv + dt;
            } else {
                if (isQ) xv = xv + (dt * u + dt * dt / 2 * 0.0);
                else if (isTm) xv = xv + dt;
            }
        }
    }
    if (inst_ok && isS) *oX = xv;  // x_{T-1}
    if (inst_ok && l == 0) {       // the early stop was decided by k_select_x
        a.cur[bb] = 1 - cur;
        a.pend[bb] = 0;
    }
}
#undef AD_LO
#undef AD_HI

template <class S>
static void launch_lin_sys(const Bufs& a, int B, int T, hipStream_t st, const FwdArgs& f, int which) {
    if (which == KER_FWD_SPEC) {
        const dim3 grid(grid_x8((B + 7) / 8)), block(64);
        if (f.n_alpha <= 1) hipLaunchKernelGGL((k_forward_lin<S, 1>), grid, block, 0, st, a, f);
        else if (f.n_alpha <= 11) hipLaunchKernelGGL((k_forward_lin<S, 11>), grid, block, 0, st, a, f);
        else hipLaunchKernelGGL((k_forward_lin<S, 16>), grid, block, 0, st, a, f);
    } else {
        hipLaunchKernelGGL((k_blend<S>), dim3((B + 255) / 256, T), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_flip, dim3((B + 255) / 256), dim3(256), 0, st, a);
    }
}

bool forward_lin_supported(int kind, int nd, int n_alpha) { return kind == 0 && nd == 2 && n_alpha <= 16; }  // (PosOrn-1: k_forward_wg)

void launch_forward_lin(int nd, int which, const Bufs& a, int B, int T, hipStream_t st, const FwdArgs& f) {
    (void)nd;
    launch_lin_sys<Sys<0, 2>>(a, B, T, st, f, which);
}

void launch_apply_rows_tm(int kind, int nd, const Bufs& a, int B, hipStream_t st, const FwdArgs& f) {
    if (f.apply_dpp) {  // 16 lanes per instance on registers
        const dim3 g4(grid_x8((B + 3) / 4)), blk(64);
        if (kind == 3) hipLaunchKernelGGL((k_apply_dpp_tm<Sys<3, 1>>), g4, blk, 0, st, a, f);
        else if (nd == 1) hipLaunchKernelGGL((k_apply_dpp_tm<Sys<1, 1>>), g4, blk, 0, st, a, f);
        else hipLaunchKernelGGL((k_apply_dpp_tm<Sys<1, 2>>), g4, blk, 0, st, a, f);
        return;
    }
    const dim3 grid(grid_x8((B + 7) / 8)), block(64);
    if (kind == 3) hipLaunchKernelGGL((k_apply_rows_tm<Sys<3, 1>>), grid, block, 0, st, a, f);
    else if (nd == 1) hipLaunchKernelGGL((k_apply_rows_tm<Sys<1, 1>>), grid, block, 0, st, a, f);
    else hipLaunchKernelGGL((k_apply_rows_tm<Sys<1, 2>>), grid, block, 0, st, a, f);
}

}  // namespace ilqr
