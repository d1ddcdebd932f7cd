// ilqr_batchcp.hip -- BatchILQRCP on the device (reference src/solver/BatchILQRCP.cpp:109-175), one lane per instance.
//
// Gauss-Newton on the whole control sequence in the primitive basis u = PSI w.  Per iteration and instance:
//   k_cp_linearize  System::fpBatch (System.cpp:181-211): rollout of the current u; at the keypoint steps the residual e,
//                   C = J'QJ + L and r = J'Q e + L ql (limits on the PRE-step state, System.cpp:158).  Su (n_kp n_x x
//                   n_u (T-1), BatchILQRCP.cpp:61-97) is never materialised: W = Su PSI obeys the same recurrence
//                   W_{i+1} = A_i W_i + B_i PSI_i (W_1 = 0, the reference's shifted seeding: block j of PSI meets
//                   B_j = dx_j/du_{j-1}, quirk D-1) and lives in LDS as an n_x x Kw tile per lane.  Also PSI'R u.
//   k_cp_solve      H = sum_kp W'CW + PSI'R PSI,  g = sum_kp W'r - PSI'R u,  dw = H^-1 g by partial-pivot LU in LDS
//                   (BatchILQRCP.cpp:129-133; the reference forms the inverse explicitly, same pivoting).
//   k_cp_linesearch backtracking on the true cost (:138-158): u + alpha PSI dw, accept if cost < cost0 or alpha < 1e-3.
// The contraction is a 14x14 (Kw x Kw) matrix per instance at the BASELINE config: far too small for an MFMA tile to
// pay; it is VALU work and the pass is bound by the rollout, like the recursive solver.
#include "ilqr_batchcp.hpp"

#include <cstdlib>
#include <cstring>

#include "ilqr_batch_dev.hpp"

namespace ilqr {

#define PSI(k, q) c.psi[(size_t)(k) * KWP + (q)]

// Lanes (instances) per workgroup of the one-lane-per-instance kernels.  These kernels are latency-bound chains (rollout, FK,
// LU); at B = 8192 full waves would give 128 workgroups for 256 CUs.  Quarter waves spread the same lanes over four times as
// many SIMDs, which run them at the same speed (a wave costs the same with 16 or 64 active lanes).
#ifndef CP_LPB
#define CP_LPB 16
#endif
constexpr int LPB = CP_LPB;

// KWP lanes per instance: lane q owns column q of W = Su PSI (n_x values in registers) and of PSI'R u, and reads its entries of the
// basis rows as coalesced vector loads (with one lane per instance they are uniform -> one scalar-load latency per entry: 24 us per
// step on the 2nd-order time system).  All lanes walk the rollout and the keypoint evaluation; lane 0 writes the shared results.
template <class S, int KWP>
__global__ __launch_bounds__(64) void k_cp_linearize(Bufs a, CPArgs c) {
    constexpr int NX = S::NX, NU = S::NU, IPB = 64 / KWP;
    static_assert(64 % KWP == 0, "KWP lanes per instance");
    const DevDesc& d = *a.desc;
    const int q = threadIdx.x % KWP, b = blockIdx.x * IPB + threadIdx.x / KWP;
    if (b >= d.B || !a.active[b]) return;
    const int Bp = d.Bp, T = d.T;
    double W[NX], x[NX], xp[NX], u[NU], gu = 0;
    UNR for (int r = 0; r < NX; r++) W[r] = 0;
    init_state<S>(d, a, b, x);
    UNR for (int i = 0; i < NX; i++) xp[i] = x[i];
    double cost_e = 0, cost_u = 0, cost_l = 0;
    int kpi = 0;
    const double* U = a.U[0];

    auto record = [&](int i) {  // keypoint kpi sits at step i: x = x_i, xp = x_{i-1}
        double lxx[NX][NX], lx[NX], Ld[NX], ql[NX];
        stage_derivs<S, false>(d, a, b, x, kpi, lxx, lx);  // lxx = J'QJ, lx = -J'Q e
        if (i > 0) limit_terms<S>(d, xp, Ld, ql);
        else { UNR for (int r = 0; r < NX; r++) { Ld[r] = 0; ql[r] = 0; } }
        double tg[S::NF];
        UNR for (int r = 0; r < S::NF; r++) tg[r] = AT(a.kp_tg, kpi * S::NF + r, b);
        cost_e += kp_cost<S>(d, kpi, tg, x, nullptr);
        UNR for (int r = 0; r < NX; r++) cost_l += ql[r] * Ld[r] * ql[r];
        double* Wk = c.Wkp + (size_t)kpi * NX * KWP * Bp;
        UNR for (int r = 0; r < NX; r++) AT(Wk, r * KWP + q, b) = W[r];
        if (q == 0) {
            double* Ck = c.Ckp + (size_t)kpi * NX * NX * Bp;
            double* rk = c.rkp + (size_t)kpi * NX * Bp;
            UNR for (int r = 0; r < NX; r++) {
                UNR for (int s = 0; s < NX; s++) AT(Ck, r * NX + s, b) = lxx[r][s] + ((r == s) ? Ld[r] : 0.0);
                AT(rk, r, b) = -lx[r] + Ld[r] * ql[r];
            }
        }
        kpi++;
    };

    if (kpi < d.n_kp && d.kp_t[kpi] == 0) record(0);
    // the controls and basis rows of G steps are fetched together and consumed before any branch: one memory latency per group.
    // Basis row-set s+1 serves the W update of step s and the PSI'R u term of step s+1.
    constexpr int G = 4;
    for (int s0 = 0; s0 < T - 1; s0 += G) {
        double ug[G][NU], pg[G + 1][NU];
        UNR for (int k = 0; k < G; k++) {
            const int s = (s0 + k < T - 1) ? s0 + k : T - 2;
            UNR for (int i = 0; i < NU; i++) ug[k][i] = AT(U, s * NU + i, b);
        }
        UNR for (int k = 0; k <= G; k++) {
            const int s = (s0 + k < T - 1) ? s0 + k : T - 2;
            UNR for (int i = 0; i < NU; i++) pg[k][i] = PSI(s * NU + i, q);
        }
        UNR for (int k = 0; k < G; k++) { UNR for (int i = 0; i < NU; i++) asm volatile("" : "+v"(ug[k][i])); }
        UNR for (int k = 0; k <= G; k++) { UNR for (int i = 0; i < NU; i++) asm volatile("" : "+v"(pg[k][i])); }
        UNR for (int k = 0; k < G; k++) {
            const int s = s0 + k;
            if (s < T - 1) {
                UNR for (int i = 0; i < NU; i++) u[i] = ug[k][i];
                UNR for (int i = 0; i < NU; i++) cost_u += u[i] * d.R_diag[i] * u[i];
                UNR for (int i = 0; i < NU; i++) gu += pg[k][i] * (d.R_diag[i] * u[i]);
                StepAB<S> ab;
                double xn[NX];
                step_ab<S>(d, x, u, xn, ab);
                UNR for (int i = 0; i < NX; i++) { xp[i] = x[i]; x[i] = xn[i]; }
                const int i = s + 1;
                if (kpi < d.n_kp && d.kp_t[kpi] == i) record(i);
                if (i <= T - 2) {  // W_{i+1} = A_i W_i + B_i PSI_i
                    if (S::ND == 2) { UNR for (int r = 0; r < DOF; r++) W[r] += ab.dt * W[DOF + r]; }
                    UNR for (int r = 0; r < DOF; r++) {
                        const double ps = pg[k + 1][r];
                        if (S::ND == 1) W[r] += ab.dt * ps;
                        else { W[r] += ab.hdt2 * ps; W[DOF + r] += ab.dt * ps; }
                    }
                    if (S::TM) {
                        const double pl = pg[k + 1][NU - 1];
                        UNR for (int r = 0; r < NX; r++) W[r] += ab.bc[r] * pl;
                    }
                }
            }
        }
    }
    AT(c.gu, q, b) = gu;
    if (q == 0) a.cost[b] = cost_e + cost_u + cost_l;  // cost0 of this iteration (BatchILQRCP.cpp:135)
}

template <class S, int KWP>
__global__ __launch_bounds__(LPB) void k_cp_solve(Bufs a, CPArgs c) {
    constexpr int NX = S::NX;
    extern __shared__ double lds[];
    const DevDesc& d = *a.desc;
    const int lane = threadIdx.x, b = blockIdx.x * LPB + lane;
    if (b >= d.B || !a.active[b]) return;
    const int Bp = d.Bp;
#define HL(r, q) lds[((r) * KWP + (q)) * LPB + lane]
    double g[KWP];
    UNR for (int q = 0; q < KWP; q++) g[q] = -AT(c.gu, q, b);
    for (int r = 0; r < KWP; r++)
        for (int q = 0; q < KWP; q++) HL(r, q) = c.H0[r * KWP + q];
    for (int t = 0; t < d.n_kp; t++) {
        const double* Ck = c.Ckp + (size_t)t * NX * NX * Bp;
        const double* rk = c.rkp + (size_t)t * NX * Bp;
        const double* Wk = c.Wkp + (size_t)t * NX * KWP * Bp;
        for (int q = 0; q < KWP; q++) {
            double wq[NX], cw[NX];  // column q of W_t and of C W_t
            UNR for (int i = 0; i < NX; i++) wq[i] = AT(Wk, i * KWP + q, b);
            UNR for (int i = 0; i < NX; i++) {
                double s = 0;
                UNR for (int j = 0; j < NX; j++) s += AT(Ck, i * NX + j, b) * wq[j];
                cw[i] = s;
            }
            for (int r = 0; r < KWP; r++) {
                double s = 0;
                UNR for (int i = 0; i < NX; i++) s += AT(Wk, i * KWP + r, b) * cw[i];
                HL(r, q) += s;
            }
            double s = 0;
            UNR for (int i = 0; i < NX; i++) s += wq[i] * AT(rk, i, b);
            g[q] += s;
        }
    }
    // partial-pivot LU in place (Eigen PartialPivLU), rows of g permuted along
    for (int k = 0; k < KWP; k++) {
        int pr = k;
        double best = fabs(HL(k, k));
        for (int i = k + 1; i < KWP; i++) {
            const double v = fabs(HL(i, k));
            if (v > best) { best = v; pr = i; }
        }
        if (pr != k) {
            for (int q = 0; q < KWP; q++) { const double t0 = HL(k, q); HL(k, q) = HL(pr, q); HL(pr, q) = t0; }
        }
        // g lives in registers: swap by select over the unrolled index
        {
            double gk = 0, gp = 0;
            UNR for (int q = 0; q < KWP; q++) { if (q == k) gk = g[q]; if (q == pr) gp = g[q]; }
            UNR for (int q = 0; q < KWP; q++) { if (q == k) g[q] = gp; else if (q == pr) g[q] = gk; }
        }
        const double pv = HL(k, k);
        for (int i = k + 1; i < KWP; i++) {
            const double fct = HL(i, k) / pv;
            HL(i, k) = fct;
            for (int q = k + 1; q < KWP; q++) HL(i, q) -= fct * HL(k, q);
        }
    }
    // L y = P g ; U dw = y
    UNR for (int i = 0; i < KWP; i++) {
        double s = g[i];
        UNR for (int j = 0; j < KWP; j++) if (j < i) s -= HL(i, j) * g[j];
        g[i] = s;
    }
    UNR for (int i = KWP - 1; i >= 0; i--) {
        double s = g[i];
        UNR for (int j = 0; j < KWP; j++) if (j > i) s -= HL(i, j) * g[j];
        g[i] = s / HL(i, i);
    }
    UNR for (int q = 0; q < KWP; q++) AT(c.dw, q, b) = g[q];
#undef HL
}

// The same solve with the rows of [H | -g] in registers: a wave takes G = 64 / KWP instances, lane (g, r) owns row r of instance g.
//   assembly   lane (g, q) holds column q of W, forms column q of C W (the only exchange through LDS: every row needs all of C W)
//              and adds its row of W'(C W)
//   LU         the pivot search is a max-butterfly over the KWP lanes of an instance (DPP within a 16-lane row) and a ballot for
//              the FIRST row that holds the maximum; a round of lane permutes exchanges rows k and pr, a second one hands the
//              pivot row (now in lane k) to every lane.  The multiplier of a row is local to its lane; no LDS, no barriers.
//   back subst lane i finishes x_i (sum over j ascending, as before) and the instance's lanes fetch it
// Every entry sees the same operations in the same order as in k_cp_solve (sums over i, j ascending; first-maximum pivot; the
// right-hand side eliminated along as an extra column = the forward substitution with the stored multipliers), so the two kernels
// agree bit for bit.  (The earlier form -- one wave per instance, the system in LDS, a lane per column -- spent its time in the LDS
// pipe: ~1000 64-bit LDS instructions per instance, 69 us for 8192 instances; this one 30 us, 14 of them the assembly.)
template <int CTRL>
__device__ __forceinline__ double cp_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double cp_fetch(int src_lane, double v) {
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <class S, int KWP>
__global__ __launch_bounds__(64) void k_cp_solve_w(Bufs a, CPArgs c) {
    constexpr int NX = S::NX, G = 64 / KWP;
    static_assert(KWP == 16 || KWP == 32, "a row of the system per lane, whole instances per wave");
    __shared__ double sCW[G][NX][KWP];
    __shared__ __attribute__((aligned(16))) double sC[G][(NX * NX + 1) & ~1];
    const DevDesc& d = *a.desc;
    const int lane = threadIdx.x, g = lane / KWP, r = lane % KWP, base = lane - r;
    const bool live = (int)(blockIdx.x * G + g) < d.B;
    const int b = live ? blockIdx.x * G + g : d.B - 1;  // lanes past the batch redo the last instance and store nothing
    const bool act = live && a.active[b];
    if (!__ballot(act ? 1 : 0)) return;  // uniform
    const int Bp = d.Bp;
    double h[KWP + 1];  // row r of H, then -g_r
    UNR for (int q = 0; q < KWP; q++) h[q] = c.H0[r * KWP + q];
    h[KWP] = -AT(c.gu, r, b);
    for (int t = 0; t < d.n_kp; t++) {
        const double* Ck = c.Ckp + (size_t)t * NX * NX * Bp;
        const double* rk = c.rkp + (size_t)t * NX * Bp;
        double wc[NX];  // column r of W
        UNR for (int j = 0; j < NX; j++)
            wc[j] = c.wref ? c.wref[(size_t)t * NX * KWP + j * KWP + r] : AT(c.Wkp + (size_t)t * NX * KWP * Bp, j * KWP + r, b);
        __syncthreads();  // (one wave: orders the reads of the previous keypoint before these writes)
        // C of the instance goes through LDS: every lane of the instance needs all of it, and as loads that is NX^2 vector-memory
        // instructions per lane with one useful address per 16 lanes (the address unit, not the data, was the cost: 49 loads -> 4)
        for (int e = r; e < NX * NX; e += KWP) sC[g][e] = AT(Ck, e, b);
        __syncthreads();
        UNR for (int i = 0; i < NX; i++) {
            double s = 0;
            UNR for (int j = 0; j < NX; j++) s = fma(sC[g][i * NX + j], wc[j], s);
            sCW[g][i][r] = s;
        }
        __syncthreads();
        UNR for (int q = 0; q < KWP; q++) {
            double s = 0;
            UNR for (int i = 0; i < NX; i++) s = fma(wc[i], sCW[g][i][q], s);
            h[q] += s;
        }
        {
            double s = 0;
            UNR for (int i = 0; i < NX; i++) s = fma(wc[i], AT(rk, i, b), s);
            h[KWP] += s;
        }
    }
    UNR for (int k = 0; k < KWP; k++) {  // partial-pivot LU (Eigen PartialPivLU)
        // pivot = FIRST row with the largest |H[i][k]|, i >= k (NaN candidates lose; a NaN on the diagonal stays the pivot)
        const bool cand = r >= k;
        const double av = cand ? fabs(h[k]) : -1.0;
        double mx = av;
        mx = fmax(mx, cp_dpp<0xB1>(mx));   // quad_perm [1,0,3,2]
        mx = fmax(mx, cp_dpp<0x4E>(mx));   // quad_perm [2,3,0,1]
        mx = fmax(mx, cp_dpp<0x141>(mx));  // row_half_mirror: the other quad of the half
        mx = fmax(mx, cp_dpp<0x140>(mx));  // row_mirror: the other half of the 16-lane row
        if (KWP == 32) mx = fmax(mx, cp_fetch(lane ^ 16, mx));
        const unsigned long long eq = __ballot((cand && av == mx) ? 1 : 0);
        const unsigned long long dn = __ballot((r == k && isnan(h[k])) ? 1 : 0);
        const unsigned eqg = (unsigned)(eq >> base) & (KWP == 32 ? 0xffffffffu : 0xffffu);
        const bool diag_nan = (dn >> (base + k)) & 1;
        const int pr = (eqg && !diag_nan) ? (__ffs((int)eqg) - 1) : k;
        // two rounds of lane permutes and no selects: rows k and pr change places (the other lanes fetch themselves), then every lane
        // fetches the pivot row from lane k.  (One round -- lane pr fetches row k, the others row pr -- needs a select per register
        // between fetched and own values afterwards, and those selects cost more than the second round: 45 us against 30.)
        const int src = base + ((r == pr) ? k : ((r == k) ? pr : r));
        double prow[KWP + 1];
        UNR for (int j = k; j <= KWP; j++) h[j] = cp_fetch(src, h[j]);
        UNR for (int j = k; j <= KWP; j++) prow[j] = cp_fetch(base + k, h[j]);
        if (r > k) {
            const double m = h[k] / prow[k];
            UNR for (int j = k + 1; j <= KWP; j++) h[j] = fma(-m, prow[j], h[j]);
        }
    }
    double x[KWP], mine = 0;
    UNR for (int i = KWP - 1; i >= 0; i--) {
        double s = h[KWP];
        UNR for (int j = i + 1; j < KWP; j++) s = fma(-h[j], x[j], s);  // (written out: with the product shared between the chains of
                                                                            // several rows the compiler keeps a separate multiply)
        const double xi = s / h[i];
        if (r == i) mine = xi;
        x[i] = cp_fetch(base + i, xi);
    }
    if (act) AT(c.dw, r, b) = mine;
}

// du = PSI dw for every control entry (a.U[1]), one lane per (instance, step): the line search then needs no basis at all
template <class S, int KWP>
__global__ __launch_bounds__(64) void k_cp_du(Bufs a, CPArgs c) {
    constexpr int NU = S::NU;
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * 64 + threadIdx.x, s = blockIdx.y;
    if (b >= d.B || !a.active[b]) return;
    const int Bp = d.Bp;
    double dw[KWP];
    UNR for (int q = 0; q < KWP; q++) dw[q] = AT(c.dw, q, b);
    UNR for (int i = 0; i < NU; i++) {
        double du = 0;
        UNR for (int q = 0; q < KWP; q++) du += PSI(s * NU + i, q) * dw[q];
        AT(a.U[1], s * NU + i, b) = du;
    }
}

// controls <- U0, solver state reset
template <class S>
__global__ void k_cp_init(Bufs a) {
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

// final rollout of the solution so that X (and the cost of the returned u) can be read back
template <class S>
__global__ __launch_bounds__(LPB) void k_cp_final(Bufs a) {
    constexpr int NX = S::NX, NU = S::NU;
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * LPB + threadIdx.x;
    if (b >= d.B) return;
    const int Bp = d.Bp, T = d.T;
    double x[NX], u[NU], xn[NX];
    init_state<S>(d, a, b, x);
    constexpr int G = 8;  // the controls of G steps are fetched together: one memory latency per group instead of one per step
    for (int s0 = 0; s0 < T - 1; s0 += G) {
        double ug[G][NU];
        UNR for (int k = 0; k < G; k++) {
            const int s = (s0 + k < T - 1) ? s0 + k : T - 2;
            UNR for (int i = 0; i < NU; i++) ug[k][i] = AT(a.U[0], s * NU + i, b);
        }
        UNR for (int k = 0; k < G; k++) { UNR for (int i = 0; i < NU; i++) asm volatile("" : "+v"(ug[k][i])); }
        UNR for (int k = 0; k < G; k++) {
            const int s = s0 + k;
            if (s >= T - 1) break;
            UNR for (int i = 0; i < NX; i++) AT(a.X[0], s * NX + i, b) = x[i];
            UNR for (int i = 0; i < NU; i++) u[i] = ug[k][i];
            dyn_step<S>(d, x, u, xn);
            UNR for (int i = 0; i < NX; i++) x[i] = xn[i];
        }
    }
    UNR for (int i = 0; i < NX; i++) AT(a.X[0], (T - 1) * NX + i, b) = x[i];
}

// ------------------------------------------------------------------------------------------------ PosOrn systems: coefficient space
// For constant A, B the controls stay in the affine family u = u0 + PSI w and the trajectory is affine in w:
//     x_t(w) = x_t(u0) + Wt_t w,   Wt_{i+1} = A Wt_i + B PSI_i  (Wt_0 = 0; the TRUE sensitivity, not the reference's shifted Su)
//     sum_k u_k' R u_k = u0'R u0 + 2 w.(PSI'R u0) + w'(PSI'R PSI) w,   ||PSI dw||^2 = dw'(PSI'PSI) dw,   PSI'R u = PSI'R u0 + (PSI'R PSI) w
// so an iteration needs the states at the keypoint steps only: no pass over the horizon at all.  The shifted W = Su PSI of the
// reference (what H and g are built from, quirk D-1) is the same for every instance and is broadcast once; k_cp_solve is shared
// with the general path.  The horizon is walked at the start of a solve (k_cpl_states, k_cpl_quad: rollout of u0 and the quadratic forms of its cost) and at its end
// (k_cpl_final: u and X out).
#define WT(kp, which, r, q) c.wt[((((size_t)(kp) * 2 + (which)) * NX + (r)) * KWP) + (q)]

template <class S, int KWP>
ILQR_DEV void cpl_states(const DevDesc& d, const CPArgs& c, int b, int kpi, const double* w, double* x, double* xp) {
    constexpr int NX = S::NX;
    const int Bp = d.Bp;
    const double* xb = c.xbk + (size_t)kpi * 2 * NX * Bp;
    UNR for (int r = 0; r < NX; r++) {
        double s0 = AT(xb, r, b), s1 = AT(xb, NX + r, b);
        UNR for (int q = 0; q < KWP; q++) { s0 += WT(kpi, 0, r, q) * w[q]; s1 += WT(kpi, 1, r, q) * w[q]; }
        x[r] = s0;
        xp[r] = s1;
    }
}
// e'Qe at the keypoints + limit term of the pre-step states + control cost, all as functions of w (BatchILQRCP.cpp:135,150)
template <class S, int KWP>
ILQR_DEV double cpl_cost(const DevDesc& d, const Bufs& a, const CPArgs& c, int b, const double* w, double c00) {
    constexpr int NX = S::NX;
    const int Bp = d.Bp;
    double cost_e = 0, cost_l = 0;
    for (int kpi = 0; kpi < d.n_kp; kpi++) {
        double x[NX], xp[NX], tg[S::NF];
        cpl_states<S, KWP>(d, c, b, kpi, w, x, xp);
        UNR for (int r = 0; r < S::NF; r++) tg[r] = AT(a.kp_tg, kpi * S::NF + r, b);
        cost_e += kp_cost<S>(d, kpi, tg, x, nullptr);
        if (d.kp_t[kpi] > 0) {
            double Ld[NX], ql[NX];
            limit_terms<S>(d, xp, Ld, ql);
            UNR for (int r = 0; r < NX; r++) cost_l += ql[r] * Ld[r] * ql[r];
        }
    }
    // g0 = PSI'R u0 is read HERE, behind the keypoint code (held from the start of the kernel its 16 values were part of what pushed the line-search
    // kernel over the register file: 132 bytes of scratch in the code-object audit of round 3), and used behind the quadratic form, which covers the
    // loads' latency.  lin and quad are separate sums: the same bits in either order.
    double g0v[KWP];
    UNR for (int q = 0; q < KWP; q++) g0v[q] = AT(c.g0, q, b);
    double lin = 0, quad = 0;
    UNR for (int q = 0; q < KWP; q++) {
        double s = 0;
        UNR for (int r = 0; r < KWP; r++) s += c.H0[q * KWP + r] * w[r];
        quad += w[q] * s;
    }
    UNR for (int q = 0; q < KWP; q++) lin += g0v[q] * w[q];
    // the padded diagonal of H0 is 1 (keeps H regular); the padded w entries are 0, so it adds nothing
    return cost_e + ((c00 + 2 * lin) + quad) + cost_l;
}

// KWP lanes per instance: lane q accumulates PSI'R u0 for its own column q (the long part of this pass: (T-1) n_u terms per column,
// basis rows read as coalesced vector loads; with one lane per instance they are uniform -> scalar loads whose latency, one per
// control entry, bounds the kernel).  All lanes walk the rollout of u0, lane 0 records the states at the keypoint steps.
// the reference's W = Su PSI at the keypoint steps is the same for all instances of an LTI system: fill the per-instance table
// k_cp_solve reads (one lane per (instance, entry))
__global__ void k_cpl_bcast(const double* __restrict__ wref, double* __restrict__ Wkp, int n_entries, int B, int Bp) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x, e = blockIdx.y;
    if (b < B && e < n_entries) Wkp[(size_t)e * Bp + b] = wref[e];
}

// One lane per (instance, keypoint): the two keypoint linearisations of an instance (FK, Jacobian, J'QJ) run side by side.  The lane
// of keypoint 0 also forms PSI'R u and, in the first iteration, cost0; later cost0 is the cost of the trial the line search
// accepted -- the same function of the same w, already in a.cost.
template <class S, int KWP>
__global__ __launch_bounds__(LPB) void k_cpl_linearize(Bufs a, CPArgs c) {
    constexpr int NX = S::NX;
    constexpr int ROLL = (!S::JOINT && S::ND == 1) ? LPB : 0;  // rolled FK joint loop, its per-joint values in LDS (ilqr_device.hpp: fk)
    __shared__ double sj[ROLL ? 7 * DOF : 1][LPB];  // (nothing reserved where the loop is not rolled)
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * LPB + threadIdx.x, kpi = blockIdx.y;
    if (b >= d.B || !a.active[b]) return;
    const int Bp = d.Bp;
    double w[KWP];
    UNR for (int q = 0; q < KWP; q++) w[q] = AT(c.wv, q, b);
    if (kpi < d.n_kp) {
        double x[NX], xp[NX], lxx[NX][NX], lx[NX], Ld[NX], ql[NX];
        cpl_states<S, KWP>(d, c, b, kpi, w, x, xp);
        stage_derivs<S, false, true, ROLL>(d, a, b, x, kpi, lxx, lx, &sj[0][threadIdx.x]);  // lxx = J'QJ, lx = -J'Q e
        if (d.kp_t[kpi] > 0) limit_terms<S>(d, xp, Ld, ql);
        else { UNR for (int r = 0; r < NX; r++) { Ld[r] = 0; ql[r] = 0; } }
        double* Ck = c.Ckp + (size_t)kpi * NX * NX * Bp;
        double* rk = c.rkp + (size_t)kpi * NX * Bp;
        UNR for (int r = 0; r < NX; r++) {
            UNR for (int s = 0; s < NX; s++) AT(Ck, r * NX + s, b) = lxx[r][s] + ((r == s) ? Ld[r] : 0.0);
            AT(rk, r, b) = -lx[r] + Ld[r] * ql[r];
        }
    }
    if (kpi != 0) return;
    double g0[KWP];
    UNR for (int q = 0; q < KWP; q++) g0[q] = AT(c.g0, q, b);
    if (c.it == 0) a.cost[b] = cpl_cost<S, KWP>(d, a, c, b, w, c.c00[b]);  // cost0 (BatchILQRCP.cpp:135)
    UNR for (int q = 0; q < KWP; q++) {  // PSI'R u = PSI'R u0 + (PSI'R PSI) w   (padded rows of H0: identity x 0)
        double s = g0[q];
        UNR for (int r = 0; r < KWP; r++) s += ((q < c.Kw && r < c.Kw) ? c.H0[q * KWP + r] : 0.0) * w[r];
        AT(c.gu, q, b) = s;
    }
}

// Backtracking with several step sizes at once, in two passes: lane l of an instance tries alpha = 2^-l (the reference's loop stops at the
// first alpha whose cost improves or at alpha < 1e-3, BatchILQRCP.cpp:138-158 -- the first such lane wins), so the chain a lane walks
// is one cost evaluation instead of up to eleven.
//   pass 1 (LPI = 4, L0 = 0)   alpha = 1 .. 1/8, 16 instances per wave: 99.3 % of the C5 line searches end here (profiles/r02_batch_scan.txt)
//   pass 2 (LPI = 8, L0 = 4)   alpha = 1/16 .. 2^-10 for the instances pass 1 left without a winner (iters[b] still at the old count);
//                              waves without such an instance leave after two loads
// (One pass with 16 lanes per instance -- all eleven step sizes -- evaluated 16 costs per instance where 1.7 are needed on average: 72 us
// per launch against 42 + 24 us for the two passes.  What a pass waits for is the chain of ONE cost evaluation, ~20 us with a wave per SIMD;
// sharing the keypoints of an evaluation between two lanes was tried and did not shorten it: 41 + 22 us, at 512 VGPRs with spills.)
template <class S, int KWP, int LPI, int L0>
__global__ __launch_bounds__(64) void k_cpl_linesearch(Bufs a, CPArgs c) {
    const DevDesc& d = *a.desc;
    const int lane = threadIdx.x, l = L0 + (lane & (LPI - 1)), bq = blockIdx.x * (64 / LPI) + lane / LPI;
    bool ok = bq < d.B && a.active[bq < d.B ? bq : 0];
    if (L0 > 0) ok = ok && a.iters[bq < d.B ? bq : 0] != c.it + 1;  // pass 2: only what pass 1 left undecided
    if (!__ballot(ok ? 1 : 0)) return;
    const int b = ok ? bq : 0;
    const int Bp = d.Bp;
    double wn[KWP];
    const double cost0 = a.cost[b], c00 = c.c00[b];
    const double alpha = ldexp(1.0, -(l < 11 ? l : 10));
    UNR for (int q = 0; q < KWP; q++) wn[q] = AT(c.wv, q, b) + alpha * AT(c.dw, q, b);  // (w, dw are not held: dw is read again for the stop test)
    const double cost = cpl_cost<S, KWP>(d, a, c, b, wn, c00);
    const bool take = (l < 11) && ((cost < cost0) || (alpha < 1e-3));
    const unsigned long long m = __ballot(take ? 1 : 0);
    const unsigned grp = (unsigned)((m >> (lane & ~(LPI - 1))) & ((1u << LPI) - 1));  // this instance's lanes; lane l = 10 always votes
    const int win = L0 + __ffs(grp) - 1;
    if (!ok || !grp || l != win) return;
    UNR for (int q = 0; q < KWP; q++) AT(c.wv, q, b) = wn[q];
    a.alpha[b] = alpha;
    a.iters[b] = c.it + 1;
    a.status[b] = (isfinite(cost) ? 0 : 1) | ((alpha < 1e-3) ? 2 : 0);
    if (a.cost_trace) {
        a.cost_trace[(size_t)c.it * Bp + b] = cost0;  // the reference prints the PRE-step cost (BatchILQRCP.cpp:160)
        a.alpha_trace[(size_t)c.it * Bp + b] = alpha;
    }
    a.cost[b] = cost;
    if (c.early_stop) {
        double dun2 = 0;  // sum_k ||PSI_k dw||^2
        UNR for (int q = 0; q < KWP; q++) {
            double s = 0;
            UNR for (int r = 0; r < KWP; r++) s += c.pp[q * KWP + r] * AT(c.dw, r, b);
            dun2 += AT(c.dw, q, b) * s;
        }
        if (alpha * sqrt(dun2 > 0 ? dun2 : 0.0) < 1e-3) a.active[b] = 0;  // :167
    }
}

// ---- the horizon walks of the coefficient-space path with one lane per (instance, coordinate) / per (instance, chunk of steps).
// The first versions (one lane per (instance, basis column), every lane rolling the whole state and reading 14 values per step; k_cp_final: one
// lane per instance) were chains of T - 1 steps with a memory latency per few steps: 0.44 + 0.22 ms of a 2.75 ms solve at the C5 shape.
// For the constant-A, B systems the coordinates integrate independently, so:
//   k_cpl_states   lane (b, i): joint i of instance b through the horizon (dyn_step's expressions on that coordinate, controls fetched in
//                  chunks); at the keypoint steps it drops x and the state one step earlier into `xbk`.  k_cpl_final: the final rollout --
//                  the controls are U[0] and every state goes to X[0].
//   k_cpl_quad     PSI' (R u0) and u0' R u0: a GEMM over 16-instance tiles with a long inner dimension, on the matrix cores (below).
//   k_cpl_final    forms u = u0 + PSI w for its coordinate on the way (the rows of PSI it needs are uniform over a block): one pass reads U0 and
//                  writes U and X (a separate controls kernel -- also tried as a GEMM on the matrix cores, 0.089 ms -- made the rollout read U back).
template <class S, int mode, int KWP>
ILQR_DEV void cpl_walk(const Bufs& a, const CPArgs& c) {
    constexpr int NX = S::NX, NU = S::NU, ND = S::ND, CH = 8, TILE = 64;  // TILE steps of the basis per LDS image (final pass)
    static_assert(S::TM == 0, "constant A, B");
    const DevDesc& d = *a.desc;
    const int i = blockIdx.y;
    const bool live = (int)(blockIdx.x * 256 + threadIdx.x) < d.B;
    const int b = live ? blockIdx.x * 256 + threadIdx.x : d.B - 1;  // lanes past the batch walk the last instance and store nothing
    const int Bp = d.Bp, T = d.T;
    const double dt = d.dt;
    double q = AT(a.q0, i, b), v = (ND == 2) ? AT(a.dq0, i, b) : 0.0, qp = q, vp = v;
    const double* __restrict__ Uin = a.U0;
    double* __restrict__ X = a.X[0];
    double* __restrict__ Uout = a.U[0];
    // final pass: u = u0 + PSI w is formed here.  Row s n_u + i of PSI is the same for the whole block (blockIdx.y = i): the block keeps
    // TILE steps of it in LDS (two images, filled one tile ahead) and every lane reads them as broadcasts.
    __shared__ double sPsi[mode ? 2 : 1][mode ? TILE : 1][mode ? KWP : 1];
    double w[KWP];
    auto fill = [&](int tile) {
        if (!mode) return;
        for (int e = threadIdx.x; e < TILE * KWP; e += 256) {
            const int k = tile * TILE + e / KWP;
            sPsi[tile & 1][e / KWP][e % KWP] = (k < T - 1) ? c.psi[(size_t)(k * NU + i) * KWP + e % KWP] : 0.0;
        }
    };
    if (mode) {
        UNR for (int qq = 0; qq < KWP; qq++) w[qq] = AT(c.wv, qq, b);
        fill(0);
    }
    int kpi = 0;
    int kp_next = (!mode && d.n_kp > 0) ? __builtin_amdgcn_readfirstlane(d.kp_t[0]) : -1;  // wave-uniform: a scalar compare per step
    auto record = [&]() {  // keypoint step reached: x and the state one step earlier (limit terms act on the pre-step state)
        double* xb = c.xbk + (size_t)kpi * 2 * NX * Bp;
        if (live) {
            AT(xb, i, b) = q; AT(xb, NX + i, b) = qp;
            if (ND == 2) { AT(xb, DOF + i, b) = v; AT(xb, NX + DOF + i, b) = vp; }
        }
        kpi = __builtin_amdgcn_readfirstlane(kpi + 1);
        kp_next = (kpi < d.n_kp) ? __builtin_amdgcn_readfirstlane(d.kp_t[kpi]) : -1;
    };
    if (!mode && kp_next == 0) record();
    double un[CH];  // controls of the next chunk, loaded one chunk ahead
    UNR for (int j = 0; j < CH; j++) un[j] = AT(Uin, (j < T - 1 ? j : T - 2) * NU + i, b);
    for (int k0 = 0; k0 < T - 1; k0 += CH) {
        double u[CH];
        UNR for (int j = 0; j < CH; j++) u[j] = un[j];
        UNR for (int j = 0; j < CH; j++) un[j] = AT(Uin, (k0 + CH + j < T - 1 ? k0 + CH + j : T - 2) * NU + i, b);
        if (mode && k0 % TILE == 0) {
            __syncthreads();             // image k0 / TILE is complete; the other one is no longer read
            fill(k0 / TILE + 1);
        }
        UNR for (int j = 0; j < CH; j++) {
            const int k = k0 + j;
            if (k >= T - 1) break;
            if (mode) {
                double du = 0;
                UNR for (int qq = 0; qq < KWP; qq++) du += sPsi[(k0 / TILE) & 1][(k0 % TILE) + j][qq] * w[qq];
                u[j] = u[j] + du;
                if (live) {
                    AT(X, k * NX + i, b) = q; if (ND == 2) AT(X, k * NX + DOF + i, b) = v;
                    AT(Uout, k * NU + i, b) = u[j];
                }
            }
            qp = q; vp = v;
            if (ND == 1) {
                q = q + (dt * u[j] + dt * dt / 2 * 0.0);   // dyn_step, same expressions
            } else {
                q = q + (dt * v + dt * dt / 2 * u[j]);
                v = v + dt * u[j];
            }
            if (!mode && kp_next == k + 1) record();
        }
    }
    if (mode && live) { AT(X, (T - 1) * NX + i, b) = q; if (ND == 2) AT(X, (T - 1) * NX + DOF + i, b) = v; }
}
template <class S>
__global__ __launch_bounds__(256) void k_cpl_states(Bufs a, CPArgs c) { cpl_walk<S, 0, 1>(a, c); }  // rollout of u0: keypoint states
template <class S, int KWP>
__global__ __launch_bounds__(256) void k_cpl_final(Bufs a, CPArgs c) { cpl_walk<S, 1, KWP>(a, c); }  // u = u0 + PSI w and its rollout: U, X out

// g0 = PSI' (R u0) and c00 = u0' R u0 on the f64 matrix cores: for 16 instances, G [16 x 16] = PSI' [16 x rows] x (R u0) [rows x 16] is a
// plain GEMM with a long inner dimension (rows = (T-1) n_u = 2793 at the C5 shape) -- four waves share it, each walking its k-steps four rows per
// v_mfma_f64_16x16x4_f64 on two accumulators; the parts are added in wave order.  A[i = q][k = h] = PSI[row 4s + h][q] (a row of PSI is 128
// contiguous bytes), B[k = h][j = c] = R u0 of row 4s + h, instance b0 + c (16 instances = one 128-byte line).  The VALU version (a lane per
// instance, PSI through 112 scalar loads per step) took 0.12 ms for these 183 MB; this one is bound by reading them.
typedef double d4c_t __attribute__((ext_vector_type(4)));
template <class S, int KWP>
__global__ __launch_bounds__(256) void k_cpl_quad(Bufs a, CPArgs c) {
    static_assert(KWP == 16, "one 16-column tile");
    constexpr int NU = S::NU, NW = 4;  // NW waves share the inner dimension of a tile (k-step s goes to wave s mod NW): 4x the loads in flight
    __shared__ double sR[8];
    __shared__ double sPart[NW][5][64];
    const DevDesc& d = *a.desc;
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, h = l >> 4, cc = l & 15;
    const int b = blockIdx.x * 16 + cc;  // (b < Bp: Bp is a multiple of 64)
    const int Bp = d.Bp, T = d.T, rows = (T - 1) * NU;
    if (threadIdx.x < NU) sR[threadIdx.x] = d.R_diag[threadIdx.x];
    __syncthreads();
    d4c_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    double c00 = 0;
    const int nk = (rows + 3) / 4;
    int rm = (4 * wv + h) % NU;  // (4 s + h) mod n_u along s = wv, wv + NW, ...
    auto kstep = [&](int s, d4c_t& acc) {
        const int row = 4 * s + h;
        const bool ok = row < rows;
        const int rc = ok ? row : rows - 1;
        const double ps = PSI(rc, cc), u = AT(a.U0, rc, b), ru = sR[rm] * u;
        rm += (4 * NW) % NU; if (rm >= NU) rm -= NU;
        c00 = fma(ok ? u : 0.0, ru, c00);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ok ? ps : 0.0, ok ? ru : 0.0, acc, 0, 0, 0);
    };
    for (int s0 = wv; s0 < nk; s0 += 8 * NW) {  // (k-steps beyond the end multiply zeros)
        UNR for (int j = 0; j < 8; j += 2) { kstep(s0 + j * NW, acc0); kstep(s0 + (j + 1) * NW, acc1); }
    }
    c00 += __shfl_xor(c00, 16);
    c00 += __shfl_xor(c00, 32);
    UNR for (int r = 0; r < 4; r++) sPart[wv][r][l] = acc0[r] + acc1[r];
    sPart[wv][4][l] = c00;
    __syncthreads();
    if (wv != 0 || b >= d.B) return;
    UNR for (int r = 0; r < 4; r++) {  // D[row = h + 4 r][col = cc]: column q = h + 4 r of the basis, instance b; the waves' parts in wave order
        double g = sPart[0][r][l];
        UNR for (int w_ = 1; w_ < NW; w_++) g += sPart[w_][r][l];
        AT(c.g0, h + 4 * r, b) = g;
        AT(c.wv, h + 4 * r, b) = 0;
    }
    if (h == 0) {
        double cs = sPart[0][4][l];
        UNR for (int w_ = 1; w_ < NW; w_++) cs += sPart[w_][4][l];
        c.c00[b] = cs;
        a.cur[b] = 0;
        a.active[b] = 1;
        a.iters[b] = 0;
        a.status[b] = 0;
        a.alpha[b] = 1.0;
        a.pend[b] = 0;
        a.pred[b] = 0;
    }
}

// ------------------------------------------------------------------------------------------------ host side

template <class T>
static bool cp_alloc(BatchCPState& st, T** p, size_t n, hipStream_t s) {
    void* q = nullptr;
    if (hipMalloc(&q, (n ? n : 1) * sizeof(T)) != hipSuccess) return false;
    st.allocs.push_back(q);
    (void)hipMemsetAsync(q, 0, (n ? n : 1) * sizeof(T), s);
    *p = (T*)q;
    return true;
}

void batchcp_free(BatchCPState& st) {
    for (void* q : st.allocs) (void)hipFree(q);
    st = BatchCPState();
}

template <class S, int KWP>
static int run_cp(BatchCPState& st, const DevDesc& h, Bufs& bufs, int nb_iter, int early_stop, hipStream_t stream, std::string& err, const ProfHook& ph) {
    constexpr int NX = S::NX;
    const int B = h.B;
    const dim3 grid((B + LPB - 1) / LPB), block(LPB);
    const size_t lds_h = sizeof(double) * KWP * KWP * LPB;
    if (hipFuncSetAttribute((const void*)k_cp_solve<S, KWP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h) != hipSuccess) {
        err = "ilqr_solve_batch_cp: cannot reserve LDS";
        return 1;
    }
    CPArgs c;
    c.psi = st.psi; c.H0 = st.H0; c.Wkp = st.Wkp; c.Ckp = st.Ckp; c.rkp = st.rkp; c.gu = st.gu; c.dw = st.dw; c.dun = st.dun;
    c.Kw = st.Kw; c.early_stop = early_stop; c.n_alpha = 11; c.it = 0;
    c.wref = nullptr;
    const bool wave_solve = !st.xc_lane_solve;  // cross-check (ilqr_ctx_set_crosscheck): one lane per instance
    ph(ILQR_PROF_ROLLOUT);
    hipLaunchKernelGGL((k_cp_init<S>), dim3((B + 255) / 256), dim3(256), 0, stream, bufs);
    for (int it = 0; it < nb_iter; it++) {
        c.it = it;
        ph(ILQR_PROF_BACKWARD);  // linearisation + normal equations + du = PSI dw
        hipLaunchKernelGGL((k_cp_linearize<S, KWP>), dim3((B + 64 / KWP - 1) / (64 / KWP)), dim3(64), 0, stream, bufs, c);
        if (wave_solve) hipLaunchKernelGGL((k_cp_solve_w<S, KWP>), dim3((B + 64 / KWP - 1) / (64 / KWP)), dim3(64), 0, stream, bufs, c);
        else hipLaunchKernelGGL((k_cp_solve<S, KWP>), grid, block, lds_h, stream, bufs, c);
        hipLaunchKernelGGL((k_cp_du<S, KWP>), dim3((B + 63) / 64, h.T - 1), dim3(64), 0, stream, bufs, c);
        BTArgs bt;
        bt.it = it; bt.early_stop = early_stop;
        ph(ILQR_PROF_FORWARD);   // backtracking over all step sizes (rollouts)
        hipLaunchKernelGGL((k_bt_linesearch<S>), dim3((B + 3) / 4), dim3(64), 0, stream, bufs, bt);
    }
    ph(ILQR_PROF_APPLY);
    hipLaunchKernelGGL((k_cp_final<S>), grid, block, 0, stream, bufs);
    if (hipGetLastError() != hipSuccess) { err = "ilqr_solve_batch_cp: kernel launch failed"; return 1; }
    return 0;
}

// PosOrn systems: iterate in coefficient space (see the kernels above).  psip = PSI padded to KWP columns (host copy).
template <class S, int KWP>
static int run_cpl(BatchCPState& st, const DevDesc& h, Bufs& bufs, const std::vector<double>& psip, int nb_iter, int early_stop, hipStream_t stream,
                   std::string& err, const ProfHook& ph) {
    constexpr int NX = S::NX, NU = S::NU, ND = S::ND;
    const int B = h.B, T = h.T, nkp = h.n_kp;
    const double dt = h.dt, hdt2 = dt * dt / 2;
    // sensitivities by their recurrences (constant A, B): Wt_{i+1} = A Wt_i + B PSI_i from Wt_0 = 0 (true), and the reference's
    // shifted one Wr_{i+1} = A Wr_i + B PSI_i from Wr_1 = 0, i >= 1 (BatchILQRCP.cpp:61-97, quirk D-1)
    auto advance = [&](std::vector<double>& W, int i) {
        for (int q = 0; q < KWP; q++) {
            if (ND == 2) for (int r = 0; r < DOF; r++) W[r * KWP + q] += dt * W[(DOF + r) * KWP + q];
            for (int r = 0; r < DOF; r++) {
                const double ps = psip[((size_t)i * NU + r) * KWP + q];
                if (ND == 1) W[r * KWP + q] += dt * ps;
                else { W[r * KWP + q] += hdt2 * ps; W[(DOF + r) * KWP + q] += dt * ps; }
            }
        }
    };
    if (!st.cpl_tables) {
    std::vector<std::vector<double>> Wt(T, std::vector<double>(NX * KWP, 0.0)), Wr(T, std::vector<double>(NX * KWP, 0.0));
    for (int i = 0; i + 1 < T; i++) { Wt[i + 1] = Wt[i]; advance(Wt[i + 1], i); }
    for (int i = 1; i + 1 < T; i++) { Wr[i + 1] = Wr[i]; advance(Wr[i + 1], i); }
    std::vector<double> wt((size_t)(nkp > 0 ? nkp : 1) * 2 * NX * KWP, 0.0), wref((size_t)(nkp > 0 ? nkp : 1) * NX * KWP, 0.0), pp((size_t)KWP * KWP, 0.0);
    for (int t = 0; t < nkp; t++) {
        const int ts = h.kp_t[t];
        std::copy(Wt[ts].begin(), Wt[ts].end(), wt.begin() + ((size_t)t * 2 + 0) * NX * KWP);
        if (ts > 0) std::copy(Wt[ts - 1].begin(), Wt[ts - 1].end(), wt.begin() + ((size_t)t * 2 + 1) * NX * KWP);
        std::copy(Wr[ts].begin(), Wr[ts].end(), wref.begin() + (size_t)t * NX * KWP);
    }
    const int rows = (T - 1) * NU;
    for (int a_ = 0; a_ < KWP; a_++)
        for (int b_ = 0; b_ < KWP; b_++) {
            double s = 0;
            for (int k = 0; k < rows; k++) s += psip[(size_t)k * KWP + a_] * psip[(size_t)k * KWP + b_];
            pp[(size_t)a_ * KWP + b_] = s;
        }
    if (hipMemcpyAsync(st.wt, wt.data(), wt.size() * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipMemcpyAsync(st.wref, wref.data(), wref.size() * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipMemcpyAsync(st.pp, pp.data(), pp.size() * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) {
        err = "ilqr_solve_batch_cp: sensitivity upload failed";
        return 1;
    }
    st.cpl_tables = true;
    }  // tables
    const dim3 grid((B + LPB - 1) / LPB), block(LPB);
    const size_t lds_h = sizeof(double) * KWP * KWP * LPB;
    if (hipFuncSetAttribute((const void*)k_cp_solve<S, KWP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h) != hipSuccess) {
        err = "ilqr_solve_batch_cp: cannot reserve LDS";
        return 1;
    }
    CPArgs c;
    c.psi = st.psi; c.H0 = st.H0; c.Wkp = st.Wkp; c.Ckp = st.Ckp; c.rkp = st.rkp; c.gu = st.gu; c.dw = st.dw; c.dun = st.dun;
    c.wt = st.wt; c.pp = st.pp; c.wv = st.wv; c.g0 = st.g0; c.c00 = st.c00; c.xbk = st.xbk;
    c.Kw = st.Kw; c.early_stop = early_stop; c.n_alpha = 11; c.it = 0;
    c.wref = st.wref;
    const bool wave_solve = !st.xc_lane_solve;  // cross-check (ilqr_ctx_set_crosscheck): one lane per instance
    ph(ILQR_PROF_ROLLOUT);   // rollout of u0, its keypoint states and the quadratic forms of the control cost (walks the horizon)
    hipLaunchKernelGGL((k_cpl_states<S>), dim3((B + 255) / 256, DOF), dim3(256), 0, stream, bufs, c);
    hipLaunchKernelGGL((k_cpl_quad<S, KWP>), dim3((B + 15) / 16), dim3(256), 0, stream, bufs, c);
    if (nkp > 0) hipLaunchKernelGGL(k_cpl_bcast, dim3((B + 255) / 256, nkp * NX * KWP), dim3(256), 0, stream, st.wref, st.Wkp, nkp * NX * KWP, B, h.Bp);
    for (int it = 0; it < nb_iter; it++) {
        c.it = it;
        ph(ILQR_PROF_BACKWARD);  // keypoint linearisation + the Kw x Kw normal equations
        hipLaunchKernelGGL((k_cpl_linearize<S, KWP>), dim3((B + LPB - 1) / LPB, nkp > 0 ? nkp : 1), block, 0, stream, bufs, c);
        if (wave_solve) hipLaunchKernelGGL((k_cp_solve_w<S, KWP>), dim3((B + 64 / KWP - 1) / (64 / KWP)), dim3(64), 0, stream, bufs, c);
        else hipLaunchKernelGGL((k_cp_solve<S, KWP>), grid, block, lds_h, stream, bufs, c);
        ph(ILQR_PROF_FORWARD);   // all step sizes of the backtracking, in coefficient space
        hipLaunchKernelGGL((k_cpl_linesearch<S, KWP, 4, 0>), dim3((B + 15) / 16), dim3(64), 0, stream, bufs, c);
        hipLaunchKernelGGL((k_cpl_linesearch<S, KWP, 8, 4>), dim3((B + 7) / 8), dim3(64), 0, stream, bufs, c);
    }
    ph(ILQR_PROF_APPLY);     // u = u0 + PSI w and the final rollout (walks the horizon)
    hipLaunchKernelGGL((k_cpl_final<S, KWP>), dim3((B + 255) / 256, DOF), dim3(256), 0, stream, bufs, c);
    if (hipGetLastError() != hipSuccess) { err = "ilqr_solve_batch_cp: kernel launch failed"; return 1; }
    return 0;
}

int batchcp_solve(BatchCPState& st, const DevDesc& h, Bufs& bufs, int nx, int nu, int nf, int nq, const double* psi_host, int Kw,
                  int nb_iter, int early_stop, hipStream_t stream, std::string& err, const ProfHook& ph) {
    (void)nf; (void)nq;
    if (!psi_host || Kw <= 0) { err = "ilqr_solve_batch_cp: null PSI / Kw <= 0"; return 1; }
    const bool time_sys = h.kind == 1 || h.kind == 3;
    if (Kw > 32 || (Kw > 16 && !time_sys)) { err = "ilqr_solve_batch_cp: this path takes Kw <= 16 (Kw <= 32 on the time systems)"; return 1; }
    if (nb_iter < 0) { err = "nb_iter < 0"; return 1; }
    const int KWP = Kw > 16 ? 32 : 16, T = h.T, Bp = h.Bp, nkp = h.n_kp > 0 ? h.n_kp : 1;
    const int rows = (T - 1) * nu;
    if (st.KWP != KWP || st.nkp != nkp || st.nx != nx || st.Bp != Bp || st.rows != rows) {
        batchcp_free(st);
        bool ok = cp_alloc(st, &st.psi, (size_t)rows * KWP, stream) && cp_alloc(st, &st.H0, (size_t)KWP * KWP, stream) &&
                  cp_alloc(st, &st.Wkp, (size_t)nkp * nx * KWP * Bp, stream) && cp_alloc(st, &st.Ckp, (size_t)nkp * nx * nx * Bp, stream) &&
                  cp_alloc(st, &st.rkp, (size_t)nkp * nx * Bp, stream) && cp_alloc(st, &st.gu, (size_t)KWP * Bp, stream) &&
                  cp_alloc(st, &st.dw, (size_t)KWP * Bp, stream) && cp_alloc(st, &st.dun, (size_t)Bp, stream) &&
                  cp_alloc(st, &st.wt, (size_t)nkp * 2 * nx * KWP, stream) && cp_alloc(st, &st.pp, (size_t)KWP * KWP, stream) && cp_alloc(st, &st.wref, (size_t)nkp * nx * KWP, stream) &&
                  cp_alloc(st, &st.wv, (size_t)KWP * Bp, stream) && cp_alloc(st, &st.g0, (size_t)KWP * Bp, stream) &&
                  cp_alloc(st, &st.c00, (size_t)Bp, stream) && cp_alloc(st, &st.xbk, (size_t)nkp * 2 * nx * Bp, stream);
        if (!ok) { batchcp_free(st); err = "ilqr_solve_batch_cp: hipMalloc failed"; return 1; }
        st.KWP = KWP; st.nkp = nkp; st.nx = nx; st.Bp = Bp; st.rows = rows;
        st.sig.clear(); st.psi_host.clear(); st.cpl_tables = false;
    }
    st.Kw = Kw;
    // the shared tables depend on PSI and on (horizon, dt, R, keypoint steps): rebuilt only when one of them changes
    std::vector<double> sig = {(double)T, (double)nu, (double)nx, (double)Kw, h.dt, (double)h.kind, (double)h.nd, (double)h.n_kp};
    for (int i = 0; i < nu; i++) sig.push_back(h.R_diag[i]);
    for (int i = 0; i < h.n_kp; i++) sig.push_back((double)h.kp_t[i]);
    const bool same = st.sig == sig && st.psi_host.size() == (size_t)rows * Kw &&
                      std::memcmp(st.psi_host.data(), psi_host, sizeof(double) * (size_t)rows * Kw) == 0;
    if (!same) { st.cpl_tables = false; st.sig = sig; st.psi_host.assign(psi_host, psi_host + (size_t)rows * Kw); }
    // PSI padded to KWP columns; H0 = PSI' R PSI with 1 on the padded diagonal (keeps H non-singular, dw_pad = 0)
    std::vector<double> psip((size_t)rows * KWP, 0.0), H0((size_t)KWP * KWP, 0.0);
    for (int k = 0; k < rows; k++)
        for (int q = 0; q < Kw; q++) psip[(size_t)k * KWP + q] = psi_host[(size_t)k * Kw + q];
    if (!same) {
    for (int a_ = 0; a_ < Kw; a_++)
        for (int b_ = 0; b_ < Kw; b_++) {
            double s = 0;
            for (int k = 0; k < rows; k++) s += psi_host[(size_t)k * Kw + a_] * h.R_diag[k % nu] * psi_host[(size_t)k * Kw + b_];
            H0[(size_t)a_ * KWP + b_] = s;
        }
    for (int q = Kw; q < KWP; q++) H0[(size_t)q * KWP + q] = 1.0;
    if (hipMemcpyAsync(st.psi, psip.data(), psip.size() * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipMemcpyAsync(st.H0, H0.data(), H0.size() * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) {
        err = "ilqr_solve_batch_cp: PSI upload failed";
        st.sig.clear();
        return 1;
    }
    }  // !same
    const bool general = st.xc_general;  // cross-check path (ilqr_ctx_set_crosscheck)
    if (KWP == 32) {  // wider bases on the time systems: the same kernels with 32 lanes per instance
        if (h.kind == 3) return run_cp<Sys<3, 1>, 32>(st, h, bufs, nb_iter, early_stop, stream, err, ph);
        if (h.nd == 1) return run_cp<Sys<1, 1>, 32>(st, h, bufs, nb_iter, early_stop, stream, err, ph);
        return run_cp<Sys<1, 2>, 32>(st, h, bufs, nb_iter, early_stop, stream, err, ph);
    }
    if (h.kind == 3) return run_cp<Sys<3, 1>, 16>(st, h, bufs, nb_iter, early_stop, stream, err, ph);
    if (h.kind == 2 && !general) return run_cpl<Sys<2, 1>, 16>(st, h, bufs, psip, nb_iter, early_stop, stream, err, ph);
    if (h.kind == 2) return run_cp<Sys<2, 1>, 16>(st, h, bufs, nb_iter, early_stop, stream, err, ph);
    if (h.kind == 0 && h.nd == 1 && !general) return run_cpl<Sys<0, 1>, 16>(st, h, bufs, psip, nb_iter, early_stop, stream, err, ph);
    if (h.kind == 0 && h.nd == 2 && !general) return run_cpl<Sys<0, 2>, 16>(st, h, bufs, psip, nb_iter, early_stop, stream, err, ph);
    if (h.kind == 0 && h.nd == 1) return run_cp<Sys<0, 1>, 16>(st, h, bufs, nb_iter, early_stop, stream, err, ph);
    if (h.kind == 0 && h.nd == 2) return run_cp<Sys<0, 2>, 16>(st, h, bufs, nb_iter, early_stop, stream, err, ph);
    if (h.kind == 1 && h.nd == 1) return run_cp<Sys<1, 1>, 16>(st, h, bufs, nb_iter, early_stop, stream, err, ph);
    return run_cp<Sys<1, 2>, 16>(st, h, bufs, nb_iter, early_stop, stream, err, ph);
}

}  // namespace ilqr
