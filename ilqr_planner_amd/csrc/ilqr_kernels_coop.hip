// ilqr_kernels_coop.hip -- wave-cooperative closed-form backward Riccati sweep (gfx950, fp64).
//
// Same mathematics as k_backward_si (ilqr_kernels_v2.hip): single-integrator dynamics, S = D + dt^2 P, M = S^-1,
//   K = (M D - I)/dt,  d = -M Qu,  P' = l_xx + [D - D M D - reg (D M^2 D - D M - M D + I)]/dt^2,
//   p' = l_x + p - (Qu + D d)/dt - reg (D M d - d)/dt
// but mapped for a batch that is too small to fill the chip with one lane per instance (B = 4096 is 64 waves):
// LPI lanes (32, 16 or 8) cooperate on ONE instance, 64/LPI instances per single-wave workgroup.
//   every lane owns EPL = ceil(28/LPI) of the 28 unique entries (i >= j) of the symmetric 7x7 matrices P, S, M, M^2
//   lanes 0..6 additionally own component l of the vectors p, x, u, Qu, d
// The full matrix lives in LDS (rows padded to 64 B) so that any lane can read any row; the inverse is the symmetric
// sweep operator applied to the 7 pivots in order (a_cc <- -1/a_cc, a_ic <- a_ic/a_cc, a_ij <- a_ij - a_ic a_jc/a_cc;
// after all pivots the matrix holds -S^-1), one LDS round trip per pivot; S is SPD so no pivoting is needed.  (Forming the
// reciprocal of the NEXT pivot early from two extra broadcast reads shortened the chain when the launch was thought latency-bound; it
// is issue-bound at two waves per SIMD, and reading the pivot after the update -- three instructions less per pivot -- measured 2 % faster.)
// All exchange is wave-local: LDS operations of one wave execute in order, so there is no barrier anywhere.
//
// FUSED = true (the sweep follows k_forward_wg + k_select, ilqr_kernels_wave.hip): the acceptance of the previous iteration's line
// search is folded into the sweep's load path instead of being a pass of its own (k_apply moved 190 MB per iteration that this
// kernel re-read anyway).  Per step the vector lanes load xbar, ubar AND the alpha = 1 rollout x(1), u(1), form the accepted
// trajectory x = xbar + alpha (x(1) - xbar) (the expression of k_apply, bit for bit), store it over x(1), and do the AL bookkeeping
// of AL-ILQR.cpp:190,202-208 on it in registers: g = A x - b, I_k = penalty (g<0 && lambda==0 ? 0 : 1) with the multipliers
// BEFORE the update, lambda_k = max(0, lambda_k + penalty' g) on update iterations -- the I_k buffer is neither written nor read.
// At the end the instance's buffers flip.  Instances that stopped early but still have a pending acceptance run the same pass
// (their gains are not stored).  The last iteration of a solve is finished by k_apply.
#include <cstdlib>

#include "ilqr_kernels.hpp"
#include "ilqr_step.hpp"

namespace ilqr {

__device__ __forceinline__ double rcp_nr(double x) {  // 1/x to the last bit: v_rcp_f64 (24 good bits: measured 4.6e-8) and ONE cubic
    const double r = __builtin_amdgcn_rcp(x);               // step r (1 + e + e^2), e = 1 - x r  -- max error 1.1e-16 over 2^20 samples, one
    const double e = fma(-x, r, 1.0);                       // FMA less than two Newton steps (no IEEE division sequence)
    return fma(fma(e, e, e), r, r);
}
#define LDS_ORDER() asm volatile("" ::: "memory")

// DPP move of a double (two 32-bit halves); quad_perm / row_half_mirror build an 8-lane butterfly without LDS traffic.
template <int CTRL>
__device__ __forceinline__ double dpp64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);  // (every lane has a source in these patterns: no "old" value to keep, no copy)
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double oct_sum(double v) {  // sum over lanes 8m .. 8m+7, result in all eight
    v += dpp64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp64<0x141>(v);  // row_half_mirror
    return v;
}

template <int MR, bool FUSED>
__global__ __launch_bounds__(64) void k_backward_si_coop(Bufs a, SweepArgs sw) {
    constexpr int LPI = 32;  // lanes per instance (8 and 16 measured slower at B = 4096: one wave per SIMD or less exposes the chain)
    constexpr int N = 7, IPW = 64 / LPI, EPL = (28 + LPI - 1) / LPI;
    constexpr int MRR = MR > 0 ? MR : 1;
    constexpr int ROWP = kd_rowp(N), RS = N * ROWP;
    __shared__ __attribute__((aligned(16))) double sA[IPW][N][10];  // symmetric matrix, full storage; rows padded to 80 B: row starts fall
                                                                     // in distinct banks (20 i mod 64) and stay 16-B aligned for ds_read_b128
    __shared__ double sV[IPW][3][8];   // 0: x   1: Qu   2: dv
    // gain records of the wave's instances as they lie in memory (IPW records of RS doubles, adjacent): the lanes write their entries here
    // and the record goes out as 16-byte pieces of whole lines -- one store per step instead of three 8-byte ones (K_ij, K_ji, d)
    __shared__ __attribute__((aligned(16))) double sK[IPW][RS];
    const DevDesc& d = *a.desc;
    const int lane = threadIdx.x, g = lane / LPI, l = lane % LPI;
    const int b = xcd_tile() * IPW + g;
    const int Bp = d.Bp, T = d.T;
    const bool ok = (b < d.B) && a.active[b < d.B ? b : 0];
    const int bb = (b < d.B) ? b : 0;           // clamp so that every address stays valid; stores are guarded by `ok` / `acc`
    // FUSED: winner of the previous line search still to be applied (k_select), 0 = none (first iteration, or nothing ran)
    const int pendw = (FUSED && b < d.B) ? a.pend[bb] : 0;
    const bool acc = pendw > 0;
    if (__ballot((ok || acc) ? 1 : 0) == 0ull) return;  // wave-uniform

    const bool isV = l < N;
    const int v = isV ? l : 0;
    const int cur = a.cur[bb];
    const double* X = a.X[cur];
    const double* U = a.U[cur];
    // accepted trajectory x = xbar + alpha (x(1) - xbar): x(1) lives in the other buffer and is overwritten by x.  Without a pending
    // acceptance both pointers read xbar (x(1) - xbar = 0 reproduces xbar exactly) and nothing is stored.
    const double aacc = acc ? ldexp(1.0, -(pendw - 1)) : 1.0;
    const bool blend = acc && pendw > 1;        // alpha = 1: the other buffer already holds the accepted trajectory
    double* X1 = blend ? a.X[1 - cur] : a.X[cur];
    double* U1 = blend ? a.U[1 - cur] : a.U[cur];
    if (FUSED && acc && !blend) { X = a.X[1 - cur]; U = a.U[1 - cur]; X1 = a.X[1 - cur]; U1 = a.U[1 - cur]; }
    const bool upd = FUSED && acc && sw.do_update_prev;  // multiplier update of the iteration whose acceptance is applied here
    const double dt = d.dt, idt = 1.0 / dt, idt2 = idt * idt, reg = d.reg, dt2 = dt * dt;
    const double Rv = d.R_diag[v], Dv = Rv + reg;
    const int lim_on = d.limits_set;
    const double pen = d.penalty, pen_xx = d.pen_xx;
    // bounds with the weight folded in: an unweighted coordinate gets (+inf, -inf), which no x violates (a NaN neither) -- the step below
    // forms the limit terms by max() arithmetic without testing the weight or the kind of entry
    const int lw_v = d.lw[v];
    const double smax_v = lw_v != 0 ? d.smax[v] : __builtin_inf(), smin_v = lw_v != 0 ? d.smin[v] : -__builtin_inf();

    // entries owned by this lane: n = l + e*LPI (entries beyond 27 shadow entry 0 and never store)
    bool isE[EPL], dg[EPL];
    int ei[EPL], ej[EPL];
    double Di[EPL], Dj[EPL], del[EPL], smax_i[EPL], smin_i[EPL];
    int lw_i[EPL];
    double *sAij[EPL], *sAji[EPL];
    const double *rowI[EPL], *rowJ[EPL];
    double *Kij[EPL], *Kji[EPL];  // entries (i, j) and (j, i) of the gain record image in LDS
    UNR for (int e = 0; e < EPL; e++) {
        const int n = l + e * LPI;
        isE[e] = n < 28;
        const int le = isE[e] ? n : 0;
        const int i = (le >= 1) + (le >= 3) + (le >= 6) + (le >= 10) + (le >= 15) + (le >= 21);
        const int j = le - i * (i + 1) / 2;
        ei[e] = i; ej[e] = j; dg[e] = (i == j);
        Di[e] = d.R_diag[i] + reg; Dj[e] = d.R_diag[j] + reg; del[e] = dg[e] ? 1.0 : 0.0;
        lw_i[e] = d.lw[i];
        const bool we = dg[e] && lw_i[e] != 0;  // l_xx of the limits is diagonal
        smax_i[e] = we ? d.smax[i] : __builtin_inf(); smin_i[e] = we ? d.smin[i] : -__builtin_inf();
        sAij[e] = &sA[g][i][j]; sAji[e] = &sA[g][j][i];
        rowI[e] = &sA[g][i][0]; rowJ[e] = &sA[g][j][0];
        Kij[e] = &sK[g][i * ROWP + j];
        Kji[e] = &sK[g][j * ROWP + i];
    }
    const double* rowV = &sA[g][v][0];
    // sweep operands per pivot c: a_ic, a_jc from the lower triangle (rows are padded: column 9 of row 0 holds the constant -1 the entries
    // of the pivot row / column read instead, see the sweeps below)
    const double* pvI[N][EPL];
    const double* pvJ[N][EPL];
    double pvM[N][EPL];
    UNR for (int c = 0; c < N; c++) {
        UNR for (int e = 0; e < EPL; e++) {
            const double* neg = &sA[g][0][9];
            pvI[c][e] = (ei[e] == c) ? neg : ((c <= ei[e]) ? &sA[g][ei[e]][c] : &sA[g][c][ei[e]]);
            pvJ[c][e] = (ej[e] == c) ? neg : ((c <= ej[e]) ? &sA[g][ej[e]][c] : &sA[g][c][ej[e]]);
            pvM[c][e] = (ei[e] == c || ej[e] == c) ? 0.0 : 1.0;
        }
    }
    if (l == 0) sA[g][0][9] = -1.0;
    // constraint rows (state part only: checked on the host)
    const int m = a.m;
    double Ai[MRR][EPL], Aj[MRR][EPL], Av[MRR], bbr[MRR], Arow[MRR][N];
    UNR for (int r = 0; r < MRR; r++) {
        Av[r] = bbr[r] = 0;
        UNR for (int e = 0; e < EPL; e++) Ai[r][e] = Aj[r][e] = 0;
        UNR for (int q = 0; q < N; q++) Arow[r][q] = 0;
        if (MR > 0 && r < m) {
            const double* Ar = a.conA + (size_t)r * 2 * N;
            UNR for (int e = 0; e < EPL; e++) { Ai[r][e] = Ar[ei[e]]; Aj[r][e] = Ar[ej[e]]; }
            Av[r] = Ar[v]; bbr[r] = a.conb[r];
            UNR for (int q = 0; q < N; q++) Arow[r][q] = Ar[q];
        }
    }
    // running pointers (decremented by one timestep per iteration: no 64-bit multiplies in the loop)
    const ptrdiff_t Kstep = (ptrdiff_t)Bp * RS, Vstep = (ptrdiff_t)N * Bp, Lstep = (ptrdiff_t)m * Bp;
    double* Dv_out = &sK[g][v * ROWP + N];
    // lane c < IPW * RS / 2 stores piece c of the image: piece c belongs to instance c / (RS / 2) of the wave
    constexpr int PCS = RS / 2;
    static_assert(RS % 2 == 0 && IPW * PCS <= 64, "one 16-byte piece per lane");
    const int gc = (lane / PCS < IPW) ? lane / PCS : 0, cc = lane % PCS;
    const unsigned long long okm = __ballot(ok ? 1 : 0);
    const bool okc = lane < IPW * PCS && ((okm >> (gc * LPI)) & 1ull);
    const int bc = xcd_tile() * IPW + gc;
    double* Kout = KD_REC(a.KD, Bp, RS, T - 2, (bc < d.B) ? bc : 0) + cc * 2;
    const double* Kimg = &sK[gc][cc * 2];

    int kpi = d.n_kp - 1;
    int kp_next = (kpi >= 0) ? d.kp_t[kpi] : -1;
    const size_t kpd_stride = (size_t)(N + N * N) * Bp;

    // terminal values: P = l_xx(x_{T-1}), p = l_x(x_{T-1})
    double P[EPL], p = 0;
    UNR for (int e = 0; e < EPL; e++) P[e] = 0;
    {
        double xv = AT(X, (T - 1) * N + v, bb);
        if (FUSED) {
            const double x1 = AT(X1, (T - 1) * N + v, bb);
            xv = fma(aacc, x1 - xv, xv);
            if (blend && isV) AT(X1, (T - 1) * N + v, bb) = xv;
        }
        if (kp_next == T - 1) {
            const double* src = a.kpd + (size_t)kpi * kpd_stride;
            UNR for (int e = 0; e < EPL; e++) P[e] = AT(src, N + ei[e] * N + ej[e], bb);
            p = AT(src, v, bb);
            kpi--;
            kp_next = (kpi >= 0) ? d.kp_t[kpi] : -1;
        } else if (lim_on) {
            UNR for (int e = 0; e < EPL; e++) {
                double xi = AT(X, (T - 1) * N + ei[e], bb);
                if (FUSED) xi = fma(aacc, AT(X1, (T - 1) * N + ei[e], bb) - xi, xi);
                if (dg[e] && lw_i[e] != 0 && (xi > smax_i[e] || xi < smin_i[e])) P[e] = pen_xx;
            }
            if (lw_v != 0) {
                if (xv > smax_v) p = -pen * (smax_v - xv);
                else if (xv < smin_v) p = -pen * (smin_v - xv);
            }
        }
    }
    // PF-steps-ahead prefetch ring (vmcnt retires loads and stores in issue order, so a load issued only one step ahead
    // would wait for all of the previous step's gain stores to be acknowledged)
    constexpr int PF = 4;
    const double* Xp = X + (size_t)v * Bp + bb + (size_t)(T - 2) * Vstep;
    const double* Up = U + (size_t)v * Bp + bb + (size_t)(T - 2) * Vstep;
    double* X1p = X1 + (size_t)v * Bp + bb + (size_t)(T - 2) * Vstep;
    double* U1p = U1 + (size_t)v * Bp + bb + (size_t)(T - 2) * Vstep;
    double* Lp = a.lambda + bb + (size_t)(T - 2) * Lstep;
    const double* Ip = a.Is + bb + (size_t)(T - 2) * Lstep;
    double xr[PF], ur[PF], x1r[PF], u1r[PF], lr[PF][MRR], ir[PF][MRR];
    // Every load of the ring is unconditional and the unrolled group has no early exit: a CFG path that skips a fetch makes
    // the waitcnt pass fall back to vmcnt(0), which also waits for the previous step's gain stores.  Rows r >= m re-read row 0
    // (never used); steps below 0 re-read step 0 (the pointers stop there) and their work is skipped.
    size_t rofs[MRR];
    UNR for (int r = 0; r < MRR; r++) rofs[r] = (size_t)((MR > 0 && r < m) ? r : 0) * Bp;
    auto fetch = [&](int slot, int kk) {  // loads of timestep max(kk, 0) into ring slot
        xr[slot] = *Xp;
        ur[slot] = *Up;
        x1r[slot] = u1r[slot] = 0;
        if (FUSED) { x1r[slot] = *X1p; u1r[slot] = *U1p; }
        UNR for (int r = 0; r < MRR; r++) {
            lr[slot][r] = ir[slot][r] = 0;
            if (MR > 0) { lr[slot][r] = Lp[rofs[r]]; if (!FUSED) ir[slot][r] = Ip[rofs[r]]; }
        }
        if (kk > 0) { Xp -= Vstep; Up -= Vstep; Lp -= Lstep; Ip -= Lstep; if (FUSED) { X1p -= Vstep; U1p -= Vstep; } }  // uniform; no load inside the branch
    };
    UNR for (int q = 0; q < PF; q++) { fetch(q, T - 2 - q); __builtin_amdgcn_sched_barrier(0); }  // keep issue order (vmcnt)
    // store side of the fused acceptance: the accepted x, u of step k go where x(1), u(1) were read (PF steps behind the loads)
    double* Xw = X1 + (size_t)v * Bp + bb + (size_t)(T - 2) * Vstep;
    double* Uw = U1 + (size_t)v * Bp + bb + (size_t)(T - 2) * Vstep;
    double* Lw = a.lambda + bb + (size_t)(T - 2) * Lstep;

    for (int k0 = T - 2; k0 >= 0; k0 -= PF) {
      UNR for (int jj = 0; jj < PF; jj++) {
        const int k = k0 - jj;
        double xv = xr[jj], uv = ur[jj];
        if (FUSED) {  // accepted trajectory (k_apply's expression)
            xv = fma(aacc, x1r[jj] - xv, xv);
            uv = fma(aacc, u1r[jj] - uv, uv);
        }
        double lam[MRR], Isk[MRR];
        UNR for (int r = 0; r < MRR; r++) { lam[r] = lr[jj][r]; Isk[r] = ir[jj][r]; }
        fetch(jj, k - PF);
        if (k < 0) continue;  // uniform: dummy step of the last group
        if (FUSED) {
            if (blend && isV) { *Xw = xv; *Uw = uv; }
            Xw -= Vstep; Uw -= Vstep;
        }
        // ---- vectors in: x, Qu = R u + dt p
        const double Qu = Rv * uv + dt * p;
        if (isV) { sV[g][0][v] = xv; sV[g][1][v] = Qu; }
        // ---- S = D + dt^2 P.  The kernel is bound by LDS bandwidth (128 B/clk per CU: a 64-lane b64 access is 4 clk whatever
        // the lanes do), so during the sweeps only the lower triangle is kept (ONE write per entry and sweep; the reads pick
        // sA[i][c] or sA[c][i]) and the last sweep's result goes out once, negated, into the full matrix for the row products.
        double s[EPL];
        UNR for (int e = 0; e < EPL; e++) {
            s[e] = dt2 * P[e] + (dg[e] ? Di[e] : 0.0);
            *sAij[e] = s[e];
        }
        LDS_ORDER();
        // ---- symmetric sweeps: after the 7 pivots the matrix holds -S^-1
        UNR for (int c = 0; c < N; c++) {
            double aic[EPL], ajc[EPL];
            UNR for (int e = 0; e < EPL; e++) {
                aic[e] = *pvI[c][e];
                ajc[e] = *pvJ[c][e];
            }
            const double r = rcp_nr(sA[g][c][c]);
            LDS_ORDER();
            UNR for (int e = 0; e < EPL; e++) {
                // ONE expression for all four kinds of entries: the entries of the pivot row / column read -1 for "their" a_ic / a_jc (pvI, pvJ)
                // and take 0 for their own value (pvM), which turns  s - (a_ic r) a_jc  into  a_ic r  (entry (i, c)),  a_jc r  (entry (c, j))
                // and  -r  (the pivot), each with the bits of the direct expression.  Three selects per entry and pivot (six v_cndmask)
                // became one multiplication.
                const double t = aic[e] * r;
                const double val = fma(-t, ajc[e], s[e] * pvM[c][e]);
                s[e] = val;
                if (c + 1 < N) *sAij[e] = val;
            }
            LDS_ORDER();
        }
        double M[EPL];
        UNR for (int e = 0; e < EPL; e++) {
            M[e] = -s[e];
            *sAij[e] = M[e];
            *sAji[e] = M[e];
        }
        LDS_ORDER();
        // ---- M^2 entries = row_i . row_j ; vector lanes: dv = -row_v . Qu  (two partial sums each: shorter chains)
        double m2[EPL];
        UNR for (int e = 0; e < EPL; e++) {
            double s0 = 0, s1 = 0;
            UNR for (int q = 0; q < N; q += 2) s0 += rowI[e][q] * rowJ[e][q];
            UNR for (int q = 1; q < N; q += 2) s1 += rowI[e][q] * rowJ[e][q];
            m2[e] = s0 + s1;
        }
        double rv[N];  // row v of M, read once for dv and M d
        UNR for (int q = 0; q < N; q++) rv[q] = rowV[q];
        // (Qu_q and dv_q could come from lane q by DPP row_newbcast instead of LDS: measured 3 % slower -- the two broadcasts sit
        // on the dv -> M d dependency chain.)
        double dv, Md;
        {
            double s0 = 0, s1 = 0;
            UNR for (int q = 0; q < N; q += 2) s0 += rv[q] * sV[g][1][q];
            UNR for (int q = 1; q < N; q += 2) s1 += rv[q] * sV[g][1][q];
            dv = -(s0 + s1);
        }
        if (isV) sV[g][2][v] = dv;
        LDS_ORDER();
        {
            double s0 = 0, s1 = 0;
            UNR for (int q = 0; q < N; q += 2) s0 += rv[q] * sV[g][2][q];
            UNR for (int q = 1; q < N; q += 2) s1 += rv[q] * sV[g][2][q];
            Md = s0 + s1;
        }
        // ---- gains out
        UNR for (int e = 0; e < EPL; e++) {
            if (isE[e]) {
                *Kij[e] = (M[e] * Dj[e] - del[e]) * idt;
                if (!dg[e]) *Kji[e] = (M[e] * Di[e]) * idt;
            }
        }
        if (isV) *Dv_out = dv;
        LDS_ORDER();
        if (okc) {
            const double k0v = Kimg[0], k1v = Kimg[1];
            Kout[0] = k0v; Kout[1] = k1v;
        }
        Kout -= Kstep;
        // ---- stage derivatives l_xx (entries), l_x (vector component)
        double lxx[EPL], lx = 0;
        UNR for (int e = 0; e < EPL; e++) lxx[e] = 0;
        if (k == kp_next) {  // uniform: keypoint step, precomputed by k_kp_derivs (incl. limits)
            const double* src = a.kpd + (size_t)kpi * kpd_stride;
            UNR for (int e = 0; e < EPL; e++) lxx[e] = AT(src, N + ei[e] * N + ej[e], bb);
            lx = AT(src, v, bb);
            // consume the loads inside the branch: otherwise their wait lands after the join and every step drains vmcnt to 0
            UNR for (int e = 0; e < EPL; e++) asm volatile("" : "+v"(lxx[e]));
            asm volatile("" : "+v"(lx));
            kpi--;
            kp_next = (kpi >= 0) ? __builtin_amdgcn_readfirstlane(d.kp_t[kpi]) : -1;
        } else if (lim_on) {  // uniform.  inspectJointLimit (System.cpp:121-142) branch-free: the nested tests were five exec-mask regions per step
            UNR for (int e = 0; e < EPL; e++) {
                const double xi = sV[g][0][ei[e]];
                const double beyond = fmax(xi - smax_i[e], 0.0) + fmax(smin_i[e] - xi, 0.0);
                lxx[e] = (beyond > 0.0) ? pen_xx : 0.0;
            }
            // l_x = -L q, q = limit - x on the violated side: pen (x - max) above, -pen (min - x) below -- the bits of -pen (max - x), -pen (min - x)
            lx = pen * fmax(xv - smax_v, 0.0) - pen * fmax(smin_v - xv, 0.0);
        }
        if (MR > 0) {
            UNR for (int rr = 0; rr < MRR; rr++) {
                if (rr < m) {
                    double gr = -bbr[rr];
                    if (LPI >= 16) {  // A_r . x: every vector lane multiplies its own component, 8-lane butterfly (lane 7 adds 0)
                        gr += oct_sum(isV ? Av[rr] * xv : 0.0);
                    } else {
                        UNR for (int q = 0; q < N; q++) gr += Arow[rr][q] * sV[g][0][q];
                    }
                    if (FUSED) {  // AL-ILQR.cpp:190 (mask with the multipliers before the update), :202-208 (update)
                        // gr is the constraint value only in the instance's first 8 lanes (oct_sum): lane 0's test is taken for all 32
                        const unsigned long long fb = __ballot((gr < 0 && lam[rr] == 0) ? 1 : 0);
                        Isk[rr] = sw.pen_in * (((fb >> (g * LPI)) & 1ull) ? 0.0 : 1.0);
                        if (upd) {
                            const double nv = lam[rr] + sw.pen_update_prev * gr;
                            lam[rr] = nv > 0 ? nv : 0;
                            if (l == 0) Lw[rofs[rr]] = lam[rr];
                        }
                    }
                    UNR for (int e = 0; e < EPL; e++) lxx[e] += Ai[rr][e] * Isk[rr] * Aj[rr][e];
                    lx += Av[rr] * (lam[rr] + Isk[rr] * gr);
                }
            }
        }
        // ---- P', p'
        UNR for (int e = 0; e < EPL; e++) {
            const double t1 = del[e] * Di[e] - Di[e] * M[e] * Dj[e] - reg * (Di[e] * m2[e] * Dj[e] - Di[e] * M[e] - M[e] * Dj[e] + del[e]);
            P[e] = lxx[e] + t1 * idt2;
        }
        p = lx + p - (Qu + Dv * dv) * idt - reg * (Dv * Md - dv) * idt;
        if (FUSED && MR > 0) Lw -= Lstep;
        LDS_ORDER();
      }
    }
    if (FUSED && acc && l == 0) {  // the acceptance is complete: the other buffer is the instance's trajectory now
        a.cur[bb] = 1 - cur;
        a.pend[bb] = 0;
    }
}

template <bool FUSED>
static void launch_coop(bool al, const Bufs& a, int B, hipStream_t st, const SweepArgs& sw) {
    const dim3 grid(grid_x8((B + 1) / 2)), block(64);  // two instances per single-wave workgroup
    if (!al || a.m == 0) hipLaunchKernelGGL((k_backward_si_coop<0, FUSED>), grid, block, 0, st, a, sw);
    else if (a.m <= 1) hipLaunchKernelGGL((k_backward_si_coop<1, FUSED>), grid, block, 0, st, a, sw);
    else hipLaunchKernelGGL((k_backward_si_coop<4, FUSED>), grid, block, 0, st, a, sw);
}

void launch_backward_si_coop(bool al, bool fused, const Bufs& a, int B, hipStream_t st, const SweepArgs& sw) {
    if (fused) launch_coop<true>(al, a, B, st, sw);
    else launch_coop<false>(al, a, B, st, sw);
}

}  // namespace ilqr
