// ilqr_kernels_gen.hip -- wave-cooperative backward Riccati sweep for every system kind (gfx950, fp64)
//
// k_backward (ilqr_kernels.hip) gives an instance one lane and keeps P (15 x 15 for PosOrnTime-2), Qxx, Qux, ... in registers:
// 450+ VGPRs worth of state, i.e. scratch traffic for every product, on 64 waves for B = 4096 -- 47 ms per sweep.  Here ONE
// WAVE owns an instance, the matrices live in LDS and the lanes split the entries (ILQRRecursive.cpp:68-97, AL terms of
// AL-ILQR.cpp:110-134):
//   B^T P, Qux = B^T P A, Quu = R + B^T P B          entrywise from the block structure A = [[I, dt I, 0], [0, I, 0], [0, 0, 1]],
//                                                    B = [[c1 I, bq], [c2 I, bv], [0, 2s]] (dt = s^2 for time systems); only the
//                                                    time column bc needs dot products (NX of them with P, NU with B^T P)
//   Qxx = l_xx + A^T P A                             in registers: a lane owns a 2x2 block (1x1 for NX <= 8) of Qxx and of P'
//   -(Quu + reg I)^-1                                symmetric sweep operator, one lane per entry, NU pivots, no pivoting (SPD)
//   K = Quu_inv Qux, d = Quu_inv Qu                  8-term dot products over contiguous LDS rows (K, Qux, T1 stored transposed)
//   P' = Qxx + K^T (Quu K + Qux) + Qxu K             16 terms per entry, operands shared inside a lane's block
//   p' = Qx + K^T (Quu d + Qu) + Qxu d
// Qxu = A^T P B is formed on its own, as the reference does.  Taking Qux^T instead looks harmless (P is symmetric in exact
// arithmetic) but is not: P is symmetric only up to rounding relative to its LARGEST entries (1e4 against 1e-5), and the
// substitution moved final costs by up to 8e-2 on 10 % of the time-system / second-order instances (measured; the oracle has
// the same switch, orc_set_variant, to demonstrate it).  All exchange is wave-local: LDS operations of one wave execute in
// order, no barrier anywhere.
#include <cstdlib>
#include <cstring>

#include "ilqr_step.hpp"

namespace ilqr {

#define LDS_ORDER() asm volatile("" ::: "memory")

__device__ __forceinline__ double rcp_nr_g(double x) {  // 1/x: v_rcp_f64 + two Newton steps
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}
__device__ __forceinline__ double wave_sum(double v) {  // all 64 lanes get the sum
    UNR for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

template <class S, bool AL>
__global__ __launch_bounds__(64) void k_backward_gen(Bufs a) {
    constexpr int NX = S::NX, NU = S::NU, ND = S::ND, TM = S::TM;
    constexpr int PS = 18;  // row stride of sP / sBtP (doubles): 16-byte aligned rows whose starts fall in distinct banks
    constexpr int TS = 10;  // row stride of the NU-wide matrices stored transposed ([state index][control index])
    constexpr int NE_UX = NU * NX, T_UX = (NE_UX + 63) / 64;
    constexpr int BLK = (NX > 8) ? 2 : 1, NB = (NX + BLK - 1) / BLK;
    constexpr int ROWP = kd_rowp(NX), RS = NU * ROWP;
    constexpr int MMAX = 16;
    static_assert(NB * NB <= 64 && NU * NU <= 64 && NX <= 16 && NU <= 8, "lane maps");
    __shared__ __attribute__((aligned(16))) double sP[16][PS], sBtP[8][PS];
    __shared__ __attribute__((aligned(16))) double sS[8][TS], sQuu[8][TS], sKt[16][TS], sT1t[16][TS], sQuxt[16][TS], sQxu[16][TS];
    __shared__ __attribute__((aligned(16))) double sx[16], su[8], sbc[16], sp[16], sQu[8], sQx[16], sd[8], stq[8], slam[MMAX], sIs[MMAX];

    const DevDesc& d = *a.desc;
    const int l = threadIdx.x;
    const int b = xcd_tile();
    if (b >= d.B) return;
    if (!a.active[b]) return;  // wave-uniform: one instance per wave
    const int Bp = d.Bp, T = d.T;
    const int cur = a.cur[b];
    const double* X = a.X[cur];
    const double* U = a.U[cur];
    const double reg = d.reg, pen = d.penalty, pen_xx = d.pen_xx;
    const int lim_on = d.limits_set;
    const int m = AL ? a.m : 0;

    // ---- lane maps
    int uxi[T_UX], uxj[T_UX];
    bool uxv[T_UX];
    UNR for (int t = 0; t < T_UX; t++) {
        const int e = l + 64 * t;
        uxv[t] = e < NE_UX;
        const int ee = uxv[t] ? e : 0;
        uxi[t] = ee / NX;
        uxj[t] = ee % NX;
    }
    const bool uuv = l < NU * NU;
    const int ui = uuv ? l / NU : 0, uj = uuv ? l % NU : 0;
    const bool blv = l < NB * NB;
    const int bi = blv ? l / NB : 0, bj = blv ? l % NB : 0;
    const bool isX = l < NX, isU = l < NU;
    const int vx = isX ? l : 0, vu = isU ? l : 0;
    const double Ru = d.R_diag[vu];
    const double smax_v = d.smax[vx], smin_v = d.smin[vx];
    const int lw_v = d.lw[vx];

    auto is_vrow = [](int i) { return ND == 2 && i >= DOF && i < 2 * DOF; };
    // diagonal entries of this lane's block that carry a limit weight (inspectJointLimit: l_xx_ii = penalty^2 when violated)
    bool dgl[BLK * BLK];
    int dgi[BLK * BLK];
    double dmx[BLK * BLK], dmn[BLK * BLK];
    UNR for (int r = 0; r < BLK; r++)
        UNR for (int c = 0; c < BLK; c++) {
            const int i = bi * BLK + r, j = bj * BLK + c, q = r * BLK + c;
            const bool on = blv && i == j && i < NX;
            dgi[q] = on ? i : 0;
            dgl[q] = on && d.lw[dgi[q]] != 0;
            dmx[q] = d.smax[dgi[q]];
            dmn[q] = d.smin[dgi[q]];
        }

    // ---- terminal values: P = l_xx(x_{T-1}), p = l_x(x_{T-1})
    int kpi = d.n_kp - 1;
    int kp_next = (kpi >= 0) ? d.kp_t[kpi] : -1;
    const size_t kpd_stride = (size_t)(NX + NX * NX) * Bp;
    // l_x (vector lanes) and l_xx (block entries) of step k; keypoint steps come from k_kp_derivs (limits included)
    auto stage_terms = [&](int k, double xv, double* lxxb, double& lxv) {
        UNR for (int q = 0; q < BLK * BLK; q++) lxxb[q] = 0;
        lxv = 0;
        if (k == kp_next) {  // uniform
            const double* src = a.kpd + (size_t)kpi * kpd_stride;
            UNR for (int r = 0; r < BLK; r++)
                UNR for (int c = 0; c < BLK; c++) {
                    const int i = bi * BLK + r, j = bj * BLK + c;
                    if (blv && i < NX && j < NX) lxxb[r * BLK + c] = AT(src, NX + i * NX + j, b);
                }
            if (isX) lxv = AT(src, vx, b);
            kpi--;
            kp_next = (kpi >= 0) ? d.kp_t[kpi] : -1;
        } else if (lim_on) {
            UNR for (int q = 0; q < BLK * BLK; q++)
                if (dgl[q]) {
                    const double xi = sx[dgi[q]];
                    if (xi > dmx[q] || xi < dmn[q]) lxxb[q] = pen_xx;
                }
            if (isX && lw_v != 0) {
                if (xv > smax_v) lxv = -pen * (smax_v - xv);
                else if (xv < smin_v) lxv = -pen * (smin_v - xv);
            }
        }
    };
    {
        const double xv = AT(X, (T - 1) * NX + vx, b);
        if (isX) sx[vx] = xv;
        LDS_ORDER();
        double lxxb[BLK * BLK], lxv;
        stage_terms(T - 1, xv, lxxb, lxv);
        UNR for (int r = 0; r < BLK; r++)
            UNR for (int c = 0; c < BLK; c++) {
                const int i = bi * BLK + r, j = bj * BLK + c;
                if (blv && i < NX && j < NX) sP[i][j] = lxxb[r * BLK + c];
            }
        if (isX) sp[vx] = lxv;
        LDS_ORDER();
    }

    // ---- prefetch ring: x, u (and lambda, I for AL) of the next steps; every load unconditional (see ilqr_kernels_coop.hip)
    constexpr int PF = 3;
    const size_t Xstep = (size_t)NX * Bp, Ustep = (size_t)NU * Bp, Lstep = (size_t)m * Bp;
    const int lr_ = (AL && l < m) ? l : 0;
    const double* Xp = X + (size_t)vx * Bp + b + (size_t)(T - 2) * Xstep;
    const double* Up = U + (size_t)vu * Bp + b + (size_t)(T - 2) * Ustep;
    const double* Lp = AL ? a.lambda + (size_t)lr_ * Bp + b + (size_t)(T - 2) * Lstep : nullptr;
    const double* Ip = AL ? a.Is + (size_t)lr_ * Bp + b + (size_t)(T - 2) * Lstep : nullptr;
    double xr[PF], ur[PF], lmr[PF], isr[PF];
    auto fetch = [&](int slot, int kk) {
        xr[slot] = *Xp;
        ur[slot] = *Up;
        lmr[slot] = isr[slot] = 0;
        if (AL) { lmr[slot] = *Lp; isr[slot] = *Ip; }
        if (kk > 0) { Xp -= Xstep; Up -= Ustep; if (AL) { Lp -= Lstep; Ip -= Lstep; } }  // uniform; no load inside
    };
    UNR for (int q = 0; q < PF; q++) { fetch(q, T - 2 - q); __builtin_amdgcn_sched_barrier(0); }

    double* Kout = KD_REC(a.KD, Bp, RS, T - 2, b);
    const ptrdiff_t Kstep = (ptrdiff_t)Bp * RS;

    for (int k0 = T - 2; k0 >= 0; k0 -= PF) {
      UNR for (int jj = 0; jj < PF; jj++) {
        const int k = k0 - jj;
        const double xv = xr[jj], uv = ur[jj], lamv = lmr[jj], isv = isr[jj];
        fetch(jj, k - PF);
        if (k < 0) continue;  // uniform: dummy step of the last group
        // ---- 1. x, u, (lambda, I) into LDS; time column of B
        if (isX) sx[vx] = xv;
        if (isU) su[vu] = uv;
        if (AL && l < m) { slam[l] = lamv; sIs[l] = isv; }
        LDS_ORDER();
        const double dts = TM ? su[NU - 1] : 0.0;
        const double dt = TM ? dts * dts : d.dt;
        const double hdt2 = dt * dt / 2;
        const double c1 = (ND == 1) ? dt : hdt2, c2 = dt;  // B = [c1 I ; c2 I] on the joint block
        if (TM) {
            double bcv = 0;
            if (l < DOF) {
                if (ND == 1) bcv = 2 * dts * su[l];
                else {
                    const double dqn = sx[DOF + l] + dt * su[l];  // velocity AFTER the step (PosOrnTimePlannerSys.cpp:176)
                    bcv = 2 * dts * dqn + 2 * dts * dts * dts * su[l];
                }
            } else if (ND == 2 && l < 2 * DOF) {
                bcv = 2 * dts * su[l - DOF];
            } else if (l == NX - 1) {
                bcv = 2 * dts;
            }
            if (isX) sbc[vx] = bcv;
        }
        double lxxb[BLK * BLK], lxv;
        stage_terms(k, xv, lxxb, lxv);
        LDS_ORDER();
        // ---- 2. B^T P  (NU x NX)
        UNR for (int t = 0; t < T_UX; t++) {
            const int i = uxi[t], j = uxj[t];
            double v;
            if (TM && i == NU - 1) {
                double s0 = 0;
                _Pragma("unroll 3") for (int q = 0; q < NX; q++) s0 += sbc[q] * sP[q][j];
                v = s0;
            } else {
                v = (ND == 1) ? dt * sP[i][j] : hdt2 * sP[i][j] + dt * sP[DOF + i][j];
            }
            if (uxv[t]) sBtP[i][j] = v;
        }
        LDS_ORDER();
        // ---- 3. Qux (transposed), Quu, Qu, Qx, Qxx
        // AL rows: g = A [x;u] - b, w = lambda + I g (AL-ILQR.cpp:110-134)
        double Qux_e[T_UX];
        UNR for (int t = 0; t < T_UX; t++) {
            const int i = uxi[t], j = uxj[t];
            Qux_e[t] = is_vrow(j) ? sBtP[i][j - DOF] * dt + sBtP[i][j] : sBtP[i][j];
        }
        double Quu_e;
        {
            if (TM && uj == NU - 1) {
                double s0 = 0;
                _Pragma("unroll 3") for (int q = 0; q < NX; q++) s0 += sBtP[ui][q] * sbc[q];
                Quu_e = s0;
            } else {
                Quu_e = (ND == 1) ? sBtP[ui][uj] * dt : sBtP[ui][uj] * hdt2 + sBtP[ui][DOF + uj] * dt;
            }
            if (ui == uj) Quu_e = d.R_diag[ui] + Quu_e;
        }
        // Qxu = A^T P B (NX x NU), formed on its own as the reference does: P is symmetric only up to rounding relative to its
        // LARGEST entries, and Qux^T in its place changes the small entries of P' by far more than an ulp
        double Qxu_e[T_UX];
        UNR for (int t = 0; t < T_UX; t++) {
            const int i = uxj[t], j = uxi[t];  // state row i, control column j
            auto atp = [&](int cc) { return is_vrow(i) ? dt * sP[i - DOF][cc] + sP[i][cc] : sP[i][cc]; };
            if (TM && j == NU - 1) {
                double s0 = 0;
                _Pragma("unroll 3") for (int q = 0; q < NX; q++) s0 += atp(q) * sbc[q];
                Qxu_e[t] = s0;
            } else {
                Qxu_e[t] = (ND == 1) ? atp(j) * dt : atp(j) * hdt2 + atp(DOF + j) * dt;
            }
        }
        double Qu_v = 0, Qx_v = 0;
        if (TM && l == NU - 1) {
            double s0 = 0;
            _Pragma("unroll 3") for (int q = 0; q < NX; q++) s0 += sbc[q] * sp[q];
            Qu_v = Ru * uv + s0;
        } else {
            const double pv = (ND == 1) ? dt * sp[vu] : hdt2 * sp[vu] + dt * sp[DOF + vu];
            Qu_v = Ru * uv + pv;
        }
        Qx_v = lxv + (is_vrow(vx) ? dt * sp[vx - DOF] + sp[vx] : sp[vx]);
        double Qxx_b[BLK * BLK];
        UNR for (int r = 0; r < BLK; r++)
            UNR for (int c = 0; c < BLK; c++) {
                const int i = (bi * BLK + r) < NX ? bi * BLK + r : 0, j = (bj * BLK + c) < NX ? bj * BLK + c : 0;
                // AtP[i][j] = P[i][j] (+ dt P[i-DOF][j] on velocity rows); Qxx = lxx + AtP[i][j] (+ dt AtP[i][j-DOF] on velocity columns)
                const double atp = is_vrow(i) ? dt * sP[i - DOF][j] + sP[i][j] : sP[i][j];
                double v = atp;
                if (is_vrow(j)) {
                    const double atp2 = is_vrow(i) ? dt * sP[i - DOF][j - DOF] + sP[i][j - DOF] : sP[i][j - DOF];
                    v = atp2 * dt + atp;
                }
                Qxx_b[r * BLK + c] = lxxb[r * BLK + c] + v;
            }
        if (AL) {
            const int ns = NX + NU;
            for (int r = 0; r < m; r++) {  // uniform
                const double* Ar = a.conA + ((size_t)(a.per_step ? k : 0) * m + r) * ns;
                const double part = (isX ? Ar[vx] * xv : 0.0) + (isU ? Ar[NX + vu] * uv : 0.0);
                const double g = wave_sum(part) - a.conb[(size_t)(a.per_step ? k : 0) * m + r];
                const double Ik = sIs[r], wv = slam[r] + Ik * g;
                UNR for (int t = 0; t < T_UX; t++) { Qux_e[t] += Ar[NX + uxi[t]] * Ik * Ar[uxj[t]]; Qxu_e[t] += Ar[uxj[t]] * Ik * Ar[NX + uxi[t]]; }
                Quu_e += Ar[NX + ui] * Ik * Ar[NX + uj];
                Qu_v += Ar[NX + vu] * wv;
                Qx_v += Ar[vx] * wv;
                UNR for (int rr = 0; rr < BLK; rr++)
                    UNR for (int c = 0; c < BLK; c++) {
                        const int i = (bi * BLK + rr) < NX ? bi * BLK + rr : 0, j = (bj * BLK + c) < NX ? bj * BLK + c : 0;
                        Qxx_b[rr * BLK + c] += Ar[i] * Ik * Ar[j];
                    }
            }
        }
        UNR for (int t = 0; t < T_UX; t++)
            if (uxv[t]) { sQuxt[uxj[t]][uxi[t]] = Qux_e[t]; sQxu[uxj[t]][uxi[t]] = Qxu_e[t]; }
        if (uuv) { sQuu[ui][uj] = Quu_e; sS[ui][uj] = Quu_e + ((ui == uj) ? reg : 0.0); }
        if (isU) sQu[vu] = Qu_v;
        if (isX) sQx[vx] = Qx_v;
        LDS_ORDER();
        // ---- 4. symmetric sweeps on Quu + reg I: afterwards sS = -(Quu + reg I)^-1 = Quu_inv of the reference
        {
            double sv = sS[ui][uj];
            UNR for (int c = 0; c < NU; c++) {
                const double aic = sS[ui][c], acj = sS[c][uj], acc = sS[c][c];
                LDS_ORDER();
                const double r = rcp_nr_g(acc);
                const double tt = aic * r;
                double val = fma(-tt, acj, sv);
                if (uj == c) val = tt;
                if (ui == c) val = acj * r;
                if (ui == c && uj == c) val = -r;
                sv = val;
                if (uuv) sS[ui][uj] = val;
                LDS_ORDER();
            }
        }
        // ---- 5. K = Quu_inv Qux, d = Quu_inv Qu  (transposed K in LDS, gain record to HBM)
        UNR for (int t = 0; t < T_UX; t++) {
            const int i = uxi[t], j = uxj[t];
            double s0 = 0, s1 = 0;
            UNR for (int q = 0; q < NU; q += 2) s0 += sS[i][q] * sQuxt[j][q];
            UNR for (int q = 1; q < NU; q += 2) s1 += sS[i][q] * sQuxt[j][q];
            const double kv = s0 + s1;
            if (uxv[t]) { sKt[j][i] = kv; Kout[i * ROWP + j] = kv; }
        }
        {
            double s0 = 0;
            UNR for (int q = 0; q < NU; q++) s0 += sS[vu][q] * sQu[q];
            if (isU) { sd[vu] = s0; Kout[vu * ROWP + NX] = s0; }
        }
        Kout -= Kstep;
        LDS_ORDER();
        // ---- 6. T1 = Quu K + Qux (transposed), tq = Quu d + Qu   (un-regularised Quu: ILQRRecursive.cpp:94-95)
        UNR for (int t = 0; t < T_UX; t++) {
            const int i = uxi[t], j = uxj[t];
            double s0 = 0, s1 = 0;
            UNR for (int q = 0; q < NU; q += 2) s0 += sQuu[i][q] * sKt[j][q];
            UNR for (int q = 1; q < NU; q += 2) s1 += sQuu[i][q] * sKt[j][q];
            if (uxv[t]) sT1t[j][i] = (s0 + s1) + Qux_e[t];
        }
        {
            double s0 = 0;
            UNR for (int q = 0; q < NU; q++) s0 += sQuu[vu][q] * sd[q];
            if (isU) stq[vu] = s0 + Qu_v;
        }
        LDS_ORDER();
        // ---- 7. P' = Qxx + K^T T1 + Qux^T K ; p' = Qx + K^T tq + Qux^T d
        {
            // rank-1 accumulation over the control index: 4 BLK operands live at a time (a fully unrolled version keeps 64 doubles
            // in flight and pushes the kernel to 256 VGPRs = one wave per SIMD)
            int ri[BLK], cj[BLK];
            UNR for (int r = 0; r < BLK; r++) { ri[r] = (bi * BLK + r) < NX ? bi * BLK + r : 0; cj[r] = (bj * BLK + r) < NX ? bj * BLK + r : 0; }
            double acc0[BLK * BLK], acc1[BLK * BLK];
            UNR for (int q = 0; q < BLK * BLK; q++) acc0[q] = acc1[q] = 0;
            double ps0 = 0, ps1 = 0;
            _Pragma("unroll 2") for (int q = 0; q < NU; q++) {
                double kr[BLK], qr[BLK], tc[BLK], kc[BLK];
                UNR for (int r = 0; r < BLK; r++) { kr[r] = sKt[ri[r]][q]; qr[r] = sQxu[ri[r]][q]; tc[r] = sT1t[cj[r]][q]; kc[r] = sKt[cj[r]][q]; }
                UNR for (int r = 0; r < BLK; r++)
                    UNR for (int c = 0; c < BLK; c++) { acc0[r * BLK + c] += kr[r] * tc[c]; acc1[r * BLK + c] += qr[r] * kc[c]; }
                ps0 += sKt[vx][q] * stq[q];
                ps1 += sQxu[vx][q] * sd[q];
            }
            const double pv = (Qx_v + ps0) + ps1;
            LDS_ORDER();
            UNR for (int r = 0; r < BLK; r++)
                UNR for (int c = 0; c < BLK; c++) {
                    const int i = bi * BLK + r, j = bj * BLK + c;
                    if (blv && i < NX && j < NX) sP[i][j] = (Qxx_b[r * BLK + c] + acc0[r * BLK + c]) + acc1[r * BLK + c];
                }
            if (isX) sp[vx] = pv;
        }
        LDS_ORDER();
      }
    }
}

bool backward_gen_supported(int kind, int nd, bool al, int m) {
    static const bool off = std::getenv("ILQR_BWD") && !std::strcmp(std::getenv("ILQR_BWD"), "v1");
    (void)nd;
    return !off && kind != 2 && (!al || m <= 16);
}

template <class S>
static void launch_gen_sys(bool al, const Bufs& a, int B, hipStream_t st) {
    const dim3 grid(grid_x8(B)), block(64);
    if (al) hipLaunchKernelGGL((k_backward_gen<S, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_backward_gen<S, false>), grid, block, 0, st, a);
}

void launch_backward_gen(int kind, int nd, bool al, const Bufs& a, int B, hipStream_t st) {
    if (kind == 3) launch_gen_sys<Sys<3, 1>>(al, a, B, st);
    else if (kind == 0 && nd == 1) launch_gen_sys<Sys<0, 1>>(al, a, B, st);
    else if (kind == 0 && nd == 2) launch_gen_sys<Sys<0, 2>>(al, a, B, st);
    else if (kind == 1 && nd == 1) launch_gen_sys<Sys<1, 1>>(al, a, B, st);
    else launch_gen_sys<Sys<1, 2>>(al, a, B, st);
}

}  // namespace ilqr
