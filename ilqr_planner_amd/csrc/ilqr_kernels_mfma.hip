// ilqr_kernels_mfma.hip -- wave-per-instance backward Riccati sweep with the dense products on the f64 matrix cores
//
// The Riccati step of ILQRRecursive.cpp:68-97 (AL terms AL-ILQR.cpp:110-134) for every system kind, mapped so that
// the three genuinely dense products of a step run as v_mfma_f64_16x16x4_f64 and chain through registers.  Everything is
// carried with ONE extra "affine" column NX (all systems have n_x <= 15):
//     Qux~ = [Qux | Qu]   K~ = [K | d] = Quu_inv Qux~   T1~ = Quu K~ + Qux~ = [Quu K + Qux | Quu d + Qu]
//     P~'  = [Qxx | Qx] + K~^T T1~ + [Qxu ; 0] K~       = [P' | p']   (row NX and rows/columns beyond are never read)
// Lane l = (h = l >> 4, c = l & 15).  The f64 MFMA takes A[i = c][k = h], B[k = h][j = c] and returns D[row = h + 4 r][col = c]
// in register r -- so the D registers r = 0, 1 of a product ARE the A/B operands of k-step 0, 1 of the next one:
//     K~  = S (LDS)  x Qux~ (registers, "U-map": lane owns control rows h, h+4 of column c)
//     T1~ = Quu (LDS) x K~ (D registers of the first product) + Qux~
//     P~' = K~^T T1~ (both D registers) + Qxu (registers: the U-map transposed) x K~ + [Qxx | Qx] ("P-map": rows h+4r, column c)
// No LDS traffic at all for the products (the LDS-based predecessor of this kernel, removed, spent 250 of its 410 LDS instructions per step there and was bound by
// LDS bandwidth).  What is left in LDS: P (read entrywise for the structured A^T P A, B^T P, ...), B^T P, the Quu sweep, and
// the few vectors.  The time column of B needs dot products with P and B^T P: they are split over all 64 lanes and reduced
// with lane shuffles instead of being walked by the 16 lanes that own the results.
// MI355X runs f64 MFMA at the vector rate -- the gain is the removed LDS traffic, not FLOPs.
//
// Register budget: 2 waves per SIMD (256 VGPRs; 238 used).  The step is written branch-free where a choice depends on the lane (loads hoisted
// out of the conditions with clamped indices, `fma(mask ? dt : 0, up, own)` instead of `mask ? dt * up + own : own`): the first version spent
// ~1500 of its ~4000 instructions per three steps on saveexec / cbranch / restore sequences around ~110 divergent regions per step.  That costs
// registers -- at the former budget of 168 VGPRs (3 waves per SIMD) the branch-free form spills 78 of them into the loop and is 2x slower, at
// 256 it is 30 % faster than the branchy one (B = 2048: 985 -> 684 us, B = 4096: 1881 -> 1351 us; profiles/r02_batch_scan.txt).
#include <cstdlib>
#include <cstring>

#include "ilqr_step.hpp"

namespace ilqr {

#define LDS_ORDER() asm volatile("" ::: "memory")
typedef double d4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double rcp_nr_m(double x) {  // 1/x to the last bit: v_rcp_f64 (24 good bits: measured 4.6e-8) and ONE cubic
    const double r = __builtin_amdgcn_rcp(x);               // step r (1 + e + e^2), e = 1 - x r  -- max error 1.1e-16 over 2^20 samples, one
    const double e = fma(-x, r, 1.0);                       // FMA less than two Newton steps (no IEEE division sequence)
    return fma(fma(e, e, e), r, r);
}
template <int CTRL>
__device__ __forceinline__ double dpp64m(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);  // (every lane has a source in these patterns: no "old" value, no copy)
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double oct_sum_m(double v) {  // sum over lanes 8m .. 8m+7, result in all eight
    v += dpp64m<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp64m<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp64m<0x141>(v);  // row_half_mirror
    return v;
}
__device__ __forceinline__ double row16_sum(double v) {  // sum over the 16 lanes of a DPP row, result in all sixteen
    v = oct_sum_m(v);
    v += dpp64m<0x140>(v);  // row_mirror
    return v;
}
__device__ __forceinline__ double cross_rows_sum(double v) {  // sum over lanes c, c+16, c+32, c+48
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ double wave_sum_m(double v) { return cross_rows_sum(row16_sum(v)); }


// ---- Quu + reg I inverted in REGISTERS (round 3).  The 8 x 8 matrix was swept in LDS, one entry per lane: per pivot three dependent LDS
// reads, the reciprocal, one FMA and an LDS write -- ~280 clocks of a lone wave's chain, eight times per step, more than half of the
// step.  Now every lane reads ROW (c16 & 7) of the matrix once (the four DPP rows of the wave hold four copies) and the pivots run as in
// ilqr_kernels_dpp.hip: the pivot row is the DPP operand of the FMA (v_fmac_f64_dpp ... row_newbcast:c), deferred row scaling, no LDS.
// NP pivots (= n_u: 7 or 8) over NP columns; hazards as explained there (two wait states before a DPP read of a freshly written register).
#define MPV_ALL_ " row_mask:0xf bank_mask:0xf"
#define MPV_HEAD_(C) "v_mov_b64_dpp %[acc], %[s" #C "] row_newbcast:" #C MPV_ALL_ "\n\t"
#define MPV_RCP_(C)                                                                        \
    "v_rcp_f64 %[rc], %[acc]\n\t"                                                          \
    "v_fma_f64 %[t], %[s" #C "], %[n" #C "], %[s" #C "]\n\t"                               \
    "v_fma_f64 %[e], -%[acc], %[rc], 1.0\n\t"                                              \
    "v_fma_f64 %[e], %[e], %[e], %[e]\n\t"                                                 \
    "v_fma_f64 %[rc], %[e], %[rc], %[rc]\n\t"                                              \
    "v_mul_f64 %[t], %[t], %[rc]\n\t"
#define MPV_F_(J, C) "v_fmac_f64_dpp %[s" #J "], %[s" #J "], -%[t] row_newbcast:" #C MPV_ALL_ "\n\t"
#define MPV_TAIL_(C) "v_add_f64 %[s" #C "], %[t], %[n" #C "]\n\tv_fma_f64 %[myrc], -%[rc], %[n" #C "], %[myrc]\n\t"
#define MPV7_(C, A, B, D, E, F, G) MPV_HEAD_(C) MPV_RCP_(C) MPV_F_(A, C) MPV_F_(B, C) MPV_F_(D, C) MPV_F_(E, C) MPV_F_(F, C) MPV_F_(G, C) MPV_TAIL_(C)
#define MPV8_(C, A, B, D, E, F, G, H) MPV_HEAD_(C) MPV_RCP_(C) MPV_F_(A, C) MPV_F_(B, C) MPV_F_(D, C) MPV_F_(E, C) MPV_F_(F, C) MPV_F_(G, C) MPV_F_(H, C) MPV_TAIL_(C)
template <int NP>
__device__ __forceinline__ void quu_pivots(double (&s)[8], const double (&nm1)[8], double& myrc) {
    double acc, rc, e, t;
    if (NP == 8) {
        asm volatile("s_nop 1\n\t" MPV8_(0, 1, 2, 3, 4, 5, 6, 7) MPV8_(1, 0, 2, 3, 4, 5, 6, 7) MPV8_(2, 0, 1, 3, 4, 5, 6, 7) MPV8_(3, 0, 1, 2, 4, 5, 6, 7)
                         MPV8_(4, 0, 1, 2, 3, 5, 6, 7) MPV8_(5, 0, 1, 2, 3, 4, 6, 7) MPV8_(6, 0, 1, 2, 3, 4, 5, 7) MPV8_(7, 0, 1, 2, 3, 4, 5, 6) "s_nop 0"
                     : [acc] "=&v"(acc), [rc] "=&v"(rc), [e] "=&v"(e), [t] "=&v"(t), [s0] "+v"(s[0]), [s1] "+v"(s[1]), [s2] "+v"(s[2]), [s3] "+v"(s[3]),
                       [s4] "+v"(s[4]), [s5] "+v"(s[5]), [s6] "+v"(s[6]), [s7] "+v"(s[7]), [myrc] "+v"(myrc)
                     : [n0] "v"(nm1[0]), [n1] "v"(nm1[1]), [n2] "v"(nm1[2]), [n3] "v"(nm1[3]), [n4] "v"(nm1[4]), [n5] "v"(nm1[5]), [n6] "v"(nm1[6]), [n7] "v"(nm1[7]));
    } else {
        asm volatile("s_nop 1\n\t" MPV7_(0, 1, 2, 3, 4, 5, 6) MPV7_(1, 0, 2, 3, 4, 5, 6) MPV7_(2, 0, 1, 3, 4, 5, 6) MPV7_(3, 0, 1, 2, 4, 5, 6) MPV7_(4, 0, 1, 2, 3, 5, 6)
                         MPV7_(5, 0, 1, 2, 3, 4, 6) MPV7_(6, 0, 1, 2, 3, 4, 5) "s_nop 0"
                     : [acc] "=&v"(acc), [rc] "=&v"(rc), [e] "=&v"(e), [t] "=&v"(t), [s0] "+v"(s[0]), [s1] "+v"(s[1]), [s2] "+v"(s[2]), [s3] "+v"(s[3]),
                       [s4] "+v"(s[4]), [s5] "+v"(s[5]), [s6] "+v"(s[6]), [myrc] "+v"(myrc)
                     : [n0] "v"(nm1[0]), [n1] "v"(nm1[1]), [n2] "v"(nm1[2]), [n3] "v"(nm1[3]), [n4] "v"(nm1[4]), [n5] "v"(nm1[5]), [n6] "v"(nm1[6]));
    }
}

template <class S, bool AL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_backward_mfma(Bufs a) {
    constexpr int NX = S::NX, NU = S::NU, ND = S::ND, TM = S::TM;
    constexpr int PS = 18;  // row stride of sP / sBtP (doubles): rows 16-byte aligned, starts in distinct banks
    constexpr int TS = 10;  // row stride of the NU x NU matrices
    constexpr int ROWP = kd_rowp(NX), RS = NU * ROWP;
    constexpr int MMAX = 16;
    static_assert(NX <= 15 && NU <= 8, "one affine column next to the state, two k-steps of 4");
    __shared__ __attribute__((aligned(16))) double sP[16][PS], sBtP[8][PS], sS[8][TS], sQuu[8][TS];
    __shared__ double sLx[16], sLxx[16];  // limit terms of the step (entry 15: constant 0)
    __shared__ __attribute__((aligned(16))) double sx[16], su[8], sbc[16], sp[16], slam[MMAX], sIs[MMAX];
    __shared__ double sDump[64];  // target of the stores of lanes that own nothing: an unconditional ds_write is cheaper than an exec-mask branch

    const DevDesc& d = *a.desc;
    const int l = threadIdx.x, h = l >> 4, c16 = l & 15;
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

    // zero the LDS images once: padding rows/columns feed the matrix cores (0 x garbage must not be NaN)
    for (int i = l; i < 16 * PS; i += 64) (&sP[0][0])[i] = 0;
    for (int i = l; i < 8 * PS; i += 64) (&sBtP[0][0])[i] = 0;
    for (int i = l; i < 8 * TS; i += 64) { (&sS[0][0])[i] = 0; (&sQuu[0][0])[i] = 0; }
    if (l < 16) { sx[l] = 0; sbc[l] = 0; sp[l] = 0; slam[l] = 0; sIs[l] = 0; sLx[l] = 0; sLxx[l] = 0; }
    if (l < 8) su[l] = 0;
    LDS_ORDER();

    // ---- lane maps
    const bool colS = c16 < NX;      // this lane's column is a state column
    const bool colA = c16 == NX;     // ... the affine column
    const int cj = colS ? c16 : 0;   // clamped state index of the column
    int ui[2];
    bool uv[2];                      // U-map: control rows h, h+4
    UNR for (int r = 0; r < 2; r++) { uv[r] = (h + 4 * r) < NU; ui[r] = uv[r] ? h + 4 * r : 0; }
    int pi[4];
    bool pv[4];                      // P-map: state rows h + 4r
    UNR for (int r = 0; r < 4; r++) { pv[r] = (h + 4 * r) < NX; pi[r] = pv[r] ? h + 4 * r : 0; }
    const bool qv = (l >> 3) < NU && (l & 7) < NU;  // Quu map: (row l >> 3, column l & 7)
    const int qi = qv ? l >> 3 : 0, qj = qv ? l & 7 : 0;
    const bool isX = l < NX, isU = l < NU;
    const int vx = isX ? l : 0, vu = isU ? l : 0;
    auto is_vrow = [](int i) { return ND == 2 && i >= DOF && i < 2 * DOF; };
    double* const dump = &sDump[l];
    double* wBtP[2];
    UNR for (int r = 0; r < 2; r++) wBtP[r] = (uv[r] && colS) ? &sBtP[ui[r]][c16] : dump;
    double* const wS = qv ? &sS[qi][qj] : dump;
    double* const wQuu = qv ? &sQuu[qi][qj] : dump;
    double* wP[4];
    UNR for (int r = 0; r < 4; r++) wP[r] = (pv[r] && colS) ? &sP[pi[r]][c16] : ((pv[r] && colA) ? &sp[pi[r]] : dump);
    double* const wx = isX ? &sx[vx] : dump;
    double* const wu = isU ? &su[vu] : dump;
    double* const wbc = isX ? &sbc[vx] : dump;

    // limit terms (inspectJointLimit, System.cpp:121-142).  Lane i < n_x forms l_x_i and l_xx_ii of ITS coordinate from the x_i it has in a
    // register (bounds with the weight folded in: (+inf, -inf) where there is none) and drops them into LDS; the P-map picks them up
    // through per-lane read pointers -- the affine column reads l_x, the diagonal entries l_xx, every other entry a constant zero.  (Each
    // lane forming the terms of its four P-map rows and selecting the one it needs, if any, was 56 instructions per step.)
    const bool wl = isX && d.lw[vx] != 0;
    const double mxl = wl ? d.smax[vx] : __builtin_inf(), mnl = wl ? d.smin[vx] : -__builtin_inf();
    double* const wLx = isX ? &sLx[vx] : dump;
    double* const wLxx = isX ? &sLxx[vx] : dump;
    const double* rdL[4];
    UNR for (int r = 0; r < 4; r++)
        rdL[r] = (pv[r] && colA) ? &sLx[pi[r]] : ((pv[r] && colS && pi[r] == c16) ? &sLxx[pi[r]] : &sLxx[15]);

    double nm1q[8];  // quu_pivots: -1 in the lane whose row is the pivot row of pivot c, 0 elsewhere
    UNR for (int c = 0; c < 8; c++) nm1q[c] = ((c16 & 7) == c) ? -1.0 : 0.0;
    // control weights of the rows this lane works on: out of the descriptor once (read inside the loop they are a global load and an
    // s_waitcnt vmcnt(0) per step -- which also waits for the whole prefetch ring)
    const double Ru[2] = {d.R_diag[ui[0]], d.R_diag[ui[1]]};
    const double Rq = d.R_diag[qi];

    int kpi = d.n_kp - 1;
    int kp_next = (kpi >= 0) ? d.kp_t[kpi] : -1;
    const size_t kpd_stride = (size_t)(NX + NX * NX) * Bp;
    // [l_xx | l_x] of step k in the P-map (xk: this lane's coordinate of x_k); keypoint steps come from k_kp_derivs (limits included)
    auto stage_terms = [&](int k, double xk, double* lq) {
        if (k == kp_next) {  // uniform
            const double* src = a.kpd + (size_t)kpi * kpd_stride;
            UNR for (int r = 0; r < 4; r++) {
                lq[r] = 0;
                if (pv[r] && colS) lq[r] = AT(src, NX + pi[r] * NX + c16, b);
                if (pv[r] && colA) lq[r] = AT(src, pi[r], b);
            }
            // consume the loads inside the branch: otherwise their wait lands after the join and every step drains vmcnt to 0 (the prefetch ring)
            UNR for (int r = 0; r < 4; r++) asm volatile("" : "+v"(lq[r]));
            kpi--;
            kp_next = (kpi >= 0) ? __builtin_amdgcn_readfirstlane(d.kp_t[kpi]) : -1;  // (into a scalar register HERE: kept in a vector register, the
                                                                                         // step's `k == kp_next` waits for this load -- and the ring)
        } else if (lim_on) {  // uniform
            const double over = fmax(xk - mxl, 0.0), under = fmax(mnl - xk, 0.0);   // at most one of them is non-zero
            *wLx = pen * over - pen * under;                      // l_x_i = -L q, q = limit - x (same bits as -pen (max - x))
            *wLxx = (over + under > 0.0) ? pen_xx : 0.0;
            LDS_ORDER();
            UNR for (int r = 0; r < 4; r++) lq[r] = *rdL[r];
        } else {
            UNR for (int r = 0; r < 4; r++) lq[r] = 0;
        }
    };
    {   // terminal values: P = l_xx(x_{T-1}), p = l_x(x_{T-1})
        const double xv = AT(X, (T - 1) * NX + vx, b);
        if (isX) sx[vx] = xv;
        LDS_ORDER();
        double lq[4];
        stage_terms(T - 1, xv, lq);
        UNR for (int r = 0; r < 4; r++) {
            if (pv[r] && colS) sP[pi[r]][c16] = lq[r];
            if (pv[r] && colA) sp[pi[r]] = lq[r];
        }
        LDS_ORDER();
    }

    // ---- prefetch ring: x, u (and lambda, I for AL); every load unconditional (see ilqr_kernels_dpp.hip)
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
        const double xv = xr[jj], uv_ = ur[jj], lamv = lmr[jj], isv = isr[jj];
        fetch(jj, k - PF);
        if (k < 0) continue;  // uniform: dummy step of the last group
        // ---- 1. x, u, (lambda, I) into LDS; time column of B
        *wx = xv;
        *wu = uv_;
        if (AL && l < m) { slam[l] = lamv; sIs[l] = isv; }
        LDS_ORDER();
        const double dts = TM ? su[NU - 1] : 0.0;
        const double dt = TM ? dts * dts : d.dt;
        const double hdt2 = dt * dt / 2;
        const double c1 = (ND == 1) ? dt : hdt2, c2 = dt;  // B = [c1 I ; c2 I] on the joint block
        if (TM) {  // lane l < n_x forms bc_l (PosOrnTimePlannerSys.cpp:161-162,176); all candidates are computed, the lane's one is selected
            const int lj = l < DOF ? l : ((ND == 2 && l < 2 * DOF) ? l - DOF : 0);
            const double sul = su[lj];
            double bcv;
            if (ND == 1) {
                bcv = 2 * dts * sul;
            } else {
                const double dqn = sx[DOF + lj] + dt * sul;  // velocity AFTER the step
                const double bq = 2 * dts * dqn + 2 * dts * dts * dts * sul;
                bcv = (l < DOF) ? bq : 2 * dts * sul;
            }
            bcv = (l == NX - 1) ? 2 * dts : ((l < NX - 1) ? bcv : 0.0);
            *wbc = bcv;
        }
        double lq[4];
        stage_terms(k, xv, lq);
        LDS_ORDER();
        auto atp = [&](int i, int cc) {  // (A^T P)[i][cc]; both loads unconditional (the row above a velocity row, else the row itself)
            const bool vr = is_vrow(i);
            const double own = sP[i][cc], up = sP[vr ? i - DOF : i][cc];
            return fma(vr ? dt : 0.0, up, own);  // (+ 0 x up on the other rows: exact)
        };
        // ---- 2. time column: wT[j] = sum_q bc_q P[q][j] (row NU-1 of B^T P), wA[i] = sum_q (A^T P)[i][q] bc_q (column NU-1 of
        //         Qxu), bp = bc . p; every lane adds 4 terms of the sum for column / row c16, lanes c16 + 16 h' hold the rest
        double wT = 0, wA = 0, bp = 0;
        if (TM) {
            UNR for (int t = 0; t < 4; t++) {
                const int q = h + 4 * t;                          // 0..15: the entries beyond n_x of sbc / sp / sP are the zeros of the start
                const double bq = sbc[q];
                const double pq = sp[q], Pq = sP[q][cj];
                wT += bq * (colA ? pq : Pq);                     // the affine column carries p: its "column sum" is bc . p
                wA += atp(cj, q) * bq;
            }
            wT = cross_rows_sum(wT);
            wA = cross_rows_sum(wA);
            bp = wT;  // meaningful on the lanes of the affine column, which are the ones that form Qu
        }
        // ---- 3. B^T P in the U-map
        double btp[2];
        UNR for (int r = 0; r < 2; r++) {
            const int i = ui[r];
            const bool trow = TM && i == NU - 1;   // the time control's row of B^T P is the reduced column sum
            const int ij = trow ? 0 : i;
            const double joint = (ND == 1) ? dt * sP[ij][cj] : hdt2 * sP[ij][cj] + dt * sP[DOF + ij][cj];
            btp[r] = trow ? wT : joint;
            *wBtP[r] = btp[r];
        }
        LDS_ORDER();
        // ---- 4. Qux~ = [B^T P A | Qu] (U-map), Qxu (U-map transposed: state c16, control h + 4r), Quu, [Qxx | Qx] (P-map)
        double qux[2], qxu[2];
        UNR for (int r = 0; r < 2; r++) {
            const int i = ui[r];
            const bool trow = TM && i == NU - 1;
            const int ij = trow ? 0 : i;
            const bool vr = is_vrow(cj);
            // state column: Qux[i][c] = (B^T P A)[i][c],  Qxu[c][i] = (A^T P B)[c][i]
            const double left = sBtP[i][vr ? cj - DOF : cj];
            const double vS = fma(vr ? dt : 0.0, left, btp[r]);
            const double a1 = atp(cj, ij), a2 = (ND == 2) ? atp(cj, DOF + ij) : 0.0;
            const double wJ = (ND == 1) ? a1 * dt : a1 * hdt2 + a2 * dt;
            const double wS = trow ? wA : wJ;
            // affine column: Qu_i = R_i u_i + (B^T p)_i
            const double pj = (ND == 1) ? dt * sp[ij] : hdt2 * sp[ij] + dt * sp[DOF + ij];
            const double vA = Ru[r] * su[i] + (trow ? bp : pj);
            const double v = colS ? vS : (colA ? vA : 0.0);
            qux[r] = uv[r] ? v : 0.0;
            qxu[r] = (uv[r] && colS) ? wS : 0.0;
        }
        double quu;
        {
            double tot = 0;
            if (TM) {  // column NU-1: sum_q BtP[qi][q] bc_q; the 8 lanes of a row add two terms each (q = qj, qj + 8), DPP butterfly
                const bool second = qj + 8 < NX;
                const int q2 = second ? qj + 8 : qj;
                double part = sBtP[qi][qj] * sbc[qj];
                const double part2 = sBtP[qi][q2] * sbc[q2];
                part += second ? part2 : 0.0;
                tot = oct_sum_m(qv ? part : 0.0);
            }
            const int qjj = (TM && qj == NU - 1) ? 0 : qj;
            const double joint = (ND == 1) ? sBtP[qi][qjj] * dt : sBtP[qi][qjj] * hdt2 + sBtP[qi][DOF + qjj] * dt;
            quu = (TM && qj == NU - 1) ? tot : joint;
            quu = (qi == qj) ? Rq + quu : quu;
        }
        double qxx[4];
        UNR for (int r = 0; r < 4; r++) {
            const int i = pi[r];
            const bool vc = is_vrow(cj), vi = is_vrow(i);
            const double t0 = atp(i, cj), t1 = atp(i, vc ? cj - DOF : cj);
            const double vS = fma(vc ? dt : 0.0, t1, t0);                      // state column: (A^T P A)[i][c]
            const double pown = sp[i], pup = sp[vi ? i - DOF : i];
            const double vA = fma(vi ? dt : 0.0, pup, pown);                   // affine column: Qx_i = l_x_i + (A^T p)_i
            const double v = colS ? vS : (colA ? vA : 0.0);
            qxx[r] = pv[r] ? lq[r] + v : 0.0;
        }
        if (AL) {
            const int ns = NX + NU;
            for (int rr = 0; rr < m; rr++) {  // uniform; g = A [x;u] - b, w = lambda + I g
                const double* Ar = a.conA + ((size_t)(a.per_step ? k : 0) * m + rr) * ns;
                const double part = (isX ? Ar[vx] * xv : 0.0) + (isU ? Ar[NX + vu] * uv_ : 0.0);
                const double g = wave_sum_m(part) - a.conb[(size_t)(a.per_step ? k : 0) * m + rr];
                const double Ik = sIs[rr], wv = slam[rr] + Ik * g;
                const double axc = Ar[cj];
                UNR for (int r = 0; r < 2; r++) {
                    const double au = Ar[NX + ui[r]];
                    if (uv[r] && colS) { qux[r] += au * Ik * axc; qxu[r] += axc * Ik * au; }
                    if (uv[r] && colA) qux[r] += au * wv;
                }
                quu += Ar[NX + qi] * Ik * Ar[NX + qj];
                UNR for (int r = 0; r < 4; r++) {
                    if (pv[r] && colS) qxx[r] += Ar[pi[r]] * Ik * axc;
                    if (pv[r] && colA) qxx[r] += Ar[pi[r]] * wv;
                }
            }
        }
        *wQuu = quu;
        const double s0 = quu + ((qi == qj) ? reg : 0.0);
        *wS = s0;
        LDS_ORDER();
        // ---- 5. Quu_inv = -(Quu + reg I)^-1: row (c16 & 7) of Quu + reg I out of LDS once, the pivots in registers (quu_pivots above);
        //         afterwards myrc * srow = this lane's row of Quu_inv
        double srow[8], myrc = 0.0;
        {
            const d4_t* rp = reinterpret_cast<const d4_t*>(&sS[c16 & 7][0]);  // rows are 80 bytes apart: 16-byte aligned
            const double2 r0 = reinterpret_cast<const double2*>(rp)[0], r1 = reinterpret_cast<const double2*>(rp)[1], r2 = reinterpret_cast<const double2*>(rp)[2],
                          r3 = reinterpret_cast<const double2*>(rp)[3];
            srow[0] = r0.x; srow[1] = r0.y; srow[2] = r1.x; srow[3] = r1.y; srow[4] = r2.x; srow[5] = r2.y; srow[6] = r3.x; srow[7] = r3.y;
        }
        quu_pivots<NU>(srow, nm1q, myrc);
        // ---- 6. K~ = Quu_inv Qux~ ; T1~ = Quu K~ + Qux~ ; P~' = [Qxx | Qx] + K~^T T1~ + Qxu K~   (f64 matrix cores)
        const bool rowU = c16 < NU;
        const int cu = rowU ? c16 : 0;
        const double mU = rowU ? 1.0 : 0.0;  // (a factor, not a select: a select of a load is compiled as an exec-mask region around the load)
        // A operand: Quu_inv[c16][h], Quu_inv[c16][4 + h] -- entries h, 4 + h of this lane's row (h is the lane's DPP row: a select among four)
        const double lo01 = (h & 1) ? srow[1] : srow[0], lo23 = (h & 1) ? srow[3] : srow[2], hi01 = (h & 1) ? srow[5] : srow[4], hi23 = (h & 1) ? srow[7] : srow[6];
        const double msc = (c16 < 8 ? myrc : 0.0) * mU;                       // (rows live in lanes c16 < 8; the copies in 8..15 feed zeros)
        const double sa0 = ((h & 2) ? lo23 : lo01) * msc, sa1 = ((h & 2) ? hi23 : hi01) * msc;
        const double qa0 = sQuu[cu][h] * mU, qa1 = sQuu[cu][4 + h] * mU;   // A operand: Quu[c16][4c + h]
        d4_t Kt = {0, 0, 0, 0};
        Kt = __builtin_amdgcn_mfma_f64_16x16x4f64(sa0, qux[0], Kt, 0, 0, 0);
        Kt = __builtin_amdgcn_mfma_f64_16x16x4f64(sa1, qux[1], Kt, 0, 0, 0);
        UNR for (int r = 0; r < 2; r++)
            if (uv[r] && c16 <= NX) Kout[ui[r] * ROWP + c16] = Kt[r];  // gain record row {K[i][0..NX-1], d[i]}
        Kout -= Kstep;
        // T1~ and the Qxu K~ part of P~' depend on K~ only: two independent accumulator chains, then K~^T T1~ on top
        d4_t T1 = {qux[0], qux[1], 0, 0};
        d4_t Pn = {qxx[0], qxx[1], qxx[2], qxx[3]};
        T1 = __builtin_amdgcn_mfma_f64_16x16x4f64(qa0, Kt[0], T1, 0, 0, 0);
        Pn = __builtin_amdgcn_mfma_f64_16x16x4f64(qxu[0], Kt[0], Pn, 0, 0, 0);
        T1 = __builtin_amdgcn_mfma_f64_16x16x4f64(qa1, Kt[1], T1, 0, 0, 0);
        Pn = __builtin_amdgcn_mfma_f64_16x16x4f64(qxu[1], Kt[1], Pn, 0, 0, 0);
        Pn = __builtin_amdgcn_mfma_f64_16x16x4f64(Kt[0], T1[0], Pn, 0, 0, 0);
        Pn = __builtin_amdgcn_mfma_f64_16x16x4f64(Kt[1], T1[1], Pn, 0, 0, 0);
        LDS_ORDER();
        UNR for (int r = 0; r < 4; r++) *wP[r] = Pn[r];
        LDS_ORDER();
      }
    }
}

bool backward_mfma_supported(int kind, int nd, bool al, int m) {
    (void)nd;
    return kind != 2 && (!al || m <= 16);
}

template <class S>
static void launch_mfma_sys(bool al, const Bufs& a, int B, hipStream_t st) {
    const dim3 grid(grid_x8(B)), block(64);
    if (al) hipLaunchKernelGGL((k_backward_mfma<S, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_backward_mfma<S, false>), grid, block, 0, st, a);
}

void launch_backward_mfma(int kind, int nd, bool al, const Bufs& a, int B, hipStream_t st) {
    if (kind == 3) launch_mfma_sys<Sys<3, 1>>(al, a, B, st);
    else if (kind == 0 && nd == 1) launch_mfma_sys<Sys<0, 1>>(al, a, B, st);
    else if (kind == 0 && nd == 2) launch_mfma_sys<Sys<0, 2>>(al, a, B, st);
    else if (kind == 1 && nd == 1) launch_mfma_sys<Sys<1, 1>>(al, a, B, st);
    else launch_mfma_sys<Sys<1, 2>>(al, a, B, st);
}

}  // namespace ilqr
