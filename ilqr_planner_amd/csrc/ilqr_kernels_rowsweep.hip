// ilqr_kernels_rowsweep.hip -- backward Riccati sweep of the general systems (PosOrn order 2, PosOrnTime order 1 / 2, JointSpaceTime) with the
// matrices in REGISTERS, one ROW per lane, rows broadcast as DPP operands (gfx950, fp64).  Round 3; the mapping of ilqr_kernels_dpp.hip carried
// over to n_x <= 15, n_u <= 8.
//
// The step (ILQRRecursive.cpp:68-97, AL terms AL-ILQR.cpp:110-134), with the affine parts carried as one extra column (index NX):
//     Qux~ = [B'PA | Qu]   Quu = R + B'PB   Qxx~ = [l_xx + A'PA | l_x + A'p]   Qxu = A'PB
//     Quu_inv = -(Quu + reg I)^-1    K~ = [K | d] = Quu_inv Qux~    T1~ = Quu K~ + Qux~    P~' = [P' | p'] = Qxx~ + K~' T1~ + Qxu K~
// -- the reference's own expression, term for term: the un-regularised Quu in T1~ formed explicitly (first-order insensitivity of P' to the error of
// K~, see step 6), Qux and Qxu both formed from P as it is (on ill-conditioned time-system steps replacing one by the transpose of the other moves the
// result by 1e-7 .. 1e-5).
//
// Mapping.  16 lanes = one DPP row own an instance (4 instances per wave; the MFMA sweep this replaces gave an instance a whole wave and ran
// two rounds of 2048 waves at B = 4096).  Lane l < n_x holds ROW l of P~ (n_x + 1 doubles).  The control rows (B'P, Qux~, Quu, K~, T1~) live in
// the lanes whose P rows they are made of: joint control i in lane CB + i (CB = 7 for the 2nd-order systems, where (B'P)_i = dt^2/2 P_i + dt P_{7+i}
// needs row i shifted up by 7 lanes -- the same shifted copy A'P needs --, CB = 0 for order 1), the time control in lane 15.
//   * A, B are never formed: A'P, (.)A, (.)B on the joint block are a 7-lane row shift (two 32-bit DPP moves per entry) and in-lane column operations.
//   * the time column b of B: every lane forms its b_l; b is then broadcast into every lane (n_x moves), P b = (b'P)' by the symmetry of P is an
//     in-lane dot product, its entries are broadcast back into a row for lane 15.
//   * Quu + reg I is swept in registers exactly as in ilqr_kernels_dpp.hip (pivot row = DPP operand of the FMA, deferred row scaling).
//   * the four dense products (K~, Quu K~, K~'T1~, Qxu K~) are "row += own[k] * broadcast(row of control k)": n_u x (n_x + 1) broadcast FMAs each.
//     K~' T1~ needs COLUMN l of K~ in lane l: the only transposition of the step, through a 1-KiB LDS image per instance.
//   * gains leave through a double-buffered LDS image of the wave's records as whole lines, a step behind (see ilqr_kernels_dpp.hip).
// Hazards of the inline-assembly DPP blocks as explained there: every statement starts with the two wait states a DPP read needs after a VALU
// write of any of its operands.
#include "ilqr_kernels.hpp"
#include "ilqr_step.hpp"

namespace ilqr {

namespace {

#define RW_ALL_ " row_mask:0xf bank_mask:0xf"

__device__ __forceinline__ double rw_rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(fma(e, e, e), r, r);
}
template <int L>
__device__ __forceinline__ double rw_bcast(double v) {  // lane L of the DPP row to all sixteen (compiler-visible: hazards handled by the compiler)
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + L, 0xf, 0xf, false);
}
__device__ __forceinline__ double rw_shr7(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x117, 0xf, 0xf, false); }  // lane l <- lane l - 7 (0 below)
__device__ __forceinline__ double rw_shl7(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x107, 0xf, 0xf, false); }  // lane l <- lane l + 7 (0 above)
template <int CTRL>
__device__ __forceinline__ double rw_dpp32(double v) {
    return __builtin_amdgcn_update_dpp(0.0, v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ double rw_row_sum(double v) {  // sum over the 16 lanes of the DPP row, result in all sixteen
    v += rw_dpp32<0xB1>(v);   // quad_perm [1,0,3,2]
    v += rw_dpp32<0x4E>(v);   // quad_perm [2,3,0,1]
    v += rw_dpp32<0x141>(v);  // row_half_mirror
    v += rw_dpp32<0x140>(v);  // row_mirror
    return v;
}

__device__ __forceinline__ double rw_take(double v) {
    double r;
    asm volatile("v_mov_b64_e32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// acc[c] += bcast_L(src[c]) * mul for N consecutive columns starting at acc / src (N = 8, 7 or 1): one statement, two wait states first
template <int L, int N>
__device__ __forceinline__ void rw_fmac(double* acc, const double* src, double mul) {
#define F_(I) "v_fmac_f64_dpp %[a" #I "], %[s" #I "], %[m] row_newbcast:%[L]" RW_ALL_ "\n\t"
    if (N == 8)
        asm volatile("s_nop 1\n\t" F_(0) F_(1) F_(2) F_(3) F_(4) F_(5) F_(6) F_(7) ""
                     : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [a4] "+v"(acc[4]), [a5] "+v"(acc[5]), [a6] "+v"(acc[6]), [a7] "+v"(acc[7])
                     : [s0] "v"(src[0]), [s1] "v"(src[1]), [s2] "v"(src[2]), [s3] "v"(src[3]), [s4] "v"(src[4]), [s5] "v"(src[5]), [s6] "v"(src[6]), [s7] "v"(src[7]), [m] "v"(mul),
                       [L] "n"(L));
    else if (N == 7)
        asm volatile("s_nop 1\n\t" F_(0) F_(1) F_(2) F_(3) F_(4) F_(5) F_(6) ""
                     : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [a4] "+v"(acc[4]), [a5] "+v"(acc[5]), [a6] "+v"(acc[6])
                     : [s0] "v"(src[0]), [s1] "v"(src[1]), [s2] "v"(src[2]), [s3] "v"(src[3]), [s4] "v"(src[4]), [s5] "v"(src[5]), [s6] "v"(src[6]), [m] "v"(mul), [L] "n"(L));
    else
        asm volatile("s_nop 1\n\t" F_(0) "" : [a0] "+v"(acc[0]) : [s0] "v"(src[0]), [m] "v"(mul), [L] "n"(L));
#undef F_
}

// One pivot of the symmetric sweep with deferred row scaling on an NP-column row (NP = 7 or 8), pivot row in lane L (see ilqr_kernels_dpp.hip):
//   acc = bcast_L(s[C]); rc = 1/acc; t = (s[C] - s[C] dc) rc; s[j] -= t bcast_L(s[j]) (j != C); s[C] = t - dc; myrc += rc dc   (dc = 1 in the pivot lane, 0 elsewhere)
template <int NP, int C, int L>
__device__ __forceinline__ void rw_pivot(double (&s)[8], double dc, double& myrc) {
    double acc, rc, e, t;
    constexpr int J0 = (C == 0) ? 1 : 0, J1 = J0 + 1 + (C == J0 + 1), J2 = J1 + 1 + (C == J1 + 1), J3 = J2 + 1 + (C == J2 + 1), J4 = J3 + 1 + (C == J3 + 1),
                  J5 = J4 + 1 + (C == J4 + 1), J6 = J5 + 1 + (C == J5 + 1);  // the other columns in ascending order
#define HEAD_                                                                                                  \
    "s_nop 1\n\tv_mov_b64_dpp %[acc], %[sc] row_newbcast:%[L]" RW_ALL_ "\n\t"                                    \
    "v_rcp_f64 %[rc], %[acc]\n\tv_fma_f64 %[t], %[sc], -%[n], %[sc]\n\tv_fma_f64 %[e], -%[acc], %[rc], 1.0\n\t" \
    "v_fma_f64 %[e], %[e], %[e], %[e]\n\tv_fma_f64 %[rc], %[e], %[rc], %[rc]\n\tv_mul_f64 %[t], %[t], %[rc]\n\t"
#define F_(N) "v_fmac_f64_dpp %[s" #N "], %[s" #N "], -%[t] row_newbcast:%[L]" RW_ALL_ "\n\t"
#define TAIL_ "v_add_f64 %[sc], %[t], -%[n]\n\tv_fma_f64 %[myrc], %[rc], %[n], %[myrc]"
    if (NP == 8)
        asm volatile(HEAD_ F_(0) F_(1) F_(2) F_(3) F_(4) F_(5) F_(6) TAIL_
                     : [acc] "=&v"(acc), [rc] "=&v"(rc), [e] "=&v"(e), [t] "=&v"(t), [sc] "+v"(s[C]), [s0] "+v"(s[J0]), [s1] "+v"(s[J1]), [s2] "+v"(s[J2]), [s3] "+v"(s[J3]),
                       [s4] "+v"(s[J4]), [s5] "+v"(s[J5]), [s6] "+v"(s[J6]), [myrc] "+v"(myrc)
                     : [n] "v"(dc), [L] "n"(L));
    else
        asm volatile(HEAD_ F_(0) F_(1) F_(2) F_(3) F_(4) F_(5) TAIL_
                     : [acc] "=&v"(acc), [rc] "=&v"(rc), [e] "=&v"(e), [t] "=&v"(t), [sc] "+v"(s[C]), [s0] "+v"(s[J0]), [s1] "+v"(s[J1]), [s2] "+v"(s[J2]), [s3] "+v"(s[J3]),
                       [s4] "+v"(s[J4]), [s5] "+v"(s[J5]), [myrc] "+v"(myrc)
                     : [n] "v"(dc), [L] "n"(L));
#undef HEAD_
#undef F_
#undef TAIL_
}

}  // namespace

template <class S, bool AL>
__global__ __launch_bounds__(64) void k_backward_rows(Bufs a) {
    constexpr int NX = S::NX, NU = S::NU, ND = S::ND, TM = S::TM;
    constexpr int NC = NX + 1;                  // columns of a row: the state columns and the affine one
    constexpr int CB = (ND == 2) ? DOF : 0;     // lane of joint control 0
    constexpr int H1 = NC - 8;                  // columns of the second half of a row (1, 7 or 8)
    constexpr int ROWP = kd_rowp(NX), RS = NU * ROWP, IPW = 4;
    constexpr int MMAX = 16;
    static_assert(NX <= 15 && NX >= 8 && NU <= 8 && (H1 == 1 || H1 == 7 || H1 == 8), "row halves of 8 + {1, 7, 8} columns");
#define CLANE(k) (((k) < DOF) ? CB + (k) : 15)  // DPP lane that holds control row k
    // LDS rows are ROWP + 2 doubles apart: the control lanes of a wave write 16-byte pieces of their rows at the same column, and at a stride of
    // 128 bytes all of them fall on the same four banks (measured: bank conflicts were 69 % of the LDS cycles of the first version)
    constexpr int RSTR = ROWP + 2, KSTR = 18;
    __shared__ __attribute__((aligned(16))) double sKT[IPW][NU][KSTR];          // K~ of the step, for its transposition
    __shared__ __attribute__((aligned(16))) double sK[2][IPW * NU * RSTR];      // two images of the wave's gain records (rows padded)
    const DevDesc& d = *a.desc;
    const int lane = threadIdx.x, g = lane >> 4, l = lane & 15;
    const int b = xcd_tile() * IPW + g;
    const int Bp = d.Bp, T = d.T;
    const bool ok = (b < d.B) && a.active[b < d.B ? b : 0];
    if (__ballot(ok ? 1 : 0) == 0ull) return;  // wave-uniform
    const int bb = (b < d.B) ? b : 0;
    const bool isX = l < NX;
    const bool isJ = l >= CB && l < CB + DOF;           // joint-control lane
    const bool isT = TM && l == 15;                     // time-control lane
    const bool isC = isJ || isT;
    const int ci = isT ? NU - 1 : (isJ ? l - CB : 0);   // control row of this lane
    const bool isV = ND == 2 && l >= DOF && l < 2 * DOF;  // velocity row of the state
    const int vx = isX ? l : 0;
    const int cur = a.cur[bb];
    const double* X = a.X[cur];
    const double* U = a.U[cur];
    const double reg = d.reg, pen = d.penalty, pen_xx = d.pen_xx;
    const int lim_on = d.limits_set;
    const int m = AL ? a.m : 0;
    const double xm = isX ? 1.0 : 0.0, jm = isJ ? 1.0 : 0.0, tmk = isT ? 1.0 : 0.0, vm = isV ? 1.0 : 0.0;
    const double Rc = isC ? d.R_diag[ci] : 0.0;
    const bool wl = isX && d.lw[vx] != 0;
    const double mxl = wl ? d.smax[vx] : __builtin_inf(), mnl = wl ? d.smin[vx] : -__builtin_inf();
    double dci[8];  // dci[k] = 1 in the lane of control row k (also the sweep's lane constant)
    UNR for (int k = 0; k < 8; k++) dci[k] = (isC && ci == k && k < NU) ? 1.0 : 0.0;

    // gain records of the wave's instances (adjacent in memory) leave as 16-byte pieces of an LDS image
    constexpr int PCS = RS / 2, RPC = ROWP / 2, NPQ = (IPW * PCS + 63) / 64;  // pieces per record / per row; pieces of the wave per lane
    static_assert(RS % 2 == 0 && ROWP % 2 == 0, "16-byte pieces");
    const unsigned long long okm = __ballot(ok ? 1 : 0);
    const bool kfull = okm == ~0ull;
    bool pst[NPQ];
    int pof[NPQ];  // where piece lane + 64 q of the linear image lies in the padded one (doubles)
    UNR for (int q = 0; q < NPQ; q++) {
        const int c = lane + 64 * q, cc = (c < IPW * PCS) ? c : IPW * PCS - 1, gi = cc / PCS, rw = (cc % PCS) / RPC, jp = cc % RPC;
        pst[q] = c < IPW * PCS && ((okm >> (gi * 16)) & 1ull);
        pof[q] = (gi * NU + rw) * RSTR + 2 * jp;
    }
    double* Kout = KD_REC(a.KD, Bp, RS, T - 2, xcd_tile() * IPW);
    const ptrdiff_t Kstep = (ptrdiff_t)Bp * RS;

    int kpi = d.n_kp - 1;
    int kp_next = (kpi >= 0) ? d.kp_t[kpi] : -1;
    const size_t kpd_stride = (size_t)(NX + NX * NX) * Bp;
    // [l_xx row | l_x] of step k for this lane's coordinate (xk): keypoint steps come from k_kp_derivs (limits included)
    auto stage_terms = [&](int k, double xk, double* lq) {
        UNR for (int c = 0; c < NC; c++) lq[c] = 0;
        if (k == kp_next) {  // uniform
            const double* src = a.kpd + (size_t)kpi * kpd_stride;
            UNR for (int c = 0; c < NX; c++) lq[c] = xm * AT(src, NX + vx * NX + c, bb);
            lq[NX] = xm * AT(src, vx, bb);
            UNR for (int c = 0; c < NC; c++) asm volatile("" : "+v"(lq[c]));  // consume the loads inside the branch (see ilqr_kernels_dpp.hip)
            kpi--;
            kp_next = (kpi >= 0) ? __builtin_amdgcn_readfirstlane(d.kp_t[kpi]) : -1;
        } else if (lim_on) {  // uniform.  inspectJointLimit (System.cpp:121-142): l_x_i = -L q, l_xx_ii = L^2 on a violated side
            const double over = fmax(xk - mxl, 0.0), under = fmax(mnl - xk, 0.0);
            lq[NX] = pen * over - pen * under;
            if (__ballot(over + under > 0.0 ? 1 : 0) != 0ull) {  // uniform, rare: the diagonal entry is register l of lane l
                const double lv = (over + under > 0.0) ? pen_xx : 0.0;
                UNR for (int c = 0; c < NX; c++) lq[c] = (c == l) ? lv : 0.0;
            }
        }
    };

    // terminal values: P~ = [l_xx | l_x](x_{T-1})
    double Pt[16];
    UNR for (int c = 0; c < 16; c++) Pt[c] = 0;
    {
        const double xv = AT(X, (T - 1) * NX + vx, bb);
        double lq[NC];
        stage_terms(T - 1, xv, lq);
        UNR for (int c = 0; c < NC; c++) Pt[c] = lq[c];
    }

    // prefetch ring: x (state lanes), u (control lanes), lambda / I (AL: lane r < m carries row r); every load unconditional
    constexpr int PF = 3;
    const size_t Xstep = (size_t)NX * Bp, Ustep = (size_t)NU * Bp, Lstep = (size_t)m * Bp;
    const int lr_ = (AL && l < m) ? l : 0;
    const double* Xp = X + (size_t)vx * Bp + bb + (size_t)(T - 2) * Xstep;
    const double* Up = U + (size_t)ci * Bp + bb + (size_t)(T - 2) * Ustep;
    const double* Lp = AL ? a.lambda + (size_t)lr_ * Bp + bb + (size_t)(T - 2) * Lstep : nullptr;
    const double* Ip = AL ? a.Is + (size_t)lr_ * Bp + bb + (size_t)(T - 2) * Lstep : nullptr;
    double xr[PF], ur[PF], lmr[PF], isr[PF];
    auto fetch = [&](int slot, int kk) {
        xr[slot] = *Xp;
        ur[slot] = *Up;
        lmr[slot] = isr[slot] = 0;
        if (AL) { lmr[slot] = *Lp; isr[slot] = *Ip; }
        if (kk > 0) { Xp -= Xstep; Up -= Ustep; if (AL) { Lp -= Lstep; Ip -= Lstep; } }
    };
    UNR for (int q = 0; q < PF; q++) { fetch(q, T - 2 - q); __builtin_amdgcn_sched_barrier(0); }

    for (int k0 = T - 2; k0 >= 0; k0 -= PF) {
      UNR for (int jj = 0; jj < PF; jj++) {
        const int k = k0 - jj;
        // (ring values leave their slots by an opaque move before the slot's next load is issued: see ring_take in ilqr_kernels_dpp.hip)
        const double xv = xm * rw_take(xr[jj]), uv = (isC ? 1.0 : 0.0) * rw_take(ur[jj]), lamv = AL ? rw_take(lmr[jj]) : 0.0, isv = AL ? rw_take(isr[jj]) : 0.0;
        __builtin_amdgcn_sched_barrier(0);
        fetch(jj, k - PF);
        // the gain image of the step before this one (in time) leaves now: LDS -> registers here, registers -> memory after the pivots
        const bool kprev = k < T - 2;
        const int ib = (T - 2 - k) & 1;  // image written by this step; the previous step wrote the other one
        double kqa[NPQ], kqb[NPQ];
        UNR for (int q = 0; q < NPQ; q++) {
            const double2 t2 = *reinterpret_cast<const double2*>(&sK[ib ^ 1][pof[q]]);
            kqa[q] = t2.x; kqb[q] = t2.y;
        }
#define SEND_()                                                                                                         \
        {                                                                                                               \
            double2* dst = reinterpret_cast<double2*>(Kout + Kstep) + lane;                                             \
            if (kfull) {                                                                                                \
                UNR for (int q = 0; q + 1 < NPQ; q++) dst[64 * q] = make_double2(kqa[q], kqb[q]);                       \
                if (lane + 64 * (NPQ - 1) < IPW * PCS) dst[64 * (NPQ - 1)] = make_double2(kqa[NPQ - 1], kqb[NPQ - 1]);  \
            } else {                                                                                                    \
                UNR for (int q = 0; q < NPQ; q++) if (pst[q]) dst[64 * q] = make_double2(kqa[q], kqb[q]);               \
            }                                                                                                           \
        }
        if (k < 0) {  // uniform: dummy step of the last group; the one right behind the last real step sends its image out
            if (k == -1) SEND_()
            continue;
        }
        // ---- 1. step sizes, the time column b of B (PosOrnTimePlannerSys.cpp:161-162,176: the velocity AFTER the step)
        const double dts = TM ? rw_bcast<15>(uv) : 0.0;
        const double dt = TM ? dts * dts : d.dt;
        const double c1 = (ND == 1) ? dt : dt * dt / 2, c2 = (ND == 1) ? 0.0 : dt;  // B = [c1 I ; c2 I] on the joint block
        double bq = 0.0;
        if (TM) {
            if (ND == 1) {
                bq = 2 * dts * uv;                                   // q rows: lane i is also control lane i
            } else {
                const double uq = rw_shl7(uv), dq = rw_shl7(xv);     // q rows: u_i and dq_i live 7 lanes up
                const double dqn = dq + dt * uq;
                const double bqq = 2 * dts * dqn + 2 * dts * dts * dts * uq;
                bq = (l < DOF) ? bqq : 2 * dts * uv;                 // v rows: lane 7 + i is control lane i
            }
            bq = (l == NX - 1) ? 2 * dts : ((l < NX - 1) ? bq : 0.0);
        }
        double lq[NC];
        stage_terms(k, xv, lq);
        // ---- 2. A'P~ (row l), B'P~ (control lanes) from the row, the row 7 lanes below and b
        double AtP[16], BtP[16];
        UNR for (int c = 0; c < 16; c++) AtP[c] = BtP[c] = 0;
        {
            const double vdt = vm * dt, jc1 = jm * c1, jc2 = jm * c2, jdt = jm * dt;
            UNR for (int c = 0; c < NC; c++) {
                if (ND == 2) {
                    const double sh = rw_shr7(Pt[c]);
                    AtP[c] = fma(vdt, sh, Pt[c]);
                    BtP[c] = fma(jc1, sh, jc2 * Pt[c]);
                } else {
                    AtP[c] = Pt[c];
                    BtP[c] = jdt * Pt[c];
                }
            }
        }
        double brep[NX > 0 ? NX : 1];
        if (TM) {
#define BR_(R) if (R < NX) brep[R] = rw_bcast<R>(bq);
            BR_(0) BR_(1) BR_(2) BR_(3) BR_(4) BR_(5) BR_(6) BR_(7) BR_(8) BR_(9) BR_(10) BR_(11) BR_(12) BR_(13) BR_(14)
#undef BR_
            double Pb = 0;
            UNR for (int r = 0; r < NX; r++) Pb = fma(Pt[r], brep[r], Pb);      // (P b)_l = (b'P)_l by the symmetry of P
            const double bp = rw_row_sum(bq * Pt[NX]);                          // b . p
#define TB_(C) if (C < NX) BtP[C] = fma(tmk, rw_bcast<C>(Pb), BtP[C]);   // the time control's row, in lane 15
            TB_(0) TB_(1) TB_(2) TB_(3) TB_(4) TB_(5) TB_(6) TB_(7) TB_(8) TB_(9) TB_(10) TB_(11) TB_(12) TB_(13) TB_(14)
#undef TB_
            BtP[NX] = fma(tmk, bp, BtP[NX]);
        }
        // ---- 3. Qux~ (control lanes), Quu row, Qxu row, Qxx~ row
        double Qux[16], quu[8], qxu[8], Qxx[16];
        UNR for (int c = 0; c < 16; c++) Qux[c] = Qxx[c] = 0;
        UNR for (int c = 0; c < NX; c++) {
            const bool vc = ND == 2 && c >= DOF && c < 2 * DOF;
            Qux[c] = vc ? fma(dt, BtP[c - DOF], BtP[c]) : BtP[c];
            Qxx[c] = lq[c] + (vc ? fma(dt, AtP[c - DOF], AtP[c]) : AtP[c]);
        }
        Qux[NX] = fma(Rc, uv, BtP[NX]);      // Qu_i = R_i u_i + (B'p)_i
        Qxx[NX] = lq[NX] + AtP[NX];          // Qx_l = l_x_l + (A'p)_l
        UNR for (int kk = 0; kk < 8; kk++) quu[kk] = qxu[kk] = 0;
        UNR for (int kk = 0; kk < DOF; kk++) {
            quu[kk] = (ND == 1) ? c1 * BtP[kk] : fma(c1, BtP[kk], c2 * BtP[DOF + kk]);
            qxu[kk] = (ND == 1) ? c1 * AtP[kk] : fma(c1, AtP[kk], c2 * AtP[DOF + kk]);
        }
        if (TM) {
            double s0 = 0, s1 = 0;
            UNR for (int c = 0; c < NX; c++) { s0 = fma(BtP[c], brep[c], s0); s1 = fma(AtP[c], brep[c], s1); }
            quu[NU - 1] = s0;
            qxu[NU - 1] = s1;
        }
        if (AL) {
            const int ns = NX + NU;
            for (int rr = 0; rr < m; rr++) {  // uniform; g = A [x; u] - b, w = lambda + I g (AL-ILQR.cpp:110-134)
                const double* Ar = a.conA + ((size_t)(a.per_step ? k : 0) * m + rr) * ns;
                const double ax = isX ? Ar[vx] : 0.0, au = isC ? Ar[NX + ci] : 0.0;
                const double gsum = rw_row_sum(ax * xv + au * uv) - a.conb[(size_t)(a.per_step ? k : 0) * m + rr];
                const double Ik = __shfl(isv, (lane & ~15) + rr), lam = __shfl(lamv, (lane & ~15) + rr);  // row rr travels in lane rr of the group
                const double wv = lam + Ik * gsum;
                const double auI = au * Ik, axI = ax * Ik;
                UNR for (int c = 0; c < NX; c++) { Qux[c] += auI * Ar[c]; Qxx[c] += axI * Ar[c]; }
                Qux[NX] += au * wv;
                Qxx[NX] += ax * wv;
                UNR for (int kk = 0; kk < NU; kk++) { quu[kk] += auI * Ar[NX + kk]; qxu[kk] += axI * Ar[NX + kk]; }
            }
        }
        UNR for (int kk = 0; kk < NU; kk++) quu[kk] = fma(dci[kk], Rc, quu[kk]);  // + R on the diagonal (register ci of control lane ci)
        // ---- 4. Quu_inv = -(Quu + reg I)^-1: symmetric sweep with deferred row scaling; afterwards myrc * s = this lane's row of Quu_inv
        double s[8], myrc = 0.0;
        UNR for (int kk = 0; kk < 8; kk++) s[kk] = (kk < NU) ? fma(dci[kk], reg, quu[kk]) : 0.0;
        rw_pivot<NU, 0, CLANE(0)>(s, dci[0], myrc);
        rw_pivot<NU, 1, CLANE(1)>(s, dci[1], myrc);
        rw_pivot<NU, 2, CLANE(2)>(s, dci[2], myrc);
        rw_pivot<NU, 3, CLANE(3)>(s, dci[3], myrc);
        rw_pivot<NU, 4, CLANE(4)>(s, dci[4], myrc);
        rw_pivot<NU, 5, CLANE(5)>(s, dci[5], myrc);
        rw_pivot<NU, 6, CLANE(6)>(s, dci[6], myrc);
        if (NU == 8) rw_pivot<NU, 7, CLANE(7)>(s, dci[7], myrc);
        if (kprev) SEND_()
        // ---- 5. K~ = Quu_inv Qux~ (control lanes): sum_k s[k] (row of control k), scaled by myrc
        double Kt[16];
        UNR for (int c = 0; c < 16; c++) Kt[c] = 0;
#define BLK_(ACC, SRC, MUL, KK)                                                                \
        if (KK < NU) {                                                                          \
            rw_fmac<CLANE(KK), 8>(&ACC[0], &SRC[0], MUL[KK]);                                   \
            rw_fmac<CLANE(KK), H1>(&ACC[8], &SRC[8], MUL[KK]);                                  \
        }
#define BLK_ALL_(ACC, SRC, MUL) BLK_(ACC, SRC, MUL, 0) BLK_(ACC, SRC, MUL, 1) BLK_(ACC, SRC, MUL, 2) BLK_(ACC, SRC, MUL, 3) BLK_(ACC, SRC, MUL, 4) BLK_(ACC, SRC, MUL, 5) BLK_(ACC, SRC, MUL, 6) BLK_(ACC, SRC, MUL, 7)
        double sm[8];
        UNR for (int kk = 0; kk < 8; kk++) sm[kk] = myrc * s[kk];  // this lane's row of Quu_inv
        BLK_ALL_(Kt, Qux, sm)
        // ---- gains out (the lane's row {K_i0 .. K_i,nx-1, d_i} is ROWP contiguous doubles of the record) + the image for the transposition
        if (isC) {
            double* w = &sK[ib][(g * NU + ci) * RSTR];
            UNR for (int c = 0; c + 1 < ROWP; c += 2) *reinterpret_cast<double2*>(w + c) = make_double2(Kt[c], (c + 1 < NC) ? Kt[c + 1] : 0.0);
            double* wt = &sKT[g][ci][0];
            UNR for (int c = 0; c < 16; c += 2) *reinterpret_cast<double2*>(wt + c) = make_double2(Kt[c], Kt[c + 1]);
        }
        Kout -= Kstep;
        // ---- 6. T1~ = Quu K~ + Qux~ (control lanes; the UN-regularised Quu), grown in the registers of Qux~ (not needed again).  The product is
        //         formed explicitly although T1~ = -reg K~ in exact arithmetic: with it P~' = Qxx~ + K~'T1~ + Qxu K~ is insensitive to first order to
        //         the error of K~ (the "Joseph form" property of the reference's expression); the shortcut was measured -- 478 instead of 597 us at
        //         B = 4096 -- and moves ill-conditioned time-system steps by 1e-9 .. 2e-8 where the oracle's neutral variants move them by 1e-11 .. 1e-10.
        BLK_ALL_(Qux, Kt, quu)
        // ---- 7. P~' = Qxx~ + K~' T1~ + Qxu K~: column l of K~ out of the LDS image (state lanes; zeros elsewhere)
        double kt[8];
        UNR for (int kk = 0; kk < 8; kk++) kt[kk] = (kk < NU) ? xm * sKT[g][kk][l] : 0.0;
        BLK_ALL_(Qxx, Qux, kt)
        BLK_ALL_(Qxx, Kt, qxu)
        UNR for (int c = 0; c < NC; c++) Pt[c] = Qxx[c];
#undef BLK_
#undef BLK_ALL_
      }
    }
    if ((T - 2) % PF == PF - 1) {  // the last real step closed its group: no dummy step sent its image out
        const int ib = (T - 2) & 1;
        UNR for (int q = 0; q < NPQ; q++) if (pst[q]) reinterpret_cast<double2*>(Kout + Kstep)[lane + 64 * q] = *reinterpret_cast<const double2*>(&sK[ib][pof[q]]);
    }
#undef SEND_
#undef CLANE
}

bool backward_rows_supported(int kind, int nd, bool al, int m) {
    if (kind == 2) return false;                 // JointSpace order 1: the single-integrator sweep
    if (kind == 0 && nd == 1) return false;      // PosOrn order 1 with control rows in its constraints: the matrix-core sweep (n_x = 7 < 8)
    return !al || m <= 16;
}

template <class S>
static void launch_rows_sys(bool al, const Bufs& a, int B, hipStream_t st) {
    const dim3 grid(grid_x8((B + 3) / 4)), block(64);
    if (al) hipLaunchKernelGGL((k_backward_rows<S, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_backward_rows<S, false>), grid, block, 0, st, a);
}

void launch_backward_rows(int kind, int nd, bool al, const Bufs& a, int B, hipStream_t st) {
    if (kind == 3) launch_rows_sys<Sys<3, 1>>(al, a, B, st);
    else if (kind == 0) launch_rows_sys<Sys<0, 2>>(al, a, B, st);
    else if (nd == 1) launch_rows_sys<Sys<1, 1>>(al, a, B, st);
    else launch_rows_sys<Sys<1, 2>>(al, a, B, st);
}

}  // namespace ilqr
