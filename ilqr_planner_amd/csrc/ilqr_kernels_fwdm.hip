// ilqr_kernels_fwdm.hip -- line search of the time systems on the f64 matrix cores: all step sizes of an instance as one product per step
//
// The trials of the step-halving search (ILQRRecursive.cpp:101-155) differ in alpha only, so the control law of one timestep,
//     u_k(alpha) = ubar_k + K_k (x_k(alpha) - xbar_k) + alpha d_k        for the 16 step sizes alpha = 2^-c,
// is ONE product  [K_k | d_k] (8 x 16)  x  [dx_k(alpha_0) .. dx_k(alpha_15) ; alpha_0 .. alpha_15] (16 x 16): the gain record of the
// sweep, exactly as it lies in memory, times the state deviations of the 16 rollouts with the step sizes as 16th row -- four
// v_mfma_f64_16x16x4_f64.  The kernel this replaces (k_forward_tile: lane = (instance, alpha), every lane the whole 8 x 15 product) spent 120
// FMAs and 150 LDS reads per lane and step and kept the LDS pipe of a CU busy for 2400 clocks per step.
//
// A wave owns FM_TI = 4 consecutive instances and, per timestep, works through them one after the other with all 64 lanes each
// (lane l = (h = l >> 4, c = l & 15); the wave holds the states of all four in registers):
//   * column c is the rollout with alpha = 2^-c;
//   * the f64 MFMA takes A[i = c][k = h], B[k = h][j = c] per k-step s = 0..3 and returns D[row = h + 4 r][col = c] in register r.
//     The 16 k-slots (s, h) are assigned so that a lane holds the state entries of the joints it also gets the controls of:
//         slot (0, h) = q_h      slot (1, h) = q_{h+4}  (h = 3: the time state)
//         slot (2, h) = dq_h     slot (3, h) = dq_{h+4} (h = 3: the constant alpha -- the record's d column)      [2nd order]
//     (1st order: slots 0, 1 as above, slot (3, 3) = alpha, the rest empty), and D register 0 / 1 is the control of joint h / h + 4
//     (h = 3, register 1: the time control).  Rows 8..11 of A repeat the time control's gain row, so register 2 holds the time
//     control in EVERY lane: dt = u_t^2 needs no exchange between lanes, and the dynamics of a step are local to the lane;
//   * memory: per step and instance the 1 KiB gain record comes in as ONE 16-byte load per lane (the four records of a wave are
//     4 KiB contiguous), (xbar, ubar) of the four instances as 23 x 4 = 92 doubles in two loads per lane (32-byte segments of the
//     [row][instance] arrays), both PF steps ahead in registers; they pass through the wave's LDS image (dropped a step before use; LDS
//     operations of a wave execute in order: no barrier) and are read from there in the operand layout.  The writer column's (x, u) go out
//     through LDS the same way: 32-byte segments, two store instructions per step.  (One instance per wave was measured too: four times
//     the load / store instructions per CU and 8-byte scattered stores -- slower than the kernel it was to replace.)
// The pass yields the costs of all step sizes (limit part here, task part from the exported keypoint states in k_select_x); the predicted
// winner's column writes its trajectory.  Instances whose prediction lost are re-rolled by k_apply_rows_tm (8 lanes per instance: this
// kernel with every column on the accepted step size was measured against it and is slower, 0.29 against 0.21 ms).
#include "ilqr_kernels.hpp"
#include "ilqr_step.hpp"

namespace ilqr {

typedef double d4f_t __attribute__((ext_vector_type(4)));
#define FM_TI 4  // instances per wave (measured on C4, ms per 20-iteration solve: 2 -> 34.2, 4 -> 32.8, 8 -> 41.6: register spills)

template <class S>
__global__ __launch_bounds__(64) void k_forward_mfma(Bufs a, FwdArgs f) {
    constexpr int NX = S::NX, NU = S::NU, ND = S::ND, TI = FM_TI;
    static_assert(S::TM == 1 && NU == 8 && NX <= 15, "time systems: 8 controls, the state and the alpha row fill the 16 k-slots");
    constexpr int ROWP = kd_rowp(NX), RS = NU * ROWP;
    constexpr int PF = 3;  // steps in flight
    constexpr int KS = 18, NXU = NX + NU, XUS = NXU + 1;           // row stride of the record image (bank spread), entries of (x, u), their stride
    constexpr int NPC = (NXU * TI + 63) / 64;                       // (x, u) pieces per lane
    static_assert(RS * 8 <= 64 * 16, "one 16-byte piece of a record per lane");
    __shared__ __attribute__((aligned(16))) double sK[TI][NU * KS];
    __shared__ double sXU[TI][XUS], sOut[2][TI][XUS];

    const DevDesc& d = *a.desc;
    const int l = threadIdx.x, h = l >> 4, c = l & 15;
    const int b0 = xcd_tile() * TI;
    const int Bp = d.Bp, T = d.T, B = d.B;
    if (b0 >= B) return;
    const int n_alpha = f.n_alpha;
    // per instance (wave-uniform)
    bool act[TI];
    int curI[TI];
    double alphaI[TI];
    bool writerI[TI];  // (per lane)
    bool any = false;
    UNR for (int i = 0; i < TI; i++) {
        const int b = b0 + i;
        const bool in = b < B;
        const int bb = in ? b : b0;
        act[i] = in && (a.active[bb] != 0);
        any = any || act[i];
        curI[i] = a.cur[bb];
        alphaI[i] = ldexp(1.0, -c);
        const int pr = a.pred[bb];
        writerI[i] = act[i] && (c == (pr < n_alpha ? pr : n_alpha - 1));
    }
    if (!any) return;  // wave-uniform
    const bool part = c < n_alpha;
    const bool last = (h == 3);  // the lane row that owns joint 3, the time state / control and the alpha slot

    // slot s of this lane: state index (or none), column of the gain record
    int sidx[4];
    bool sst[4];
    sidx[0] = h; sst[0] = true;
    sidx[1] = last ? NX - 1 : h + 4; sst[1] = true;
    if (ND == 2) {
        sidx[2] = DOF + h; sst[2] = true;
        sidx[3] = last ? 0 : DOF + h + 4; sst[3] = !last;
    } else {
        sidx[2] = 0; sst[2] = false;
        sidx[3] = 0; sst[3] = false;
    }
    const int rowA = (c < NU) ? c : NU - 1;
    constexpr int NS = (ND == 2) ? 4 : 3;  // k-steps: the 1st-order systems skip slot 2 (empty in every lane)
    auto slot = [](int q) { return (ND == 2 || q < 2) ? q : 3; };

    // bounds of the slots (unweighted or empty: (+inf, -inf), which costs 0 for every x -- see LimRegs in ilqr_kernels_v2.hip)
    const int lim_on = d.limits_set;
    const double pen = d.penalty;
    double smx[4], smn[4];
    UNR for (int s = 0; s < 4; s++) {
        const bool w = sst[s] && d.lw[sidx[s]] != 0;
        smx[s] = w ? d.smax[sidx[s]] : __builtin_inf();
        smn[s] = w ? d.smin[sidx[s]] : -__builtin_inf();
    }

    // ---- loader.  Record of instance i: lane l holds bytes 16 l .. 16 l + 15; (x, u): piece p of lane l is entry (l + 64 p) / TI of instance
    // (l + 64 p) % TI.  Every load is unconditional (steps beyond the end re-read the last one, instances beyond the batch the padding).
    const int prow = (2 * l) / ROWP, pcol = (2 * l) % ROWP;  // (ROWP is even: a piece never straddles two rows)
    const bool pk = 2 * l < RS;
    const double2* Kp = reinterpret_cast<const double2*>(KD_REC(a.KD, Bp, RS, 0, b0)) + (pk ? l : 0);
    double* const wK = &sK[0][pk ? prow * KS + pcol : 0];
    const size_t Kstep2 = (size_t)Bp * RS / 2;
    const double* XUp[NPC];
    double* XUo[NPC];
    size_t XUstep[NPC];
    bool pxu[NPC], pxs[NPC];
    int pdst[NPC];
    UNR for (int p = 0; p < NPC; p++) {
        const int idx = l + 64 * p;
        const bool in = idx < NXU * TI;
        const int e = in ? idx / TI : 0, ii = in ? idx % TI : 0;
        int cu_ = curI[0];
        bool ac_ = act[0];
        UNR for (int i = 1; i < TI; i++) { if (ii == i) { cu_ = curI[i]; ac_ = act[i]; } }
        const bool isx = e < NX;
        XUp[p] = (isx ? a.X[cu_] + (size_t)e * Bp : a.U[cu_] + (size_t)(e - NX) * Bp) + b0 + ii;
        XUo[p] = (isx ? a.X[1 - cu_] + (size_t)e * Bp : a.U[1 - cu_] + (size_t)(e - NX) * Bp) + b0 + ii;
        XUstep[p] = (size_t)(isx ? NX : NU) * Bp;
        pxu[p] = in && ac_;   // stores only for instances at work
        pxs[p] = isx;
        pdst[p] = ii * XUS + e;
    }
    double rK0[PF][TI], rK1[PF][TI], rXU[PF][NPC];
    auto fetch = [&](int j, int kk) {
        UNR for (int i = 0; i < TI; i++) {
            const double2 v2 = Kp[(size_t)i * (RS / 2)];
            rK0[j][i] = v2.x;
            rK1[j][i] = v2.y;
        }
        UNR for (int p = 0; p < NPC; p++) rXU[j][p] = *XUp[p];
        if (kk < T - 2) {  // uniform
            Kp += Kstep2;
            UNR for (int p = 0; p < NPC; p++) XUp[p] += XUstep[p];
        }
    };
    auto stage = [&](int j) {  // ring slot j -> LDS image
        if (pk) { UNR for (int i = 0; i < TI; i++) *reinterpret_cast<double2*>(wK + i * NU * KS) = make_double2(rK0[j][i], rK1[j][i]); }
        UNR for (int p = 0; p < NPC; p++) { if (l + 64 * p < NXU * TI) (&sXU[0][0])[pdst[p]] = rXU[j][p]; }
    };
    UNR for (int j = 0; j < PF; j++) { fetch(j, j); __builtin_amdgcn_sched_barrier(0); }

    // operand reads of instance 0 (instance i: constant offsets)
    const double* rdA[4];
    const double* rdX[4];
    UNR for (int s = 0; s < 4; s++) {
        rdA[s] = &sK[0][rowA * KS + (sst[s] ? sidx[s] : NX)];  // empty slots read the d column against a zero in B
        rdX[s] = &sXU[0][sidx[s]];
    }
    const double *rdU0 = &sXU[0][NX + h], *rdU1 = &sXU[0][NX + h + 4], *rdUt = &sXU[0][NX + NU - 1];
    double* wO[4];
    UNR for (int s = 0; s < 4; s++) wO[s] = &sOut[0][0][sst[s] ? sidx[s] : NXU];  // (slot without a state: the spare entry)
    double* const wOu0 = &sOut[0][0][NX + h];
    double* const wOu1 = &sOut[0][0][NX + h + 4];

    // states of the four instances in the slots
    double xs[TI][4];
    UNR for (int i = 0; i < TI; i++) {
        const int bb = (b0 + i < B) ? b0 + i : b0;
        UNR for (int s = 0; s < 4; s++) {
            const int e = sidx[s];
            double v = 0;
            if (e < DOF) v = AT(a.q0, e, bb);
            else if (ND == 2 && e < 2 * DOF) v = AT(a.dq0, e - DOF, bb);
            xs[i][s] = sst[s] ? v : 0.0;  // (the time state starts at 0: init_state)
        }
        if (last) xs[i][3] = alphaI[i];  // the alpha slot: "state" alpha against an xbar of 0
    }
    double lc[TI], dun[TI];
    UNR for (int i = 0; i < TI; i++) lc[i] = dun[i] = 0;
    const int n_kp = d.n_kp;
    int kpi = 0, kp_next = (n_kp > 0) ? d.kp_t[0] : -1;
    const int want_dun = f.early_stop;  // sum ||du|| is read by the early-stop test only

    auto limit_cost = [&](int i) {
        UNR for (int q = 0; q < NS; q++) {
            const int s = slot(q);
            const double v = fmax(xs[i][s] - smx[s], 0.0) + fmax(smn[s] - xs[i][s], 0.0);
            lc[i] = __builtin_fma(v * pen, v, lc[i]);
        }
    };
    auto export_kp = [&](int i, int k, double u0, double u1) {  // (x, u) of this step size for the task cost (k_select_x)
        if (part) {
            double* o = a.kpx + ((size_t)kpi * 16 + c) * (NX + NU) * Bp + b0 + i;
            UNR for (int q = 0; q < NS; q++) {
                const int s = slot(q);
                if (sst[s]) o[(size_t)sidx[s] * Bp] = xs[i][s];
            }
            if (k < T - 1) { o[(size_t)(NX + h) * Bp] = u0; o[(size_t)(NX + h + 4) * Bp] = u1; }
        }
    };
    auto put_out = [&](int par, int i, double u0, double u1) {  // the writer column of instance i into the output image
        if (writerI[i]) {
            const int o = (par * TI + i) * XUS;
            UNR for (int q = 0; q < NS; q++) wO[slot(q)][o] = xs[i][slot(q)];
            wOu0[o] = u0;
            wOu1[o] = u1;
        }
    };
    auto store_out = [&](int par, bool with_u) {  // output image -> trajectory arrays (32-byte segments)
        UNR for (int p = 0; p < NPC; p++) {
            const double v = (&sOut[par][0][0])[pdst[p]];
            if (pxu[p] && (with_u || pxs[p])) *XUo[p] = v;
            XUo[p] += XUstep[p];
        }
    };

    stage(0);
    asm volatile("" ::: "memory");
    const int nsteps = T - 1;
    for (int k0 = 0; k0 < nsteps; k0 += PF) {
        UNR for (int j = 0; j < PF; j++) {
            const int k = k0 + j;
            if (k < nsteps) {  // uniform (the last group may be short)
                const bool at_kp = (k == kp_next);  // uniform
                // the four instances side by side: operands of all, then the four accumulator chains interleaved (a dependent MFMA finds its
                // predecessor finished after the other three instances' -- issued one instance after the other each chain stalls the wave), then
                // the per-instance arithmetic
                double Av[TI][4], xb[TI][4], ub0[TI], ub1[TI], ubt[TI];
                UNR for (int i = 0; i < TI; i++) {
                    UNR for (int q = 0; q < NS; q++) { Av[i][q] = rdA[slot(q)][i * NU * KS]; xb[i][q] = rdX[slot(q)][i * XUS]; }
                    ub0[i] = rdU0[i * XUS]; ub1[i] = rdU1[i * XUS]; ubt[i] = rdUt[i * XUS];
                }
                d4f_t D[TI];
                UNR for (int i = 0; i < TI; i++) D[i] = (d4f_t){0, 0, 0, 0};
                UNR for (int q = 0; q < NS; q++) {
                    const int s = slot(q);
                    UNR for (int i = 0; i < TI; i++) {
                        const double dx = xs[i][s] - xb[i][q];
                        const double bv = sst[s] ? dx : ((s == 3 && last) ? xs[i][3] : 0.0);
                        D[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(Av[i][q], bv, D[i], 0, 0, 0);
                    }
                }
                UNR for (int i = 0; i < TI; i++) {
                    if (!act[i]) continue;  // uniform (an instance that is not at work still ran its products: on harmless data)
                    const double u0 = ub0[i] + D[i][0], u1 = ub1[i] + D[i][1], ut = ubt[i] + D[i][2];
                    if (want_dun) {  // uniform
                        double n2 = __builtin_fma(D[i][1], D[i][1], D[i][0] * D[i][0]);
                        n2 += __shfl_xor(n2, 16);
                        n2 += __shfl_xor(n2, 32);
                        dun[i] += sqrt(n2);
                    }
                    put_out(j & 1, i, u0, u1);
                    if (at_kp) export_kp(i, k, u0, u1);
                    if (lim_on) limit_cost(i);  // uniform
                    // dynamics (SimulationInterface.cpp:19-31; dt = u_t^2, PosOrnTimePlannerSys.cpp:139-150)
                    const double dts = ut, dt = dts * dts;
                    if (ND == 1) {
                        const double q0n = xs[i][0] + (dt * u0 + dt * dt / 2 * 0.0);
                        const double q1n = xs[i][1] + (dt * u1 + dt * dt / 2 * 0.0);
                        const double tn = xs[i][1] + dt;
                        xs[i][0] = q0n;
                        xs[i][1] = last ? tn : q1n;
                    } else {
                        const double v0 = xs[i][2], v1 = xs[i][3];
                        const double q0n = xs[i][0] + (dt * v0 + dt * dt / 2 * u0);
                        const double q1n = xs[i][1] + (dt * v1 + dt * dt / 2 * u1);
                        const double tn = xs[i][1] + dt;
                        xs[i][0] = q0n;
                        xs[i][2] = v0 + dt * u0;
                        xs[i][1] = last ? tn : q1n;
                        xs[i][3] = last ? v1 : v1 + dt * u1;
                    }
                }
                if (at_kp) {
                    kpi++;
                    kp_next = (kpi < n_kp) ? __builtin_amdgcn_readfirstlane(d.kp_t[kpi]) : -1;
                }
                asm volatile("" ::: "memory");
                store_out(j & 1, true);            // (x_k, u_k) of the writer columns (a delayed store -- a step later, from the other image -- was
                                                   // measured slower: more registers)
            }
            asm volatile("" ::: "memory");
            stage((j + 1) % PF);                   // step k + 1 into the image (its operands are read at the top of the next step)
            asm volatile("" ::: "memory");
            fetch(j, k + PF);
        }
    }
    {   // terminal state
        UNR for (int i = 0; i < TI; i++) { if (act[i]) put_out(0, i, 0.0, 0.0); }
        asm volatile("" ::: "memory");
        store_out(0, false);
        UNR for (int i = 0; i < TI; i++) {
            if (!act[i]) continue;
            if (kp_next == T - 1) export_kp(i, T - 1, 0.0, 0.0);
            if (lim_on) limit_cost(i);
        }
    }

    UNR for (int i = 0; i < TI; i++) {
        if (!act[i]) continue;
        double v = lc[i];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (part && h == 0) {
            AT(a.lsc, c, b0 + i) = v;
            AT(a.dunA, c, b0 + i) = dun[i];
        }
    }
}

template <class S>
static void launch_fwdm(const Bufs& a, int B, hipStream_t st, const FwdArgs& f) {
    const dim3 grid(grid_x8((B + FM_TI - 1) / FM_TI)), block(64);
    hipLaunchKernelGGL((k_forward_mfma<S>), grid, block, 0, st, a, f);
}

void launch_forward_mfma(int kind, int nd, const Bufs& a, int B, hipStream_t st, const FwdArgs& f) {
    if (kind == 3) launch_fwdm<Sys<3, 1>>(a, B, st, f);
    else if (nd == 1) launch_fwdm<Sys<1, 1>>(a, B, st, f);
    else launch_fwdm<Sys<1, 2>>(a, B, st, f);
}

}  // namespace ilqr
