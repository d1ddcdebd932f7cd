// ilqr_kernels_v2.hip -- chip-filling kernels of the batched iLQR hot path (gfx950, fp64).
//
// k_forward_tile   the whole step-halving line search of ILQRRecursive.cpp:101-155 in ONE pass over the gains:
//                  16 lanes per instance, lane a rolls the horizon out with alpha = 2^-a, so the n_alpha trials the
//                  reference runs one after the other run side by side and K_k, d_k, xbar_k, ubar_k are read from
//                  HBM once instead of once per trial.  A 256-thread workgroup owns 16 instances; each timestep's
//                  block (70 doubles per instance for n_x = n_u = 7) is fetched with coalesced loads 4 steps ahead,
//                  staged through LDS and broadcast-read by the 16 lanes of each instance.  The alpha = 1 lane
//                  writes its trajectory speculatively (it wins most iterations); the winner is picked with a wave
//                  ballot exactly as the do/while would have (first alpha whose cost is below cost0 and not NaN,
//                  else the last one).  B = 4096 gives 256 workgroups = one per CU, 1024 waves.
//                  APPLY mode re-rolls the winning alpha for the instances whose winner was not alpha = 1.
// k_al_post        AL bookkeeping on the ACCEPTED trajectory, one lane per (instance, k): the active-set weights
//                  penalty * I_k the next backward sweep needs (AL-ILQR.cpp:21-44,190 -- every trial overwrites them, so
//                  only the accepted trial's values survive in the reference too) and, every lag_update_step
//                  iterations, the multiplier update (AL-ILQR.cpp:202-208).  Keeps the rollout kernels free of AL.
// k_backward_si    backward Riccati sweep specialised to single-integrator dynamics (PosOrnPlannerSys nb_deriv = 1:
//                  A = I, B = dt I, l_ux = 0), one lane per instance.  With S = Quu + reg I = D + dt^2 P,
//                  D = R + reg I and M = S^-1 the reference's update collapses, exactly, to
//                      K  = (M D - I)/dt
//                      d  = -M Qu
//                      P' = l_xx + [D - D M D - reg (D M^2 D - D M - M D + I)]/dt^2
//                      p' = l_x + p - (Qu + D d)/dt - reg (D M d - d)/dt
//                  (K'QuuK + K'Qux = -reg K'K because (Quu + reg I) K = -Qux; Qxu K = dt P K).  One SPD inverse
//                  (LDL^T, no pivoting needed) and one symmetric 7x7x7 product replace the partial-pivot LU and
//                  five dense products of the generic sweep (ilqr_kernels.hip k_backward), all in registers.
// Everything a loop needs from the shared descriptor is copied into registers before the loop: the compiler cannot
// prove that the trajectory stores do not alias the descriptor and would otherwise re-issue scalar loads every step.
#include <cstdlib>
#include <cstring>

#include "ilqr_kernels.hpp"
#include "ilqr_step.hpp"

namespace ilqr {

// Workgroup barrier that only drains LDS traffic: __syncthreads() also emits s_waitcnt vmcnt(0), which would wait for
// the global prefetches issued several timesteps ahead and serialise every step behind an HBM round trip.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// joint/velocity limits held in registers (inspectJointLimit, System.cpp:121-142)
template <int NX>
struct LimRegs {
    int on;
    double penalty, pen_xx;
    double smax[NX], smin[NX];
    __device__ __forceinline__ void load(const DevDesc& d) {
        on = d.limits_set;
        penalty = d.penalty;
        pen_xx = d.pen_xx;
        // an unweighted coordinate gets the bounds (+inf, -inf): its distance beyond them is max(x - inf, 0) + max(-inf - x, 0) = 0 for every x
        // (NaN and +-inf included: max() drops a NaN operand), so it adds q pen q = +0 -- no test of the weight inside the rollout
        UNR for (int i = 0; i < NX; i++) {
            const bool w = d.lw[i] != 0;
            smax[i] = w ? d.smax[i] : __builtin_inf();
            smin[i] = w ? d.smin[i] : -__builtin_inf();
        }
    }
    // q' L q of inspectJointLimit, branch-free: the distance beyond the bound is max(x - max, 0) + max(min - x, 0), and (-q) pen (-q) == q pen q
    // bit for bit (a NaN state adds 0 in both forms).  The nested `if (x > max) .. else if (x < min)` compiled to two exec-mask regions per
    // coordinate -- 19 instructions each, a quarter of k_forward_tile's step; a (uniform) test of the weight per coordinate made every coordinate
    // its own basic block: 7 dependent instructions through the same two temporaries, no overlap between coordinates.  Here the 15 distances
    // are independent of each other and only the final sum is a chain.
    __device__ __forceinline__ double cost(const double* x) const {
        double a = 0;
        if (on) {
            double q[NX], t[NX];
            UNR for (int i = 0; i < NX; i++) {
                q[i] = fmax(x[i] - smax[i], 0.0) + fmax(smin[i] - x[i], 0.0);
                t[i] = q[i] * penalty;
            }
            UNR for (int i = 0; i < NX; i++) a = __builtin_fma(t[i], q[i], a);  // (`a += q * penalty * q` as the compiler contracts it)
        }
        return a;
    }
};

// ------------------------------------------------------------------------------------------------ forward, alpha-parallel

// instances per workgroup of k_forward_tile: 4 (one wave, 11 KB of LDS for the 2nd-order time system) gives 1024 workgroups
// at B = 4096 -- four per CU, whose barriers and load latencies overlap -- where 16 gave one per CU (C4: 0.85 -> 0.70 ms; 8: 0.75, 2: 1.36)
#ifndef FT_TI
#define FT_TI 4
#endif
template <class S, bool APPLY>
__global__ __launch_bounds__(FT_TI * 16) void k_forward_tile(Bufs a, FwdArgs f) {
    constexpr int NX = S::NX, NU = S::NU, NK = NU * NX;
    constexpr int NC = NK + NU + NX + NU;          // doubles per instance-step: K | d | xbar | ubar
    constexpr int NO = NX + NU;                    // doubles written per instance-step: x | u
    constexpr int PF = 3;                          // prefetch distance in timesteps
    constexpr int TI = FT_TI, NT = TI * 16;        // instances per workgroup (x 16 step sizes each) and its threads
    __shared__ double s_in[2][NC][TI];
    __shared__ double s_out[2][NO][TI];
    __shared__ int s_wr[TI];
    __shared__ double s_dump[NT];  // where loader slots without a destination (padding of a gain record, lanes beyond the last row) write:
                                   // an unconditional ds_write instead of one exec-mask region per slot and step

    const DevDesc& d = *a.desc;
    const int tid = threadIdx.x, il = tid >> 4, ai = tid & 15;
    const int b0 = xcd_tile() * TI, b = b0 + il;
    const int Bp = d.Bp, T = d.T, B = d.B;

    // SPEC: lane `ai` tries alpha = 2^-ai; the lane whose index equals the instance's PREDICTED winner (the winner of
    // its previous iteration: alpha = 1 early on, the alpha floor once converged) writes its trajectory speculatively.
    // APPLY: lane 0 re-rolls the actual winner (pend = winner + 1) when the prediction was wrong.
    bool part, writer;
    double alpha;
    if (!APPLY) {
        part = (b < B) && a.active[b] && (ai < f.n_alpha);
        alpha = ldexp(1.0, -ai);
        const int pr = (b < B) ? a.pred[b] : 0;
        writer = part && (ai == (pr < f.n_alpha ? pr : f.n_alpha - 1));
    } else {
        const int w = (b < B) ? a.pend[b] : 0;
        part = (w > 0) && (ai == 0);
        alpha = ldexp(1.0, -(w - 1));
        writer = part;
    }
    if (!__syncthreads_or(part ? 1 : 0)) return;  // nothing to do for these 16 instances (uniform)
    if (writer) s_wr[il] = 1;
    else if (ai == 0 && !(part && !APPLY)) s_wr[il] = 0;  // instance without any writer lane

    // descriptor -> registers
    LimRegs<NX> lim;
    lim.load(d);
    const double dt_fixed = d.dt;
    const int n_kp = d.n_kp;
    int kp_next = (n_kp > 0) ? d.kp_t[0] : -1;

    // loader.  Gains: the records of the tile's 16 instances at one timestep are ONE contiguous run of 16 RS doubles; thread t
    // fetches flat elements t + 256 q (fully coalesced) and knows where each lands in the [component][instance] LDS image.
    // xbar | ubar: component (tid >> 4) + 16 j of instance tid & 15 (16 instances of a row = one 128-byte line).
    // Every load is unconditional (timesteps beyond the end re-read the last one): see ilqr_kernels_coop.hip on vmcnt.
    constexpr int ROWP = kd_rowp(NX), RS = NU * ROWP;
    constexpr int NLG = (TI * RS + NT - 1) / NT;         // gain loads per thread per step
    constexpr int NLX = ((NX + NU) * TI + NT - 1) / NT;  // xbar/ubar loads per thread per step
    constexpr int NLD = NLG + NLX;
    const int li = tid % TI, lb = b0 + li, lc0 = tid / TI;
    const int lcur = a.cur[lb];
    const double* Xc = a.X[lcur];
    const double* Uc = a.U[lcur];
    const double* gbase = KD_REC(a.KD, Bp, RS, 0, b0);
    const size_t gstep = (size_t)Bp * RS;
    int goff[NLG], gdst[NLG];  // flat source offset (clamped into the run) and LDS destination (-1: padding / out of range)
    UNR for (int q = 0; q < NLG; q++) {
        const int fl = tid + NT * q;
        const bool in = fl < TI * RS;
        const int flc = in ? fl : 0;
        const int inst = flc / RS, w = flc % RS, i = w / ROWP, jj = w % ROWP;
        goff[q] = flc;
        gdst[q] = !in ? -1 : (jj < NX ? (i * NX + jj) * TI + inst : (jj == NX ? (NK + i) * TI + inst : -1));

    }
    const double* xptr[NLX];
    size_t xstep[NLX];
    int xkmax[NLX], xdst[NLX];
    UNR for (int j = 0; j < NLX; j++) {
        const int c = lc0 + 16 * j;          // 0 .. NX+NU-1 valid
        const bool in = c < NX + NU;
        const int cc = in ? c : 0;
        if (cc < NX) { xptr[j] = Xc + (size_t)cc * Bp + lb; xstep[j] = (size_t)NX * Bp; xkmax[j] = T - 1; }
        else { xptr[j] = Uc + (size_t)(cc - NX) * Bp + lb; xstep[j] = (size_t)NU * Bp; xkmax[j] = T - 2; }
        xdst[j] = in ? (NK + NU + cc) * TI + li : -1;
    }
    // running pointers: load_step is called for k = 0, 1, 2, ... in order; a pointer stops at its array's last timestep (steps beyond the end
    // re-read it).  One 64-bit add per pointer and step instead of a 64-bit multiply-add chain per address.
    const double* gp[NLG];
    UNR for (int q = 0; q < NLG; q++) gp[q] = gbase + goff[q];
    auto load_step = [&](int k, double* r) {
        UNR for (int q = 0; q < NLG; q++) r[q] = *gp[q];
        UNR for (int j = 0; j < NLX; j++) r[NLG + j] = *xptr[j];
        const size_t gadv = (k < T - 2) ? gstep : 0;  // uniform
        UNR for (int q = 0; q < NLG; q++) gp[q] += gadv;
        UNR for (int j = 0; j < NLX; j++) xptr[j] += (k < xkmax[j]) ? xstep[j] : 0;
    };
    auto stage_step = [&](int buf, const double* r) {
        double* dst = &s_in[buf][0][0];
        UNR for (int q = 0; q < NLG; q++) *(gdst[q] >= 0 ? dst + gdst[q] : &s_dump[tid]) = r[q];
        UNR for (int j = 0; j < NLX; j++) *(xdst[j] >= 0 ? dst + xdst[j] : &s_dump[tid]) = r[NLG + j];
    };
    // output stage: thread t stores component (t >> 4) + 16 j of instance t & 15
    constexpr int NST = (NO + 15) / 16;
    double* sptr[NST];
    size_t sstep[NST];
    UNR for (int j = 0; j < NST; j++) {
        const int cc = lc0 + 16 * j;
        const bool isx = cc < NX;
        sptr[j] = (isx ? a.X[1 - lcur] + (size_t)cc * Bp : a.U[1 - lcur] + (size_t)(cc - NX) * Bp) + lb;
        sstep[j] = (size_t)(isx ? NX : NU) * Bp;
    }
    auto store_step = [&](int k, int buf) {  // x_k (k <= T-1), u_k (k <= T-2)
        if (s_wr[li]) {
            UNR for (int j = 0; j < NST; j++) {
                const int cc = lc0 + 16 * j;
                if (cc < NX || (cc < NO && k < T - 1)) sptr[j][(size_t)k * sstep[j]] = s_out[buf][cc][li];
            }
        }
    };

    double pre[PF][NLD];
    UNR for (int j = 0; j < PF; j++) load_step(j, pre[j]);

    double x[NX];
    if (part) init_state<S>(d, a, b, x);
    else { UNR for (int i = 0; i < NX; i++) x[i] = 0; }
    const double cost0 = (b < B) ? a.cost[b] : 0.0;
    double newCost = 0, dun = 0;
    int kpi = 0;

    stage_step(0, pre[0]);
    lds_barrier();

    const int nsteps = T - 1;
    for (int k0 = 0; k0 < nsteps; k0 += PF) {
        UNR for (int j = 0; j < PF; j++) {
            const int k = k0 + j;
            if (k < nsteps) {  // uniform
                const int buf = k & 1;
                // stage step k+1 (loaded PF-1 steps ago), prefetch step k+PF into the slot just consumed
                double nxt[NLD];
                UNR for (int q = 0; q < NLD; q++) nxt[q] = pre[(j + 1) % PF][q];
                load_step(k + PF, pre[j]);
                if (part) {
                    // u = ubar + K (x - xbar) + alpha d, two control rows at a time.  The LDS reads of the NEXT two rows are issued before the
                    // products of the current two and fenced there (sched_barrier): left to itself the compiler issues one ds_read2 two
                    // instructions ahead of its use -- 60 exposed LDS latencies per step, and a wave is alone on its SIMD here
                    const double* sb = &s_in[buf][0][il];
                    auto SI = [&](int c) { return sb[c * TI]; };
                    double u[NU], dx[NX], xb[NX], kr[2][2][NX], dd[2][2], ub[2][2];
                    auto rows = [&](int s, int i0) {
                        UNR for (int r = 0; r < 2; r++) {
                            if (i0 + r < NU) {
                                UNR for (int q = 0; q < NX; q++) kr[s][r][q] = SI((i0 + r) * NX + q);
                                dd[s][r] = SI(NK + i0 + r);
                                ub[s][r] = SI(NK + NU + NX + i0 + r);
                            }
                        }
                    };
                    UNR for (int i = 0; i < NX; i++) xb[i] = SI(NK + NU + i);
                    rows(0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    UNR for (int i = 0; i < NX; i++) dx[i] = x[i] - xb[i];
                    double n2 = 0;
                    UNR for (int i0 = 0; i0 < NU; i0 += 2) {
                        const int s = (i0 >> 1) & 1;
                        if (i0 + 2 < NU) rows(s ^ 1, i0 + 2);
                        __builtin_amdgcn_sched_barrier(0);
                        double acc[2] = {0, 0};
                        UNR for (int q = 0; q < NX; q++) {
                            UNR for (int r = 0; r < 2; r++)
                                if (i0 + r < NU) acc[r] += kr[s][r][q] * dx[q];
                        }
                        UNR for (int r = 0; r < 2; r++) {
                            if (i0 + r < NU) {
                                const double du = acc[r] + alpha * dd[s][r];
                                n2 += du * du;
                                u[i0 + r] = ub[s][r] + du;
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    dun += sqrt(n2);
                    if (writer) {
                        UNR for (int i = 0; i < NX; i++) s_out[buf][i][il] = x[i];
                        UNR for (int i = 0; i < NU; i++) s_out[buf][NX + i][il] = u[i];
                    }
                    if (k == kp_next) {  // keypoint step (rare): hand (x, u) of this alpha to k_select_x -- the FK stays out of this kernel
                        if (!APPLY) {
                            double* o = a.kpx + ((size_t)kpi * 16 + ai) * (NX + NU) * Bp;
                            UNR for (int i = 0; i < NX; i++) AT(o, i, b) = x[i];
                            UNR for (int i = 0; i < NU; i++) AT(o, NX + i, b) = u[i];
                        }
                    }
                    if (!APPLY) newCost += lim.cost(x);
                    // dynamics step (SimulationInterface.cpp:19-31)
                    const double dts = S::TM ? u[NU - 1] : 0.0;
                    const double dt = S::TM ? dts * dts : dt_fixed;
                    if (S::ND == 1) {
                        UNR for (int i = 0; i < DOF; i++) x[i] = x[i] + (dt * u[i] + dt * dt / 2 * 0.0);
                    } else {
                        UNR for (int i = 0; i < DOF; i++) {
                            const double v = x[DOF + i];
                            x[i] = x[i] + (dt * v + dt * dt / 2 * u[i]);
                            x[DOF + i] = v + dt * u[i];
                        }
                    }
                    if (S::TM) x[NX - 1] = x[NX - 1] + dt;
                }
                if (k == kp_next) {  // uniform.  The index of the next keypoint step goes into a scalar register here: left in a vector register (and
                    kpi++;           // updated under `part`), every step's `k == kp_next` waited for that load -- and with it for all but the newest
                    kp_next = (kpi < n_kp) ? __builtin_amdgcn_readfirstlane(d.kp_t[kpi]) : -1;  // loads of the prefetch ring
                }
                stage_step(buf ^ 1, nxt);
                lds_barrier();
                store_step(k, buf);
            }
        }
    }
    // terminal state and cost
    {
        const int buf = nsteps & 1;
        if (part) {
            if (writer) { UNR for (int i = 0; i < NX; i++) s_out[buf][i][il] = x[i]; }
            if (!APPLY && kp_next == T - 1) {
                double* o = a.kpx + ((size_t)kpi * 16 + ai) * (NX + NU) * Bp;
                UNR for (int i = 0; i < NX; i++) AT(o, i, b) = x[i];
            }
            if (!APPLY) newCost += lim.cost(x);
        }
        lds_barrier();
        store_step(T - 1, buf);
    }

    if (APPLY) {
        if (writer) {
            a.cur[b] = 1 - a.cur[b];
            a.pend[b] = 0;
        }
        return;
    }
    // ---- limit cost and sum ||du|| of this alpha; the task cost and the decision are k_select_x's
    if (part) {
        AT(a.lsc, ai, b) = newCost;
        AT(a.dunA, ai, b) = dun;
    }
}

// Line-search decision for the alpha-parallel rollouts of k_forward_tile, one lane per (instance, alpha): task cost at the
// keypoints from the exported (x, u) + the limit cost; the first alpha whose cost is below the current one and not NaN wins,
// else the last one tried (ILQRRecursive.cpp:101-155).  If the winner is the lane that wrote its trajectory speculatively
// (the predicted winner) the buffers flip here, otherwise `pend` asks the APPLY pass to re-roll it.
template <class S>
__global__ __launch_bounds__(64) void k_select_x(Bufs a, FwdArgs f) {
    constexpr int NX = S::NX, NU = S::NU;
    const DevDesc& d = *a.desc;
    const int lane = threadIdx.x, gi = lane >> 4, ai = lane & 15;
    const int b = blockIdx.x * 4 + gi;
    const int Bp = d.Bp, T = d.T, B = d.B;
    const bool inst_ok = (b < B) && (a.active[b < B ? b : 0] != 0);
    const int bb = (b < B) ? b : 0;
    const bool part = inst_ok && ai < f.n_alpha;
    double c = 0;
    if (part) {
        for (int kpi = 0; kpi < d.n_kp; kpi++) {
            const int k = d.kp_t[kpi];
            const double* o = a.kpx + ((size_t)kpi * 16 + ai) * (NX + NU) * Bp;
            double xt[NX], ut[NU], tg[S::NF];
            UNR for (int i = 0; i < NX; i++) xt[i] = AT(o, i, bb);
            UNR for (int i = 0; i < NU; i++) ut[i] = (k < T - 1) ? AT(o, NX + i, bb) : 0.0;
            UNR for (int i = 0; i < S::NF; i++) tg[i] = AT(a.kp_tg, kpi * S::NF + i, bb);
            c += kp_cost<S>(d, kpi, tg, xt, ut);
        }
        c += AT(a.lsc, ai, bb);
    }
    const double cost0 = a.cost[bb];
    const bool ok = part && !((c >= cost0) || isnan(c));
    const unsigned m16 = (unsigned)((__ballot(ok ? 1 : 0) >> (gi * 16)) & 0xffffull);
    const int w = m16 ? (__ffs(m16) - 1) : (f.n_alpha - 1);
    const double wcost = __shfl(c, gi * 16 + w);
    if (inst_ok && ai == 0) {
        const double walpha = ldexp(1.0, -w);
        const double wdun = AT(a.dunA, w, bb);
        const int pr = a.pred[bb] < f.n_alpha ? a.pred[bb] : f.n_alpha - 1;
        a.cost[bb] = wcost;
        a.alpha[bb] = walpha;
        a.iters[bb] = f.it + 1;
        a.status[bb] = (isfinite(wcost) ? 0 : 1) | ((walpha <= d.alpha_floor) ? 2 : 0);
        if (a.cost_trace) {
            a.cost_trace[(size_t)f.it * Bp + bb] = wcost;
            a.alpha_trace[(size_t)f.it * Bp + bb] = walpha;
        }
        if (w == pr) a.cur[bb] = 1 - a.cur[bb];  // the speculatively written trajectory is the accepted one
        else a.pend[bb] = w + 1;
        a.pred[bb] = w;
        bool stop = f.early_stop && (walpha * sqrt(wdun) < d.stop_tol);
        if (!f.al) stop = stop && (wcost < 1e-3);  // ILQRRecursive.cpp:174 vs AL-ILQR.cpp:225
        if (stop) a.active[bb] = 0;
    }
}

// AL bookkeeping on the accepted trajectory of the instances that ran iteration `it`:
//   I_k      = penalty_roll * (g<0 && lambda==0 ? 0 : 1)          with the multipliers BEFORE the update
//   lambda_k = max(0, lambda_k + penalty_update * g)               only on update iterations
template <class S>
__global__ __launch_bounds__(256) void k_al_post(Bufs a, FwdArgs f) {
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (b >= d.B) return;
    if (a.iters[b] != f.it + 1) return;
    const int Bp = d.Bp;
    const int cur = a.cur[b];
    double x[S::NX], u[S::NU];
    UNR for (int i = 0; i < S::NX; i++) x[i] = AT(a.X[cur], k * S::NX + i, b);
    UNR for (int i = 0; i < S::NU; i++) u[i] = AT(a.U[cur], k * S::NU + i, b);
    for (int r = 0; r < a.m; r++) {
        const double g = con_g<S>(a, k, r, x, u);
        const double lam = AT(a.lambda, k * a.m + r, b);
        AT(a.Is, k * a.m + r, b) = f.penalty_roll * ((g < 0 && lam == 0) ? 0.0 : 1.0);
        if (f.do_update) {
            const double v = lam + f.penalty_update * g;
            AT(a.lambda, k * a.m + r, b) = v > 0 ? v : 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------ launchers

template <class S>
static void launch_tile(const Bufs& a, int B, hipStream_t st, const FwdArgs& f) {  // all step sizes of the time systems' line search + the decision
    const dim3 gridT(grid_x8((B + FT_TI - 1) / FT_TI)), blockT(FT_TI * 16);
    hipLaunchKernelGGL((k_forward_tile<S, false>), gridT, blockT, 0, st, a, f);
    hipLaunchKernelGGL((k_select_x<S>), dim3((B + 3) / 4), dim3(64), 0, st, a, f);
}
template <class S>
static void launch_al_post(const Bufs& a, int B, int T, hipStream_t st, const FwdArgs& f) {
    hipLaunchKernelGGL((k_al_post<S>), dim3((B + 255) / 256, T - 1), dim3(256), 0, st, a, f);
}

// closed-form sweep (k_backward_si_coop): usable for single-integrator dynamics when no constraint row touches the controls, the rows
// are shared over k and there are at most 4 of them (they live in registers)
bool backward_si_supported(int kind, int nd, bool al, int m, int per_step, bool con_state_only) {
    if (!((kind == 0 || kind == 2) && nd == 1)) return false;  // PosOrn-1 and JointSpace-1
    if (!al) return true;
    return con_state_only && per_step == 0 && m <= 4;
}

void launch_solver_v2(int kind, int nd, int which, bool al, const Bufs& a, int B, int T, hipStream_t st, const FwdArgs& f) {
    (void)al;
    if (which == KER_FWD_SPEC) {  // time systems only (PosOrn: k_forward_wg / k_forward_lin)
        if (kind == 3) launch_tile<Sys<3, 1>>(a, B, st, f);
        else if (kind == 1 && nd == 1) launch_tile<Sys<1, 1>>(a, B, st, f);
        else if (kind == 1) launch_tile<Sys<1, 2>>(a, B, st, f);
    } else if (which == KER_AL_UPDATE) {
        if (kind == 2) launch_al_post<Sys<2, 1>>(a, B, T, st, f);
        else if (kind == 3) launch_al_post<Sys<3, 1>>(a, B, T, st, f);
        else if (kind == 0 && nd == 1) launch_al_post<Sys<0, 1>>(a, B, T, st, f);
        else if (kind == 0 && nd == 2) launch_al_post<Sys<0, 2>>(a, B, T, st, f);
        else if (kind == 1 && nd == 1) launch_al_post<Sys<1, 1>>(a, B, T, st, f);
        else launch_al_post<Sys<1, 2>>(a, B, T, st, f);
    }
}

}  // namespace ilqr
