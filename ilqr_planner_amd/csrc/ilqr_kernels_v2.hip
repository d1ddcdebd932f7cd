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
// k_al_update      multiplier update of AL-ILQR.cpp:202-208 on the accepted trajectory, one lane per (instance, k).
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
#include "ilqr_kernels.hpp"
#include "ilqr_step.hpp"

namespace ilqr {

// ------------------------------------------------------------------------------------------------ forward, alpha-parallel

template <class S, bool AL, bool APPLY>
__global__ __launch_bounds__(256) void k_forward_tile(Bufs a, FwdArgs f) {
    constexpr int NX = S::NX, NU = S::NU, NK = NU * NX;
    constexpr int NC = NK + NU + NX + NU;          // doubles per instance-step: K | d | xbar | ubar
    constexpr int NO = NX + NU;                    // doubles written per instance-step: x | u
    constexpr int NLD = (NC * 16 + 255) / 256;     // cooperative loads per thread per step
    constexpr int PF = 4;                          // prefetch distance in timesteps
    __shared__ double s_in[2][NC][16];
    __shared__ double s_out[2][NO][16];
    __shared__ int s_wr[16];

    const DevDesc& d = *a.desc;
    const int tid = threadIdx.x, il = tid >> 4, ai = tid & 15;
    const int b0 = blockIdx.x * 16, b = b0 + il;
    const int Bp = d.Bp, T = d.T;

    bool part;
    double alpha;
    if (!APPLY) {
        part = (b < d.B) && a.active[b] && (ai < f.n_alpha);
        alpha = ldexp(1.0, -ai);
    } else {
        const int w = (b < d.B) ? a.pend[b] : 0;
        part = (w > 0) && (ai == 0);
        alpha = ldexp(1.0, -w);
    }
    if (!__syncthreads_or(part ? 1 : 0)) return;  // nothing to do for these 16 instances (uniform)
    const bool writer = part && (ai == 0);
    if (ai == 0) s_wr[il] = writer ? 1 : 0;

    // loader role: this thread always fetches for instance (tid & 15), components (tid >> 4) + 16 j
    const int li = tid & 15, lb = b0 + li, lc0 = tid >> 4;
    const int lcur = a.cur[lb];
    const double* Xc = a.X[lcur];
    const double* Uc = a.U[lcur];
    auto load_step = [&](int k, double* r) {
        UNR for (int j = 0; j < NLD; j++) {
            const int c = lc0 + 16 * j;
            r[j] = 0;
            if (c < NC && k < T - 1) {
                const double* base;
                int row;
                if (c < NK) { base = a.K; row = k * NK + c; }
                else if (c < NK + NU) { base = a.D; row = k * NU + (c - NK); }
                else if (c < NK + NU + NX) { base = Xc; row = k * NX + (c - NK - NU); }
                else { base = Uc; row = k * NU + (c - NK - NU - NX); }
                r[j] = base[(size_t)row * Bp + lb];
            }
        }
    };
    auto stage_step = [&](int buf, const double* r) {
        UNR for (int j = 0; j < NLD; j++) {
            const int c = lc0 + 16 * j;
            if (c < NC) s_in[buf][c][li] = r[j];
        }
    };

    // writer role of the output stage: thread t < NO*16 stores component t>>4 of instance t&15
    const int cur_b = (b < d.B) ? a.cur[b] : 0;
    double* Xn = a.X[1 - lcur];  // for the store stage (instance li)
    double* Un = a.U[1 - lcur];
    auto store_step = [&](int k, int buf) {  // x_k (k <= T-1), u_k (k <= T-2)
        if (tid < NO * 16 && s_wr[li]) {
            const int c = lc0;
            UNR for (int j = 0; j < (NO + 15) / 16; j++) {
                const int cc = c + 16 * j;
                if (cc < NX) Xn[(size_t)(k * NX + cc) * Bp + lb] = s_out[buf][cc][li];
                else if (cc < NO && k < T - 1) Un[(size_t)(k * NU + (cc - NX)) * Bp + lb] = s_out[buf][cc][li];
            }
        }
    };
    (void)cur_b;

    double pre[PF][NLD];
    UNR for (int j = 0; j < PF; j++) load_step(j, pre[j]);

    double x[NX];
    if (part) init_state<S>(d, a, b, x);
    else { UNR for (int i = 0; i < NX; i++) x[i] = 0; }
    const double cost0 = (b < d.B) ? a.cost[b] : 0.0;
    double newCost = 0, dun = 0;
    int kpi = 0;

    stage_step(0, pre[0]);
    __syncthreads();

    const int nsteps = T - 1;
    for (int k0 = 0; k0 < nsteps; k0 += PF) {
        UNR for (int j = 0; j < PF; j++) {
            const int k = k0 + j;
            if (k < nsteps) {  // uniform
                const int buf = k & 1;
                // prefetch step k+PF into the slot just consumed, stage step k+1 (loaded PF-1 steps ago)
                double nxt[NLD];
                UNR for (int q = 0; q < NLD; q++) nxt[q] = pre[(j + 1) % PF][q];
                load_step(k + PF, pre[j]);
                if (part) {
                    double u[NU], xn[NX], dx[NX];
                    UNR for (int i = 0; i < NX; i++) dx[i] = x[i] - s_in[buf][NK + NU + i][il];
                    double n2 = 0;
                    UNR for (int i = 0; i < NU; i++) {
                        double s = 0;
                        UNR for (int q = 0; q < NX; q++) s += s_in[buf][i * NX + q][il] * dx[q];
                        const double du = s + alpha * s_in[buf][NK + i][il];
                        n2 += du * du;
                        u[i] = s_in[buf][NK + NU + NX + i][il] + du;
                    }
                    dun += sqrt(n2);
                    if (writer) {
                        UNR for (int i = 0; i < NX; i++) s_out[buf][i][il] = x[i];
                        UNR for (int i = 0; i < NU; i++) s_out[buf][NX + i][il] = u[i];
                        if (AL) {
                            for (int r = 0; r < a.m; r++) {
                                const double g = con_g<S>(a, k, r, x, u);
                                const double lam = AT(a.lambda, k * a.m + r, b);
                                AT(a.Is, k * a.m + r, b) = f.penalty_roll * ((g < 0 && lam == 0) ? 0.0 : 1.0);
                            }
                        }
                    }
                    const bool iskp = (kpi < d.n_kp && d.kp_t[kpi] == k);
                    newCost += stage_cost<S>(d, a, b, iskp ? kpi : -1, x, u);
                    if (iskp) kpi++;
                    dyn_step<S>(d, x, u, xn);
                    UNR for (int i = 0; i < NX; i++) x[i] = xn[i];
                }
                stage_step(buf ^ 1, nxt);
                __syncthreads();
                store_step(k, buf);
            }
        }
    }
    // terminal state and cost
    {
        const int buf = nsteps & 1;
        if (part) {
            if (writer) { UNR for (int i = 0; i < NX; i++) s_out[buf][i][il] = x[i]; }
            const bool iskp = (kpi < d.n_kp && d.kp_t[kpi] == T - 1);
            double zu[NU];
            UNR for (int i = 0; i < NU; i++) zu[i] = 0;
            newCost += stage_cost<S>(d, a, b, iskp ? kpi : -1, x, zu);
        }
        __syncthreads();
        store_step(T - 1, buf);
    }

    if (APPLY) {
        if (writer) {
            a.cur[b] = 1 - a.cur[b];
            a.pend[b] = 0;
        }
        return;
    }
    // ---- winner of the line search: first alpha with !(cost >= cost0 || isnan(cost)), else the last one tried
    const bool ok = part && !((newCost >= cost0) || isnan(newCost));
    const unsigned long long bal = __ballot(ok ? 1 : 0);
    const int lane = tid & 63, g0 = lane & ~15;
    const unsigned grp = (unsigned)((bal >> g0) & 0xFFFFull);
    const int w = grp ? (__ffs((int)grp) - 1) : (f.n_alpha - 1);
    const double wcost = __shfl(newCost, g0 + w);
    const double wdun = __shfl(dun, g0 + w);
    if (writer) {  // lane 0 of an active instance
        const double walpha = ldexp(1.0, -w);
        a.cost[b] = wcost;
        a.alpha[b] = walpha;
        a.iters[b] = f.it + 1;
        a.status[b] = (isfinite(wcost) ? 0 : 1) | ((walpha <= d.alpha_floor) ? 2 : 0);
        if (a.cost_trace) {
            a.cost_trace[(size_t)f.it * Bp + b] = wcost;
            a.alpha_trace[(size_t)f.it * Bp + b] = walpha;
        }
        if (w == 0) a.cur[b] = 1 - a.cur[b];  // the speculatively written alpha = 1 trajectory is the accepted one
        else a.pend[b] = w;
        bool stop = f.early_stop && (walpha * sqrt(wdun) < d.stop_tol);
        if (!AL) stop = stop && (wcost < 1e-3);
        if (stop) a.active[b] = 0;
    }
}

// multipliers: lambda <- max(0, lambda + penalty * g(x_k, u_k)) on the accepted trajectory of the instances that ran
// iteration `it` (AL-ILQR.cpp:202-208); one lane per (instance, timestep)
template <class S>
__global__ __launch_bounds__(256) void k_al_update(Bufs a, FwdArgs f) {
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
        const double v = AT(a.lambda, k * a.m + r, b) + f.penalty_update * g;
        AT(a.lambda, k * a.m + r, b) = v > 0 ? v : 0;
    }
}

// ------------------------------------------------------------------------------------------------ backward, closed form

__device__ __forceinline__ constexpr int sym(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// AL rows must not touch the controls for the closed form (checked on the host); they then only add to l_x, l_xx.
template <bool AL>
__global__ __launch_bounds__(64) void k_backward_si(Bufs a) {
    using S = Sys<0, 1>;
    constexpr int N = 7, NS = N * (N + 1) / 2;
    const DevDesc& d = *a.desc;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= d.B) return;
    if (!a.active[b]) return;
    const int Bp = d.Bp, T = d.T;
    const int cur = a.cur[b];
    const double* X = a.X[cur];
    const double* U = a.U[cur];
    const double dt = d.dt, idt = 1.0 / dt, idt2 = idt * idt, reg = d.reg;
    double Dg[N];
    UNR for (int i = 0; i < N; i++) Dg[i] = d.R_diag[i] + reg;

    double P[NS], p[N], x[N], u[N];
    int kpi = d.n_kp - 1;
    // terminal: P = l_xx(x_{T-1}), p = l_x(x_{T-1})   (no constraint on the final state, AL-ILQR.cpp:96)
    {
        UNR for (int i = 0; i < N; i++) x[i] = AT(X, (T - 1) * N + i, b);
        double lxx[N][N], lx[N];
        const bool iskp = (kpi >= 0 && d.kp_t[kpi] == T - 1);
        stage_derivs<S>(d, a, b, x, iskp ? kpi : -1, lxx, lx);
        if (iskp) kpi--;
        UNR for (int i = 0; i < N; i++) {
            p[i] = lx[i];
            UNR for (int j = 0; j <= i; j++) P[sym(i, j)] = lxx[i][j];
        }
    }
    // prefetch of the next timestep's xbar, ubar
    double xn_[N], un_[N];
    UNR for (int i = 0; i < N; i++) { xn_[i] = AT(X, (T - 2) * N + i, b); un_[i] = AT(U, (T - 2) * N + i, b); }

    for (int k = T - 2; k >= 0; k--) {
        UNR for (int i = 0; i < N; i++) { x[i] = xn_[i]; u[i] = un_[i]; }
        if (k > 0) { UNR for (int i = 0; i < N; i++) { xn_[i] = AT(X, (k - 1) * N + i, b); un_[i] = AT(U, (k - 1) * N + i, b); } }

        // S = D + dt^2 P   (= Quu + reg I)
        double Sm[NS];
        UNR for (int i = 0; i < N; i++)
            UNR for (int j = 0; j <= i; j++) Sm[sym(i, j)] = dt * (dt * P[sym(i, j)]) + ((i == j) ? Dg[i] : 0.0);
        // LDL^T:  S = L diag(e) L^T  (L unit lower, stored in Sm below the diagonal; 1/e on the diagonal)
        double ie[N];
        UNR for (int j = 0; j < N; j++) {
            double le[N];  // L_jq * e_q
            double ej = Sm[sym(j, j)];
            UNR for (int q = 0; q < j; q++) {
                le[q] = Sm[sym(j, q)] * Sm[sym(q, q)];  // the diagonal slot holds e_q once column q is done
                ej -= Sm[sym(j, q)] * le[q];
            }
            Sm[sym(j, j)] = ej;
            ie[j] = 1.0 / ej;
            UNR for (int i = j + 1; i < N; i++) {
                double s = Sm[sym(i, j)];
                UNR for (int q = 0; q < j; q++) s -= Sm[sym(i, q)] * le[q];
                Sm[sym(i, j)] = s * ie[j];
            }
        }
        // Li = L^-1 (unit lower)
        double Li[NS];
        UNR for (int i = 0; i < N; i++) {
            Li[sym(i, i)] = 1.0;
            UNR for (int j = 0; j < i; j++) {
                double s = Sm[sym(i, j)];
                UNR for (int q = j + 1; q < i; q++) s += Sm[sym(i, q)] * Li[sym(q, j)];
                Li[sym(i, j)] = -s;
            }
        }
        // M = S^-1 = Li^T diag(1/e) Li  (symmetric)
        double M[NS];
        UNR for (int i = 0; i < N; i++)
            UNR for (int j = 0; j <= i; j++) {
                double s = 0;
                UNR for (int q = i; q < N; q++) s += (Li[sym(q, i)] * ie[q]) * Li[sym(q, j)];
                M[sym(i, j)] = s;
            }
        // Qu = R u + dt p ; dv = -M Qu ; Md = M dv
        double Qu[N], dv[N], Md[N];
        UNR for (int i = 0; i < N; i++) Qu[i] = d.R_diag[i] * u[i] + dt * p[i];
        UNR for (int i = 0; i < N; i++) {
            double s = 0;
            UNR for (int j = 0; j < N; j++) s += M[sym(i, j)] * Qu[j];
            dv[i] = -s;
        }
        UNR for (int i = 0; i < N; i++) {
            double s = 0;
            UNR for (int j = 0; j < N; j++) s += M[sym(i, j)] * dv[j];
            Md[i] = s;
        }
        // K = (M D - I)/dt ; store K, d
        UNR for (int i = 0; i < N; i++) {
            UNR for (int j = 0; j < N; j++)
                AT(a.K, k * N * N + i * N + j, b) = (M[sym(i, j)] * Dg[j] - ((i == j) ? 1.0 : 0.0)) * idt;
            AT(a.D, k * N + i, b) = dv[i];
        }
        // stage derivatives (keypoint / limits / AL rows)
        double lx[N];
        UNR for (int i = 0; i < N; i++) lx[i] = 0;
        double lxxs[NS];
        UNR for (int i = 0; i < NS; i++) lxxs[i] = 0;
        {
            const bool iskp = (kpi >= 0 && d.kp_t[kpi] == k);
            if (iskp) {
                double lxx[N][N], lxf[N];
                stage_derivs<S>(d, a, b, x, kpi, lxx, lxf);  // includes the limit terms
                UNR for (int i = 0; i < N; i++) {
                    lx[i] = lxf[i];
                    UNR for (int j = 0; j <= i; j++) lxxs[sym(i, j)] = lxx[i][j];
                }
                kpi--;
            } else if (d.limits_set) {
                UNR for (int i = 0; i < N; i++) {
                    if (d.lw[i] != 0) {
                        double qv = 0, L = 0;
                        if (x[i] > d.smax[i]) { qv = d.smax[i] - x[i]; L = d.penalty; }
                        else if (x[i] < d.smin[i]) { qv = d.smin[i] - x[i]; L = d.penalty; }
                        lx[i] += -L * qv;
                        lxxs[sym(i, i)] += L * L;
                    }
                }
            }
        }
        if (AL) {
            const int ns = 2 * N;
            for (int r = 0; r < a.m; r++) {
                const double* Ar = a.conA + ((size_t)(a.per_step ? k : 0) * a.m + r) * ns;
                const double Ik = AT(a.Is, k * a.m + r, b);
                const double lam = AT(a.lambda, k * a.m + r, b);
                const double g = con_g<S>(a, k, r, x, u);
                const double wv = lam + Ik * g;
                UNR for (int i = 0; i < N; i++) {
                    UNR for (int j = 0; j <= i; j++) lxxs[sym(i, j)] += Ar[i] * Ik * Ar[j];
                    lx[i] += Ar[i] * wv;
                }
            }
        }
        // M2 = M M (symmetric) and the P, p updates
        double Pn[NS];
        UNR for (int i = 0; i < N; i++)
            UNR for (int j = 0; j <= i; j++) {
                double m2 = 0;
                UNR for (int q = 0; q < N; q++) m2 += M[sym(i, q)] * M[sym(q, j)];
                const double mij = M[sym(i, j)];
                const double del = (i == j) ? 1.0 : 0.0;
                const double t = del * Dg[i] - Dg[i] * mij * Dg[j] - reg * (Dg[i] * m2 * Dg[j] - Dg[i] * mij - mij * Dg[j] + del);
                Pn[sym(i, j)] = lxxs[sym(i, j)] + t * idt2;
            }
        UNR for (int i = 0; i < N; i++) p[i] = lx[i] + p[i] - (Qu[i] + Dg[i] * dv[i]) * idt - reg * (Dg[i] * Md[i] - dv[i]) * idt;
        UNR for (int i = 0; i < NS; i++) P[i] = Pn[i];
    }
}

// ------------------------------------------------------------------------------------------------ launchers

template <class S>
static void launch_v2_kernel(int which, bool al, const Bufs& a, int B, int T, hipStream_t st, const FwdArgs& f) {
    const dim3 gridT((B + 15) / 16), blockT(256);
    switch (which) {
        case KER_FWD_SPEC:
            if (al) hipLaunchKernelGGL((k_forward_tile<S, true, false>), gridT, blockT, 0, st, a, f);
            else hipLaunchKernelGGL((k_forward_tile<S, false, false>), gridT, blockT, 0, st, a, f);
            break;
        case KER_FWD_APPLY:
            if (al) hipLaunchKernelGGL((k_forward_tile<S, true, true>), gridT, blockT, 0, st, a, f);
            else hipLaunchKernelGGL((k_forward_tile<S, false, true>), gridT, blockT, 0, st, a, f);
            break;
        case KER_AL_UPDATE:
            hipLaunchKernelGGL((k_al_update<S>), dim3((B + 255) / 256, T - 1), dim3(256), 0, st, a, f);
            break;
    }
}

void launch_solver_v2(int kind, int nd, int which, bool al, const Bufs& a, int B, int T, hipStream_t st, const FwdArgs& f) {
    if (which == KER_BACKWARD_SI) {
        const dim3 grid((B + 63) / 64), block(64);
        if (al) hipLaunchKernelGGL((k_backward_si<true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_backward_si<false>), grid, block, 0, st, a);
        return;
    }
    if (kind == 0 && nd == 1) launch_v2_kernel<Sys<0, 1>>(which, al, a, B, T, st, f);
    else if (kind == 0 && nd == 2) launch_v2_kernel<Sys<0, 2>>(which, al, a, B, T, st, f);
    else if (kind == 1 && nd == 1) launch_v2_kernel<Sys<1, 1>>(which, al, a, B, T, st, f);
    else launch_v2_kernel<Sys<1, 2>>(which, al, a, B, T, st, f);
}

}  // namespace ilqr
