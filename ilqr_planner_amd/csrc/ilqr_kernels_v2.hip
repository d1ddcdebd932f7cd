// ilqr_kernels_v2.hip -- decision and bookkeeping kernels of the cooperative pipelines (gfx950, fp64).
//
// k_select_x       line-search decision of the time systems: task cost of every step size from the keypoint states the rollout kernel
//                  exported (k_forward_mfma, ilqr_kernels_fwdm.hip) + its limit cost; the winner as the do/while of ILQRRecursive.cpp:101-155
//                  would have found it.
// k_al_post        AL bookkeeping on the ACCEPTED trajectory, one lane per (instance, k): the active-set weights
//                  penalty * I_k the next backward sweep needs (AL-ILQR.cpp:21-44,190 -- every trial overwrites them, so
//                  only the accepted trial's values survive in the reference too) and, every lag_update_step
//                  iterations, the multiplier update (AL-ILQR.cpp:202-208).  Keeps the rollout kernels free of AL.
// + the launchers of this file's kernels and the rule that selects the closed-form sweep (backward_si_supported).
#include <cstdlib>
#include <cstring>

#include "ilqr_kernels.hpp"
#include "ilqr_step.hpp"

namespace ilqr {

// Line-search decision for the alpha-parallel rollouts of k_forward_mfma (ilqr_kernels_fwdm.hip), one lane per (instance, alpha): task cost at the
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
static void launch_select(const Bufs& a, int B, hipStream_t st, const FwdArgs& f) {
    hipLaunchKernelGGL((k_select_x<S>), dim3((B + 3) / 4), dim3(64), 0, st, a, f);
}
template <class S>
static void launch_al_post(const Bufs& a, int B, int T, hipStream_t st, const FwdArgs& f) {
    hipLaunchKernelGGL((k_al_post<S>), dim3((B + 255) / 256, T - 1), dim3(256), 0, st, a, f);
}

// closed-form sweep (k_backward_si_dpp): usable for single-integrator dynamics when no constraint row touches the controls, the rows
// are shared over k and there are at most 4 of them (they live in registers)
bool backward_si_supported(int kind, int nd, bool al, int m, int per_step, bool con_state_only) {
    if (!((kind == 0 || kind == 2) && nd == 1)) return false;  // PosOrn-1 and JointSpace-1
    if (!al) return true;
    return con_state_only && per_step == 0 && m <= 4;
}

void launch_solver_v2(int kind, int nd, int which, bool al, const Bufs& a, int B, int T, hipStream_t st, const FwdArgs& f) {
    (void)al;
    if (which == KER_FWD_SPEC) {  // time systems only (PosOrn: k_forward_wg / k_forward_lin): all step sizes of the line search + the decision
        launch_forward_mfma(kind, nd, a, B, st, f);
        if (kind == 3) launch_select<Sys<3, 1>>(a, B, st, f);
        else if (kind == 1 && nd == 1) launch_select<Sys<1, 1>>(a, B, st, f);
        else if (kind == 1) launch_select<Sys<1, 2>>(a, B, st, f);
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
