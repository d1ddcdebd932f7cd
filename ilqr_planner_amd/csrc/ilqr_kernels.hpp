// ilqr_kernels.hpp -- device buffer table + kernel launchers shared by ilqr_kernels.hip and ilqr_capi.cpp
#pragma once
#include <hip/hip_runtime.h>

#include "ilqr_device.hpp"

namespace ilqr {

// All trajectory-like buffers are [row][Bp] with the (padded) batch innermost; row = t * DIM + component.
struct Bufs {
    const DevDesc* desc;
    double* X[2];  // [T][NX][Bp]   double-buffered; cur[b] says which one holds instance b's accepted trajectory
    double* U[2];  // [T-1][NU][Bp]
    const double* U0;  // [T-1][NU][Bp]
    double* KD;    // gains, one record per (timestep, instance): [T-1][Bp][NU][ROWP], row i = { K_k[i][0..NX-1], d_k[i], pad };
                   // ROWP = NX+1 rounded up to even, so rows are 16-byte aligned: the cooperative kernels read a gain row
                   // with 16-byte loads and the sweep writes an instance's block as one contiguous 448-byte run
    const double* q0;   // [DOF][Bp]
    const double* dq0;  // [DOF][Bp]
    const double* kp_tg;  // [n_kp][NF][Bp]
    double* cost;   // [Bp]
    double* alpha;  // [Bp]
    int* cur;
    int* active;
    int* iters;
    int* status;
    double* kpd;    // [n_kp][NX + NX*NX][Bp] l_x | l_xx of the keypoint steps of the current trajectory (k_kp_derivs)
    int* pend;      // [Bp] line-search winner index + 1 still to be applied by the APPLY pass (0 = nothing pending)
    int* pred;      // [Bp] predicted winner index of the next line search (= winner of the previous iteration)
    double* lsc;    // [16][Bp] limit cost of the alpha = 1 rollout blended to every step size (k_forward_wg -> k_select)
    double* dun;    // [Bp] sum_k ||du_k(1)|| of that rollout
    double* kpdev;  // [n_kp][NX+NU][Bp] deviation (dx, du) of that rollout at the keypoint steps
    double* kpx;    // [n_kp][16][NX+NU][Bp] state | control of every alpha's rollout at the keypoint steps (k_forward_mfma -> k_select_x)
    double* dunA;   // [16][Bp] sum_k ||du_k|| of every alpha's rollout
    double* ws;     // [backward_ws_entries][Bp] matrices of a step of the generic sweep (k_backward), allocated at its first use
    double* cost_trace;   // [nb_iter][Bp] or null
    double* alpha_trace;  // [nb_iter][Bp] or null
    // augmented Lagrangian (shared constraint rows, per-instance multipliers)
    int m, per_step;
    const double* conA;  // [T-1 or 1][m][NX+NU]
    const double* conb;  // [T-1 or 1][m]
    double* lambda;      // [T-1][m][Bp]
    double* Is;          // [T-1][m][Bp]  penalty * active-set mask at rollout time
    int kd_sym;          // the records of KD are in the packed symmetric form of the last solve (see KD_SYM_RS), set by the host per solve
};

constexpr int kd_rowp(int nx) { return (nx + 2) & ~1; }
// Packed record of the single-integrator sweep with uniform control weights (k_backward_si_dpp<.., UNIF = true>): there K = N / dt = ((R + reg) M - I) / dt
// is symmetric (M = S^-1 is, and R + reg is a multiple of I), so the record holds the upper triangle of K row by row (28 entries), d (7) and one pad:
// 36 doubles = 288 bytes instead of 56 = 448.  Gains are 2/3 of what the forward pass reads and 57 % of what the sweep writes.  Entry (i, j), i > j, is
// read as (j, i): the two differ in the last bits (the rows of the swept matrix are formed by different lanes), which the parity gates see as rounding.
constexpr int KD_SYM_RS = 36, KD_SYM_D = 28;
__host__ __device__ constexpr int kd_sym_tri(int i, int j) { return i * 7 - i * (i - 1) / 2 + (j - i); }          // i <= j < 7
__host__ __device__ constexpr int kd_sym_off(int i, int j) { return j >= 7 ? KD_SYM_D + i : (i <= j ? kd_sym_tri(i, j) : kd_sym_tri(j, i)); }  // j = 7: d_i
// offset of entry (i, j) (j = n_x: the feed-forward) in a record of either form, and the record length
__host__ __device__ inline int kd_off(int sym, int rowp, int i, int j) { return sym ? kd_sym_off(i, j) : i * rowp + j; }
__host__ __device__ inline int kd_rs(int sym, int nu, int rowp) { return sym ? KD_SYM_RS : nu * rowp; }
#define KD_REC(kd, Bp, RS, k, b) ((kd) + ((size_t)(k) * (size_t)(Bp) + (size_t)(b)) * (size_t)(RS))

struct FwdArgs {
    int it, line_search, early_stop, do_update, nb_iter;
    int n_kp;     // number of keypoints (grid of KER_KP_DERIVS)
    int al;       // 1 = AL_ILQR semantics (early stop without the cost test)
    int n_alpha;  // number of step sizes 1, 1/2, ... the line search may try (11 for alpha_floor = 1e-3)
    double penalty_roll, penalty_update;
    int kp_ext;   // some keypoint has a dead zone, an object frame or its own control penalty (selects the full keypoint code)
    int apply_dpp;  // time systems: the re-roll of the winner by k_apply_dpp_tm (16 lanes per instance on registers) instead of k_apply_rows_tm
    int limits;   // the descriptor's limits_set (host copy: selects kernel instantiations)
    int small;    // batch of at most one wave per SIMD: the latency-built forward pass (k_forward_dpp) instead of the bandwidth-built one
    int fused;    // the acceptance of the line search is applied by the next sweep (k_backward_si_dpp<.., true>): until then the accepted
                  // trajectory of an instance with pend > 0 is xbar + alpha (x(1) - xbar) over its two buffers
};

// Fused acceptance in the cooperative sweep: what k_apply would have used at the end of the PREVIOUS iteration
struct SweepArgs {
    double pen_in;           // penalty the active-set weights of the incoming trajectory are formed with (AL-ILQR.cpp:190)
    double pen_update_prev;  // penalty of the previous iteration's multiplier update (AL-ILQR.cpp:203-205)
    int do_update_prev;      // the previous iteration was an update iteration
};

// v1 (one lane per instance, generic): KER_INIT, KER_BACKWARD, KER_FORWARD
// v2: KER_FWD_SPEC  = all n_alpha line-search trials of an instance at once (16 lanes per instance), K read once
//     KER_FWD_APPLY = re-roll the winning step size for the instances whose winner was not alpha = 1
//     KER_AL_UPDATE = multiplier update on the accepted trajectory
//     KER_KP_DERIVS = l_x, l_xx at the keypoint steps (FK + Jacobian), one lane per (instance, keypoint), feeding every sweep (launch_solver)
enum { KER_INIT = 0, KER_BACKWARD = 1, KER_FORWARD = 2, KER_FWD_SPEC = 3, KER_FWD_APPLY = 4, KER_AL_UPDATE = 5, KER_KP_DERIVS = 7 };

void launch_solver(int kind, int nd, int which, bool al, const Bufs& a, int B, hipStream_t st, const FwdArgs& f);
bool backward_si_supported(int kind, int nd, bool al, int m, int per_step, bool con_state_only);
void launch_solver_v2(int kind, int nd, int which, bool al, const Bufs& a, int B, int T, hipStream_t st, const FwdArgs& f);
void launch_backward_si_dpp(bool al, bool fused, bool uniform_R, int lpi, const Bufs& a, int B, hipStream_t st, const SweepArgs& sw);  // rows in registers, DPP broadcasts (ilqr_kernels_dpp.hip)
bool backward_mfma_supported(int kind, int nd, bool al, int m);
bool backward_rows_supported(int kind, int nd, bool al, int m);  // row-per-lane register sweep of the general systems (ilqr_kernels_rowsweep.hip)
void launch_backward_rows(int kind, int nd, bool al, const Bufs& a, int B, hipStream_t st);  // needs KER_KP_DERIVS first
int backward_ws_entries(int kind, int nd);  // doubles per instance of k_backward's workspace
void launch_backward_mfma(int kind, int nd, bool al, const Bufs& a, int B, hipStream_t st);  // needs KER_KP_DERIVS first
bool forward_lin_supported(int kind, int nd, int n_alpha);
void launch_forward_mfma(int kind, int nd, const Bufs& a, int B, hipStream_t st, const FwdArgs& f);  // time systems: all step sizes of an instance as one matrix-core product per step (ilqr_kernels_fwdm.hip)
void launch_apply_rows_tm(int kind, int nd, const Bufs& a, int B, hipStream_t st, const FwdArgs& f);  // time systems: re-roll of the winner, 8 lanes per instance
void launch_forward_lin(int nd, int which, const Bufs& a, int B, int T, hipStream_t st, const FwdArgs& f);
bool forward_wave_supported(int kind, int nd, int n_alpha);
bool init_lti_supported(int kind, int nd);
void launch_init_lti(int kind, int nd, const Bufs& a, int B, hipStream_t st);  // followed by KER_AL_UPDATE (it = -1) for AL solves
void launch_forward_wave(int kind, const Bufs& a, int B, hipStream_t st, const FwdArgs& f);
void launch_apply_wave(int kind, const Bufs& a, int B, int T, hipStream_t st, const FwdArgs& f);  // blend + AL bookkeeping + flip
void launch_fx_all(int kind, int nd, const Bufs& a, int B, int T, double* out, hipStream_t st);
void launch_to_soa(const double* src, double* dst, int B, int Bp, int rows, hipStream_t st);
void launch_from_soa(const double* src, double* dst, int B, int Bp, int rows, hipStream_t st);
void launch_from_soa_cur(const double* s0, const double* s1, const int* cur, double* dst, int B, int Bp, int rows, hipStream_t st);
void launch_from_soa_scaled(const double* src, const double* alpha, const int* iters, double* dst, int B, int Bp, int rows, hipStream_t st);
void launch_get_gains(const double* kd, int kd_sym, const double* alpha, const int* iters, double* K_out, double* d_out, int B, int Bp, int T1, int nu, int nx, hipStream_t st);
void launch_warm_start(const Bufs& a, double* U0, double* q0, double* dq0, int shift, int B, int T, int nx, int nu, int nd, hipStream_t st);
void launch_track(const Bufs& a, const double* x_meas, int k, int with_ff, double* u_out, int B, int nx, int nu, hipStream_t st);
void launch_fk_batch(const DevDesc* dd, int n, const double* q, double* pos, double* quat, double* jac, hipStream_t st);

}  // namespace ilqr
