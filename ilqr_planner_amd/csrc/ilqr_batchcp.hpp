// ilqr_batchcp.hpp -- BatchILQRCP on the device (reference src/solver/BatchILQRCP.cpp:109-175)
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/ilqr_hip.h"  // ILQR_PROF_* categories
#include "ilqr_kernels.hpp"

namespace ilqr {

// Per-launch timing hook handed in by the C ABI layer (HIP-event marks of ilqr_profile_*): hook(ILQR_PROF_x) in front of a launch charges
// the time until the next mark to category x.  Empty hook = profiling off.
struct ProfHook {
    void (*mark)(void*, int) = nullptr;
    void* ctx = nullptr;
    void operator()(int which) const { if (mark) mark(ctx, which); }
};

// Device buffers of the control-primitive solver; sized for one (problem, Kw) pair and reused across solves.
struct BatchCPState {
    int KWP = 0, Kw = 0, nkp = 0, nx = 0, Bp = 0, rows = 0;
    bool xc_lane_solve = false, xc_general = false;  // cross-check variants, set from the context before every solve (ilqr_ctx_set_crosscheck)
    double* psi = nullptr;  // [(T-1) n_u][KWP]   PSI, columns zero-padded to KWP
    double* H0 = nullptr;   // [KWP][KWP]         PSI' R PSI (identity on the padded diagonal)
    double* Wkp = nullptr;  // [n_kp][NX*KWP][Bp] rows of Su PSI at the keypoint steps
    double* Ckp = nullptr;  // [n_kp][NX*NX][Bp]  J'QJ + L
    double* rkp = nullptr;  // [n_kp][NX][Bp]     J'Q e + L ql
    double* gu = nullptr;   // [KWP][Bp]          PSI' R u
    double* dw = nullptr;   // [KWP][Bp]
    double* dun = nullptr;  // [Bp]               ||du||
    // coefficient-space path of the PosOrn systems (constant A, B): u = u0 + PSI w
    double* wt = nullptr;   // [n_kp][2][NX][KWP] true sensitivities d x_t / d w and d x_{t-1} / d w at the keypoint steps (shared)
    double* pp = nullptr;   // [KWP][KWP]         PSI' PSI (shared)
    double* wref = nullptr; // [n_kp][NX][KWP]    the reference's shifted W = Su PSI at the keypoint steps (shared; broadcast into Wkp)
    double* wv = nullptr;   // [KWP][Bp]          w
    double* g0 = nullptr;   // [KWP][Bp]          PSI' R u0
    double* c00 = nullptr;  // [Bp]               u0' R u0
    double* xbk = nullptr;  // [n_kp][2][NX][Bp]  x_t, x_{t-1} of the rollout of u0
    std::vector<void*> allocs;
    // what the shared tables on the device (psi, H0, wt, wref, pp) were built from: a solve with the same basis on the same system skips
    // the host-side table construction, the uploads and their stream synchronisations (0.25 ms of a 3.3 ms C5 solve)
    std::vector<double> psi_host;
    std::vector<double> sig;
    bool cpl_tables = false;  // wt / wref / pp of the coefficient-space path are valid for (psi_host, sig)
};

// Args shared by the three kernels
struct CPArgs {
    const double* psi;
    const double* H0;
    double *Wkp, *Ckp, *rkp, *gu, *dw, *dun;
    const double *wt, *pp;
    const double* wref;  // shared W = Su PSI at the keypoint steps (constant-dt systems) or null: per-instance Wkp
    double *wv, *g0, *c00, *xbk;
    int Kw, it, early_stop, n_alpha;
};

// Wide bases (Kw > 16) and BatchILQR's identity basis (psi_host == nullptr, Kw = (T-1) n_u): ilqr_batchwide.hip
struct BatchWideState {
    std::vector<void*> allocs;
    long long key = -1;  // (m, N, Bp, kind) the buffers were sized for
    std::vector<double> tab_sig, tab_psi;  // what the shared tables on the device were built from (rebuilt only when it changes)
    // LTI systems: shared tables
    double *G = nullptr, *Et = nullptr, *ZPZ = nullptr, *PZ = nullptr;
    // per instance
    double *xbk = nullptr, *av = nullptr, *v0 = nullptr, *p0 = nullptr, *scal = nullptr, *cv = nullptr, *beta = nullptr, *dvb = nullptr, *sc = nullptr,
           *Ckp = nullptr, *rkp = nullptr, *u0hat = nullptr, *g0 = nullptr, *y0 = nullptr, *psi = nullptr, *h0inv = nullptr;
};
int batchwide_solve(BatchWideState& st, const DevDesc& h, Bufs& bufs, int nx, int nu, const double* psi_host, int Kw, int nb_iter, int early_stop,
                    bool u0_zero, hipStream_t stream, std::string& err, const ProfHook& ph = ProfHook());
void batchwide_free(BatchWideState& st);

int batchcp_solve(BatchCPState& st, const DevDesc& h, Bufs& bufs, int nx, int nu, int nf, int nq, const double* psi_host, int Kw,
                  int nb_iter, int early_stop, hipStream_t stream, std::string& err, const ProfHook& ph = ProfHook());
void batchcp_free(BatchCPState& st);

}  // namespace ilqr
