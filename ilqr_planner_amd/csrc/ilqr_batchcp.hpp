// ilqr_batchcp.hpp -- BatchILQRCP on the device (reference src/solver/BatchILQRCP.cpp:109-175)
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "ilqr_kernels.hpp"

namespace ilqr {

// Device buffers of the control-primitive solver; sized for one (problem, Kw) pair and reused across solves.
struct BatchCPState {
    int KWP = 0, Kw = 0, nkp = 0, nx = 0, Bp = 0, rows = 0;
    double* psi = nullptr;  // [(T-1) n_u][KWP]   PSI, columns zero-padded to KWP
    double* H0 = nullptr;   // [KWP][KWP]         PSI' R PSI (identity on the padded diagonal)
    double* Wkp = nullptr;  // [n_kp][NX*KWP][Bp] rows of Su PSI at the keypoint steps
    double* Ckp = nullptr;  // [n_kp][NX*NX][Bp]  J'QJ + L
    double* rkp = nullptr;  // [n_kp][NX][Bp]     J'Q e + L ql
    double* gu = nullptr;   // [KWP][Bp]          PSI' R u
    double* dw = nullptr;   // [KWP][Bp]
    double* dun = nullptr;  // [Bp]               ||du||
    std::vector<void*> allocs;
};

// Args shared by the three kernels
struct CPArgs {
    const double* psi;
    const double* H0;
    double *Wkp, *Ckp, *rkp, *gu, *dw, *dun;
    int Kw, it, early_stop, n_alpha;
};

int batchcp_solve(BatchCPState& st, const DevDesc& h, Bufs& bufs, int nx, int nu, int nf, int nq, const double* psi_host, int Kw,
                  int nb_iter, int early_stop, hipStream_t stream, std::string& err);
void batchcp_free(BatchCPState& st);

}  // namespace ilqr
