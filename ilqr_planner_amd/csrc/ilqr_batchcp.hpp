// ilqr_batchcp.hpp -- BatchILQRCP on the device (reference src/solver/BatchILQRCP.cpp:109-175)
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "ilqr_kernels.hpp"

namespace ilqr {

struct BatchCPState {
    double* psi = nullptr;  // device copy of PSI ((T-1) n_u x Kw)
    double* work = nullptr;
    size_t psi_elems = 0, work_elems = 0;
};

int batchcp_solve(BatchCPState& st, const DevDesc& h, Bufs& bufs, int nx, int nu, int nf, int nq, const double* psi_host, int Kw,
                  int nb_iter, int early_stop, hipStream_t stream, std::string& err);
void batchcp_free(BatchCPState& st);

}  // namespace ilqr
