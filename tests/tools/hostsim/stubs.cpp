// Everything of the library that is NOT the lane-per-instance kernel set: refused on the host harness (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>

#include <string>

#include "ilqr_batchcp.hpp"
#include "ilqr_kernels.hpp"

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace ilqr {
[[noreturn]] static void refuse(const char* what) {
    std::fprintf(stderr, "hostsim: %s is a cooperative GPU kernel and cannot run on the sanitizer harness (use ILQR_HIP_PATH=v1)\n", what);
    std::abort();
}
bool backward_si_supported(int, int, bool, int, int, bool) { return false; }
bool backward_mfma_supported(int, int, bool, int) { return false; }
bool forward_lin_supported(int, int, int) { return false; }
bool forward_wave_supported(int, int, int) { return false; }
bool init_lti_supported(int, int) { return false; }
void launch_solver_v2(int, int, int, bool, const Bufs&, int, int, hipStream_t, const FwdArgs&) { refuse("launch_solver_v2"); }
void launch_backward_si_dpp(bool, bool, bool, int, const Bufs&, int, hipStream_t, const SweepArgs&) { refuse("k_backward_si_dpp"); }
void launch_backward_mfma(int, int, bool, const Bufs&, int, hipStream_t) { refuse("k_backward_mfma"); }
bool backward_rows_supported(int, int, bool, int) { return false; }
void launch_backward_rows(int, int, bool, const Bufs&, int, hipStream_t) { refuse("k_backward_rows"); }
void launch_apply_rows_tm(int, int, const Bufs&, int, hipStream_t, const FwdArgs&) { refuse("k_apply_rows_tm"); }
void launch_forward_mfma(int, int, const Bufs&, int, hipStream_t, const FwdArgs&) { refuse("k_forward_mfma"); }
void launch_forward_lin(int, int, const Bufs&, int, int, hipStream_t, const FwdArgs&) { refuse("k_forward_lin"); }
void launch_init_lti(int, int, const Bufs&, int, hipStream_t) { refuse("k_init_roll_lti"); }
void launch_forward_wave(int, const Bufs&, int, hipStream_t, const FwdArgs&) { refuse("k_forward_wg"); }
void launch_apply_wave(int, const Bufs&, int, int, hipStream_t, const FwdArgs&) { refuse("k_apply"); }
int batchwide_solve(BatchWideState&, const DevDesc&, Bufs&, int, int, const double*, int, int, int, bool, hipStream_t, std::string&, const ProfHook&) { refuse("batchwide_solve"); }
void batchwide_free(BatchWideState&) {}
int batchcp_solve(BatchCPState&, const DevDesc&, Bufs&, int, int, int, int, const double*, int, int, int, hipStream_t, std::string&, const ProfHook&) { refuse("batchcp_solve"); }
void batchcp_free(BatchCPState&) {}
}  // namespace ilqr
