// ilqr_batchcp.hip -- BatchILQRCP on the device (reference src/solver/BatchILQRCP.cpp:109-175)
#include "ilqr_batchcp.hpp"

namespace ilqr {

int batchcp_solve(BatchCPState&, const DevDesc&, Bufs&, int, int, int, int, const double*, int, int, int, hipStream_t, std::string& err) {
    err = "ilqr_solve_batch_cp: not implemented in this build";
    return 1;
}
void batchcp_free(BatchCPState& st) {
    if (st.psi) (void)hipFree(st.psi);
    if (st.work) (void)hipFree(st.work);
    st = BatchCPState();
}

}  // namespace ilqr
