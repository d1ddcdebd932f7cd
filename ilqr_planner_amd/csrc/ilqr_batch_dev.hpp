// ilqr_batch_dev.hpp -- device helpers shared by the batch (Gauss-Newton on the whole control sequence) solvers:
// ilqr_batchcp.hip (narrow bases, Kw <= 16) and ilqr_batchwide.hip (wide bases and BatchILQR's identity basis).
#pragma once
#include "ilqr_step.hpp"

namespace ilqr {

// dynamics step + the pieces of A, B the W recurrence needs (forwardPass, PosOrn*PlannerSys.cpp)
template <class S>
struct StepAB {
    double dt, dts, hdt2;
    double bc[S::NX];  // last column of B (time systems)
};

template <class S>
ILQR_DEV void step_ab(const DevDesc& d, const double* x, const double* u, double* xn, StepAB<S>& ab) {
    constexpr int NX = S::NX, NU = S::NU;
    dyn_step<S>(d, x, u, xn);
    ab.dts = S::TM ? u[NU - 1] : 0.0;
    ab.dt = S::TM ? ab.dts * ab.dts : d.dt;
    ab.hdt2 = ab.dt * ab.dt / 2;
    if (S::TM) {
        if (S::ND == 1) {
            UNR for (int i = 0; i < DOF; i++) ab.bc[i] = 2 * ab.dts * u[i];
        } else {
            UNR for (int i = 0; i < DOF; i++) {
                ab.bc[i] = 2 * ab.dts * xn[DOF + i] + 2 * ab.dts * ab.dts * ab.dts * u[i];  // velocity AFTER the step
                ab.bc[DOF + i] = 2 * ab.dts * u[i];
            }
        }
        ab.bc[NX - 1] = 2 * ab.dts;
    }
}

// limits on a state: diag(L) and q (inspectJointLimit)
template <class S>
ILQR_DEV void limit_terms(const DevDesc& d, const double* x, double* Ld, double* q) {
    UNR for (int i = 0; i < S::NX; i++) {
        Ld[i] = 0;
        q[i] = 0;
        if (d.limits_set && d.batch_limits && d.lw[i] != 0) {
            if (x[i] > d.smax[i]) { q[i] = d.smax[i] - x[i]; Ld[i] = d.penalty; }
            else if (x[i] < d.smin[i]) { q[i] = d.smin[i] - x[i]; Ld[i] = d.penalty; }
        }
    }
}

}  // namespace ilqr
