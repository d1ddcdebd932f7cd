"""Drives the sanitizer harness (build.sh) through the C ABI: the AL tutorial solve that ended in a GPU memory fault in round 1
(k_backward<Sys<0,1>, AL>, m = 14 rows, T = 400, 3 instances), after an AL batch in the same process and after a context teardown,
then every other system shape of the lane-per-instance kernel set, a second limit set, and re-specified constraint sets.
Results are checked against the notebook traces / the oracle, so the run is also a functional test of the same source.

    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tests/tools/hostsim/run_v1_under_asan.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.environ["ILQR_HIP_PATH"] = "v1"

import numpy as np  # noqa: E402

from ilqr_planner_amd import capi  # noqa: E402

capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libilqr_hostsim.so")  # harness, not the product library

from ilqr_planner_amd import workloads  # noqa: E402
from tests.helpers import assert_trace, golden, oracle_solve_instance  # noqa: E402
from tests.test_gpu_parity import _tutorial_problem  # noqa: E402


def al_tutorial(ctx, nit):
    case = golden()["cases"]["POS_ORN_SYS_AL_ILQR"]
    sv = case["solves"][1]
    B = 3
    p = _tutorial_problem(ctx, case, B)
    m = sv["m"]
    A, b = np.zeros((m, p.dims.n_x + p.dims.n_u)), np.zeros(m)
    for i, j, val in sv["A_nonzero"]:
        A[i, j] = val
    for i, val in sv["b_nonzero"]:
        b[i] = val
    p.set_constraints(A, b, np.tile(b, (B, p.T - 1, 1)))
    p.solve_al(nit, sv["lag_update_step"], sv["penalty"], sv["scaling_factor"], True, True)
    ct, at = p.trace(nit)
    n = min(nit, len(sv["trace"]))
    assert_trace(ct[0, :n], at[0, :n], sv["trace"][:n])
    print(f"AL tutorial: {n} iterations reproduce the notebook trace", flush=True)
    return p


def batch(ctx, name, B, nit, limits="inactive"):
    cfg = workloads.config(name)
    desc, inp = workloads.make_batch(ctx, cfg, B=B, limits=limits)
    p = workloads.load_batch(ctx, desc, inp, B)
    workloads.run_solver(p, cfg, nb_iter=nit, early_stop=True)
    cost = p.cost()
    r = oracle_solve_instance(cfg, inp, 0, nit, True)
    rel = abs(cost[0] - r["cost"]) / max(abs(r["cost"]), 1e-12) if np.isfinite(r["cost"]) else 0.0
    print(f"{name} B={B} limits={limits}: instance 0 rel err vs oracle {rel:.1e}", flush=True)
    assert rel < 1e-6 or not np.isfinite(r["cost"]), (name, rel)
    return p, cfg, inp


def main():
    nit = int(os.environ.get("NIT", "12"))
    # (1) the failing sequence of round 1: an AL batch, context closed, a new context, the AL tutorial solve
    ctx = capi.Context(0)
    p, _, _ = batch(ctx, "C3", 70, 3)
    p.close()
    ctx.close()
    ctx = capi.Context(0)
    p = al_tutorial(ctx, nit)
    p.iters()
    p.close()
    # (2) the same solve twice in one context, the first problem still alive; then closed out of order
    p1 = al_tutorial(ctx, 3)
    p2 = al_tutorial(ctx, 3)
    p1.close()
    p2.lam()
    p2.close()
    # (3) constraint sets re-specified with other shapes on a live problem (buffers of the old shape are released)
    p, cfg, inp = batch(ctx, "C3", 5, 2)
    T = cfg["T"]
    for m, per_step in ((3, False), (2, True), (3, True), (1, False)):
        A = np.zeros((m, 14))
        for r in range(m):
            A[r, (5 + r) % 7] = 1.0
        b = np.full(m, 2.0)
        if per_step:
            A, b = np.tile(A, (T - 1, 1, 1)), np.tile(b, (T - 1, 1))
        p.set_constraints(A, b, None)
        p.solve_al(2, 5, 0.25, 1.1, True, False)
        p.cost()
    p.close()
    # (4) every other system shape of the lane-per-instance kernels, active limits, second limit set, dead zones, hybrid keypoints
    for name, B, n, lim in (("C2", 67, 3, "urdf"), ("C2nd", 5, 3, "urdf"), ("C4t1", 5, 3, "inactive"), ("C4", 3, 2, "inactive"), ("C2hl", 5, 3, "urdf"),
                            ("C3d", 5, 3, "inactive"), ("C2ndal", 4, 3, "inactive"), ("C4t1al", 4, 3, "inactive"), ("C1j", 5, 3, "active"),
                            ("C1tal", 4, 3, "inactive"), ("C4h", 4, 3, "inactive")):
        p, _, _ = batch(ctx, name, B, n, lim)
        p.X(), p.U(), p.K(), p.d(), p.fX()
        p.warm_start(2)
        p.close()
    ctx.close()
    print("hostsim: all sequences clean under ASan/UBSan")


if __name__ == "__main__":
    main()
