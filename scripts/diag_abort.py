"""Bisect an abort seen when the v1 AL tutorial solve follows a large AL batch in the same process (children run one variant each)."""
import os
import subprocess
import sys

VARIANTS = {
    "base": dict(B1=4096, solver1="al", path2="v1"),
    "smallB": dict(B1=64, solver1="al", path2="v1"),
    "rec1": dict(B1=4096, solver1="recursive", path2="v1"),
    "v2second": dict(B1=4096, solver1="al", path2="v2"),
    "B1024": dict(B1=1024, solver1="al", path2="v1"),
    "only2": dict(B1=0, solver1="al", path2="v1"),
}


def child(name):
    sys.path.insert(0, ".")
    import numpy as np
    from ilqr_planner_amd import capi, workloads
    from tests.helpers import golden
    from tests.test_gpu_parity import _tutorial_problem

    v = VARIANTS[name]
    if v["B1"]:
        ctx = capi.Context(0)
        cfg = workloads.config("C3" if v["solver1"] == "al" else "C3r")
        desc, inp = workloads.make_batch(ctx, cfg, B=v["B1"])
        p = workloads.load_batch(ctx, desc, inp, v["B1"])
        workloads.run_solver(p, cfg, nb_iter=3, early_stop=False)
        p.cost()
        p.close()
        ctx.close()
    os.environ["ILQR_HIP_PATH"] = v["path2"]
    ctx = capi.Context(0)
    case = golden()["cases"]["POS_ORN_SYS_AL_ILQR"]
    sv = case["solves"][1]
    p = _tutorial_problem(ctx, case, 3)
    m = sv["m"]
    A, b = np.zeros((m, p.dims.n_x + p.dims.n_u)), np.zeros(m)
    for i, j, val in sv["A_nonzero"]:
        A[i, j] = val
    for i, val in sv["b_nonzero"]:
        b[i] = val
    p.set_constraints(A, b, np.tile(b, (3, p.T - 1, 1)))
    print(name, "solving", flush=True)
    p.solve_al(int(os.environ.get("NIT", "5")), sv["lag_update_step"], sv["penalty"], sv["scaling_factor"], True, True)
    print(name, "iters", p.iters(), flush=True)
    p.close()
    ctx.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for n in VARIANTS:
            r = subprocess.run([sys.executable, __file__, n], capture_output=True, text=True)
            print(f"== {n}: rc={r.returncode}", r.stdout.strip().replace("\n", " | "), (r.stderr.strip().splitlines() or [""])[0][:200], flush=True)
