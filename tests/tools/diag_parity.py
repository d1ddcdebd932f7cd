"""[test tooling: compares the GPU path with the oracle; lives under tests/ because only tests may use oracle/]
Diagnostic: per-instance comparison of the HIP path with the oracle on a seeded batch (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ilqr_planner_amd import capi, workloads
from tests.helpers import oracle_solve_instance, panda_segs

name, B, nb_iter = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
limits = sys.argv[4] if len(sys.argv) > 4 else "inactive"
ctx = capi.Context(0)
cfg = workloads.config(name)
desc, inp = workloads.make_batch(ctx, cfg, B=B, limits=limits)
p = workloads.load_batch(ctx, desc, inp, B)
workloads.run_solver(p, cfg, nb_iter=nb_iter, early_stop=True)
cost, iters = p.cost(), p.iters()
ct, at = p.trace(nb_iter)
segs = panda_segs()
nd = 0
for i in range(B):
    r = oracle_solve_instance(cfg, inp, i, nb_iter, True, segs)
    rel = abs(cost[i] - r["cost"]) / max(abs(r["cost"]), 1e-12)
    same = np.array_equal(at[i, : r["iters"]], r["trace_alpha"])
    if not same or rel > 1e-6:
        nd += 1
        k = next((j for j in range(r["iters"]) if at[i, j] != r["trace_alpha"][j]), -1)
        print(f"inst {i}: rel {rel:.2e} same_path {same} first_diff_iter {k} gpu_cost {cost[i]:.6g} ref {r['cost']:.6g}")
        if k >= 0:
            print("   gpu alpha", at[i, max(0,k-1):k+3], "cost", ct[i, max(0,k-1):k+3])
            print("   ref alpha", r["trace_alpha"][max(0,k-1):k+3], "cost", r["trace_cost"][max(0,k-1):k+3])
print("differing:", nd, "of", B)
