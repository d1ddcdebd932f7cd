"""[test tooling: compares the GPU path with the oracle; lives under tests/ because only tests may use oracle/]
Per-iteration cost differences GPU (low-rank batch solve) vs oracle (dense), identity basis."""
import sys
import numpy as np
sys.path.insert(0, ".")
from ilqr_planner_amd import capi, workloads
from tests.helpers import oracle_system_of_instance, orc

ctx = capi.Context(0)
for cfg_name, T, limits, u0s in (("C4", 16, "inactive", 0.02), ("C1t", 20, "inactive", 0.01), ("C4t1", 24, "urdf", 0.02)):
    B, nb_iter = 12, 5
    cfg = dict(workloads.config(cfg_name), T=T)
    desc, inp = workloads.make_batch(ctx, cfg, B=B, limits=limits)
    rng = np.random.default_rng(3)
    inp["U0"] = inp["U0"] + u0s * rng.standard_normal(inp["U0"].shape)
    p = workloads.load_batch(ctx, desc, inp, B)
    p.solve_batch(nb_iter, False)
    ct, at = p.trace(nb_iter)
    U = p.U()
    print(cfg_name)
    for i in range(B):
        s = oracle_system_of_instance(cfg, inp, i)
        r = orc.solve_batch(s, inp["U0"][i].reshape(-1), nb_iter, False)
        rel = np.abs(ct[i] - r["trace_cost"]) / np.abs(r["trace_cost"])
        print(i, " ".join("%.1e" % v for v in rel), " alpha", at[i], r["trace_alpha"] if not np.array_equal(at[i], r["trace_alpha"]) else "", "dU %.1e" % np.abs(U[i].reshape(-1) - r["u"]).max())
    p.close()
