"""Whole-solve wall time of a configuration with the two-stream split off / where measured to pay / forced (ilqr_ctx_set_split 0 / 1 / 2).
usage: CFG=C3 B=4096 python scripts/time_split.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_planner_amd import capi, workloads
ctx = capi.Context(0)
cfg = workloads.config(os.environ.get("CFG", "C3"))
for B in [int(b) for b in os.environ.get("B", str(cfg["B"])).split(",")]:
    desc, inp = workloads.make_batch(ctx, cfg, B=B)
    p = workloads.load_batch(ctx, desc, inp, B)
    out = {}
    costs = {}
    for mode in (0, 1, 2, 0, 2):
        ctx.set_split(mode)
        ts = []
        for rep in range(6):
            if cfg["solver"] == "al": p.reset_multipliers()
            ctx.synchronize()
            t0 = time.perf_counter()
            workloads.run_solver(p, cfg, early_stop=False)
            ctx.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        out.setdefault(mode, []).append(round(min(ts), 3))
        costs[mode] = p.cost().copy()
    same = {m: bool(np.array_equal(costs[0], costs[m], equal_nan=True)) for m in costs}
    print(B, "ms per solve by split mode:", out, "costs equal to mode 0:", same, flush=True)
    p.close()
