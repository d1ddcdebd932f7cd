"""Whole-solve wall time (profiling off, synchronise before and after) of library variants selected by environment switches, interleaved in one
process.  usage: CFG=C2 B=256 VARIANTS="wg:ILQR_FWD=wg,dpp:ILQR_FWD=dpp" python scripts/time_solve.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_planner_amd import capi, workloads
ctx = capi.Context(0)
cfg = workloads.config(os.environ.get("CFG", "C3"))
variants = [v.split(":") for v in os.environ.get("VARIANTS", "default:").split(",")]
psi = workloads.psi_of(cfg["psi"], cfg["T"], 7) if cfg["solver"] == "batch_cp" else None
for B in [int(b) for b in os.environ.get("B", str(cfg["B"])).split(",")]:
    desc, inp = workloads.make_batch(ctx, cfg, B=B)
    p = workloads.load_batch(ctx, desc, inp, B)
    out, costs = {}, {}
    for rnd in range(int(os.environ.get("ROUNDS", "3"))):
        for name, env in variants:
            for kv in filter(None, env.split(";")):
                k, v = kv.split("=")
                os.environ[k] = v
            ts = []
            for rep in range(6):
                if cfg["solver"] == "al": p.reset_multipliers()
                ctx.synchronize()
                t0 = time.perf_counter()
                workloads.run_solver(p, cfg, early_stop=False, psi=psi)
                ctx.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            out.setdefault(name, []).append(round(min(ts), 3))
            costs[name] = p.cost().copy()
            for kv in filter(None, env.split(";")):
                os.environ.pop(kv.split("=")[0], None)
    ref = costs[variants[0][0]]
    dev = {n: float(np.nanmax(np.abs(c - ref) / np.maximum(np.abs(ref), 1e-300))) for n, c in costs.items()}
    print(B, "ms per solve:", out, "max rel cost deviation from the first variant:", dev, flush=True)
    p.close()
