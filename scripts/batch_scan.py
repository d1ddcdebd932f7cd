"""Kernel times of one solve against the batch size (CFG=C3|C4 BATCHES=512,1024,...): which kernels are chain-bound (flat), which issue-bound (linear).
Per-category HIP-event averages in microseconds per launch; the data behind DESIGN.md 5.3 (profiles/r02_batch_scan.txt)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ilqr_planner_amd import capi, workloads
ctx = capi.Context(0)
for B in [int(b) for b in os.environ.get("BATCHES", "512,1024,2048,4096,8192,16384").split(",")]:
    cfg = workloads.config(os.environ.get("CFG", "C3"))
    desc, inp = workloads.make_batch(ctx, cfg, B=B)
    p = workloads.load_batch(ctx, desc, inp, B)
    for rep in range(3):
        if rep == 1:
            ctx.profile_reset(); ctx.profile(True)
        if cfg["solver"] == "al": p.reset_multipliers()
        workloads.run_solver(p, cfg, nb_iter=10, early_stop=False)
        ctx.synchronize()
    ctx.profile(False)
    out = {}
    for n, w in (("rollout", 0), ("backward", 1), ("forward", 2), ("other", 3), ("apply", 4)):
        ms, k = ctx.profile_get(w)
        out[n] = round(ms / k * 1e3, 1) if k else None
    print(B, json.dumps(out), flush=True)
    p.close()
