"""A/B of library variants selected by environment switches, interleaved in ONE process: per-category HIP-event averages (us per launch)
of the C3 / C2 solves.  usage: CFG=C3 B=4096 VARIANTS="name:ENV=1,name2:" python scripts/ab_env.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ilqr_planner_amd import capi, workloads
ctx = capi.Context(0)
cfg = workloads.config(os.environ.get("CFG", "C3"))
variants = [v.split(":") for v in os.environ.get("VARIANTS", "default:").split(",")]
for B in [int(b) for b in os.environ.get("B", str(cfg["B"])).split(",")]:
    desc, inp = workloads.make_batch(ctx, cfg, B=B)
    p = workloads.load_batch(ctx, desc, inp, B)
    res = {n: [] for n, _ in variants}
    for rnd in range(int(os.environ.get("ROUNDS", "3"))):
        for name, env in variants:
            for kv in filter(None, env.split(";")):
                k, v = kv.split("=")
                os.environ[k] = v
            for rep in range(3):
                if rep == 1:
                    ctx.profile_reset(); ctx.profile(True)
                if cfg["solver"] == "al": p.reset_multipliers()
                workloads.run_solver(p, cfg, nb_iter=10, early_stop=False)
                ctx.synchronize()
            ctx.profile(False)
            cats = [int(c) for c in os.environ.get("CATS", "1").split(",")]  # profile categories (capi.PROF_*): 1 = backward sweep, 2 = forward, 4 = other
            vals = []
            for cat in cats:
                ms, k = ctx.profile_get(cat)
                vals.append(round(ms / max(k, 1) * 1e3, 1))
            res[name].append(vals[0] if len(vals) == 1 else vals)
            for kv in filter(None, env.split(";")):
                os.environ.pop(kv.split("=")[0], None)
    print(B, json.dumps(res), "cost", float(p.cost()[:8].sum()), flush=True)
    p.close()
