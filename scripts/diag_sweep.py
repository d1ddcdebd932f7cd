"""Compare the gains of one iteration between sweep variants selected by environment switches (development aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ilqr_planner_amd import capi, workloads
ctx = capi.Context(0)
cfg = workloads.config(os.environ.get("CFG", "C2"))
B = int(os.environ.get("B", "8"))
desc, inp = workloads.make_batch(ctx, cfg, B=B, seed=11)
out = {}
for name, env in (("coop", "ILQR_SWEEP_COOP"), ("lpi16", "ILQR_SWEEP_LPI16"), ("lpi8", None)):
    if env: os.environ[env] = "1"
    p = workloads.load_batch(ctx, desc, inp, B)
    workloads.run_solver(p, cfg, nb_iter=int(os.environ.get("IT", "1")), early_stop=False)
    out[name] = (p.K(), p.d(), p.cost(), p.alpha())
    p.close()
    if env: os.environ.pop(env)
K0, d0, c0, a0 = out["coop"]
np.set_printoptions(precision=3, linewidth=200)
for name in ("lpi16", "lpi8"):
    K, d, c, a = out[name]
    dK = np.abs(K - K0) / (np.abs(K0).max() + 1e-300)
    dd = np.abs(d - d0) / (np.abs(d0).max() + 1e-300)
    print(name, "max rel dK", dK.max(), "max rel dd", dd.max(), "cost", c[:4], "vs", c0[:4], "alpha", a[:4], a0[:4])
    if dK.max() > 1e-8:
        i, k = np.unravel_index(np.argmax(dK.reshape(B, -1).max(1)), (B,))[0], None
        per_t = dK[i].reshape(dK.shape[1], -1).max(1)
        print("  instance", i, "first bad t from the end:", [int(t) for t in np.where(per_t > 1e-8)[0][-5:]], "of", len(per_t))
        t = int(np.where(per_t > 1e-8)[0][-1])
        print("  K coop\n", K0[i, t], "\n  K", name, "\n", K[i, t], "\n  d coop", d0[i, t], "\n  d", name, d[i, t])
