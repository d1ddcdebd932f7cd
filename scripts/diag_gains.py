"""Gains of the first iteration: default HIP path vs ILQR_HIP_PATH=v1 on a seeded batch (per-step max relative difference)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_planner_amd import capi, workloads

name, B, nb_iter = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ctx = capi.Context(0)
cfg = workloads.config(name)
desc, inp = workloads.make_batch(ctx, cfg, B=B)
res = {}
for path in ("v2", "v1"):
    os.environ["ILQR_HIP_PATH"] = path
    p = workloads.load_batch(ctx, desc, inp, B)
    workloads.run_solver(p, cfg, nb_iter=nb_iter, early_stop=False)
    res[path] = (p.K(), p.d(), p.cost(), p.alpha() if hasattr(p, "alpha") else None)
    p.close()
K2, d2, c2, a2 = res["v2"]
K1, d1, c1, a1 = res["v1"]
print("cost rel diff max", np.nanmax(np.abs(c2 - c1) / np.abs(c1)))
sc = np.max(np.abs(K1), axis=(2, 3), keepdims=True) + 1e-300
dK = np.abs(K2 - K1) / sc
print("K: max rel diff (per instance-step scale)", np.nanmax(dK))
i, k = np.unravel_index(np.nanargmax(dK.max(axis=(2, 3))), dK.shape[:2])
print("worst at instance", i, "step", k)
np.set_printoptions(precision=4, linewidth=220)
print("per-step max over instances (every 10th):", dK.max(axis=(0, 2, 3))[::10])
r, c = np.unravel_index(np.nanargmax(dK[i, k]), dK[i, k].shape)
print("entry", r, c, K2[i, k, r, c], K1[i, k, r, c])
print("row-wise max rel diff at worst:", dK[i, k].max(axis=1))
print("col-wise max rel diff at worst:", dK[i, k].max(axis=0))
