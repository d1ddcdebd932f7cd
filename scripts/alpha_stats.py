"""How predictable is the line-search winner? (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_planner_amd import capi, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
ctx = capi.Context(0)
cfg = workloads.config(name)
B = 4096
desc, inp = workloads.make_batch(ctx, cfg, B=B)
p = workloads.load_batch(ctx, desc, inp, B)
psi = workloads.psi_of(cfg['psi'], p.T, p.dims.n_u) if cfg['solver'] == 'batch_cp' else None
n = workloads.run_solver(p, cfg, early_stop=False, psi=psi)
ct, at = p.trace(n)
idx = np.round(-np.log2(at)).astype(int)
print("mean trials", (idx + 1).mean(), " winner histogram", np.bincount(idx.reshape(-1), minlength=11) / idx.size)
for it in range(n):
    prev = idx[:, it - 1] if it > 0 else np.zeros(B, int)
    mis = (idx[:, it] != prev)
    wg = mis.reshape(-1, 16).any(1).mean()
    print(f"it {it:2d}: winner==alpha1 {np.mean(idx[:, it] == 0):.2f}  floor {np.mean(idx[:, it] == 10):.2f}  mispredict {mis.mean():.2f}  workgroups with a pending instance {wg:.2f}")
