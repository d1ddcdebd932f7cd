"""Create / solve / destroy many problems of every solver and check that device memory comes back (no leak) and results repeat."""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
from ilqr_planner_amd import capi, workloads

sys.path.insert(0, "ilqr_planner_amd/pylqr")
from PyLQR.utils import primitives

ctx = capi.Context(0)
free0 = torch.cuda.mem_get_info()[0]
ref = {}
for rep in range(40):
    for name, T, solver in (("C3", 60, "al"), ("C4", 40, "rec"), ("C2", 50, "cp"), ("C2", 30, "batch"), ("C4t1", 30, "batch"), ("C1j", 40, "rec")):
        cfg = dict(workloads.config(name), T=T)
        B = 96
        desc, inp = workloads.make_batch(ctx, cfg, B=B)
        p = workloads.load_batch(ctx, desc, inp, B)
        if solver == "al":
            al = cfg["al"]
            p.solve_al(6, al["lag"], al["penalty"], al["scaling"], True, False)
        elif solver == "rec":
            p.solve_recursive(6, True, False)
        elif solver == "cp":
            p.solve_batch_cp(np.kron(np.asarray(primitives.build_psi_unitstep(T - 1, 2)), np.eye(p.dims.n_u)), 5, False)
        else:
            p.solve_batch(4, False)
        c = p.cost()
        key = (name, T, solver)
        if key in ref:
            assert np.array_equal(np.nan_to_num(c), np.nan_to_num(ref[key])), key
        ref[key] = c
        p.close()
    if rep % 10 == 9:
        ctx.synchronize()
        print(rep + 1, "cycles, free delta MB", (free0 - torch.cuda.mem_get_info()[0]) / 2**20, flush=True)
ctx.close()
print("ok")
