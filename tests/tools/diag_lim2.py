"""[test tooling] second limit set: initial cost and first iterations, GPU vs oracle."""
import sys
import numpy as np
sys.path.insert(0, ".")
from ilqr_planner_amd import capi, workloads
from tests.helpers import oracle_solve_instance, panda_segs

ctx = capi.Context(0)
cfg = workloads.config("C2hl")
B = 6
desc, inp = workloads.make_batch(ctx, cfg, B=B, limits="urdf")
print("desc limits2", desc.limits2_set, desc.is_sequence, desc.limit_multiplicity, desc.limit_multiplicity2, list(desc.state_max2)[:8], list(desc.state_min2)[:8], list(desc.limit_weight2)[:8])
segs = panda_segs()
for n in (0, 1, 2):
    p = workloads.load_batch(ctx, desc, inp, B)
    p.solve_recursive(n, True, False)
    c = p.cost()
    p.close()
    for i in range(3):
        r = oracle_solve_instance(cfg, inp, i, n, False, segs)
        print(n, i, c[i], r["cost"], abs(c[i] - r["cost"]) / abs(r["cost"]))
