"""[test tooling: compares the GPU path with the oracle; lives under tests/ because only tests may use oracle/]
Cost trace of one tutorial case at full precision: HIP path (whatever ILQR_HIP_PATH / ILQR_BWD select) next to the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))        # tests/
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))  # repo root
from helpers import golden
import test_gpu_parity as tg
from ilqr_planner_amd import capi

name = sys.argv[1] if len(sys.argv) > 1 else "POS_ORN_TIME_SYS"
idx = int(sys.argv[2]) if len(sys.argv) > 2 else 0
case = golden()["cases"][name]
sv = case["solves"][idx]
ctx = capi.Context(0)
p = tg._tutorial_problem(ctx, case, 1)
p.solve_recursive(sv["nb_iter"], sv["line_search"], sv["early_stop"])
ct, at = p.trace(sv["nb_iter"])
for i, (c_ref, a_ref) in enumerate(sv["trace"]):
    print(i + 1, repr(float(ct[0, i])), at[0, i], c_ref, a_ref)
