"""Per-instance parity proof of chosen instances of a bench batch, with the failing steps printed (development aid; uses the oracle like the tests).
usage: CFG=C4 IDX=12,57 python scripts/diag_proof.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ilqr_planner_amd import capi, workloads
from tests import parity_proof as pp
from tests.helpers import oracle_solve_instance, panda_segs
ctx = capi.Context(0)
cfg = workloads.config(os.environ.get("CFG", "C4"))
B = int(os.environ.get("B", cfg["B"]))
nb_iter = int(os.environ.get("IT", cfg["nb_iter"]))
idx = [int(i) for i in os.environ["IDX"].split(",")]
desc, inp = workloads.make_batch(ctx, cfg, B=B)
p = workloads.load_batch(ctx, desc, inp, B)
workloads.run_solver(p, cfg, nb_iter=nb_iter, early_stop=False)
segs = panda_segs()
summ, rel, failures = pp.check_batch(p, cfg, inp, nb_iter, False, workloads.run_solver, lambda i: oracle_solve_instance(cfg, inp, i, nb_iter, False, segs), always=tuple(idx), indices=idx)
print(summ)
for f in failures:
    print("instance", f["i"], "rel", f["rel"])
    for st in f["steps"]:
        print("   ", json.dumps({k: (v if not isinstance(v, float) else float("%.6g" % v)) for k, v in st.items()}))
