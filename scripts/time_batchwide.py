"""Wall time of the wide-basis batch solvers (BatchILQR / BatchILQRCP with Kw > 16) at batch sizes of the BASELINE configs."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from ilqr_planner_amd import capi, workloads
sys.path.insert(0, "ilqr_planner_amd/pylqr")
from PyLQR.utils import primitives  # the product's own basis builders (no oracle outside tests/)

ctx = capi.Context(0)


def run(label, cfg, B, nb_iter, psi=None):
    desc, inp = workloads.make_batch(ctx, cfg, B=B)
    p = workloads.load_batch(ctx, desc, inp, B)
    go = (lambda: p.solve_batch(nb_iter, False)) if psi is None else (lambda: p.solve_batch_cp(psi, nb_iter, False))
    go()
    ctx.synchronize()
    ts = []
    for _ in range(3):
        p.set_controls(inp["U0"])
        ctx.synchronize()
        t = time.perf_counter()
        go()
        ctx.synchronize()
        ts.append(time.perf_counter() - t)
    c = p.cost()
    print(f"{label}: B={B} T={cfg['T']} iters={nb_iter}  {min(ts)*1e3:.2f} ms/solve  ({B*nb_iter/min(ts)/1e6:.2f} M problem-iterations/s)  median cost {np.median(c):.3e}", flush=True)
    p.close()


run("BatchILQR PosOrn-1 (tutorial shape, 693 controls)", dict(workloads.config("C2"), T=100), 4096, 10)
run("BatchILQR PosOrn-1 C5 shape (2793 controls)", workloads.config("C5"), 8192, 10)
run("BatchILQR PosOrn-2 (T=100)", dict(workloads.config("C2nd"), T=100), 4096, 10)
run("BatchILQR PosOrnTime-1 (792 controls)", dict(workloads.config("C4t1"), T=100), 4096, 10)
run("BatchILQR PosOrnTime-2 (T=50, 392 controls)", dict(workloads.config("C4"), T=50), 4096, 10)
cfg = workloads.config("C5")
psi = np.kron(np.asarray(primitives.build_psi_RBF(cfg["T"] - 1, 32)), np.eye(7))
run("BatchILQRCP rbf K=32 (Kw=224) C5 shape", cfg, 8192, 10, psi)
