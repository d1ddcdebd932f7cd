"""Data collection for the per-instance parity proof (tests/parity_proof.py): runs seeded batches on the GPU on both kernel sets,
replays every GPU iteration with the oracle from the GPU's own state and writes the distributions to gpurun_out/parity_probe.json
(step agreement, tie margins, verdict counts).  Usage: python tests/tools/parity_probe.py [cfg:B:iters ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import numpy as np  # noqa: E402

from tests import parity_proof as pp  # noqa: E402
from tests.helpers import oracle_solve_instance, panda_segs  # noqa: E402


def run(ctx, name, B, nb_iter, path, limits="inactive", early_stop=False):
    from ilqr_planner_amd import workloads

    os.environ["ILQR_HIP_PATH"] = path
    cfg = workloads.config(name)
    desc, inp = workloads.make_batch(ctx, cfg, B=B, limits=limits)
    p = workloads.load_batch(ctx, desc, inp, B)
    workloads.run_solver(p, cfg, nb_iter=nb_iter, early_stop=early_stop)
    cost, iters = p.cost(), p.iters()
    ct, at = p.trace(nb_iter)
    states = pp.gpu_states(p, cfg, nb_iter, early_stop, workloads.run_solver)
    p.close()
    segs = panda_segs()
    t0 = time.time()
    res, rels, steps_rel, ties, fails, c0 = [], [], [], [], [], []
    for i in range(B):
        r = oracle_solve_instance(cfg, inp, i, nb_iter, early_stop, segs)
        if not (np.isfinite(r["cost"]) and np.isfinite(cost[i])):
            rel = 0.0 if (not np.isfinite(r["cost"]) and not np.isfinite(cost[i])) else np.inf
        else:
            rel = abs(cost[i] - r["cost"]) / max(abs(r["cost"]), 1e-12)
        rels.append(rel)
        pf = pp.prove_instance(cfg, inp, i, states, ct, at, iters, segs)
        res.append((rel <= 1e-4, pf))
        for st in pf["steps"]:
            if st["how"] == "same":
                steps_rel.append(st["rel"])
            elif st["how"].startswith("tie"):
                td = dict(i=i, it=st["it"], how=st["how"], tie_margin=st.get("tie_margin"), ag=st["alpha_gpu"], ao=st["alpha_orc"], rel=st["rel"],
                          margins=st.get("margins"))
                if st["how"] == "tie:mask_in" and cfg["solver"] == "al" and len(ties) < 6:  # diagnosis: the rows behind the margin
                    it = st["it"]
                    r0 = pp.one_step(cfg, inp, i, it, states, segs)  # X of the incoming trajectory = rollout of U_it
                    s_ = __import__("tests.helpers", fromlist=["x"]).oracle_system_of_instance(cfg, inp, i, segs)
                    X0 = __import__("tests.helpers", fromlist=["x"]).orc.solve_al(s_, inp["A"], inp["b"], states[it]["lam"][i], states[it]["U"][i].reshape(-1), 0,
                                                                                     cfg["al"]["lag"], 0.25, 1.1, True, False)["X"]
                    g = X0[:-1] @ inp["A"][0][: X0.shape[1]] - inp["b"][0]
                    lp, ln = states[it - 1]["lam"][i][:, 0], states[it]["lam"][i][:, 0]
                    k = int(np.argmin(np.where(lp == 0, np.abs(g), np.inf)))
                    td["diag"] = dict(k=k, g=g[max(0, k - 3): k + 4].tolist(), lam_prev=lp[max(0, k - 3): k + 4].tolist(), lam=ln[max(0, k - 3): k + 4].tolist(),
                                      n_lam0=int((lp == 0).sum()))
                ties.append(td)
            else:
                fails.append(dict(i=i, it=st["it"], ag=st["alpha_gpu"], ao=st["alpha_orc"], rel=st["rel"], tie_margin=st.get("tie_margin"),
                                  cost0_rel=st.get("cost0_rel"), margins=st.get("margins"), cg=st["cost_gpu"], co=st["cost_orc"]))
            if "cost0_rel" in st:
                c0.append(st["cost0_rel"])
    rels = np.asarray(rels)
    sr = np.asarray(steps_rel) if steps_rel else np.zeros(1)
    out = dict(cfg=name, B=B, nb_iter=nb_iter, path=path, limits=limits, summary=pp.summarize(res),
               final_rel=dict(median=float(np.median(rels)), p90=float(np.quantile(rels, .9)), max=float(rels.max())),
               step_rel=dict(n=len(steps_rel), median=float(np.median(sr)), p99=float(np.quantile(sr, .99)), max=float(sr.max())),
               cost0_rel_max=float(max(c0) if c0 else 0.0), n_ties=len(ties), n_fails=len(fails), ties=ties[:40], fails=fails[:60],
               oracle_s=round(time.time() - t0, 1))
    print(json.dumps({k: out[k] for k in ("cfg", "path", "limits", "summary", "final_rel", "step_rel", "cost0_rel_max", "n_ties", "n_fails")}), flush=True)
    return out


def main():
    from ilqr_planner_amd import capi

    specs = sys.argv[1:] or ["C3:256:20", "C3r:128:12", "C2:128:20", "C2nd:48:10", "C4t1:48:12", "C4:32:6"]
    ctx = capi.Context(0)
    outs = []
    for sp in specs:
        parts = sp.split(":")
        name, B, n = parts[0], int(parts[1]), int(parts[2])
        lim = parts[3] if len(parts) > 3 else "inactive"
        for path in ("v2", "v1"):
            outs.append(run(ctx, name, B, n, path, lim))
    ctx.close()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(outs, open("gpurun_out/parity_probe.json", "w"), indent=1)


if __name__ == "__main__":
    main()
