"""Per-instance parity proof for the Riccati solvers (ILQRRecursive / AL_ILQR) -- test infrastructure, uses the oracle.

Why a final-cost comparison alone cannot be the gate.  The reference's iteration is a DISCONTINUOUS and, far from convergence,
strongly expanding map: the line search accepts the first step size whose cost is below the previous one (ILQRRecursive.cpp:155) and
accepts the last one anyway; AL_ILQR masks a row when `g < 0 && lambda == 0` (AL-ILQR.cpp:38-42) and clamps the multipliers at zero
(:205).  Two correct implementations with different rounding (FMA contraction, another libm, an algebraically equivalent sweep) can
therefore part ways on some instances and end at different costs.  What CAN be checked, instance by instance and iteration by
iteration, is that every iteration the GPU made is the reference's iteration:

  for it = 0 .. iters-1:
      take the GPU's own state after `it` iterations  (U_it, and for AL the multipliers lambda_it and lambda_{it-1})
      run ONE oracle iteration from that state        (orc_set_resume: same iteration index, same penalties, same mask inputs)
      the oracle's (cost, alpha) must equal the GPU's trace entry `it`:   same alpha and |dcost| <= STEP_RTOL * |cost|
      -- or the oracle's own decision at that iteration must be a TIE: the line-search trial at which the two part has
         |newCost - cost0| <= TIE_RTOL * |cost0| (the comparison is decided by rounding), or an active-set / clamp / limit test
         sits within MASK_ATOL of its threshold (orc_probe_rec).

An instance whose every iteration passes is PROVEN: its GPU trajectory is a chain of reference iterations (up to rounding-level
ties), and any distance between its final cost and the oracle's own end-to-end run is the reference map's own sensitivity.  An
iteration that fails both tests is a kernel bug.  The GPU states come from deterministic re-runs with nb_iter = it (the kernels use
no atomics: a solve with fewer iterations reproduces the prefix of a longer one bit for bit, which the proof also checks through the
cost the oracle's re-rollout of U_it must reproduce).
"""
from __future__ import annotations

import numpy as np

from tests.helpers import oracle_system_of_instance, orc, panda_segs

STEP_RTOL = 1e-9    # one iteration from the same state: relative cost agreement (measured on MI355X over 2e4 steps of every system
                    # shape and both kernel sets: median 1e-13, p99 2e-11, max 5e-10 -- profiles/r02_parity_probe.json)
STEP_RTOL_ILL = 1e-3  # ... up to here if the oracle's own rounding sensitivity at that step explains it: algebraically neutral variants of
                    # its sweep (Qxu := Qux^T; Quu inverted by the pivot-free symmetric sweep operator instead of partial-pivot LU -- both
                    # equal in exact arithmetic, orc_set_variant bits 0 and 1) move the step's cost by at least 1/100 of the GPU's
                    # deviation.  Time-system instances reach cond(Quu) ~ 1e9: one iteration then amplifies 1e-16 to 1e-7 .. 1e-5.
COST0_RTOL = 1e-9   # the oracle's re-rollout of the GPU's controls must reproduce the GPU's accepted cost (measured max 9e-11)
TIE_RTOL = 1e-9     # a line-search comparison newCost < cost0 is a tie when |newCost - cost0| / |cost0| is below the step agreement
MASK_ATOL = 1e-12   # an active-set (g < 0 with lambda == 0), clamp (lambda + penalty g > 0) or limit (x > max) test is a tie within
                    # this distance of its threshold: 4500 eps -- the measured ties sit at |g| <= 3e-15, where AL_ILQR rides the bound
                    # (lambda = 0, g = +-1 ulp over a run of timesteps: the penalty is switched by the sign of rounding noise)
STOP_RTOL = 1e-9    # early-stop tests alpha sqrt(sum ||du||) < 1e-3 and cost < 1e-3: relative distance to the threshold


def gpu_states(p, cfg, nb_iter, early_stop, run_solver):
    """U (and lambda) of the whole batch after 0 .. nb_iter-1 iterations, from re-runs with fewer iterations."""
    out = []
    al = cfg["solver"] == "al"
    for it in range(nb_iter):
        if al:
            p.reset_multipliers()
        run_solver(p, cfg, nb_iter=it, early_stop=early_stop)
        out.append(dict(U=p.U(), lam=p.lam() if al else None))
    return out


def _u_oracle(cfg, inp, U):
    if cfg["kind"] in (2, 3) and inp.get("dof", 7) < 7:  # joint-space batches are padded to 7 joints on the device only
        n = inp["dof"]
        U = np.hstack([U[:, :n], U[:, 7:]])
    return U.reshape(-1)


def one_step(cfg, inp, i, it, states, segs=None, sysm=None):
    """One oracle iteration (index `it`) of instance i from the GPU's state after `it` iterations.  Returns the oracle result."""
    s = sysm or oracle_system_of_instance(cfg, inp, i, segs)
    U = _u_oracle(cfg, inp, states[it]["U"][i])
    if cfg["solver"] == "recursive":
        return orc.solve_recursive(s, U, 1, True, False, probe=True, resume=dict(it0=it) if it else None)
    al = cfg["al"]
    pen = al["penalty"] * al["scaling"] ** (it // al["lag"])          # penalty in force during iteration `it`
    pen_in = al["penalty"] * al["scaling"] ** ((it - 1) // al["lag"]) if it else pen  # ... when the incoming trajectory was rolled out
    res = dict(it0=it, init_penalty=pen_in, lambda_mask=states[it - 1]["lam"][i]) if it else None
    return orc.solve_al(s, inp["A"], inp["b"], states[it]["lam"][i], U, 1, al["lag"], pen, al["scaling"], True, False, probe=True, resume=res)


def prove_instance(cfg, inp, i, states, ct, at, iters, segs=None, nb_iter=None, early_stop=False):
    """Classifies instance i.  ct, at: the GPU's cost / alpha traces [B][nb_iter]; iters: iterations the GPU ran.
    Returns dict(verdict = 'stepwise' | 'tie' | 'unexplained', steps = [...]) -- 'stepwise': every iteration reproduced;
    'tie': every iteration reproduced or decided by a rounding-level tie in the oracle's own decision.
    With early_stop the stop / go-on decision after every iteration is checked the same way (nb_iter = the solve's iteration cap)."""
    segs = segs or panda_segs()
    s = oracle_system_of_instance(cfg, inp, i, segs)
    steps, verdict = [], "stepwise"
    n = int(iters[i])
    nb_iter = int(nb_iter if nb_iter is not None else ct.shape[1])
    for it in range(n):
        r = one_step(cfg, inp, i, it, states, segs, s)
        pr = r["probe"][0]
        cg, ag = float(ct[i, it]), float(at[i, it])
        co, ao = float(r["trace_cost"][0]), float(r["trace_alpha"][0])
        st = dict(it=it, alpha_gpu=ag, alpha_orc=ao, cost_gpu=cg, cost_orc=co)
        # the state handed over is the GPU's accepted trajectory of the previous iteration: its cost must be the GPU's previous trace entry
        if it > 0 and np.isfinite(ct[i, it - 1]):
            st["cost0_rel"] = abs(pr["cost0"] - ct[i, it - 1]) / max(abs(ct[i, it - 1]), 1e-300)
        nan_both = (not np.isfinite(cg)) and (not np.isfinite(co))
        rel = 0.0 if nan_both else (abs(cg - co) / max(abs(co), 1e-300) if np.isfinite(cg) and np.isfinite(co) else np.inf)
        st["rel"] = rel
        margins = dict(mask_in=pr["mask_margin_in"], limit_in=pr["limit_margin_in"])  # what this iteration's sweep switches on
        ill = False
        if ag == ao and STEP_RTOL < rel <= STEP_RTOL_ILL:  # an ill-conditioned sweep?  ask the oracle how much its own rounding moves this step
            sens = 0.0
            for var in (1, 2, 3):
                orc.set_variant(var)
                try:
                    rv = one_step(cfg, inp, i, it, states, segs, s)
                finally:
                    orc.set_variant(0)
                cv = float(rv["trace_cost"][0])
                if float(rv["trace_alpha"][0]) != ao or not np.isfinite(cv):
                    sens = np.inf  # the variant even changes the accepted step size
                else:
                    sens = max(sens, abs(cv - co) / max(abs(co), 1e-300))
            st["variant_rel"] = sens
            ill = sens >= rel / 100
        if ag == ao and (rel <= STEP_RTOL or ill) and st.get("cost0_rel", 0.0) <= COST0_RTOL:
            st["how"] = "same" if not ill else "same:ill-conditioned"
        else:
            how = None
            if ag != ao:  # the trial at which they part: the larger of the two step sizes (one accepted it, the other went on halving)
                a_hi = max(ag, ao)
                t = [k for k, a in enumerate(pr["alpha"]) if a == a_hi]
                if t:
                    c_t = pr["cost"][t[0]]
                    st["tie_margin"] = abs(c_t - pr["cost0"]) / max(abs(pr["cost0"]), 1e-300) if np.isfinite(c_t) else np.inf
                    if st["tie_margin"] <= TIE_RTOL:
                        how = "tie:line-search"
            if how is None and min(margins.values()) <= MASK_ATOL:
                how = "tie:" + min(margins, key=margins.get)
            st["how"] = how or "FAIL"
            st["margins"] = margins
            verdict = "unexplained" if how is None else ("tie" if verdict != "unexplained" else verdict)
        if early_stop and st["how"] != "FAIL" and it < nb_iter - 1:  # (after the last allowed iteration the decision leaves no trace)
            # the stop decision taken on this iteration's result (ILQRRecursive.cpp:174-176, AL-ILQR.cpp:225)
            crit = ao * np.sqrt(pr["dun"])
            stop_o = crit < 1e-3 and (cfg["solver"] == "al" or co < 1e-3)
            stop_g = (it == n - 1) and (n < nb_iter)
            if stop_o != stop_g and ag == ao:
                near = abs(crit - 1e-3) <= STOP_RTOL * 1e-3 or (cfg["solver"] != "al" and abs(co - 1e-3) <= STOP_RTOL * 1e-3)
                st["stop"] = "tie:early-stop" if near else "FAIL"
                st["stop_crit"] = float(crit)
                verdict = "unexplained" if not near else ("tie" if verdict != "unexplained" else verdict)
        steps.append(st)
    return dict(verdict=verdict, steps=steps)


def check_batch(p, cfg, inp, nb_iter, early_stop, run_solver, oracle_solve, always=(0, 1, 2, 3), rtol=1e-4, indices=None):
    """The parity gate of a solved batch `p` (already solved with nb_iter / early_stop): every instance is within `rtol` of the oracle's
    own end-to-end run, or is PROVEN (see the module header).  The instances in `always` are proven whatever their distance.  Returns
    (summary, rel, failures) -- failures lists the unexplained instances with their failing steps; the caller asserts it is empty.
    indices: check only these instances (a sample of a big batch); rel is then indexed like the batch, zero elsewhere."""
    cost, iters = p.cost(), p.iters()
    ct, at = p.trace(nb_iter)
    B = len(cost)
    segs = panda_segs()
    rel, flagged = np.zeros(B), []
    todo = list(range(B)) if indices is None else [int(i) for i in indices]
    for i in todo:
        r = oracle_solve(i)
        fo, fg = np.isfinite(r["cost"]), np.isfinite(cost[i])
        if fo and fg:
            rel[i] = abs(cost[i] - r["cost"]) / max(abs(r["cost"]), 1e-12)
        elif fo != fg:
            rel[i] = np.inf  # NaN on one side only: never excused, must be proven step by step
        if rel[i] > rtol or i in always:
            flagged.append(i)
    states = gpu_states(p, cfg, nb_iter, early_stop, run_solver) if flagged else None
    results, failures = [], []
    proofs = {i: prove_instance(cfg, inp, i, states, ct, at, iters, segs, nb_iter, early_stop) for i in flagged}
    for i in todo:
        results.append((rel[i] <= rtol, proofs.get(i)))
        pf = proofs.get(i)
        if pf and pf["verdict"] == "unexplained":
            failures.append(dict(i=i, rel=float(rel[i]), steps=[st for st in pf["steps"] if st["how"] == "FAIL" or st.get("stop") == "FAIL"]))
    summ = summarize(results)
    summ["n_proven_always"] = sum(1 for i in always if i in proofs and proofs[i]["verdict"] != "unexplained")
    return summ, rel, failures


def summarize(results):
    """Fractions over a list of (within_1e4: bool, proof or None) pairs."""
    n = len(results)
    within = sum(1 for w, _ in results if w)
    tie = sum(1 for w, pf in results if not w and pf and pf["verdict"] == "tie")
    stepwise = sum(1 for w, pf in results if not w and pf and pf["verdict"] == "stepwise")
    unexpl = n - within - tie - stepwise
    return dict(n=n, frac_within_1e4=within / n, frac_proven_tie=tie / n, frac_proven_stepwise=stepwise / n, frac_unexplained=unexpl / n)
