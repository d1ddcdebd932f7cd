"""Per-instance parity proof for the Riccati solvers (ILQRRecursive / AL_ILQR) and the batch solvers (BatchILQRCP / BatchILQR) --
test infrastructure, uses the oracle.

Why a final-cost comparison alone cannot be the gate.  The reference's iteration is a DISCONTINUOUS and, far from convergence,
strongly expanding map: the line search accepts the first step size whose cost is below the previous one (ILQRRecursive.cpp:155,
BatchILQRCP.cpp:152) and accepts the last one anyway; AL_ILQR masks a row when `g < 0 && lambda == 0` (AL-ILQR.cpp:38-42) and clamps
the multipliers at zero (:205).  Two correct implementations with different rounding (FMA contraction, another libm, an
algebraically equivalent sweep) can therefore part ways on some instances and end at different costs.  What CAN be checked, instance
by instance and iteration by iteration, is that every iteration the GPU made is the reference's iteration:

  for it = 0 .. iters-1:
      take the GPU's own state after `it` iterations  (U_it and its own rollout X_it of them; for AL the multipliers lambda_it, lambda_{it-1})
      run ONE oracle iteration from that state        (orc_set_resume / orc_set_resume_x: same iteration index, same penalties, the
                                                       active-set and limit tests taken on the very numbers the GPU's were)
      with every step size of the schedule evaluated  (orc_set_probe_all), and require
        (a) the oracle's cost of the state = the GPU's previous trace entry            (COST0_RTOL)
        (b) the GPU's decisions follow from the ORACLE's trial costs: every step size the GPU rejected has oracle cost >= cost0 (or
            NaN), the one it accepted has oracle cost < cost0 (or is the floor) -- a comparison may go the other way only if it is
            decided by rounding, |cost - cost0| <= TIE_RTOL |cost0|                    (constructive: the branch the GPU took is evaluated)
        (c) the oracle's cost AT THE GPU's step size = the GPU's trace entry            (STEP_RTOL; or, for an ill-conditioned sweep, a
            deviation of the size the oracle's own algebraically neutral variants move that very number: <= ILL_FACTOR x their maximum)
        (d) with early stop: the stop / go-on decision follows from the oracle's ||du|| the same way (STOP_RTOL).

An instance whose every iteration passes is PROVEN: its GPU trajectory is a chain of reference iterations, and any distance between its
final cost and the oracle's own end-to-end run is the reference map's own sensitivity.  An iteration that fails is a kernel bug.
Nothing is excused on a margin alone any more (round 2 accepted an active-set or limit test within 1e-12 of its threshold without
checking the other branch): with the GPU's own trajectory handed over, those tests see identical numbers on both sides.  The one
exception left is an AL row with more than one non-zero coefficient, whose value g = A [x; u] - b is a sum formed in another order on
the GPU: there a mask test within MASK_ATOL of zero is a tie, and it is counted.  The GPU states come from deterministic re-runs with
nb_iter = it (the kernels use no atomics: a solve with fewer iterations reproduces the prefix of a longer one bit for bit, which the
proof also checks through (a)).
"""
from __future__ import annotations

import numpy as np

from tests.helpers import oracle_system_of_instance, orc, panda_segs

STEP_RTOL = 1e-9    # one iteration from the same state: relative cost agreement (measured on MI355X over 2e4 steps of every system
                    # shape and both kernel sets: median 1e-13, p99 2e-11, max 5e-10 -- profiles/r02_parity_probe_c3.json)
STEP_RTOL_ILL = 1e-3  # ... up to here ONLY if the oracle's own rounding sensitivity at that step is of the same size: algebraically
                    # neutral variants of its arithmetic (orc_set_variant: Qxu := Qux^T; inverses by the pivot-free symmetric sweep
                    # operator instead of partial-pivot LU; products accumulated with fused multiply-adds; and their combinations) move
                    # the same number by `sens`, and the GPU's deviation is at most ILL_FACTOR x the largest of them.  Time-system
                    # instances reach cond(Quu) ~ 1e9: one iteration then amplifies 1e-16 to 1e-7 .. 1e-5.
VARIANTS = (1, 2, 3, 4, 5, 6, 7)
ILL_FACTOR = 10.0   # The GPU's arithmetic is one more neutral variant (another operation order, FMA contraction, a low-rank instead of a
                    # dense solve): its deviation is a draw from the same distribution as the measured ones, so it may exceed their
                    # maximum -- by a small factor, not by orders of magnitude (round 2 allowed 100 and did not count).  Measured on
                    # MI355X: a first try with 1 failed on a step where the GPU moved the cost by 5.978e-9 and the variants by 5.975e-9;
                    # with 4, one of 314 C4 instances failed at iteration 19 of a DIVERGING solve (cost 7.9e10, step at the floor:
                    # deviation 1.9e-7 against 2.7e-8).  The sensitivity sample is therefore widened by input perturbations (below), the
                    # factor is 10, and the number of such steps and the worst ratio are reported with every proof.
PERTURB = ((1, 1), (1, -1), (-1, 1), (-1, -1))  # sign patterns (even / odd entries) of a one-ulp relative perturbation of the controls handed
                    # to the oracle: what a backward-stable implementation may legitimately differ by is the exact result on inputs
                    # perturbed by a few ulps, i.e. condition number x eps -- this measures it directly
COST0_RTOL = 1e-9   # the oracle's cost of the GPU's state must reproduce the GPU's accepted cost (measured max 9e-11)
XDEV_TOL = 1e-9     # the GPU's trajectory must be the oracle's rollout of the GPU's controls: max |dx| / max(1, |x|)
TIE_RTOL = 1e-9     # a line-search comparison newCost < cost0 is a tie when |newCost - cost0| / |cost0| is below the step agreement
MASK_ATOL = 1e-12   # AL rows with several non-zero coefficients only (see the header)
STOP_RTOL = 1e-9    # early-stop tests alpha sqrt(sum ||du||) < 1e-3 and cost < 1e-3: relative distance to the threshold
ALPHA_FLOOR = 1e-3  # ILQRRecursive.cpp:155 / BatchILQRCP.cpp:152


def _unpad(cfg, inp, a):
    """Device arrays of joint-space batches are padded to 7 joints; the oracle works on the true number."""
    if cfg["kind"] in (2, 3) and inp.get("dof", 7) < 7:
        n = inp["dof"]
        a = np.concatenate([a[..., :n], a[..., 7:]], axis=-1)
    return np.ascontiguousarray(a)


def gpu_states(p, cfg, nb_iter, early_stop, run_solver, upto=None):
    """U, X (and lambda) of the whole batch after 0 .. nb_iter-1 iterations, from re-runs with fewer iterations (early stop off: a
    stopped instance's prefix is the same)."""
    out = []
    al = cfg["solver"] == "al"
    for it in range(nb_iter if upto is None else upto):
        if al:
            p.reset_multipliers()
        run_solver(p, cfg, nb_iter=it, early_stop=early_stop)
        out.append(dict(U=p.U(), X=p.X(), lam=p.lam() if al else None))
    return out


def _u_oracle(cfg, inp, U):
    return _unpad(cfg, inp, U).reshape(-1)


def one_step(cfg, inp, i, it, states, segs=None, sysm=None, probe="all", perturb=None):
    """One oracle iteration (index `it`) of instance i from the GPU's state after `it` iterations.  Returns the oracle result.
    perturb = (s_even, s_odd): the controls handed over are moved by one ulp (relative 2^-52) with these signs, and the oracle rolls them
    out itself (sensitivity measurement only)."""
    s = sysm or oracle_system_of_instance(cfg, inp, i, segs)
    U = _u_oracle(cfg, inp, states[it]["U"][i])
    X = _unpad(cfg, inp, states[it]["X"][i]) if states[it].get("X") is not None else None
    if perturb is not None:
        sg = np.where(np.arange(U.size) % 2 == 0, perturb[0], perturb[1])
        U = U * (1.0 + sg * 2.0 ** -52)
        X = None
    if cfg["solver"] == "recursive":
        return orc.solve_recursive(s, U, 1, True, False, probe=probe, resume=dict(it0=it, X=X))
    al = cfg["al"]
    pen = al["penalty"] * al["scaling"] ** (it // al["lag"])          # penalty in force during iteration `it`
    pen_in = al["penalty"] * al["scaling"] ** ((it - 1) // al["lag"]) if it else pen  # ... when the incoming trajectory was rolled out
    res = dict(it0=it, init_penalty=pen_in, lambda_mask=states[it - 1]["lam"][i] if it else None, X=X)
    return orc.solve_al(s, inp["A"], inp["b"], states[it]["lam"][i], U, 1, al["lag"], pen, al["scaling"], True, False, probe=probe, resume=res)


def _rel(a, b):
    fa, fb = np.isfinite(a), np.isfinite(b)
    if not fa and not fb:
        return 0.0
    if fa != fb:
        return np.inf  # NaN on one side only: never excused
    return abs(a - b) / max(abs(b), 1e-300)


def decisions_follow(pr, ag, line_search=True, floor_strict=False):
    """(b) of the header.  pr: oracle probe record with ALL trials; ag: the step size the GPU accepted.  Returns (ok, ties, detail):
    ties = number of comparisons that go the other way within TIE_RTOL.  floor_strict: the batch solvers stop at alpha < 1e-3
    (BatchILQRCP.cpp:152), the recursive ones go on while alpha > 1e-3 (ILQRRecursive.cpp:155) -- the same last step size 2^-10."""
    c0 = pr["cost0"]
    ties = 0
    for a, c in zip(pr["alpha"], pr["cost"]):
        near = np.isfinite(c) and np.isfinite(c0) and abs(c - c0) <= TIE_RTOL * abs(c0)
        if floor_strict:  # BatchILQRCP.cpp:152: if ((cost < cost0) || (alpha < 1e-3)) accept
            at_floor, improves = a < ALPHA_FLOOR, bool(c < c0)
        else:             # ILQRRecursive.cpp:155: while (((newCost >= cost0) || isNaN(newCost)) && alpha > 1e-3 && line_search)
            at_floor, improves = not (a > ALPHA_FLOOR), not (bool(c >= c0) or np.isnan(c))
        if a > ag:  # the GPU rejected this step size
            if at_floor or not line_search:
                return False, ties, dict(alpha=a, why="went below the floor")
            if improves:
                if not near:
                    return False, ties, dict(alpha=a, cost=c, cost0=c0, why="rejected an improving step")
                ties += 1
        elif a == ag:  # ... and accepted this one
            if not (improves or at_floor or not line_search):
                if not near:
                    return False, ties, dict(alpha=a, cost=c, cost0=c0, why="accepted a non-improving step above the floor")
                ties += 1
            return True, ties, None
    return False, ties, dict(alpha=ag, why="step size not in the schedule")


def _variant_sensitivity(run, pick, ref, run_perturbed=None):
    """Largest relative move of pick(result) over the oracle's neutral variants (and, if given, over one-ulp perturbations of its input);
    inf if a variant changes what pick cannot compare."""
    sens = 0.0
    for var in VARIANTS:
        orc.set_variant(var)
        try:
            v = pick(run())
        finally:
            orc.set_variant(0)
        if v is None or not np.isfinite(v):
            return np.inf
        sens = max(sens, abs(v - ref) / max(abs(ref), 1e-300))
    for pt in (PERTURB if run_perturbed is not None else ()):
        v = pick(run_perturbed(pt))
        if v is None or not np.isfinite(v):
            return np.inf
        sens = max(sens, abs(v - ref) / max(abs(ref), 1e-300))
    return sens


def _cost_at(pr, a):
    for al, c in zip(pr["alpha"], pr["cost"]):
        if al == a:
            return c
    return None


def _multi_nonzero_rows(cfg, inp):
    return cfg["solver"] == "al" and bool(np.any(np.count_nonzero(np.asarray(inp["A"]).reshape(-1, np.asarray(inp["A"]).shape[-1]), axis=1) > 1))


def prove_instance(cfg, inp, i, states, ct, at, iters, segs=None, nb_iter=None, early_stop=False):
    """Classifies instance i of a Riccati solve.  ct, at: the GPU's cost / alpha traces [B][nb_iter]; iters: iterations the GPU ran.
    Returns dict(verdict = 'stepwise' | 'tie' | 'unexplained', steps = [...], n_ties, n_ill, worst_ill_ratio) -- 'stepwise': every
    iteration reproduced; 'tie': every iteration reproduced, some decision taken inside the rounding of the oracle's own comparison.
    With early_stop the stop / go-on decision after every iteration is checked the same way (nb_iter = the solve's iteration cap)."""
    segs = segs or panda_segs()
    s = oracle_system_of_instance(cfg, inp, i, segs)
    steps, verdict = [], "stepwise"
    n = int(iters[i])
    nb_iter = int(nb_iter if nb_iter is not None else ct.shape[1])
    n_ties = n_ill = 0
    worst_ratio = 0.0
    multi = _multi_nonzero_rows(cfg, inp)
    for it in range(n):
        r = one_step(cfg, inp, i, it, states, segs, s)
        pr = r["probe"][0]
        cg, ag = float(ct[i, it]), float(at[i, it])
        ao = float(r["trace_alpha"][0])
        st = dict(it=it, alpha_gpu=ag, alpha_orc=ao, cost_gpu=cg, cost_orc=float(r["trace_cost"][0]))
        fails = []
        # the state handed over: its trajectory is the rollout of its controls, and its cost is the GPU's previous trace entry
        xdev = r.get("x_dev")
        if xdev is not None:
            st["x_dev"] = xdev
            if not (xdev <= XDEV_TOL) and np.all(np.isfinite(states[it]["X"][i])):
                fails.append("x_dev")
        if it > 0 and np.isfinite(ct[i, it - 1]):
            st["cost0_rel"] = _rel(pr["cost0"], float(ct[i, it - 1]))
            if st["cost0_rel"] > COST0_RTOL:
                fails.append("cost0")
        # (b) decisions from the oracle's own trial costs
        ok, ties, why = decisions_follow(pr, ag)
        if not ok:
            if multi and pr["mask_margin_in"] <= MASK_ATOL:  # g of a multi-coefficient row is a sum in another order on the GPU
                ties += 1
                st["mask_tie"] = pr["mask_margin_in"]
            else:
                fails.append("decision")
                st["decision"] = why
        # (c) the cost at the GPU's step size
        co_at = _cost_at(pr, ag)
        rel = _rel(cg, co_at) if co_at is not None else np.inf
        st["rel"] = rel
        if rel > STEP_RTOL:
            ill = False
            if rel <= STEP_RTOL_ILL:
                sens = _variant_sensitivity(lambda: one_step(cfg, inp, i, it, states, segs, s), lambda rv: _cost_at(rv["probe"][0], ag), co_at,
                                            lambda pt: one_step(cfg, inp, i, it, states, segs, s, perturb=pt))
                st["variant_rel"] = sens
                ill = rel <= ILL_FACTOR * sens
                if ill:
                    n_ill += 1
                    worst_ratio = max(worst_ratio, rel / sens if np.isfinite(sens) and sens > 0 else 0.0)
            if not ill:
                if multi and pr["mask_margin_in"] <= MASK_ATOL and "mask_tie" not in st:
                    ties += 1
                    st["mask_tie"] = pr["mask_margin_in"]
                else:
                    fails.append("cost")
        st["how"] = "FAIL:" + ",".join(fails) if fails else ("tie" if ties else ("same" if rel <= STEP_RTOL else "same:ill-conditioned"))
        n_ties += ties
        if fails:
            verdict = "unexplained"
        elif ties and verdict != "unexplained":
            verdict = "tie"
        if early_stop and not fails and it < nb_iter - 1 and ag == ao:  # (after the last allowed iteration the decision leaves no trace)
            # the stop decision taken on this iteration's result (ILQRRecursive.cpp:174-176, AL-ILQR.cpp:225)
            crit = ao * np.sqrt(pr["dun"])
            co = float(r["trace_cost"][0])
            stop_o = crit < 1e-3 and (cfg["solver"] == "al" or co < 1e-3)
            stop_g = (it == n - 1) and (n < nb_iter)
            if stop_o != stop_g:
                near = abs(crit - 1e-3) <= STOP_RTOL * 1e-3 or (cfg["solver"] != "al" and abs(co - 1e-3) <= STOP_RTOL * 1e-3)
                st["stop"] = "tie:early-stop" if near else "FAIL"
                st["stop_crit"] = float(crit)
                if near:
                    n_ties += 1
                verdict = "unexplained" if not near else ("tie" if verdict != "unexplained" else verdict)
        steps.append(st)
    return dict(verdict=verdict, steps=steps, n_ties=n_ties, n_ill=n_ill, worst_ill_ratio=worst_ratio)


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
    states = gpu_states(p, cfg, nb_iter, False, run_solver, upto=int(max(iters[flagged])) if flagged else 0) if flagged else None
    results, failures = [], []
    proofs = {i: prove_instance(cfg, inp, i, states, ct, at, iters, segs, nb_iter, early_stop) for i in flagged}
    for i in todo:
        results.append((rel[i] <= rtol, proofs.get(i)))
        pf = proofs.get(i)
        if pf and pf["verdict"] == "unexplained":
            failures.append(dict(i=i, rel=float(rel[i]), steps=[st for st in pf["steps"] if st["how"].startswith("FAIL") or st.get("stop") == "FAIL"]))
    summ = summarize(results)
    summ["n_proven_always"] = sum(1 for i in always if i in proofs and proofs[i]["verdict"] != "unexplained")
    return summ, rel, failures


def summarize(results):
    """Fractions over a list of (within_1e4: bool, proof or None) pairs, plus what the proofs leaned on: the number of decisions
    taken inside the rounding of the oracle's own comparison (ties) and of steps accepted as ill-conditioned, with the worst ratio of
    the GPU's deviation to the oracle's own variant sensitivity (<= ILL_FACTOR by construction)."""
    n = len(results)
    within = sum(1 for w, _ in results if w)
    tie = sum(1 for w, pf in results if not w and pf and pf["verdict"] == "tie")
    stepwise = sum(1 for w, pf in results if not w and pf and pf["verdict"] == "stepwise")
    unexpl = n - within - tie - stepwise
    proofs = [pf for _, pf in results if pf]
    return dict(n=n, frac_within_1e4=within / n, frac_proven_tie=tie / n, frac_proven_stepwise=stepwise / n, frac_unexplained=unexpl / n,
                n_proofs=len(proofs), n_steps_checked=sum(len(pf["steps"]) for pf in proofs), n_tie_decisions=sum(pf["n_ties"] for pf in proofs),
                n_steps_ill_conditioned=sum(pf["n_ill"] for pf in proofs), worst_ill_ratio=max([pf["worst_ill_ratio"] for pf in proofs], default=0.0))


# ----------------------------------------------------------------------------- batch solvers (BatchILQRCP / BatchILQR)


def gpu_states_batch(p, solve, nb_iter):
    """Controls of the whole batch after 0 .. nb_iter iterations, and the cost / alpha traces of nb_iter + 1 iterations -- all from
    re-runs WITHOUT early stop (the solves restart from the U0 of the problem; a stopped instance's prefix is the same).
    solve(p, n, early_stop) runs the batch solver under test."""
    states = []
    for it in range(nb_iter + 1):
        solve(p, it, False)
        states.append(dict(U=p.U()))
    solve(p, nb_iter + 1, False)
    ct, at = p.trace(nb_iter + 1)
    return states, ct, at


def prove_instance_batch(cfg, inp, i, psi, states, ct_ext, at_ext, n, early_stop, nb_iter, segs=None):
    """Classifies instance i of a BatchILQRCP / BatchILQR solve (psi = None: identity basis, BatchILQR.cpp:110-173).  The solver's state
    is its control vector alone (BatchILQRCP.cpp:109-175 keeps nothing else across iterations), so one oracle iteration from the GPU's
    controls after `it` iterations must give: the GPU's printed (pre-step) cost, decisions that follow from the oracle's trial costs,
    and -- at the GPU's step size -- the pre-step cost the GPU prints at iteration it + 1 (ct_ext has nb_iter + 1 entries)."""
    segs = segs or panda_segs()
    s = oracle_system_of_instance(cfg, inp, i, segs)
    steps, verdict = [], "stepwise"
    n_ties = n_ill = 0
    worst_ratio = 0.0

    def run(it, perturb=None):
        u = _u_oracle(cfg, inp, states[it]["U"][i])
        if perturb is not None:
            u = u * (1.0 + np.where(np.arange(u.size) % 2 == 0, perturb[0], perturb[1]) * 2.0 ** -52)
        return orc.solve_batch(s, u, 1, False, probe="all") if psi is None else orc.solve_batch_cp(s, psi, u, 1, False, probe="all")

    for it in range(int(n)):
        r = run(it)
        pr = r["probe"][0]
        ag, ao = float(at_ext[i, it]), float(r["trace_alpha"][0])
        st = dict(it=it, alpha_gpu=ag, alpha_orc=ao, cost0_gpu=float(ct_ext[i, it]), cost0_orc=pr["cost0"])
        fails = []
        st["cost0_rel"] = _rel(float(ct_ext[i, it]), pr["cost0"])
        if st["cost0_rel"] > COST0_RTOL:
            fails.append("cost0")
        ok, ties, why = decisions_follow(pr, ag, floor_strict=True)
        if not ok:
            fails.append("decision")
            st["decision"] = why
        co_at = _cost_at(pr, ag)
        cg_next = float(ct_ext[i, it + 1])
        rel = _rel(cg_next, co_at) if co_at is not None else np.inf
        st["rel"] = rel
        if rel > STEP_RTOL:
            ill = False
            if rel <= STEP_RTOL_ILL:  # the normal equations are ill-conditioned (R = 1e-5 against J'QJ ~ 1; overlapping bases)
                sens = _variant_sensitivity(lambda: run(it), lambda rv: _cost_at(rv["probe"][0], ag), co_at, lambda pt: run(it, pt))
                st["variant_rel"] = sens
                ill = rel <= ILL_FACTOR * sens
                if ill:
                    n_ill += 1
                    worst_ratio = max(worst_ratio, rel / sens if np.isfinite(sens) and sens > 0 else 0.0)
            if not ill:
                fails.append("cost")
        st["how"] = "FAIL:" + ",".join(fails) if fails else ("tie" if ties else ("same" if rel <= STEP_RTOL else "same:ill-conditioned"))
        n_ties += ties
        if fails:
            verdict = "unexplained"
        elif ties and verdict != "unexplained":
            verdict = "tie"
        if early_stop and not fails and it < nb_iter - 1 and ag == ao:  # BatchILQRCP.cpp:167
            crit = ao * pr["dun"]
            stop_o = crit < 1e-3
            stop_g = (it == n - 1) and (n < nb_iter)
            if stop_o != stop_g:
                near = abs(crit - 1e-3) <= STOP_RTOL * 1e-3
                st["stop"] = "tie:early-stop" if near else "FAIL"
                if near:
                    n_ties += 1
                verdict = "unexplained" if not near else ("tie" if verdict != "unexplained" else verdict)
        steps.append(st)
    return dict(verdict=verdict, steps=steps, n_ties=n_ties, n_ill=n_ill, worst_ill_ratio=worst_ratio)


def check_batch_solver(p, cfg, inp, psi, nb_iter, early_stop, solve, always=(0, 1), rtol=1e-4, indices=None):
    """The parity gate of the batch solvers: `p` has been solved with solve(p, nb_iter, early_stop).  Every instance (of `indices`) has
    the oracle's step-size sequence and a cost trace within `rtol` of the oracle's own end-to-end run, or is PROVEN iteration by
    iteration; the instances in `always` are proven whatever their distance.  No share of a batch is excused.  Returns
    (summary, rel, failures, oracle_runs)."""
    iters = p.iters()
    ct, at = p.trace(nb_iter)
    B = len(iters)
    segs = panda_segs()
    rel, flagged, runs = np.zeros(B), [], {}
    todo = list(range(B)) if indices is None else [int(i) for i in indices]
    for i in todo:
        s = oracle_system_of_instance(cfg, inp, i, segs)
        u0 = _u_oracle(cfg, inp, inp["U0"][i])
        r = orc.solve_batch(s, u0, nb_iter, early_stop) if psi is None else orc.solve_batch_cp(s, psi, u0, nb_iter, early_stop)
        runs[i] = r
        n = r["iters"]
        same = int(iters[i]) == n and np.array_equal(at[i][:n], r["trace_alpha"])
        if same:
            d = [_rel(float(a), float(b)) for a, b in zip(ct[i][:n], r["trace_cost"])]
            rel[i] = max(d) if d else 0.0
        else:
            rel[i] = np.inf
        if rel[i] > rtol or i in always:
            flagged.append(i)
    results, failures = [], []
    proofs = {}
    if flagged:
        states, ct_ext, at_ext = gpu_states_batch(p, solve, int(max(iters[flagged])))
        for i in flagged:
            proofs[i] = prove_instance_batch(cfg, inp, i, psi, states, ct_ext, at_ext, int(iters[i]), early_stop, nb_iter, segs)
            # the re-runs are the solve under test: same printed costs and step sizes over the iterations it made
            n = int(iters[i])
            if not (np.array_equal(ct_ext[i][:n], ct[i][:n], equal_nan=True) and np.array_equal(at_ext[i][:n], at[i][:n], equal_nan=True)):
                proofs[i]["verdict"] = "unexplained"
                proofs[i]["steps"].append(dict(it=-1, how="FAIL:re-run differs from the solve"))
    for i in todo:
        pf = proofs.get(i)
        results.append((rel[i] <= rtol, pf))
        if pf and pf["verdict"] == "unexplained":
            failures.append(dict(i=i, rel=float(rel[i]), steps=[st for st in pf["steps"] if st["how"].startswith("FAIL") or st.get("stop") == "FAIL"]))
    summ = summarize(results)
    summ["n_proven_always"] = sum(1 for i in always if i in proofs and proofs[i]["verdict"] != "unexplained")
    return summ, rel, failures, runs
