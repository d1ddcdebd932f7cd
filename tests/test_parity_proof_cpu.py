"""The per-instance parity proof (tests/parity_proof.py) exercised without a GPU: a stand-in "device problem" whose solves are the
oracle's own.  Shows (1) the resume / probe aids of the oracle replay a solve exactly -- one iteration from the state after `it`
iterations reproduces iteration `it` of the uninterrupted solve bit for bit, for the recursive and the AL solver, with early stop
-- and (2) the gate is not vacuous: a wrong cost, a wrong step size or a NaN in the "device" trace is reported as unexplained."""
import numpy as np
import pytest

from ilqr_planner_amd import workloads
from tests import parity_proof as pp
from tests.helpers import OracleFK, oracle_solve_instance, panda_segs


class OracleProblem:
    """Quacks like capi.BatchProblem for parity_proof.check_batch, computing with the oracle (test double, CPU)."""

    def __init__(self, cfg, inp, B):
        self.cfg, self.inp, self.B, self.segs = cfg, inp, B, panda_segs()
        self.res, self.nb = None, 0

    def reset_multipliers(self):
        pass

    def solve(self, nb_iter, early_stop):
        self.nb = nb_iter
        self.res = [oracle_solve_instance(self.cfg, self.inp, i, nb_iter, early_stop, self.segs) for i in range(self.B)]

    def cost(self):
        return np.array([r["cost"] for r in self.res])

    def iters(self):
        return np.array([r["iters"] for r in self.res], dtype=np.int32)

    def U(self):
        return np.stack([r["U"] for r in self.res])

    def X(self):
        return np.stack([r["X"] for r in self.res])

    def lam(self):
        return np.stack([r["lam"] for r in self.res])

    def trace(self, nb_iter):
        ct, at = np.full((self.B, nb_iter), np.nan), np.full((self.B, nb_iter), np.nan)
        for i, r in enumerate(self.res):
            ct[i, : r["iters"]], at[i, : r["iters"]] = r["trace_cost"], r["trace_alpha"]
        return ct, at


def _run_solver(p, cfg, nb_iter=None, early_stop=False, psi=None):
    p.solve(nb_iter, early_stop)


@pytest.mark.parametrize("name,B,nb_iter,early_stop,limits", [("C3", 3, 7, False, "inactive"), ("C2", 3, 14, True, "urdf"), ("C4t1al", 2, 6, True, "inactive")])
def test_replay_is_exact_and_gate_detects_faults(name, B, nb_iter, early_stop, limits):
    cfg = workloads.config(name)
    desc, inp = workloads.make_batch(OracleFK(), cfg, B=B, limits=limits)
    p = OracleProblem(cfg, inp, B)
    p.solve(nb_iter, early_stop)
    full = list(p.res)

    def oracle_solve(i):
        return full[i]

    summ, rel, failures = pp.check_batch(p, cfg, inp, nb_iter, early_stop, _run_solver, oracle_solve, always=tuple(range(B)))
    assert not failures and summ["frac_unexplained"] == 0.0 and np.all(rel == 0.0)
    # every replayed iteration is bit-exact: the resume aid (iteration index, penalties, mask inputs) is the uninterrupted solve
    p.res = full
    ct, at = p.trace(nb_iter)
    states = pp.gpu_states(p, cfg, nb_iter, early_stop, _run_solver)
    p.res = full
    for i in range(B):
        pf = pp.prove_instance(cfg, inp, i, states, ct, at, p.iters(), p.segs, nb_iter, early_stop)
        assert pf["verdict"] == "stepwise" and all(st["rel"] == 0.0 and st.get("cost0_rel", 0.0) == 0.0 for st in pf["steps"]), pf
    # the gate is not vacuous
    it = max(k for k in range(min(3, int(p.iters()[0]))) if np.isfinite(ct[0, k]))  # (the time-system AL case goes NaN from iteration 2 on, as the oracle does)
    bad = ct.copy()
    bad[0, it] *= 1 + 1e-6  # a cost that is off by 1e-6 relative
    assert pp.prove_instance(cfg, inp, 0, states, bad, at, p.iters(), p.segs, nb_iter, early_stop)["verdict"] == "unexplained"
    bad_a = at.copy()
    bad_a[0, it] = at[0, it] / 2 if at[0, it] > 2e-3 else at[0, it] * 2  # another step size without a tie behind it
    assert pp.prove_instance(cfg, inp, 0, states, ct, bad_a, p.iters(), p.segs, nb_iter, early_stop)["verdict"] == "unexplained"
    bad_n = ct.copy()
    bad_n[0, it] = np.nan  # a NaN only the "device" produced
    assert pp.prove_instance(cfg, inp, 0, states, bad_n, at, p.iters(), p.segs, nb_iter, early_stop)["verdict"] == "unexplained"
    if early_stop and p.iters()[0] < nb_iter:  # a solve that went on although the stop test had fired
        more = p.iters().copy()
        more[0] += 1
        ct2, at2 = ct.copy(), at.copy()
        ct2[0, more[0] - 1], at2[0, more[0] - 1] = ct[0, more[0] - 2], 1.0
        assert pp.prove_instance(cfg, inp, 0, states, ct2, at2, more, p.segs, nb_iter, early_stop)["verdict"] == "unexplained"


def test_probe_records_the_decisions():
    """The probe's trial costs are the line search the solve ran: the accepted trial is the first one below cost0 (or the last)."""
    cfg = workloads.config("C3")
    desc, inp = workloads.make_batch(OracleFK(), cfg, B=1)
    from tests.helpers import oracle_system_of_instance, orc

    s = oracle_system_of_instance(cfg, inp, 0)
    al = cfg["al"]
    r = orc.solve_al(s, inp["A"], inp["b"], inp["lambda0"][0], inp["U0"][0].reshape(-1), 8, al["lag"], al["penalty"], al["scaling"], True, False, probe=True)
    prev = None
    for it, pr in enumerate(r["probe"]):
        assert pr["alpha"][-1] == r["trace_alpha"][it] and pr["cost"][-1] == r["trace_cost"][it]
        assert all(c >= pr["cost0"] or np.isnan(c) for c in pr["cost"][:-1])
        assert pr["cost"][-1] < pr["cost0"] or pr["alpha"][-1] <= 1e-3
        if prev is not None:
            assert pr["cost0"] == prev
        prev = pr["cost"][-1]
        assert (pr["clamp_margin"] < np.inf) == ((it + 1) % al["lag"] == 0)


class OracleBatchProblem:
    """capi.BatchProblem stand-in for parity_proof.check_batch_solver: BatchILQRCP solves by the oracle (test double, CPU)."""

    def __init__(self, cfg, inp, B, psi):
        from tests.helpers import oracle_system_of_instance

        self.cfg, self.inp, self.B, self.psi = cfg, inp, B, psi
        self.sys = [oracle_system_of_instance(cfg, inp, i) for i in range(B)]
        self.res = None

    def solve(self, nb_iter, early_stop):
        from tests.helpers import orc

        self.res = [orc.solve_batch_cp(self.sys[i], self.psi, self.inp["U0"][i].reshape(-1), nb_iter, early_stop) for i in range(self.B)]

    def iters(self):
        return np.array([r["iters"] for r in self.res], dtype=np.int32)

    def U(self):
        return np.stack([r["u"].reshape(self.inp["U0"][0].shape) for r in self.res])

    def trace(self, nb_iter):
        ct, at = np.full((self.B, nb_iter), np.nan), np.full((self.B, nb_iter), np.nan)
        for i, r in enumerate(self.res):
            ct[i, : r["iters"]], at[i, : r["iters"]] = r["trace_cost"], r["trace_alpha"]
        return ct, at


@pytest.mark.parametrize("name,T,early_stop", [("C5", 60, True), ("C4cp", 30, False)])
def test_batch_solver_replay_and_gate(name, T, early_stop):
    """The batch solvers' proof: one oracle iteration from the controls after `it` iterations is iteration `it` of the uninterrupted
    solve (the solver keeps no other state), and planted faults -- a printed cost off by 1e-6, another step size, a NaN, a missed
    early stop -- are reported as unexplained."""
    from tests.helpers import psi_of

    cfg = dict(workloads.config(name), T=T)
    B, nb_iter = 2, 8
    desc, inp = workloads.make_batch(OracleFK(), cfg, B=B)
    psi = psi_of(cfg["psi"], T, 7 + (1 if cfg["kind"] == 1 else 0))
    p = OracleBatchProblem(cfg, inp, B, psi)
    solve = lambda q, n, es: q.solve(n, es)
    p.solve(nb_iter, early_stop)
    summ, rel, failures, _ = pp.check_batch_solver(p, cfg, inp, psi, nb_iter, early_stop, solve, always=tuple(range(B)))
    assert not failures and summ["frac_unexplained"] == 0.0 and np.all(rel == 0.0), (summ, failures)
    p.solve(nb_iter, early_stop)
    iters = p.iters()
    states, ct_ext, at_ext = pp.gpu_states_batch(p, solve, int(iters.max()))
    pf = pp.prove_instance_batch(cfg, inp, 0, psi, states, ct_ext, at_ext, int(iters[0]), early_stop, nb_iter)
    assert pf["verdict"] == "stepwise" and all(st["rel"] == 0.0 and st["cost0_rel"] == 0.0 for st in pf["steps"]), pf
    it = min(2, int(iters[0]) - 1)
    bad = ct_ext.copy()
    bad[0, it] *= 1 + 1e-6
    assert pp.prove_instance_batch(cfg, inp, 0, psi, states, bad, at_ext, int(iters[0]), early_stop, nb_iter)["verdict"] == "unexplained"
    bad_a = at_ext.copy()
    bad_a[0, it] = at_ext[0, it] / 2 if at_ext[0, it] > 2e-3 else at_ext[0, it] * 2
    assert pp.prove_instance_batch(cfg, inp, 0, psi, states, ct_ext, bad_a, int(iters[0]), early_stop, nb_iter)["verdict"] == "unexplained"
    bad_n = ct_ext.copy()
    bad_n[0, it + 1] = np.nan
    assert pp.prove_instance_batch(cfg, inp, 0, psi, states, bad_n, at_ext, int(iters[0]), early_stop, nb_iter)["verdict"] == "unexplained"
    if early_stop and iters[0] < nb_iter:  # a solve that went on although the stop test had fired
        assert pp.prove_instance_batch(cfg, inp, 0, psi, states, ct_ext, at_ext, int(iters[0]) + 1, early_stop, nb_iter)["verdict"] == "unexplained"
