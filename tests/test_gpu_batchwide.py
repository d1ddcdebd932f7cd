"""GPU parity of the batch solvers on wide bases: BatchILQR (= BatchILQRCP with PSI = I, reference
src/solver/BatchILQR.cpp:110-173) and BatchILQRCP with Kw > 16, through the C ABI.  The device never forms the Kw x Kw normal
matrix (low-rank form, ilqr_batchwide.hip); the oracle does what the reference does (dense H, explicit inverse)."""
import numpy as np
import pytest

from tests.helpers import assert_trace, golden, oracle_system_of_instance, orc
from tests.test_gpu_parity import _tutorial_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ilqr_planner_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


BATCH = [(n, i) for n, c in golden()["cases"].items() for i, s in enumerate(c["solves"]) if s["solver"] == "BatchILQR"]


@pytest.mark.parametrize("name,idx", BATCH, ids=[n for n, _ in BATCH])
def test_tutorial_batch_traces_on_gpu(ctx, name, idx):
    """planner3 = BatchILQR(sys) of the tutorial notebooks: printed pre-step cost to 6 significant digits, identical alpha
    sequence and iteration count."""
    case = golden()["cases"][name]
    sv = case["solves"][idx]
    B = 3
    p = _tutorial_problem(ctx, case, B)
    p.solve_batch(sv["nb_iter"], sv["early_stop"])
    iters = p.iters()
    ct, at = p.trace(sv["nb_iter"])
    nref = len(sv["trace"])
    for b in range(B):
        assert iters[b] == nref
        assert_trace(ct[b, :nref], at[b, :nref], sv["trace"])
    p.close()


def _compare(p, cfg, inp, B, nb_iter, psi, tol=1e-6, step_tol=1e-10):
    """psi = None: BatchILQR.  Every instance has the oracle's step-size sequence and a cost trace within `tol` of the oracle's
    end-to-end run (then controls and trajectory are compared too) or each of its iterations is reproduced by the oracle from the GPU's
    own controls (tests/parity_proof.py: cost to 1e-9, decisions from the oracle's trial costs): no instance is skipped, no share of the
    batch excused (round 2 let B / 8 instances with another step-size sequence go unchecked)."""
    from tests import parity_proof as pp

    U, X = p.U(), p.X()
    ct, at = p.trace(nb_iter)
    cost = p.cost()
    solve = (lambda q, n, es: q.solve_batch(n, es)) if psi is None else (lambda q, n, es: q.solve_batch_cp(psi, n, es))
    summ, rel, failures, runs = pp.check_batch_solver(p, cfg, inp, psi, nb_iter, False, solve, rtol=tol)
    print(f"parity {summ}")
    assert not failures, f"{len(failures)} instance(s) neither within {tol} nor proven: {failures[:3]}"
    for i in range(B):
        r = runs[i]
        s = oracle_system_of_instance(cfg, inp, i)
        if rel[i] <= tol:
            rl = np.abs(ct[i] - r["trace_cost"]) / np.maximum(np.abs(r["trace_cost"]), 1e-12)
            assert rl[1] <= step_tol, f"instance {i}: cost after the first step differs by {rl[1]:.2e}"  # one Gauss-Newton step: rounding only
            scale = max(1.0, np.abs(r["u"]).max())
            np.testing.assert_allclose(U[i].reshape(-1), r["u"], rtol=0, atol=tol * scale)
        # the returned controls rolled out by the oracle's dynamics give the returned trajectory
        x = np.asarray(X[i][0])
        for k in range(cfg["T"] - 1):
            x = orc.step(s, x, U[i][k])[0]
        np.testing.assert_allclose(X[i][-1], x, rtol=0, atol=1e-9 * max(1.0, np.abs(x).max()))
        assert np.isfinite(cost[i])


@pytest.mark.parametrize("cfg_name,T,limits,u0_scale,tol", [("C2", 30, "inactive", 0.0, 1e-6), ("C3r", 24, "urdf", 0.3, 1e-6), ("C2nd", 20, "inactive", 0.5, 1e-6),
                                                            ("C1j", 26, "urdf", 0.2, 1e-6), ("C4t1", 24, "urdf", 0.02, 1e-4), ("C4", 16, "inactive", 0.02, 1e-4),
                                                            ("C1t", 20, "inactive", 0.01, 1e-4)])
def test_batch_ilqr_random_batch_vs_oracle(ctx, cfg_name, T, limits, u0_scale, tol):
    """Identity basis on every system shape (PosOrn 1st / 2nd order, JointSpace: tabulated sensitivities; PosOrnTime 1st / 2nd order,
    JointSpaceTime: per-instance sensitivities), zero and random initial controls, inactive and active limits, against the dense
    restatement.  One step agrees to rounding (1e-10, checked for every instance); on the time systems the iteration amplifies that
    to ~1e-6 within five steps (the line search mostly ends at its floor there), so the trace tolerance is the north star's 1e-4."""
    from ilqr_planner_amd import workloads

    B, nb_iter = 12, 5
    cfg = dict(workloads.config(cfg_name), T=T)
    desc, inp = workloads.make_batch(ctx, cfg, B=B, limits=limits)
    rng = np.random.default_rng(3)
    inp["U0"] = inp["U0"] + u0_scale * rng.standard_normal(inp["U0"].shape)
    p = workloads.load_batch(ctx, desc, inp, B)
    p.solve_batch(nb_iter, False)
    _compare(p, cfg, inp, B, nb_iter, None, tol=tol)
    p.close()


@pytest.mark.parametrize("cfg_name,T,basis,K,u0_scale", [("C2", 40, "rbf", 5, 0.0), ("C3r", 30, "bernstein", 4, 0.3), ("C2nd", 24, "sawtooth", 3, 0.4)])
def test_wide_basis_cp_vs_oracle(ctx, cfg_name, T, basis, K, u0_scale):
    """BatchILQRCP with Kw = 7 K > 16 (overlapping bases: PSI'R PSI is dense) against the dense restatement."""
    from ilqr_planner_amd import workloads

    B, nb_iter = 10, 5
    cfg = dict(workloads.config(cfg_name), T=T)
    desc, inp = workloads.make_batch(ctx, cfg, B=B, limits="urdf")
    rng = np.random.default_rng(4)
    inp["U0"] = inp["U0"] + u0_scale * rng.standard_normal(inp["U0"].shape)
    p = workloads.load_batch(ctx, desc, inp, B)
    psi = np.kron(orc.psi(basis, T - 1, K), np.eye(7))
    assert psi.shape[1] > 16
    p.solve_batch_cp(psi, nb_iter, False)
    # overlapping bases make H ill-conditioned (cond(PSI'PSI) ~ 5e2 on top of R = 1e-5): the two linear solves agree to ~1e-6 in
    # cost after a few iterations, inside the 1e-4 the north star asks of final costs
    _compare(p, cfg, inp, B, nb_iter, psi, tol=1e-4, step_tol=1e-8)
    p.close()


def test_wide_equals_narrow_path(ctx):
    """The identity basis handed over as an explicit matrix takes the dense-PSI'R PSI route of the wide solver (Cholesky inverse,
    projection of u0 by three thin products); it must agree with the built-in identity route to rounding."""
    from ilqr_planner_amd import workloads

    B, nb_iter, T = 6, 4, 12
    cfg = dict(workloads.config("C2"), T=T)
    desc, inp = workloads.make_batch(ctx, cfg, B=B)
    p = workloads.load_batch(ctx, desc, inp, B)
    p.solve_batch(nb_iter, False)
    U1, c1 = p.U(), p.trace(nb_iter)[0]
    p.set_controls(inp["U0"])
    p.solve_batch_cp(np.eye((T - 1) * 7), nb_iter, False)
    U2, c2 = p.U(), p.trace(nb_iter)[0]
    np.testing.assert_allclose(c1, c2, rtol=1e-9)
    np.testing.assert_allclose(U1, U2, rtol=0, atol=1e-8 * max(1.0, np.abs(U1).max()))
    p.close()


@pytest.mark.parametrize("cfg_name,T,B", [("C2", 3, 67), ("C2", 9, 1), ("C4t1", 4, 5), ("C2nd", 5, 33)])
def test_batch_ilqr_edge_shapes(ctx, cfg_name, T, B):
    """Shortest horizons (the first keypoint falls on step 0 or 1, where the reference's shifted sensitivity is empty), a single
    instance, batches that are not a multiple of any lane grouping, early stop on: BatchILQR against the dense restatement."""
    from ilqr_planner_amd import workloads

    nb_iter = 4
    cfg = dict(workloads.config(cfg_name), T=T)
    desc, inp = workloads.make_batch(ctx, cfg, B=B)
    p = workloads.load_batch(ctx, desc, inp, B)
    p.solve_batch(nb_iter, True)
    ct, at = p.trace(nb_iter)
    iters, U = p.iters(), p.U()
    p.close()
    for i in sorted(set([0, B // 2, B - 1])):
        s = oracle_system_of_instance(cfg, inp, i)
        r = orc.solve_batch(s, inp["U0"][i].reshape(-1), nb_iter, True)
        n = r["iters"]
        assert iters[i] == n
        np.testing.assert_array_equal(at[i][:n], r["trace_alpha"])
        tol = 1e-4 if cfg["kind"] in (1, 3) else 1e-6  # time systems amplify the rounding of the two linear solves (see above)
        np.testing.assert_allclose(ct[i][:min(n, 2)], r["trace_cost"][:2], rtol=1e-9)  # one step: rounding only
        np.testing.assert_allclose(ct[i][:n], r["trace_cost"], rtol=tol)
        assert np.all(np.isnan(ct[i][n:]))
        np.testing.assert_allclose(U[i].reshape(-1), r["u"], rtol=0, atol=tol * max(1.0, np.abs(r["u"]).max()))


def test_wide_errors(ctx):
    from ilqr_planner_amd import workloads

    cfg = dict(workloads.config("C2"), T=20)
    desc, inp = workloads.make_batch(ctx, cfg, B=4)
    p = workloads.load_batch(ctx, desc, inp, 4)
    with pytest.raises(RuntimeError, match="full column rank"):
        p.solve_batch_cp(np.zeros((19 * 7, 28)), 1, False)
    p.close()
