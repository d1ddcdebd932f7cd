"""GPU parity of the Batch-CP solver (BatchILQRCP) through the C ABI."""
import numpy as np
import pytest

from tests.helpers import assert_trace, golden, oracle_system_of_instance, orc, psi_of
from tests.test_gpu_parity import _tutorial_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ilqr_planner_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


CP = [(n, i) for n, c in golden()["cases"].items() for i, s in enumerate(c["solves"]) if s["solver"] == "BatchILQRCP"]


@pytest.mark.parametrize("name,idx", CP, ids=[n for n, _ in CP])
def test_tutorial_cp_traces_on_gpu(ctx, name, idx):
    """The four control-primitive traces of the reference's notebooks (unit-step, sawtooth and mixed bases; 1st/2nd
    order; with and without the time state) reproduced on the GPU: printed pre-step cost to 6 significant digits,
    identical alpha sequence and iteration count (early stop)."""
    case = golden()["cases"][name]
    sv = case["solves"][idx]
    B = 3
    p = _tutorial_problem(ctx, case, B)
    psi = psi_of(sv["psi"], p.T, p.dims.n_u)
    p.solve_batch_cp(psi, sv["nb_iter"], sv["early_stop"])
    iters = p.iters()
    ct, at = p.trace(sv["nb_iter"])
    nref = len(sv["trace"])
    for b in range(B):
        assert iters[b] == nref
        assert_trace(ct[b, :nref], at[b, :nref], sv["trace"])
    p.close()


@pytest.mark.parametrize("cfg_name,B,nb_iter", [("C5", 48, 6), ("C4cp", 24, 5)])
def test_cp_random_batch_vs_oracle(ctx, cfg_name, B, nb_iter):
    from ilqr_planner_amd import workloads

    cfg = workloads.config(cfg_name)
    desc, inp = workloads.make_batch(ctx, cfg, B=B)
    p = workloads.load_batch(ctx, desc, inp, B)
    psi = psi_of(cfg["psi"], cfg["T"], p.dims.n_u)
    p.solve_batch_cp(psi, nb_iter, False)
    U = p.U()
    ct, at = p.trace(nb_iter)
    _gate(p, cfg, inp, psi, nb_iter, False, U, ct, at)
    p.close()


def _gate(p, cfg, inp, psi, nb_iter, early_stop, U, ct, at, rtol=1e-4):
    """Every instance has the oracle's step-size sequence and costs within 1e-4 of the oracle's end-to-end run (then its controls are
    compared too), or each of its iterations is reproduced by the oracle from the GPU's own controls (tests/parity_proof.py): no
    instance is skipped and no share of the batch excused."""
    from tests import parity_proof as pp

    summ, rel, failures, runs = pp.check_batch_solver(p, cfg, inp, psi, nb_iter, early_stop, lambda q, n, es: q.solve_batch_cp(psi, n, es))
    print(f"parity {summ}")
    assert not failures, f"{len(failures)} instance(s) neither within {rtol} nor proven: {failures[:3]}"
    for i, r in runs.items():
        if rel[i] <= rtol:
            np.testing.assert_allclose(U[i].reshape(-1), r["u"], rtol=0, atol=1e-4 * max(1.0, np.abs(r["u"]).max()))
    return summ


@pytest.mark.parametrize("cfg_name,B,nb_iter", [("C5", 70, 5), ("C4cp", 20, 4)])
def test_wave_solve_equals_lane_solve(ctx, cfg_name, B, nb_iter, monkeypatch):
    """The Kw x Kw normal equations solved by one wave per instance (default) and by one lane per instance (ILQR_CP_SOLVE=lane) run
    the same operations on every entry in the same order: identical results, bit for bit."""
    from ilqr_planner_amd import workloads

    cfg = workloads.config(cfg_name)
    cfg["T"] = min(cfg["T"], 60)
    desc, inp = workloads.make_batch(ctx, cfg, B=B, limits="urdf")
    psi = psi_of(cfg["psi"], cfg["T"], 7 + (1 if cfg["kind"] == 1 else 0))
    out = []
    for mode in ("wave", "lane"):
        monkeypatch.setenv("ILQR_CP_SOLVE", mode)
        p = workloads.load_batch(ctx, desc, inp, B)
        p.solve_batch_cp(psi, nb_iter, False)
        out.append((p.U(), p.trace(nb_iter)[0]))
        p.close()
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])


def test_cp_on_a_sequence_ignores_limits(ctx):
    """Batch-CP on a (hybrid) SequentialSystem: joint-space via point + pose goal, and NO limit terms although the sub-systems have
    violated limits -- SequentialSystem does not override fpBatch, which runs on the sequence object built without limits
    (SequentialSystem.cpp:12-18).  The same batch with limit_multiplicity = 1 (a plain system) does count them."""
    from ilqr_planner_amd import workloads

    cfg = workloads.config("C2h")
    B, nb_iter = 16, 5
    desc, inp = workloads.make_batch(ctx, cfg, B=B, limits="urdf")
    inp["U0"] = inp["U0"] + 0.8  # drives joints over their limits
    psi = psi_of(cfg["psi"], cfg["T"], 7)
    p = workloads.load_batch(ctx, desc, inp, B)
    p.solve_batch_cp(psi, nb_iter, False)
    ct, at = p.trace(nb_iter)
    U = p.U()
    _gate(p, cfg, inp, psi, nb_iter, False, U, ct, at)
    p.close()
    for i in range(B):
        s = oracle_system_of_instance(cfg, inp, i)
        r = orc.solve_batch_cp(s, psi, inp["U0"][i].reshape(-1), 1, False)
        np.testing.assert_allclose(ct[i][0], r["trace_cost"][0], rtol=1e-12)
    desc.limit_multiplicity = 1
    p = workloads.load_batch(ctx, desc, inp, B)
    p.solve_batch_cp(psi, 1, False)
    c_plain = p.trace(1)[0][:, 0]
    p.close()
    assert np.all(c_plain > ct[:, 0] + 1e-3)  # the limit terms of the violated joints


@pytest.mark.parametrize("cfg_name,T,K", [("C4cp", 30, 3), ("C4t1", 40, 4)])
def test_cp_time_system_basis_up_to_32_columns(ctx, cfg_name, T, K):
    """Time systems with 16 < Kw <= 32 (32 lanes per instance in the same kernels): sawtooth x controls + unit step x sqrt(dt) with K
    pieces each (Kw = 8 K), against the dense restatement; one step to rounding, the trace within the north star's tolerance."""
    from ilqr_planner_amd import workloads

    B, nb_iter = 10, 5
    cfg = dict(workloads.config(cfg_name), T=T)
    desc, inp = workloads.make_batch(ctx, cfg, B=B)
    psi = psi_of(dict(kind="sawtooth+unitstep_dt", K=K), T, 8)
    assert 16 < psi.shape[1] <= 32
    p = workloads.load_batch(ctx, desc, inp, B)
    p.solve_batch_cp(psi, nb_iter, False)
    ct, at = p.trace(nb_iter)
    U = p.U()
    for i in range(B):  # one step: rounding only
        s = oracle_system_of_instance(cfg, inp, i)
        r = orc.solve_batch_cp(s, psi, inp["U0"][i].reshape(-1), 2, False)
        np.testing.assert_allclose(ct[i][:2], r["trace_cost"][:2], rtol=1e-9)
    _gate(p, cfg, inp, psi, nb_iter, False, U, ct, at)
    p.close()


def test_cp_errors(ctx):
    from ilqr_planner_amd import workloads

    cfg = workloads.config("C4cp")  # wide bases (Kw > 16) exist for the constant-dt systems and, on time systems, for PSI = I only
    desc, inp = workloads.make_batch(ctx, cfg, B=4)
    p = workloads.load_batch(ctx, desc, inp, 4)
    with pytest.raises(RuntimeError, match="identity basis"):
        p.solve_batch_cp(np.zeros(((cfg["T"] - 1) * 8, 40)), 1, False)
    p.close()
