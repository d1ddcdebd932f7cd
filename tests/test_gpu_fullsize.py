"""Parity at BASELINE.json's full sizes (C3: B = 4096, T = 200, AL; C4: 8 x 4096 = 32768 instances of the time-augmented 2nd-order
system; C5: B = 8192, T = 400, Batch-CP; BatchILQR at the tutorial shape), where running the oracle on every instance would take
minutes: size-independent properties plus a sample of instances against the oracle.

* instances are independent: a batch whose second half repeats the first gives bit-identical results for i and i + B/2, and a
  64-instance batch cut out of it reproduces the big batch bit for bit (no result depends on the batch size, the position in a
  wave or the workgroup an instance lands in);
* the returned trajectory is the rollout of the returned controls (oracle dynamics), and for the recursive solver the returned
  cost is the reference's cost of that trajectory (oracle cost function);
* a seeded sample of instances agrees with the oracle in final cost within the north star's 1e-4 (median 1e-6) or is proven
  iteration by iteration (tests/parity_proof.py; Riccati and batch solvers alike): no share of the sample is excused."""
import numpy as np
import pytest

from tests.helpers import oracle_solve_instance, oracle_system_of_instance, orc, panda_segs, psi_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ilqr_planner_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


def _take(inp, idx):
    out = dict(inp)
    for k in ("q0", "dq0", "U0", "lambda0"):
        if k in inp:
            out[k] = np.ascontiguousarray(inp[k][idx])
    out["targets"] = [np.ascontiguousarray(t[idx]) for t in inp["targets"]]
    return out


def _solve(ctx, cfg, desc, inp, B, nb_iter, solver, keep=False):
    from ilqr_planner_amd import workloads

    p = workloads.load_batch(ctx, desc, inp, B)
    if solver == "batch":
        p.solve_batch(nb_iter, False)
    elif solver == "batch_cp":
        p.solve_batch_cp(psi_of(cfg["psi"], cfg["T"], p.dims.n_u), nb_iter, False)
    else:
        workloads.run_solver(p, cfg, nb_iter=nb_iter, early_stop=False)
    out = dict(cost=p.cost(), U=p.U(), X=p.X(), iters=p.iters(), alpha=p.alpha(), trace=p.trace(nb_iter)[0])
    if keep:
        out["p"] = p
    else:
        p.close()
    return out


@pytest.mark.parametrize("cfg_name,B,nb_iter,solver", [("C3", 4096, 20, "al"), ("C4", 32768, 8, "recursive"), ("C5", 8192, 10, "batch_cp"),
                                                       ("C2", 4096, 10, "batch")])
def test_full_size_properties(ctx, cfg_name, B, nb_iter, solver, monkeypatch):
    from ilqr_planner_amd import workloads

    if cfg_name == "C4":  # the sweep of the time systems is chosen by batch size (two kernels that agree to rounding): the bit-for-bit
        monkeypatch.setenv("ILQR_SWEEP", "rows")  # comparison of the big batch with its cut-out is made on the one the big batch takes
        monkeypatch.setenv("ILQR_APPLY", "rows")  # (likewise the re-roll of the line-search winner: k_apply_rows_tm at this size, k_apply_dpp_tm for small batches)
    if cfg_name == "C3":  # likewise the forward pass of the single-integrator systems (k_forward_wg at this size, k_forward_dpp for small batches)
        monkeypatch.setenv("ILQR_FWD", "wg")

    cfg = workloads.config(cfg_name)
    desc, inp = workloads.make_batch(ctx, cfg, B=B)
    h = B // 2
    inp = _take(inp, np.concatenate([np.arange(h), np.arange(h)]))  # second half = first half
    big = _solve(ctx, cfg, desc, inp, B, nb_iter, solver)
    fin = np.isfinite(big["cost"])
    assert fin.mean() > (0.5 if cfg_name == "C4" else 0.999)  # the reference itself diverges on many C4 instances
    # independence of the instances, bit for bit
    np.testing.assert_array_equal(big["cost"][:h], big["cost"][h:])
    np.testing.assert_array_equal(big["U"][:h], big["U"][h:])
    np.testing.assert_array_equal(big["iters"][:h], big["iters"][h:])
    rng = np.random.default_rng(7)
    idx = np.sort(rng.choice(B, 61, replace=False))  # ragged: not a multiple of the wave size
    desc_s, _ = workloads.make_batch(ctx, cfg, B=len(idx))
    inp_s = _take(inp, idx)
    small = _solve(ctx, cfg, desc_s, inp_s, len(idx), nb_iter, solver, keep=True)
    np.testing.assert_array_equal(small["cost"], big["cost"][idx])
    np.testing.assert_array_equal(small["U"], big["U"][idx])
    np.testing.assert_array_equal(small["X"], big["X"][idx])
    # the returned trajectory is the rollout of the returned controls; recursive solver: the returned cost is the cost of that trajectory
    segs = panda_segs()
    T = cfg["T"]
    for i in idx[fin[idx]][:4]:
        s = oracle_system_of_instance(cfg, inp, i, segs)
        x = np.asarray(big["X"][i][0])
        c = 0.0
        for k in range(T - 1):
            c += orc.cost(s, x, big["U"][i][k], k)
            x = orc.step(s, x, big["U"][i][k])[0]
            np.testing.assert_allclose(big["X"][i][k + 1], x, rtol=0, atol=1e-9 * max(1.0, np.abs(x).max()))
        c += orc.cost(s, x, np.zeros(s.n_u), T - 1)
        if solver == "recursive":
            assert abs(c - big["cost"][i]) <= 1e-9 * max(abs(c), 1e-9)
    # a sample against the oracle
    sample = idx[:24] if solver != "batch" else idx[:3]  # the dense 693-column restatement takes ~10 s per instance
    rel = []
    for i in sample:
        got = big["cost"][i]
        if solver in ("batch", "batch_cp"):  # the batch solvers report the cost before each step: compare the last one
            so = oracle_system_of_instance(cfg, inp, i, segs)
            u0 = inp["U0"][i].reshape(-1)
            r = orc.solve_batch(so, u0, nb_iter, False) if solver == "batch" else orc.solve_batch_cp(so, psi_of(cfg["psi"], T, 7), u0, nb_iter, False)
            ref, got = r["trace_cost"][-1], big["trace"][i][nb_iter - 1]
        else:
            ref = oracle_solve_instance(cfg, inp, i, nb_iter, False, segs)["cost"]
        if not np.isfinite(ref) or not np.isfinite(got):
            continue
        rel.append(abs(got - ref) / max(abs(ref), 1e-12))
    rel = np.asarray(rel)
    assert len(rel) >= len(sample) // 3
    assert np.median(rel) <= 1e-6, f"median rel err {np.median(rel):.2e}"
    if solver in ("al", "recursive"):
        # the 61-instance cut-out is bit-identical to the big batch (asserted above), so the per-instance proof runs on it: every instance
        # of it is within 1e-4 of the oracle's end-to-end run or each of its iterations is reproduced by the oracle from the GPU's state
        from tests import parity_proof as pp

        summ, _, failures = pp.check_batch(small["p"], cfg, inp_s, nb_iter, False, workloads.run_solver,
                                           lambda i: oracle_solve_instance(cfg, inp_s, i, nb_iter, False, segs), always=(0, 1))
        small["p"].close()
        print(f"parity {cfg_name} full size: {summ}")
        assert not failures, f"{len(failures)} instance(s) neither within 1e-4 nor proven: {failures[:3]}"
    else:  # batch solvers: the same gate on the cut-out -- within 1e-4 of the oracle's end-to-end run with its step-size sequence, or every
        # iteration reproduced by the oracle from the GPU's own controls (round 2 asked 85 % of the sample to be within 1e-4)
        from tests import parity_proof as pp

        psi = None if solver == "batch" else psi_of(cfg["psi"], T, 7)
        solve = (lambda q, n, es: q.solve_batch(n, es)) if solver == "batch" else (lambda q, n, es: q.solve_batch_cp(psi, n, es))
        sub = list(range(3)) if solver == "batch" else None  # the dense 693-column restatement takes ~10 s per instance
        summ, _, failures, _ = pp.check_batch_solver(small["p"], cfg, inp_s, psi, nb_iter, False, solve, always=(0, 1), indices=sub)
        small["p"].close()
        print(f"parity {cfg_name} full size: {summ}")
        assert not failures, f"{len(failures)} instance(s) neither within 1e-4 nor proven: {failures[:3]}"


@pytest.mark.parametrize("cfg_name,solver,B", [("C3", "al", 2048), ("C2", "recursive", 256), ("C2", "recursive", 1001)])
def test_forward_passes_agree(ctx, cfg_name, solver, B, monkeypatch):
    """The single-integrator systems have two forward passes chosen by batch size: k_forward_dpp (16 lanes per instance on registers: the chain of a
    small batch) and k_forward_wg (32 lanes per instance through LDS: the stream of a large one).  They sum the seven products of a control in a
    different order, so one iteration from the same state agrees to rounding, not bit for bit; each goes through the per-instance proof in
    test_gpu_parity.py (hip_path v2 / v2wg).  Here: ONE iteration of both on the same batch -- trajectories to 1e-12, the same step sizes, costs to 1e-12."""
    from ilqr_planner_amd import workloads

    cfg = dict(workloads.config(cfg_name), T=60)
    desc, inp = workloads.make_batch(ctx, cfg, B=B)
    res = {}
    for fwd in ("wg", "dpp"):
        monkeypatch.setenv("ILQR_FWD", fwd)
        res[fwd] = _solve(ctx, cfg, desc, inp, B, 1, solver)
    a, b = res["wg"], res["dpp"]
    np.testing.assert_array_equal(a["alpha"], b["alpha"])
    np.testing.assert_allclose(b["cost"], a["cost"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(b["U"], a["U"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(b["X"], a["X"], rtol=0, atol=1e-11)


def test_two_stream_split_does_not_change_results(ctx, monkeypatch):
    """Large batches on the matrix-core sweep (one instance per wave) are solved as two halves on two internal streams, the second one sweep behind
    the first (ilqr_ctx_set_split).  Since round 3 the time systems take the row-per-lane sweep at these sizes, so the schedule is reached through the
    cross-check switch (and by the shapes only the matrix-core sweep takes); instances are independent: bit-identical results with the split on and off."""
    from ilqr_planner_amd import workloads

    monkeypatch.setenv("ILQR_SWEEP", "mfma")
    cfg = dict(workloads.config("C4"), T=40)
    B, nb_iter = 2304, 3
    desc, inp = workloads.make_batch(ctx, cfg, B=B)
    res = {}
    for on in (1, 0):
        ctx.set_split(on)
        res[on] = _solve(ctx, cfg, desc, inp, B, nb_iter, "recursive")
    ctx.set_split(1)
    for k in ("cost", "U", "X", "iters", "alpha"):
        np.testing.assert_array_equal(res[1][k], res[0][k])


@pytest.mark.parametrize("cfg_name,B", [("C4", 256), ("C4t1", 100), ("C1t", 64)])
def test_rerolls_of_the_winner_agree(ctx, cfg_name, B, monkeypatch):
    """Time systems: the winner of the step-size-parallel line search is rolled out again where the speculated step size lost -- by k_apply_dpp_tm (16 lanes
    per instance on registers) for small batches, by k_apply_rows_tm (8 lanes per instance through LDS) for large ones.  Different summation order: after
    two iterations from the same start the trajectories agree to rounding; each runs under the per-instance proof in test_gpu_parity.py (v2 / v2rows)."""
    from ilqr_planner_amd import workloads

    cfg = dict(workloads.config(cfg_name), T=50)
    desc, inp = workloads.make_batch(ctx, cfg, B=B)
    res = {}
    for ap in ("rows", "dpp"):
        monkeypatch.setenv("ILQR_APPLY", ap)
        res[ap] = _solve(ctx, cfg, desc, inp, B, 2, "recursive")
    a, b = res["rows"], res["dpp"]
    np.testing.assert_array_equal(a["alpha"], b["alpha"])
    fin = np.isfinite(a["cost"])
    assert fin.mean() > 0.5 and np.array_equal(fin, np.isfinite(b["cost"]))
    # (two expanding iterations of a system whose step length is the square of a control: rounding differences of 1e-16 arrive as 1e-10)
    np.testing.assert_allclose(b["cost"][fin], a["cost"][fin], rtol=1e-8, atol=0)
    np.testing.assert_allclose(b["U"][fin], a["U"][fin], rtol=0, atol=1e-8)
    np.testing.assert_allclose(b["X"][fin], a["X"][fin], rtol=0, atol=1e-8)


@pytest.mark.parametrize("cfg_name,solver", [("C3", "al"), ("C2", "recursive")])
def test_sweep_lane_groupings_agree(ctx, cfg_name, solver, monkeypatch):
    """The register-resident sweep of the single-integrator systems runs with 16 lanes per instance while that gives every SIMD at most one
    wave and with 8 lanes per instance (two instances per DPP row, every broadcast issued once per half with a bank mask) beyond: a batch
    just over 4 x 1024 instances takes the second form, a 61-instance cut-out of it the first.  Same operations on the same operands in the
    same order in both: bit-identical results, whichever half of a DPP row an instance sits in; the cut-out then goes through the
    per-instance proof."""
    from ilqr_planner_amd import workloads
    from tests import parity_proof as pp

    monkeypatch.setenv("ILQR_FWD", "wg")  # the forward pass is chosen by batch size too (two kernels that agree to rounding): the big batch's one for both
    cfg = dict(workloads.config(cfg_name), T=40)
    B, nb_iter = 4200, 6
    desc, inp = workloads.make_batch(ctx, cfg, B=B)
    big = _solve(ctx, cfg, desc, inp, B, nb_iter, solver)
    rng = np.random.default_rng(11)
    idx = np.sort(rng.choice(B, 61, replace=False))
    desc_s, _ = workloads.make_batch(ctx, cfg, B=len(idx))
    inp_s = _take(inp, idx)
    small = _solve(ctx, cfg, desc_s, inp_s, len(idx), nb_iter, solver, keep=True)
    np.testing.assert_array_equal(small["cost"], big["cost"][idx])
    np.testing.assert_array_equal(small["U"], big["U"][idx])
    np.testing.assert_array_equal(small["X"], big["X"][idx])
    np.testing.assert_array_equal(small["trace"], big["trace"][idx])
    segs = panda_segs()
    summ, _, failures = pp.check_batch(small["p"], cfg, inp_s, nb_iter, False, workloads.run_solver,
                                       lambda i: oracle_solve_instance(cfg, inp_s, i, nb_iter, False, segs), always=(0, 1, 2, 3))
    small["p"].close()
    assert not failures, f"{len(failures)} instance(s) neither within 1e-4 nor proven: {failures[:3]}"
