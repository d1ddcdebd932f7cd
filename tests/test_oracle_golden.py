"""Pins the CPU oracle (oracle/ilqr_oracle.c) against every known-answer value the reference holds for the
hot path: the per-iteration cost/alpha streams stored in its tutorial notebooks (6 significant digits,
exact alpha sequence, exact iteration count incl. early stop and the -nan divergence) and the FK literal.
The reference has no tests of its own."""
import numpy as np
import pytest

from tests.helpers import assert_trace, golden, oracle_system, orc, panda_segs, psi_of, u0_of

G = golden()
CASES = [(n, i) for n, c in G["cases"].items() for i, s in enumerate(c["solves"])]


def test_fk_literal_from_notebook():
    # POS_ORN_MULTI_SYS.ipynb cell 8 stores FK(q0)'s quaternion incl. the negative w (pins KDL GetQuaternion's branch)
    ch = orc.make_chain(panda_segs())
    p, quat, J, _, _ = orc.fk(ch, G["fk_kat"]["q0"])
    np.testing.assert_allclose(quat, G["fk_kat"]["quat_from_notebook"], atol=5e-9)
    np.testing.assert_allclose(p, [0.38162676, 0.28572033, 0.57378428], atol=5e-9)
    p0, quat0, _, _, _ = orc.fk(ch, np.zeros(7))
    np.testing.assert_allclose(p0, [0.088, 0, 0.8226], atol=1e-9)
    np.testing.assert_allclose(quat0, [0, 0.92387953, 0.38268343, 0], atol=1e-8)


def test_jacobian_matches_finite_difference():
    ch = orc.make_chain(panda_segs((0.1, -0.2, 0.3), (0.01, 0.02, 0.03)))
    rng = np.random.default_rng(0)
    for _ in range(5):
        q = rng.uniform(-2, 2, 7)
        p, quat, J, _, _ = orc.fk(ch, q)
        h = 1e-6
        for j in range(7):
            qp, qm = q.copy(), q.copy()
            qp[j] += h
            qm[j] -= h
            pp, quatp, *_ = orc.fk(ch, qp)
            pm, quatm, *_ = orc.fk(ch, qm)
            np.testing.assert_allclose((pp - pm) / (2 * h), J[:3, j], atol=1e-7)
            # angular part: w = 2 H(q) dq/dt
            if np.dot(quatp, quatm) < 0:
                quatm = -quatm
            dq = (quatp - quatm) / (2 * h)
            H = np.array([[-quat[1], quat[0], -quat[3], quat[2]], [-quat[2], quat[3], quat[0], -quat[1]], [-quat[3], -quat[2], quat[1], quat[0]]])
            np.testing.assert_allclose(2 * H @ dq, J[3:, j], atol=1e-6)


@pytest.mark.parametrize("name,idx", CASES, ids=[f"{n}-{G['cases'][n]['solves'][i]['solver']}" for n, i in CASES])
def test_trace(name, idx):
    case = G["cases"][name]
    sv = case["solves"][idx]
    s = oracle_system(case["problem"])
    u0 = u0_of(case["problem"])
    if sv["solver"] == "ILQRRecursive":
        r = orc.solve_recursive(s, u0, sv["nb_iter"], sv["line_search"], sv["early_stop"])
    elif sv["solver"] == "BatchILQRCP":
        r = orc.solve_batch_cp(s, psi_of(sv["psi"], s.T, s.n_u), u0, sv["nb_iter"], sv["early_stop"])
    elif sv["solver"] == "BatchILQR":  # planner3 of the notebooks; a dense 693..792-column inverse per iteration: first 12 iterations
        n = min(sv["nb_iter"], 12)
        r = orc.solve_batch(s, u0, n, sv["early_stop"])
        assert len(r["trace_cost"]) == min(n, len(sv["trace"]))
        # explicit inverse of a 792 x 792 matrix of condition ~1e9, iterated: the 6th printed digit is within reach of the rounding
        # of the inverse (LU here, Eigen's there) -- one unit of the last printed place instead of half
        assert_trace(r["trace_cost"], r["trace_alpha"], sv["trace"][:n], ulps=1.01)
        return
    else:
        m = sv["m"]
        A, b = np.zeros((m, s.n_x + s.n_u)), np.zeros(m)
        for i, j, v in sv["A_nonzero"]:
            A[i, j] = v
        for i, v in sv["b_nonzero"]:
            b[i] = v
        r = orc.solve_al(s, A, b, np.tile(b, (s.T - 1, 1)), u0, sv["nb_iter"], sv["lag_update_step"], sv["penalty"],
                         sv["scaling_factor"], sv["line_search"], sv["early_stop"])
    assert_trace(r["trace_cost"], r["trace_alpha"], sv["trace"])


def test_initial_costs():
    # initial costs printed by the CP solver = cost of the zero-control rollout (SURVEY.md 8c KATs)
    for name, c0 in (("POS_ORN_SYS", 0.506613), ("POS_ORN_TIME_SYS", 3.41273), ("POS_ORN_TIME_SYS_2ND", 4.04153), ("POS_ORN_SYS_OBJ_FRAME", 1.11617), ("POS_ORN_MULTI_SYS", 0.174263)):
        case = G["cases"][name]
        s = oracle_system(case["problem"])
        sv = [x for x in case["solves"] if x["solver"] == "BatchILQRCP"][0]
        r = orc.solve_batch_cp(s, psi_of(sv["psi"], s.T, s.n_u), u0_of(case["problem"]), 1, False)
        assert float("%.6g" % r["trace_cost"][0]) == c0


def test_inverse_and_primitives():
    rng = np.random.default_rng(1)
    A = rng.normal(size=(8, 8))
    np.testing.assert_allclose(orc.inverse(A) @ A, np.eye(8), atol=1e-10)
    ps = orc.psi("unitstep", 99, 2)
    assert ps.shape == (99, 2) and abs(ps[:50, 0].sum() - 1.0) < 1e-12 and ps[50:, 0].sum() == 0  # bw = round(49.5) = 50
    saw = orc.psi("sawtooth", 399, 2)
    assert saw[0, 0] == -0.5 and abs(saw[199, 0] - 0.5) < 1e-12 and saw[200, 0] == 0 and saw[200, 1] == -0.5
    lin = orc.psi("linear", 10, 2)
    assert lin.shape == (10, 4)
    bern = orc.psi("bernstein", 11, 4)
    np.testing.assert_allclose(bern.sum(1), 1.0, atol=1e-12)


def test_joint_space_system_config1():
    """BASELINE configs[0] (C1): 3-joint Robot2D + JointSpacePlannerSys, T = 50.  The problem is linear-quadratic (f(x) = x, J = I),
    so the recursive solver converges in one iteration and then fails to improve -- the shape of the trace stored in
    JOINT_SPACE_SYS.ipynb (2 iterations, alpha 1 then 0.000976562; its targets are unseeded random numbers, so no values)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tests.helpers import OracleFK, oracle_solve_instance
    from ilqr_planner_amd import workloads

    cfg = workloads.config("C1")
    desc, inp = workloads._make_joint_batch(cfg, 1, cfg["seed"], "inactive")
    r = oracle_solve_instance(cfg, inp, 0, 10, True)
    assert r["iters"] == 2 and r["trace_alpha"][0] == 1.0 and r["trace_alpha"][1] == 2.0 ** -10
    # optimum of the LQ problem: the keypoints are met up to the control penalty
    np.testing.assert_allclose(r["X"][24], inp["targets"][0][0][:3], atol=2e-3)
    np.testing.assert_allclose(r["X"][49], inp["targets"][1][0][:3], atol=2e-3)
    assert r["cost"] < 1e-4


def test_hybrid_sequence_keypoints():
    """Hybrid sequences (HYBRID_SYS*.ipynb; no stored trace is reproducible: unseeded targets): the keypoint of the joint-space
    sub-system inside a PosOrn(Time) problem has f(x) = x, J = I (JointSpacePlannerSys.cpp:77-81), so at its step
    cost = e'Qe + u'R_sub u (+ limits, once per sub-system), cost_x = -Q e - L'q, cost_xx = Q + L'L with e = target - x -- checked
    in closed form; the pose keypoint of the same system is untouched by the flag."""
    rng = np.random.default_rng(8)
    for timed in (False, True):
        n = 8 if timed else 7
        T, dt = 20, (None if timed else 0.1)
        tgt = rng.uniform(-1, 1, 7)
        Q = np.diag(rng.uniform(0.5, 2.0, n))
        Q[0, 1] = Q[1, 0] = 0.3
        k2 = G["cases"]["POS_ORN_SYS"]["problem"]["keypoints"][1]
        Qp = np.diag(list(k2["Qdiag"]) + ([0.1] if timed else []))
        qmax = np.full(7, 0.5)
        kps = [dict(timestep=9, joint=True, target=tgt, Q=Q, Ru=[1e-3] * n, **(dict(ctime=2.0) if timed else {})),
               dict(timestep=19, pos=k2["pos"], orn=k2["orn"], Q=Qp, Ru=[1e-4] * n, **(dict(ctime=4.0) if timed else {}))]
        from tests.helpers import panda_segs

        s = orc.make_system(panda_segs(), orc.SYS_POS_ORN_TIME if timed else orc.SYS_POS_ORN, 1, T, dt, [1e-6] * n, kps, [0.1] * 7, [0.0] * 7,
                            qmax, -qmax, lim_mult=2)
        x = rng.uniform(-0.9, 0.9, n)
        u = rng.uniform(-0.3, 0.3, n)
        e = np.concatenate([tgt, [2.0]])[:n] - x if timed else tgt - x
        lim = np.where(x[:7] > 0.5, 0.5 - x[:7], np.where(x[:7] < -0.5, -0.5 - x[:7], 0.0))
        limf = np.concatenate([lim, np.zeros(n - 7)])
        want = e @ Q @ e + 1e-3 * (u @ u) + 2 * (limf @ limf)
        np.testing.assert_allclose(orc.cost(s, x, u, 9), want, rtol=1e-13)
        np.testing.assert_allclose(orc.cost_x(s, x, 9), -Q @ e - 2 * limf, rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(orc.cost_xx(s, x, 9), Q + 2 * np.diag((limf != 0).astype(float)), rtol=1e-13)
        # the pose keypoint: same values as a system without the joint-space one
        s2 = orc.make_system(panda_segs(), orc.SYS_POS_ORN_TIME if timed else orc.SYS_POS_ORN, 1, T, dt, [1e-6] * n, kps[1:], [0.1] * 7, [0.0] * 7,
                             qmax, -qmax, lim_mult=2)
        assert orc.cost(s, x, u, 19) == orc.cost(s2, x, u, 19)
        np.testing.assert_array_equal(orc.cost_xx(s, x, 19), orc.cost_xx(s2, x, 19))
        # Batch-CP on a sequence sees no limit terms (SequentialSystem.cpp:12-18): zero controls keep the state at q0 = 0.1 < 0.5 anyway;
        # start outside instead
        s3 = orc.make_system(panda_segs(), orc.SYS_POS_ORN_TIME if timed else orc.SYS_POS_ORN, 1, T, dt, [1e-6] * n, kps, [0.8] * 7, [0.0] * 7,
                             qmax, -qmax, lim_mult=2)
        s4 = orc.make_system(panda_segs(), orc.SYS_POS_ORN_TIME if timed else orc.SYS_POS_ORN, 1, T, dt, [1e-6] * n, kps, [0.8] * 7, [0.0] * 7,
                             qmax, -qmax, lim_mult=1)
        psi = np.kron(orc.psi("unitstep", T - 1, 2), np.eye(n))
        u0 = np.tile([0.0] * 7 + ([0.1] if timed else []), T - 1)
        c_seq = orc.solve_batch_cp(s3, psi, u0, 1, False)["trace_cost"][0]
        c_plain = orc.solve_batch_cp(s4, psi, u0, 1, False)["trace_cost"][0]
        assert c_plain > c_seq + 0.1  # 2 keypoint rows x 7 joints x (0.8 - 0.5)^2
