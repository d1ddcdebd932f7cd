"""PosOrnKeypointDistFunct::diff against hand-derived known answers (tests/golden/f3_deadzone_kat.json, derivation in
tests/golden/make_f3_kat.py) -- the oracle on the CPU, the device through the C ABI.  The reference has no fixture for this class."""
import json
import os

import numpy as np
import pytest

from tests.helpers import GOLDEN, orc, panda_segs

KAT = json.load(open(os.path.join(GOLDEN, "f3_deadzone_kat.json")))["cases"]


def qmul(a, b):
    """Hamilton product, (w, x, y, z)."""
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                     w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def target_for(p, q, case):
    """Target pose (p*, q*) that puts the state (p, q) at the case's plain residual: p* = p + r_p, q = delta (x) q*."""
    th, n = case["theta"], np.asarray(case["axis"], float)
    delta = np.concatenate([[np.cos(th / 2)], np.sin(th / 2) * n])
    qs = qmul(delta * [1, -1, -1, -1], q)  # conj(delta) (x) q
    return p + np.asarray(case["r_pos"]), qs


@pytest.mark.parametrize("case", KAT, ids=[c["name"] for c in KAT])
@pytest.mark.parametrize("state", ["identity_target", "arm_pose"])
def test_oracle_deadzone_residual(case, state):
    if state == "identity_target":  # q* = (1,0,0,0): q = delta
        p = np.array([0.2, -0.2, 0.1])
        th, n = case["theta"], np.asarray(case["axis"], float)
        q = np.concatenate([[np.cos(th / 2)], np.sin(th / 2) * n])
    else:  # an arm pose: FK of the tutorial start configuration
        pq = orc.fk(orc.make_chain(panda_segs()), [0.62991112, -0.2329776, -0.01423721, -1.70254115, 0.06251303, 1.50592777, 0.71771416])
        p, q = pq[0], pq[1]
    ps, qs = target_for(p, q, case)
    kp = dict(timestep=1, pos=ps, orn=qs, Q=np.eye(6))
    s_plain = orc.make_system(panda_segs(), orc.SYS_POS_ORN, 1, 2, 0.1, [1e-5] * 7, [kp], [0.0] * 7)
    s_dist = orc.make_system(panda_segs(), orc.SYS_POS_ORN, 1, 2, 0.1, [1e-5] * 7,
                             [dict(kp, dist=dict(pos_radius=case["pos_radius"], orn_thresh=case["orn_thresh"]))], [0.0] * 7)
    fx = np.concatenate([p, q])
    import ctypes as C

    for s, want in ((s_plain, case["plain"]), (s_dist, case["expect"])):
        e = np.zeros(6)
        orc.lib().orc_kp_diff(C.byref(s), C.byref(s.kp[0]), fx.ctypes.data_as(C.POINTER(C.c_double)), e.ctypes.data_as(C.POINTER(C.c_double)))
        np.testing.assert_allclose(e, want, rtol=0, atol=5e-15)


@pytest.mark.gpu
@pytest.mark.parametrize("hip_path", ["v2", "v1"])
def test_device_deadzone_cost(hip_path, monkeypatch):
    """The device's residual, seen through the cost of a rollout whose only keypoint sits on the start state: e' Q e for the full
    precision and for one-hot precisions (each component's magnitude), all KAT cases in one batch."""
    from ilqr_planner_amd import capi, workloads

    monkeypatch.setenv("ILQR_HIP_PATH", hip_path)
    ctx = capi.Context(0)
    q0 = np.array([0.62991112, -0.2329776, -0.01423721, -1.70254115, 0.06251303, 1.50592777, 0.71771416])
    chain = workloads.panda_chain()
    base = capi.make_desc(kind=0, nb_deriv=1, horizon=3, dt=0.1, R_diag=[1e-5] * 7, chain=chain, kp_timesteps=[], kp_Q=[])
    pos, quat, _ = ctx.fk_batch(base, q0[None, :])
    p, q = pos[0], quat[0]
    for case in KAT:
        ps, qs = target_for(p, q, case)
        dist = [dict(pos_radius=case["pos_radius"], orn_thresh=case["orn_thresh"])]
        Qs = [np.diag(case["cost_Q"])] + [np.diag(np.eye(6)[i]) for i in range(6)]
        want = [case["cost"]] + [case["expect"][i] ** 2 for i in range(6)]
        for Qm, w in zip(Qs, want):
            desc = capi.make_desc(kind=0, nb_deriv=1, horizon=3, dt=0.1, R_diag=[1e-5] * 7, chain=chain, kp_timesteps=[0], kp_Q=[Qm], kp_dist=dist)
            pr = capi.BatchProblem(ctx, desc, 2)
            pr.set_init_state(np.tile(q0, (2, 1)))
            pr.set_keypoint_targets(0, np.tile(np.concatenate([ps, qs]), (2, 1)))
            pr.set_controls(np.zeros((2, 2, 7)))
            pr.solve_recursive(0, True, False)  # rollout only: the cost is the keypoint's e' Q e (u = 0)
            np.testing.assert_allclose(pr.cost(), [w, w], rtol=0, atol=1e-14)
            pr.close()
    ctx.close()
