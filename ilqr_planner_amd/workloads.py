"""Seeded synthetic problem batches for the BASELINE.json configs.

Targets are FK(q_rand) for q_rand ~ U(joint limits), evaluated with the library's own batched FK kernel, so that
every instance is reachable.  Nothing here touches the oracle or the reference tree.
"""
from __future__ import annotations

import os

import numpy as np

from . import capi

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
Q0_TUT = np.array([0.62991112, -0.2329776, -0.01423721, -1.70254115, 0.06251303, 1.50592777, 0.71771416])


def panda_urdf_text() -> str:
    return open(os.path.join(_DATA, "panda_chain.urdf")).read()


def panda_chain(tool_rpy=None, tool_xyz=None):
    return capi.chain_from_urdf(panda_urdf_text(), "panda_link0", "panda_tip", tool_rpy, tool_xyz)


def config(name: str):
    """Static part of a BASELINE config: kind, nb_deriv, T, dt, B, seed, precision diagonals, keypoint times."""
    P = [1, 1, 1, .1, .1, .1]
    V = [1, 1, 1, .1, .1, .1]
    cfgs = {
        # C1: the reference's CPU plumbing case -- Robot2D (3 links) + JointSpacePlannerSys, T = 50, one seed.  On the device the
        # 3 joints are padded to the 7 the kernels are built for (zero precision, zero limit weight, u = 0 keeps them at rest).
        "C1": dict(kind=capi.SYS_JOINT, nb_deriv=1, T=50, dt=0.1, B=1, seed=0, dof=3, q0=[np.pi / 4] * 3, Qdiag=[[1, 1, 1]] * 2, solver="recursive",
                   nb_iter=10),
        # JointSpaceTimePlannerSys: joint targets with a free duration (time state, dt = u_last^2), AngularTimeKeypoint targets
        "C1t": dict(kind=capi.SYS_JOINT_TIME, nb_deriv=1, T=60, dt=None, B=128, seed=11, dof=7, Qdiag=[[1] * 7 + [0], [1] * 7 + [.1]], ctimes=[2.0, 4.0],
                    solver="recursive", nb_iter=10),
        "C1j": dict(kind=capi.SYS_JOINT, nb_deriv=1, T=100, dt=0.1, B=256, seed=10, dof=7, Qdiag=[[1] * 7, [1, .5, 1, .5, 1, .5, 1]], solver="recursive",
                    nb_iter=8),
        # C2: "pos-only" = zero orientation precision (POS_ORN_MULTI_SYS.ipynb cell 12)
        "C2": dict(kind=capi.SYS_POS_ORN, nb_deriv=1, T=100, dt=0.1, B=256, seed=1, Qdiag=[[1, 1, 1, 0, 0, 0]] * 2, solver="recursive", nb_iter=20),
        # the same with joint-dependent control weights: K is not symmetric then, the sweep takes its general form and writes plain gain records
        "C2r": dict(kind=capi.SYS_POS_ORN, nb_deriv=1, T=60, dt=0.1, B=64, seed=21, Qdiag=[[1, 1, 1, 0, 0, 0]] * 2, solver="recursive", nb_iter=12,
                    R_diag=[1e-5, 2e-5, 5e-6, 1e-5, 3e-5, 1e-5, 4e-6]),
        # C3: AL-iLQR with the tutorial's single row q_6 <= 2.0 (penalty .25, scaling 1.1, update every 5)
        "C3": dict(kind=capi.SYS_POS_ORN, nb_deriv=1, T=200, dt=0.05, B=4096, seed=2, Qdiag=[P, P], solver="al", nb_iter=20,
                   al=dict(row=5, bound=2.0, penalty=0.25, scaling=1.1, lag=5)),
        # AL-iLQR on the other system shapes (same single row q_6 <= 2.0): 2nd order, time state
        "C2ndal": dict(kind=capi.SYS_POS_ORN, nb_deriv=2, T=100, dt=0.05, B=64, seed=15, Qdiag=[P + [1, 1, 1, 0, 0, 0], P + V], solver="al", nb_iter=10,
                       al=dict(row=5, bound=2.0, penalty=0.25, scaling=1.1, lag=5)),
        "C4t1al": dict(kind=capi.SYS_POS_ORN_TIME, nb_deriv=1, T=100, dt=None, B=64, seed=16, Qdiag=[P + [0], P + [.1]], ctimes=[2.0, 5.0], solver="al",
                       nb_iter=10, al=dict(row=5, bound=2.0, penalty=0.25, scaling=1.1, lag=5)),
        "C4al": dict(kind=capi.SYS_POS_ORN_TIME, nb_deriv=2, T=80, dt=None, B=48, seed=19, Qdiag=[P + V + [.1], P + V + [.1]], ctimes=[2.0, 4.0], solver="al",
                     nb_iter=8, al=dict(row=5, bound=2.0, penalty=0.25, scaling=1.1, lag=3)),
        "C1jal": dict(kind=capi.SYS_JOINT, nb_deriv=1, T=100, dt=0.1, B=64, seed=17, dof=7, Qdiag=[[1] * 7, [1, .5, 1, .5, 1, .5, 1]], solver="al", nb_iter=8,
                      al=dict(row=5, bound=1.0, penalty=0.25, scaling=1.1, lag=3)),
        "C1tal": dict(kind=capi.SYS_JOINT_TIME, nb_deriv=1, T=60, dt=None, B=64, seed=18, dof=7, Qdiag=[[1] * 7 + [0], [1] * 7 + [.1]], ctimes=[2.0, 4.0],
                      solver="al", nb_iter=8, al=dict(row=5, bound=0.5, penalty=0.25, scaling=1.1, lag=3)),
        "C3r": dict(kind=capi.SYS_POS_ORN, nb_deriv=1, T=200, dt=0.05, B=4096, seed=2, Qdiag=[P, P], solver="recursive", nb_iter=20),
        # PosOrnKeypointDistFunct: dead zones of 5 cm / 0.1 rad at the via point, 1 cm / mixed thresholds at the goal
        "C3d": dict(kind=capi.SYS_POS_ORN, nb_deriv=1, T=100, dt=0.05, B=256, seed=8, Qdiag=[P, P], solver="recursive", nb_iter=15,
                    kp_dist=[dict(pos_radius=0.05, orn_thresh=[0.1, 0.1, 0.1]), dict(pos_radius=0.01, orn_thresh=[0.02, 0.2, 0.0])]),
        "C2ndd": dict(kind=capi.SYS_POS_ORN, nb_deriv=2, T=100, dt=0.05, B=256, seed=9, Qdiag=[P + [1, 1, 1, 0, 0, 0], P + V], solver="recursive",
                      nb_iter=10, kp_dist=[dict(pos_radius=0.03, orn_thresh=[0.05, 0.05, 0.05]), None]),
        "C4": dict(kind=capi.SYS_POS_ORN_TIME, nb_deriv=2, T=200, dt=None, B=4096, seed=3, Qdiag=[P + V + [.1], P + V + [.1]],
                   ctimes=[2.5, 5.0], solver="recursive", nb_iter=20),
        "C5": dict(kind=capi.SYS_POS_ORN, nb_deriv=1, T=400, dt=0.01, B=8192, seed=4, Qdiag=[P, P], solver="batch_cp", nb_iter=10,
                   psi=dict(kind="unitstep", K=2)),
        # 2nd-order PosOrn (state [q, dq], control ddq) and 1st-order time system: the remaining System shapes of SURVEY.md 8
        "C2nd": dict(kind=capi.SYS_POS_ORN, nb_deriv=2, T=100, dt=0.05, B=256, seed=6, Qdiag=[P + [1, 1, 1, 0, 0, 0], P + V], solver="recursive", nb_iter=12),
        "C4t1": dict(kind=capi.SYS_POS_ORN_TIME, nb_deriv=1, T=100, dt=None, B=256, seed=7, Qdiag=[P + [0], P + [.1]], ctimes=[2.0, 5.0],
                     solver="recursive", nb_iter=14),
        # Batch-CP on the time-augmented 2nd-order system (the C4 system shape), sawtooth x controls + unit-step x sqrt(dt)
        # hybrid sequences (HYBRID_SYS.ipynb, HYBRID_SYS_TIME.ipynb): a joint-space via point (Angular(Time)Keypoint of a JointSpace(Time)PlannerSys
        # sub-system, own control penalty 1e-3) followed by a pose goal of a PosOrn(Time)PlannerSys sub-system, limits counted twice
        "C2h": dict(kind=capi.SYS_POS_ORN, nb_deriv=1, T=100, dt=0.05, B=256, seed=12, Qdiag=[[1] * 7, P], solver="recursive", nb_iter=12, hybrid=True,
                    psi=dict(kind="unitstep", K=2)),
        # the same with two limit sets, as when the sub-systems of a sequence are given different bounds (HYBRID_SYS_TIME.ipynb does that):
        # the second sub-system's bounds are 0.3 rad inside the first one's
        "C2hl": dict(kind=capi.SYS_POS_ORN, nb_deriv=1, T=100, dt=0.05, B=128, seed=14, Qdiag=[[1] * 7, P], solver="recursive", nb_iter=12, hybrid=True,
                     limits2=True),
        "C4h": dict(kind=capi.SYS_POS_ORN_TIME, nb_deriv=1, T=100, dt=None, B=128, seed=13, Qdiag=[[1] * 7 + [0], P + [.1]], ctimes=[2.5, 5.0],
                    solver="recursive", nb_iter=12, hybrid=True),
        "C4cp": dict(kind=capi.SYS_POS_ORN_TIME, nb_deriv=2, T=50, dt=None, B=64, seed=5, Qdiag=[P + V + [.1], P + V + [.1]],
                     ctimes=[2.5, 5.0], solver="batch_cp", nb_iter=8, psi=dict(kind="sawtooth+unitstep_dt", K=2)),
    }
    return dict(cfgs[name])


def make_batch(ctx: capi.Context, cfg: dict, B: int | None = None, seed: int | None = None, limits: str = "inactive", chain=None):
    """Returns (desc, inputs) where inputs = dict(q0, dq0, targets[list per keypoint], U0, [A, b, lambda0])."""
    B = int(B if B is not None else cfg["B"])
    seed = int(seed if seed is not None else cfg["seed"])
    if cfg["kind"] in (capi.SYS_JOINT, capi.SYS_JOINT_TIME):
        return _make_joint_batch(cfg, B, seed, limits)
    chain = chain or panda_chain()
    dof = chain["dof"]
    lo, up = chain["lower"], chain["upper"]
    rng = np.random.default_rng(seed)
    kind, nd, T = cfg["kind"], cfg["nb_deriv"], cfg["T"]
    tm = 1 if kind == capi.SYS_POS_ORN_TIME else 0
    nu = dof + tm
    kp_t = [T // 2 - 1, T - 1]
    if limits == "inactive":
        qmax = np.full(dof, 10 * np.pi)
        qmin = -qmax
    else:
        qmax, qmin = up.copy(), lo.copy()
    nx = nd * dof + tm
    smax, smin, w = np.zeros(nx), np.zeros(nx), np.zeros(nx, dtype=int)
    smax[:dof], smin[:dof], w[:dof] = qmax, qmin, 1
    if nd == 2:
        smax[dof:2 * dof], smin[dof:2 * dof], w[dof:2 * dof] = 10.0, -10.0, 1
    hyb = bool(cfg.get("hybrid"))
    desc = capi.make_desc(kind=kind, nb_deriv=nd, horizon=T, dt=cfg["dt"], R_diag=control_weights(cfg, nu), chain=chain, kp_timesteps=kp_t,
                          kp_Q=[np.diag(q) for q in cfg["Qdiag"]], limits=dict(state_max=smax, state_min=smin, limit_weight=w, penalty=1.0),
                          kp_dist=cfg.get("kp_dist"), kp_joint=[1, 0] if hyb else None, kp_Ru=[[1e-3] * nu, [1e-5] * nu] if hyb else None,
                          limit_multiplicity=(1 if cfg.get("limits2") else 2) if hyb else 1,
                          limits2=dict(state_max=smax - 0.3 * (w != 0), state_min=smin + 0.3 * (w != 0), limit_weight=w, penalty=1.0) if cfg.get("limits2") else None)
    q0 = np.clip(Q0_TUT[None, :] + rng.uniform(-0.3, 0.3, (B, dof)), lo, up)
    targets = []
    for i in range(2):
        qr = rng.uniform(lo, up, (B, dof))
        pos, quat, _ = ctx.fk_batch(desc, qr)
        cols = [pos, quat]
        if nd == 2:
            cols += [np.zeros((B, 3)), np.zeros((B, 4))]
        if tm:
            cols += [np.full((B, 1), cfg["ctimes"][i])]
        if hyb and i == 0:  # the via point is a joint configuration (+ its continuous time)
            cols = [qr] + ([np.full((B, 1), cfg["ctimes"][0])] if tm else [])
        targets.append(np.ascontiguousarray(np.hstack(cols)))
    U0 = np.zeros((B, T - 1, nu))
    if tm:
        U0[:, :, -1] = 0.01
    inp = dict(q0=q0, dq0=np.zeros((B, dof)), targets=targets, U0=U0, kp_t=kp_t, limits=dict(state_max=smax, state_min=smin, limit_weight=w))
    if cfg.get("al"):
        al = cfg["al"]
        A = np.zeros((1, nx + nu))
        A[0, al["row"]] = 1.0
        inp.update(A=A, b=np.array([al["bound"]]), lambda0=np.full((B, T - 1, 1), al["bound"]))  # tutorial: init multipliers = b
    return desc, inp


def _make_joint_batch(cfg, B, seed, limits):
    """JointSpace(Time)PlannerSys batches (Angular(Time)Keypoint targets); `dof` < 7 joints are padded to the device's 7."""
    rng = np.random.default_rng(seed)
    T, dofu, D = cfg["T"], cfg.get("dof", 7), 7
    tm = 1 if cfg["kind"] == capi.SYS_JOINT_TIME else 0
    n = D + tm
    kp_t = [T // 2 - 1, T - 1]
    lim = 10 * np.pi if limits == "inactive" else 2.0
    smax, smin, w = np.zeros(n), np.zeros(n), np.zeros(n, dtype=int)
    smax[:dofu], smin[:dofu], w[:dofu] = lim, -lim, 1
    chain = dict(dof=D, seg_joint=[], seg_xyz=[], seg_R=[], seg_axis=[])  # no kinematic chain: f(x) = x
    kp_Q = []
    for q in cfg["Qdiag"]:
        Q = np.zeros((n, n))
        Q[:dofu, :dofu] = np.diag(q[:dofu])
        if tm:
            Q[D, D] = q[-1]
        kp_Q.append(Q)
    desc = capi.make_desc(kind=cfg["kind"], nb_deriv=1, horizon=T, dt=cfg["dt"], R_diag=[1e-5] * n, chain=chain, kp_timesteps=kp_t, kp_Q=kp_Q,
                          limits=dict(state_max=smax, state_min=smin, limit_weight=w, penalty=1.0))
    q0 = np.zeros((B, D))
    base = np.asarray(cfg.get("q0", [0.0] * dofu), float)
    q0[:, :dofu] = base[None, :] + (rng.uniform(-0.3, 0.3, (B, dofu)) if B > 1 else 0.0)
    targets = []
    for i in range(2):
        t = np.zeros((B, n))
        t[:, :dofu] = rng.uniform(-2.5, 2.5, (B, dofu)) if not tm else rng.uniform(-1.0, 1.0, (B, dofu))
        if tm:
            t[:, D] = cfg["ctimes"][i]
        targets.append(t)
    U0 = np.zeros((B, T - 1, n))
    if tm:
        U0[:, :, -1] = 0.01  # as the time-system tutorials start (POS_ORN_TIME_SYS.ipynb cell 8)
    inp = dict(q0=q0, dq0=np.zeros((B, D)), targets=targets, U0=U0, kp_t=kp_t, dof=dofu, limits=dict(state_max=smax, state_min=smin, limit_weight=w))
    if cfg.get("al"):  # one inequality row on a joint (7-joint problems only: the constraint rows are not padded)
        al = cfg["al"]
        A = np.zeros((1, 2 * n))
        A[0, al["row"]] = 1.0
        inp.update(A=A, b=np.array([al["bound"]]), lambda0=np.full((B, T - 1, 1), al["bound"]))
    return desc, inp


def psi_of(spec: dict, T: int, n_u: int):
    """Control-primitive basis of a batch_cp config, built with the product's own basis builders (PyLQR.utils.primitives, the mirror
    of the reference's primitives.cpp) exactly as the tutorial cells combine them: kron(basis(T-1, K), I_nu)."""
    import sys

    pyl = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pylqr")
    if pyl not in sys.path:
        sys.path.insert(0, pyl)
    from PyLQR.utils import primitives

    build = dict(unitstep=primitives.build_psi_unitstep, sawtooth=primitives.build_psi_sawtooth, rbf=primitives.build_psi_RBF,
                 bernstein=primitives.build_psi_bernstein)
    K = spec["K"]
    if spec["kind"] in build:
        return np.kron(np.asarray(build[spec["kind"]](T - 1, K)), np.eye(n_u))
    if spec["kind"] == "sawtooth+unitstep_dt":  # controls on a sawtooth basis, sqrt(dt) on unit steps (POS_ORN_TIME_SYS.ipynb)
        a = np.diag([1.0] * (n_u - 1) + [0.0])
        b = np.diag([0.0] * (n_u - 1) + [1.0])
        return np.kron(np.asarray(primitives.build_psi_sawtooth(T - 1, K)), a) + np.kron(np.asarray(primitives.build_psi_unitstep(T - 1, K)), b)
    raise KeyError(spec["kind"])


def load_batch(ctx: capi.Context, desc, inp, B: int) -> capi.BatchProblem:
    p = capi.BatchProblem(ctx, desc, B)
    p.set_init_state(inp["q0"], inp["dq0"])
    for k, t in enumerate(inp["targets"]):
        p.set_keypoint_targets(k, t)
    p.set_controls(inp["U0"])
    if "A" in inp:
        p.set_constraints(inp["A"], inp["b"], inp["lambda0"])
    return p


def control_weights(cfg: dict, nu: int):
    """R of a workload: 1e-5 I unless the configuration names its own diagonal ("R_diag": per-joint weights; a time control keeps 1e-5)."""
    r = list(cfg.get("R_diag", []))
    return [float(r[i]) if i < len(r) else 1e-5 for i in range(nu)]


def run_solver(p: capi.BatchProblem, cfg: dict, nb_iter=None, early_stop=False, psi=None):
    n = int(nb_iter if nb_iter is not None else cfg["nb_iter"])
    if cfg["solver"] == "recursive":
        p.solve_recursive(n, True, early_stop)
    elif cfg["solver"] == "al":
        al = cfg["al"]
        p.solve_al(n, al["lag"], al["penalty"], al["scaling"], True, early_stop)
    else:
        p.solve_batch_cp(psi, n, early_stop)
    return n
