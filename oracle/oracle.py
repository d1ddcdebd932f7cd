"""ctypes front-end of the CPU oracle (oracle/ilqr_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by anything under ilqr_planner_amd/.

Also holds an independent URDF -> chain reader (xml.etree), restating what TinyURDFParser +
KDLRobot's constructor do (reference src/sim/KDLRobot.cpp:45-66): the product has its own reader in
C++ (ilqr_planner_amd/csrc/urdf_chain.cpp); the two are cross-checked in tests/.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
import xml.etree.ElementTree as ET

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libilqr_oracle.so")

MAX_SEG, MAX_DOF, MAX_NX, MAX_NU, MAX_NF, MAX_NQ, MAX_KP = 24, 7, 16, 8, 16, 14, 8
SYS_POS_ORN, SYS_POS_ORN_TIME, SYS_JOINT, SYS_JOINT_TIME = 0, 1, 2, 3


class Chain(C.Structure):
    _fields_ = [
        ("n_seg", C.c_int),
        ("dof", C.c_int),
        ("seg_joint", C.c_int * MAX_SEG),
        ("seg_xyz", (C.c_double * 3) * MAX_SEG),
        ("seg_R", (C.c_double * 9) * MAX_SEG),
        ("seg_axis", (C.c_double * 3) * MAX_SEG),
    ]


class Keypoint(C.Structure):
    _fields_ = [
        ("timestep", C.c_int),
        ("pos", C.c_double * 3),
        ("orn", C.c_double * 4),
        ("dpos", C.c_double * 3),
        ("dorn", C.c_double * 4),
        ("ctime", C.c_double),
        ("Q", C.c_double * (MAX_NQ * MAX_NQ)),
        ("dist", C.c_int),
        ("pos_radius", C.c_double),
        ("orn_thresh", C.c_double * 3),
        ("has_frame", C.c_int),
        ("fR", C.c_double * 9),
        ("fp", C.c_double * 3),
        ("has_Ru", C.c_int),
        ("Ru", C.c_double * MAX_NU),
        ("jt", C.c_double * MAX_NX),
        ("joint", C.c_int),
    ]


class System(C.Structure):
    _fields_ = [
        ("chain", Chain),
        ("kind", C.c_int),
        ("nb_deriv", C.c_int),
        ("T", C.c_int),
        ("dt", C.c_double),
        ("R_diag", C.c_double * MAX_NU),
        ("limits_set", C.c_int),
        ("lim_mult", C.c_int),
        ("penalty", C.c_double),
        ("state_max", C.c_double * MAX_NX),
        ("state_min", C.c_double * MAX_NX),
        ("limit_weight", C.c_int * MAX_NX),
        ("q0", C.c_double * MAX_DOF),
        ("dq0", C.c_double * MAX_DOF),
        ("n_kp", C.c_int),
        ("kp", Keypoint * MAX_KP),
        ("dof", C.c_int),
        ("n_x", C.c_int),
        ("n_u", C.c_int),
        ("n_f", C.c_int),
        ("n_Q", C.c_int),
        ("limits2_set", C.c_int),
        ("lim2_mult", C.c_int),
        ("sequence", C.c_int),
        ("state_max2", C.c_double * MAX_NX),
        ("state_min2", C.c_double * MAX_NX),
        ("limit_weight2", C.c_int * MAX_NX),
    ]


MAX_TRIALS = 16


class ProbeRec(C.Structure):
    """orc_probe_rec: what the reference's discontinuous decisions were taken on in one iteration (test aid)."""
    _fields_ = [("cost0", C.c_double), ("n_trials", C.c_int), ("trial_alpha", C.c_double * MAX_TRIALS), ("trial_cost", C.c_double * MAX_TRIALS),
                ("dun", C.c_double), ("mask_margin_in", C.c_double), ("mask_margin", C.c_double), ("clamp_margin", C.c_double), ("limit_margin_in", C.c_double),
                ("limit_margin", C.c_double)]


class Constraints(C.Structure):
    _fields_ = [("m", C.c_int), ("per_step", C.c_int), ("A", C.POINTER(C.c_double)), ("b", C.POINTER(C.c_double))]


def build(force: bool = False) -> str:
    """Compile oracle/_build/libilqr_oracle.so with the committed Makefile (gcc)."""
    src = os.path.join(_HERE, "ilqr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "ilqr_oracle.h"))
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.orc_system_finalize.argtypes = [C.POINTER(System)]
        L.orc_fk.argtypes = [C.POINTER(Chain), dp, dp, dp, dp, dp, dp, dp]
        L.orc_get_fx_jac.argtypes = [C.POINTER(System), dp, dp, dp]
        L.orc_kp_diff.argtypes = [C.POINTER(System), C.POINTER(Keypoint), dp, dp]
        L.orc_cost.argtypes = [C.POINTER(System), dp, dp, C.c_int]
        L.orc_cost.restype = C.c_double
        L.orc_cost_x.argtypes = [C.POINTER(System), dp, C.c_int, dp]
        L.orc_cost_xx.argtypes = [C.POINTER(System), dp, C.c_int, dp]
        L.orc_step.argtypes = [C.POINTER(System), dp, dp, dp, dp, dp, dp, dp]
        L.orc_solve_recursive.argtypes = [C.POINTER(System), dp, C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp]
        L.orc_solve_recursive.restype = C.c_int
        L.orc_solve_al.argtypes = [C.POINTER(System), C.POINTER(Constraints), dp, dp, C.c_int, C.c_int, C.c_double,
                                   C.c_double, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp]
        L.orc_solve_al.restype = C.c_int
        L.orc_solve_batch_cp.argtypes = [C.POINTER(System), dp, C.c_int, dp, C.c_int, C.c_int, dp, dp]
        L.orc_solve_batch_cp.restype = C.c_int
        for n in ("rbf", "bernstein", "unitstep", "sawtooth", "linear"):
            getattr(L, "orc_psi_" + n).argtypes = [C.c_int, C.c_int, dp]
        L.orc_inverse.argtypes = [C.c_int, dp, dp]
        L.orc_inverse.restype = C.c_int
        L.orc_set_variant.argtypes = [C.c_int]
        L.orc_set_variant.restype = None
        L.orc_set_probe.argtypes = [C.POINTER(ProbeRec), C.c_int]
        L.orc_set_probe.restype = None
        L.orc_set_resume.argtypes = [C.c_int, C.c_double, dp]
        L.orc_set_resume.restype = None
        L.orc_set_probe_all.argtypes = [C.c_int]
        L.orc_set_probe_all.restype = None
        L.orc_set_resume_x.argtypes = [dp]
        L.orc_set_resume_x.restype = None
        L.orc_get_resume_x_dev.argtypes = []
        L.orc_get_resume_x_dev.restype = C.c_double
        for n in ("orc_sd_logmap", "orc_sd_expmap"):
            getattr(L, n).argtypes = [dp, dp, dp]
        L.orc_sd_transport.argtypes = [dp, dp, dp, dp]
        L.orc_sd_distance.argtypes = [dp, dp]
        L.orc_sd_distance.restype = C.c_double
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _arr(x, n=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    if n is not None:
        assert a.size == n, (a.shape, n)
    return a


# ----------------------------------------------------------------------------- URDF -> chain


def rpy_to_R(r, p, y):
    """KDL Rotation::RPY(r,p,y) = Rz(y) Ry(p) Rx(r) (URDF convention)."""
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array(
        [
            [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
            [-sp, cp * sr, cp * cr],
        ]
    )


def chain_from_urdf(urdf_text: str, base: str, tip: str, tool_rpy=(0, 0, 0), tool_xyz=(0, 0, 0)):
    """Segments base->tip plus the user tool frame Frame(EulerZYX(rpy[0],rpy[1],rpy[2]), xyz)
    (reference src/sim/KDLRobot.cpp:61-66; note rpy[0] is used as the Z angle).
    Returns dict(seg_joint, seg_xyz, seg_R, seg_axis, dof, lower, upper)."""
    root = ET.fromstring(urdf_text)
    by_child = {}
    for j in root.findall("joint"):
        by_child[j.find("child").get("link")] = j
    path = []
    link = tip
    while link != base:
        if link not in by_child:
            raise RuntimeError(f"[KDLRobot] Unable to build kinematic chain from {base} to {tip}")
        j = by_child[link]
        path.append(j)
        link = j.find("parent").get("link")
    path.reverse()
    segs = {"seg_joint": [], "seg_xyz": [], "seg_R": [], "seg_axis": [], "lower": [], "upper": []}
    dof = 0
    for j in path:
        o = j.find("origin")
        xyz = [float(v) for v in (o.get("xyz", "0 0 0") if o is not None else "0 0 0").split()]
        rpy = [float(v) for v in (o.get("rpy", "0 0 0") if o is not None else "0 0 0").split()]
        ax = j.find("axis")
        axis = [float(v) for v in (ax.get("xyz") if ax is not None else "1 0 0").split()]
        typ = j.get("type")
        if typ in ("revolute", "continuous"):
            n = math.sqrt(sum(a * a for a in axis))
            axis = [a / n for a in axis]
            segs["seg_joint"].append(dof)
            lim = j.find("limit")
            segs["lower"].append(float(lim.get("lower")) if lim is not None and lim.get("lower") else -math.inf)
            segs["upper"].append(float(lim.get("upper")) if lim is not None and lim.get("upper") else math.inf)
            dof += 1
        elif typ == "fixed":
            segs["seg_joint"].append(-1)
        else:
            raise RuntimeError(f"unsupported joint type {typ}")
        segs["seg_xyz"].append(xyz)
        segs["seg_R"].append(rpy_to_R(*rpy).reshape(-1).tolist())
        segs["seg_axis"].append(axis)
    # tool frame: EulerZYX(a,b,c) = RPY(c,b,a)
    segs["seg_joint"].append(-1)
    segs["seg_xyz"].append(list(map(float, tool_xyz)))
    segs["seg_R"].append(rpy_to_R(float(tool_rpy[2]), float(tool_rpy[1]), float(tool_rpy[0])).reshape(-1).tolist())
    segs["seg_axis"].append([0.0, 0.0, 1.0])
    segs["dof"] = dof
    return segs


def make_chain(segs) -> Chain:
    c = Chain()
    n = len(segs["seg_joint"])
    assert n <= MAX_SEG
    c.n_seg, c.dof = n, segs["dof"]
    for i in range(n):
        c.seg_joint[i] = segs["seg_joint"][i]
        for k in range(3):
            c.seg_xyz[i][k] = segs["seg_xyz"][i][k]
            c.seg_axis[i][k] = segs["seg_axis"][i][k]
        for k in range(9):
            c.seg_R[i][k] = segs["seg_R"][i][k]
    return c


# ----------------------------------------------------------------------------- problem construction


def make_system(segs, kind, nb_deriv, T, dt, R_diag, keypoints, q0, dq0=None, qMax=None, qMin=None, dqMax=None, dqMin=None, lim_mult=1,
                limits2=None) -> System:
    """Mirror of the System constructors (reference src/system/System.cpp:19-75) + localInit.
    keypoints: list of dict(timestep,pos,orn,Q[,dpos,dorn][,ctime]); sorted by timestep here (System.cpp:82)."""
    s = System()
    s.chain = make_chain(segs)
    s.kind, s.nb_deriv, s.T, s.dt = kind, nb_deriv, T, float(dt if dt is not None else 0.0)
    s.lim_mult = int(lim_mult)  # SequentialSystem: number of sub-systems (each adds the limit terms)
    L = lib()
    L.orc_system_finalize(C.byref(s))
    dof = s.dof
    for i, v in enumerate(R_diag):
        s.R_diag[i] = v
    for i in range(dof):
        s.q0[i] = q0[i]
        s.dq0[i] = 0.0 if dq0 is None else dq0[i]
    if qMax is not None:
        s.limits_set, s.penalty = 1, 1.0
        ssz = nb_deriv * dof
        smax, smin, w = np.zeros(ssz), np.zeros(ssz), np.ones(ssz, dtype=int)
        if nb_deriv == 1:
            smax[:], smin[:] = qMax, qMin
        else:
            dM = np.zeros(dof) if dqMax is None else np.asarray(dqMax, float)
            dm = np.zeros(dof) if dqMin is None else np.asarray(dqMin, float)
            smax[:] = np.concatenate([qMax, dM])
            smin[:] = np.concatenate([qMin, dm])
            # Eigen isApprox: ||a-b||^2 <= prec^2 * min(||a||^2,||b||^2), prec 1e-12 (System.cpp:58-60)
            if np.sum((dM - dm) ** 2) <= 1e-24 * min(np.sum(dM**2), np.sum(dm**2)):
                w[dof:] = 0
        for i in range(ssz):
            s.state_max[i], s.state_min[i], s.limit_weight[i] = smax[i], smin[i], int(w[i])
        # time systems append a zero-weight entry (PosOrnTimePlannerSys.cpp:72-83): arrays are zero-initialised
    else:
        s.limits_set, s.penalty = 0, 0.0
    if limits2 is not None:  # second group of sub-systems of a sequence: dict(qMax, qMin[, mult]) (nb_deriv = 1)
        s.limits2_set, s.sequence, s.lim2_mult = 1, 1, int(limits2.get("mult", 1))
        for i in range(dof):
            s.state_max2[i], s.state_min2[i], s.limit_weight2[i] = limits2["qMax"][i], limits2["qMin"][i], 1
    kps = sorted(keypoints, key=lambda k: k["timestep"])
    s.n_kp = len(kps)
    nq = s.n_Q
    for i, k in enumerate(kps):
        kp = s.kp[i]
        kp.timestep = int(k["timestep"])
        joint_kp = bool(k.get("joint"))  # Angular(Time)Keypoint of a joint-space sub-system inside a PosOrn(Time) system (hybrid sequence)
        if kind in (SYS_JOINT, SYS_JOINT_TIME) or joint_kp:  # AngularKeypoint / AngularTimeKeypoint: joint vector (+ continuous time last)
            tgt = list(k["target"]) + ([float(k["ctime"])] if kind in (SYS_JOINT_TIME, SYS_POS_ORN_TIME) else [])
            for j, v in enumerate(tgt):
                kp.jt[j] = float(v)
            kp.joint = 1 if joint_kp else 0
        else:
            for j in range(3):
                kp.pos[j] = k["pos"][j]
                kp.dpos[j] = k.get("dpos", [0, 0, 0])[j]
            for j in range(4):
                kp.orn[j] = k["orn"][j]
                kp.dorn[j] = k.get("dorn", [0, 0, 0, 0])[j]
        kp.ctime = float(k.get("ctime", 0.0))
        nqk = s.n_x if joint_kp else nq
        Q = _arr(k["Q"]).reshape(nqk, nqk)
        for a in range(nqk):
            for b in range(nqk):
                kp.Q[a * nqk + b] = Q[a, b]
        if k.get("frame") is not None:  # 4x4 pose of the frame the keypoint's sub-system works in (TransformedSimulationInterface)
            Tm = _arr(k["frame"]).reshape(4, 4)
            kp.has_frame = 1
            for a in range(3):
                kp.fp[a] = Tm[a, 3]
                for b in range(3):
                    kp.fR[a * 3 + b] = Tm[a, b]
        if k.get("Ru") is not None:  # control penalty of the owning sub-system (SequentialSystem)
            kp.has_Ru = 1
            for a, v in enumerate(k["Ru"]):
                kp.Ru[a] = float(v)
        if k.get("dist") is not None:  # PosOrnKeypointDistFunct: dict(pos_radius, orn_thresh[3])
            kp.dist = 1
            kp.pos_radius = float(k["dist"]["pos_radius"])
            for j in range(3):
                kp.orn_thresh[j] = float(k["dist"]["orn_thresh"][j])
    return s


# ----------------------------------------------------------------------------- calls


def fk(chain: Chain, q, dq=None):
    dof = chain.dof
    q = _arr(q, dof)
    dq = _arr(dq if dq is not None else np.zeros(dof), dof)
    p, quat, J, dx, w = np.zeros(3), np.zeros(4), np.zeros(6 * MAX_DOF), np.zeros(3), np.zeros(3)
    lib().orc_fk(C.byref(chain), _dp(q), _dp(dq), _dp(p), _dp(quat), _dp(J), _dp(dx), _dp(w))
    return p, quat, J[: 6 * dof].reshape(6, dof), dx, w


def get_fx_jac(s: System, x):
    x = _arr(x, s.n_x)
    fx, J = np.zeros(s.n_f), np.zeros(s.n_Q * s.n_x)
    lib().orc_get_fx_jac(C.byref(s), _dp(x), _dp(fx), _dp(J))
    return fx, J.reshape(s.n_Q, s.n_x)


def cost(s: System, x, u, k):
    x, u = _arr(x, s.n_x), _arr(u, s.n_u)
    return lib().orc_cost(C.byref(s), _dp(x), _dp(u), k)


def cost_x(s: System, x, k):
    x = _arr(x, s.n_x)
    o = np.zeros(s.n_x)
    lib().orc_cost_x(C.byref(s), _dp(x), k, _dp(o))
    return o


def cost_xx(s: System, x, k):
    x = _arr(x, s.n_x)
    o = np.zeros(s.n_x * s.n_x)
    lib().orc_cost_xx(C.byref(s), _dp(x), k, _dp(o))
    return o.reshape(s.n_x, s.n_x)


def step(s: System, x, u):
    x, u = _arr(x, s.n_x), _arr(u, s.n_u)
    xn, fx = np.zeros(s.n_x), np.zeros(s.n_f)
    A, B, J = np.zeros(s.n_x**2), np.zeros(s.n_x * s.n_u), np.zeros(s.n_Q * s.n_x)
    lib().orc_step(C.byref(s), _dp(x), _dp(u), _dp(xn), _dp(fx), _dp(A), _dp(B), _dp(J))
    return xn, fx, A.reshape(s.n_x, s.n_x), B.reshape(s.n_x, s.n_u), J.reshape(s.n_Q, s.n_x)


class _Aids:
    """Arms the oracle's test aids around one solve: the decision-margin probe and / or a resume state (see ilqr_oracle.h)."""

    def __init__(self, nb_iter, probe, resume):
        self.buf = (ProbeRec * max(nb_iter, 1))() if probe else None
        self.cap = max(nb_iter, 1)
        self.all = probe == "all"  # also record the step sizes below the accepted one
        self.resume = resume
        self._keep = self._keep_x = None
        self.x_dev = None

    def __enter__(self):
        L = lib()
        if self.buf is not None:
            L.orc_set_probe(self.buf, self.cap)
            L.orc_set_probe_all(1 if self.all else 0)
        if self.resume:
            lm = self.resume.get("lambda_mask")
            self._keep = _arr(lm) if lm is not None else None
            L.orc_set_resume(int(self.resume.get("it0", 0)), float(self.resume.get("init_penalty", 0.0)), _dp(self._keep))
            if self.resume.get("X") is not None:  # the caller's own rollout of the controls handed over
                self._keep_x = _arr(self.resume["X"])
                L.orc_set_resume_x(_dp(self._keep_x))
        return self

    def __exit__(self, *exc):
        L = lib()
        if self._keep_x is not None:
            self.x_dev = float(L.orc_get_resume_x_dev())
        L.orc_set_probe(None, 0)
        L.orc_set_probe_all(0)
        L.orc_set_resume(0, 0.0, None)
        L.orc_set_resume_x(None)

    def records(self, n):
        if self.buf is None:
            return None
        out = []
        for i in range(n):
            r = self.buf[i]
            out.append(dict(cost0=r.cost0, alpha=list(r.trial_alpha[: r.n_trials]), cost=list(r.trial_cost[: r.n_trials]), dun=r.dun,
                            mask_margin_in=r.mask_margin_in, mask_margin=r.mask_margin, clamp_margin=r.clamp_margin, limit_margin_in=r.limit_margin_in,
                            limit_margin=r.limit_margin))
        return out


def solve_recursive(s: System, U0, nb_iter, line_search=True, early_stop=True, probe=False, resume=None):
    T, nx, nu, nf = s.T, s.n_x, s.n_u, s.n_f
    U0 = _arr(U0, (T - 1) * nu)
    X, fX, U = np.zeros((T, nx)), np.zeros((T, nf)), np.zeros((T - 1, nu))
    K, d = np.zeros((T - 1, nu, nx)), np.zeros((T - 1, nu))
    cost_ = np.zeros(1)
    tc, ta = np.full(max(nb_iter, 1), np.nan), np.full(max(nb_iter, 1), np.nan)
    with _Aids(nb_iter, probe, resume) as aids:
        n = lib().orc_solve_recursive(C.byref(s), _dp(U0), nb_iter, int(line_search), int(early_stop), _dp(X), _dp(fX), _dp(U),
                                      _dp(K), _dp(d), _dp(cost_), _dp(tc), _dp(ta))
    return dict(X=X, fX=fX, U=U, K=K, d=d, cost=float(cost_[0]), iters=n, trace_cost=tc[:n], trace_alpha=ta[:n], probe=aids.records(n), x_dev=aids.x_dev)


def solve_al(s: System, A, b, lambda0, U0, nb_iter, lag_update_step, penalty, scaling, line_search=True, early_stop=True, probe=False, resume=None):
    """A: (m, n_x+n_u) or (T-1, m, n_x+n_u); b likewise; lambda0: (T-1, m) (copied; returned updated).
    probe: also return the per-iteration decision margins; resume: dict(it0, init_penalty, lambda_mask) (test aids, ilqr_oracle.h)."""
    T, nx, nu, nf = s.T, s.n_x, s.n_u, s.n_f
    A, b = _arr(A), _arr(b)
    per_step = 1 if A.ndim == 3 else 0
    m = A.shape[-2]
    c = Constraints(m, per_step, _dp(A), _dp(b))
    lam = _arr(lambda0, (T - 1) * m).copy().reshape(T - 1, m)
    U0 = _arr(U0, (T - 1) * nu)
    X, fX, U = np.zeros((T, nx)), np.zeros((T, nf)), np.zeros((T - 1, nu))
    cost_ = np.zeros(1)
    tc, ta = np.full(max(nb_iter, 1), np.nan), np.full(max(nb_iter, 1), np.nan)
    with _Aids(nb_iter, probe, resume) as aids:
        n = lib().orc_solve_al(C.byref(s), C.byref(c), _dp(lam), _dp(U0), nb_iter, lag_update_step, penalty, scaling,
                               int(line_search), int(early_stop), _dp(X), _dp(fX), _dp(U), _dp(cost_), _dp(tc), _dp(ta))
    return dict(X=X, fX=fX, U=U, cost=float(cost_[0]), iters=n, trace_cost=tc[:n], trace_alpha=ta[:n], lam=lam, probe=aids.records(n), x_dev=aids.x_dev)


def solve_batch_cp(s: System, psi, u0, nb_iter, early_stop=True, probe=False):
    """probe: True = per-iteration record of what the backtracking decided on (pre-step cost, every trial's step size and cost,
    ||du||); "all" = the trials below the accepted step size as well (test aids, ilqr_oracle.h)."""
    T, nu = s.T, s.n_u
    psi = _arr(psi)
    assert psi.shape[0] == (T - 1) * nu
    u = _arr(u0, (T - 1) * nu).copy()
    tc, ta = np.full(max(nb_iter, 1), np.nan), np.full(max(nb_iter, 1), np.nan)
    with _Aids(nb_iter, probe, None) as aids:
        n = lib().orc_solve_batch_cp(C.byref(s), _dp(psi), psi.shape[1], _dp(u), nb_iter, int(early_stop), _dp(tc), _dp(ta))
    return dict(u=u, iters=n, trace_cost=tc[:n], trace_alpha=ta[:n], probe=aids.records(n))


def solve_batch(s: System, u0, nb_iter, early_stop=True, probe=False):
    """BatchILQR::solve (reference src/solver/BatchILQR.cpp:110-173).  The file differs from BatchILQRCP.cpp only in the absence
    of PSI (lstq_A = Su'(J'QJ+L)Su + R, du = lstq_A^-1 lstq_B): it is the control-primitive solver with PSI = I, run as such."""
    return solve_batch_cp(s, np.eye((s.T - 1) * s.n_u), u0, nb_iter, early_stop, probe)


def psi(kind: str, dim: int, K: int):
    out = np.zeros((dim, 2 * K if kind == "linear" else K))
    getattr(lib(), "orc_psi_" + kind)(dim, K, _dp(out))
    return out


def inverse(A):
    A = _arr(A)
    n = A.shape[0]
    o = np.zeros((n, n))
    lib().orc_inverse(n, _dp(A), _dp(o))
    return o


def set_variant(v: int):
    """Test aid, algebraically neutral variants of the restated arithmetic: bit 0 makes the backward sweep use Qxu := Qux^T (equal to
    A'PB in exact arithmetic), bit 1 inverts by the pivot-free symmetric sweep operator instead of partial-pivot LU, bit 2 accumulates
    every product with fused multiply-adds."""
    lib().orc_set_variant(int(v))
