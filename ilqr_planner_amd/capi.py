"""ctypes binding of the C ABI in include/ilqr_hip.h (libilqr_hip.so, hand-written HIP for gfx950).

This is plumbing: every computation happens inside the shared library on the GPU.  There is no CPU fallback --
if the library is missing or no HIP device is visible, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libilqr_hip.so")

MAX_SEG, MAX_KP, MAX_NX, MAX_NU, MAX_NF, MAX_NQ = 24, 8, 15, 8, 15, 13
SYS_POS_ORN, SYS_POS_ORN_TIME, SYS_JOINT, SYS_JOINT_TIME = 0, 1, 2, 3
STATUS_OK, STATUS_NONFINITE, STATUS_ALPHA_FLOOR = 0, 1, 2
PROF_ROLLOUT, PROF_BACKWARD, PROF_FORWARD, PROF_OTHER, PROF_APPLY = 0, 1, 2, 3, 4

# every symbol include/ilqr_hip.h declares (checked by tests/test_capi_symbols.py)
EXPORTS = [
    "ilqr_dims_of", "ilqr_desc_defaults", "ilqr_ctx_create", "ilqr_ctx_destroy", "ilqr_last_error", "ilqr_ctx_set_stream",
    "ilqr_ctx_synchronize", "ilqr_version", "ilqr_problem_create", "ilqr_problem_destroy", "ilqr_problem_set_init_state",
    "ilqr_problem_set_keypoint_targets", "ilqr_problem_set_controls", "ilqr_problem_set_constraints",
    "ilqr_problem_set_init_state_dev", "ilqr_problem_set_keypoint_targets_dev", "ilqr_problem_set_controls_dev",
    "ilqr_solve_recursive", "ilqr_solve_al", "ilqr_solve_batch_cp", "ilqr_solve_batch", "ilqr_problem_get_X", "ilqr_problem_get_fX",
    "ilqr_problem_get_U", "ilqr_problem_get_K", "ilqr_problem_get_d", "ilqr_problem_get_cost", "ilqr_problem_get_alpha",
    "ilqr_problem_get_iters", "ilqr_problem_get_status", "ilqr_problem_get_lambda", "ilqr_problem_get_trace",
    "ilqr_problem_get_X_dev", "ilqr_problem_get_U_dev", "ilqr_problem_get_cost_dev", "ilqr_fk_batch",
    "ilqr_profile_enable", "ilqr_profile_reset", "ilqr_profile_get", "ilqr_chain_from_urdf", "ilqr_urdf_last_error",
    "ilqr_problem_reset_multipliers", "ilqr_problem_warm_start", "ilqr_problem_track", "ilqr_problem_track_dev", "ilqr_ctx_set_split", "ilqr_ctx_set_crosscheck",
]


class ProblemDesc(C.Structure):
    """Mirror of ilqr_problem_desc."""

    _fields_ = [
        ("kind", C.c_int),
        ("nb_deriv", C.c_int),
        ("dof", C.c_int),
        ("horizon", C.c_int),
        ("dt", C.c_double),
        ("R_diag", C.c_double * MAX_NU),
        ("limits_set", C.c_int),
        ("penalty", C.c_double),
        ("state_max", C.c_double * (MAX_NX + 1)),
        ("state_min", C.c_double * (MAX_NX + 1)),
        ("limit_weight", C.c_int * (MAX_NX + 1)),
        ("n_seg", C.c_int),
        ("seg_joint", C.c_int * MAX_SEG),
        ("seg_xyz", (C.c_double * 3) * MAX_SEG),
        ("seg_R", (C.c_double * 9) * MAX_SEG),
        ("seg_axis", (C.c_double * 3) * MAX_SEG),
        ("n_kp", C.c_int),
        ("kp_timestep", C.c_int * MAX_KP),
        ("kp_Q", (C.c_double * (MAX_NQ * MAX_NQ)) * MAX_KP),
        ("kp_dist", C.c_int * MAX_KP),
        ("kp_pos_radius", C.c_double * MAX_KP),
        ("kp_orn_thresh", (C.c_double * 3) * MAX_KP),
        ("kp_has_frame", C.c_int * MAX_KP),
        ("kp_frame_R", (C.c_double * 9) * MAX_KP),
        ("kp_frame_p", (C.c_double * 3) * MAX_KP),
        ("kp_has_Ru", C.c_int * MAX_KP),
        ("kp_Ru", (C.c_double * MAX_NU) * MAX_KP),
        ("kp_joint", C.c_int * MAX_KP),
        ("limit_multiplicity", C.c_int),
        ("is_sequence", C.c_int),
        ("limits2_set", C.c_int),
        ("penalty2", C.c_double),
        ("state_max2", C.c_double * (MAX_NX + 1)),
        ("state_min2", C.c_double * (MAX_NX + 1)),
        ("limit_weight2", C.c_int * (MAX_NX + 1)),
        ("limit_multiplicity2", C.c_int),
        ("reg", C.c_double),
        ("alpha_floor", C.c_double),
        ("stop_tol", C.c_double),
    ]


class Dims(C.Structure):
    _fields_ = [("n_x", C.c_int), ("n_u", C.c_int), ("n_f", C.c_int), ("n_Q", C.c_int)]


_lib = None


def load():
    """dlopen libilqr_hip.so and declare the prototypes.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    L.ilqr_version.restype = C.c_char_p
    L.ilqr_last_error.restype = C.c_char_p
    L.ilqr_last_error.argtypes = [vp]
    L.ilqr_desc_defaults.argtypes = [C.POINTER(ProblemDesc)]
    L.ilqr_desc_defaults.restype = None
    L.ilqr_dims_of.argtypes = [C.POINTER(ProblemDesc), C.POINTER(Dims)]
    L.ilqr_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.ilqr_ctx_destroy.argtypes = [vp]
    L.ilqr_ctx_destroy.restype = None
    L.ilqr_ctx_set_stream.argtypes = [vp, vp]
    L.ilqr_ctx_synchronize.argtypes = [vp]
    L.ilqr_ctx_set_split.argtypes = [vp, C.c_int]
    L.ilqr_ctx_set_crosscheck.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.ilqr_problem_create.argtypes = [vp, C.POINTER(ProblemDesc), C.c_int, C.POINTER(vp)]
    L.ilqr_problem_destroy.argtypes = [vp]
    L.ilqr_problem_destroy.restype = None
    for n in ("ilqr_problem_set_init_state", "ilqr_problem_set_init_state_dev"):
        getattr(L, n).argtypes = [vp, vp, vp]
    for n in ("ilqr_problem_set_keypoint_targets", "ilqr_problem_set_keypoint_targets_dev"):
        getattr(L, n).argtypes = [vp, C.c_int, vp]
    for n in ("ilqr_problem_set_controls", "ilqr_problem_set_controls_dev"):
        getattr(L, n).argtypes = [vp, vp]
    L.ilqr_problem_set_constraints.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp]
    L.ilqr_problem_reset_multipliers.argtypes = [vp]
    L.ilqr_problem_warm_start.argtypes = [vp, C.c_int]
    L.ilqr_problem_track.argtypes = [vp, C.c_int, dp, C.c_int, dp]
    L.ilqr_problem_track_dev.argtypes = [vp, C.c_int, vp, C.c_int, vp]
    L.ilqr_solve_recursive.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.ilqr_solve_al.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
    L.ilqr_solve_batch_cp.argtypes = [vp, dp, C.c_int, C.c_int, C.c_int]
    L.ilqr_solve_batch.argtypes = [vp, C.c_int, C.c_int]
    for n in ("X", "fX", "U", "K", "d", "cost", "alpha", "lambda", "X_dev", "U_dev", "cost_dev"):
        getattr(L, "ilqr_problem_get_" + n).argtypes = [vp, vp]
    L.ilqr_problem_get_iters.argtypes = [vp, ip]
    L.ilqr_problem_get_status.argtypes = [vp, ip]
    L.ilqr_problem_get_trace.argtypes = [vp, dp, dp, C.c_int]
    L.ilqr_fk_batch.argtypes = [vp, C.POINTER(ProblemDesc), C.c_int, dp, dp, dp, dp]
    L.ilqr_profile_enable.argtypes = [vp, C.c_int]
    L.ilqr_profile_reset.argtypes = [vp]
    L.ilqr_profile_get.argtypes = [vp, C.c_int, dp, ip]
    L.ilqr_chain_from_urdf.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, dp, dp, C.POINTER(ProblemDesc), dp, dp]
    L.ilqr_urdf_last_error.restype = C.c_char_p
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _f64(x, shape=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


def chain_from_urdf(urdf_text: str, base: str, tip: str, tool_rpy=None, tool_xyz=None):
    """URDF text -> chain dict (seg_joint, seg_xyz, seg_R, seg_axis, dof, lower, upper) via the C++ reader in the library."""
    L = load()
    d = ProblemDesc()
    L.ilqr_desc_defaults(C.byref(d))
    lo, up = np.zeros(MAX_SEG), np.zeros(MAX_SEG)
    rpy = _f64(tool_rpy, (3,)) if tool_rpy is not None else None
    xyz = _f64(tool_xyz, (3,)) if tool_xyz is not None else None
    if L.ilqr_chain_from_urdf(urdf_text.encode(), base.encode(), tip.encode(), _dp(rpy), _dp(xyz), C.byref(d), _dp(lo), _dp(up)):
        raise RuntimeError(L.ilqr_urdf_last_error().decode())
    n = d.n_seg
    return dict(dof=d.dof, seg_joint=[d.seg_joint[i] for i in range(n)], seg_xyz=[list(d.seg_xyz[i]) for i in range(n)],
                seg_R=[list(d.seg_R[i]) for i in range(n)], seg_axis=[list(d.seg_axis[i]) for i in range(n)],
                lower=lo[: d.dof].copy(), upper=up[: d.dof].copy())


def make_desc(*, kind, nb_deriv, horizon, dt, R_diag, chain, kp_timesteps, kp_Q, limits=None, kp_dist=None, kp_frames=None, kp_Ru=None,
              limit_multiplicity=1, kp_joint=None, limits2=None) -> ProblemDesc:
    """chain: dict(seg_joint, seg_xyz, seg_R, seg_axis, dof); limits: dict(state_max, state_min, limit_weight, penalty) or None."""
    L = load()
    d = ProblemDesc()
    L.ilqr_desc_defaults(C.byref(d))
    d.kind, d.nb_deriv, d.dof, d.horizon, d.dt = int(kind), int(nb_deriv), int(chain["dof"]), int(horizon), float(dt or 0.0)
    for i, v in enumerate(R_diag):
        d.R_diag[i] = float(v)
    n = len(chain["seg_joint"])
    if n > MAX_SEG:
        raise ValueError("too many segments")
    d.n_seg = n
    for i in range(n):
        d.seg_joint[i] = int(chain["seg_joint"][i])
        for k in range(3):
            d.seg_xyz[i][k] = float(chain["seg_xyz"][i][k])
            d.seg_axis[i][k] = float(chain["seg_axis"][i][k])
        for k in range(9):
            d.seg_R[i][k] = float(chain["seg_R"][i][k])
    dims = Dims()
    if L.ilqr_dims_of(C.byref(d), C.byref(dims)):
        raise ValueError("unsupported system kind / nb_deriv")
    if limits is not None:
        d.limits_set, d.penalty = 1, float(limits.get("penalty", 1.0))
        for i in range(len(limits["state_max"])):
            d.state_max[i] = float(limits["state_max"][i])
            d.state_min[i] = float(limits["state_min"][i])
            d.limit_weight[i] = int(limits["limit_weight"][i])
    d.n_kp = len(kp_timesteps)
    nq = dims.n_Q
    for k, (ts, Q) in enumerate(zip(kp_timesteps, kp_Q)):
        d.kp_timestep[k] = int(ts)
        jk = bool(kp_joint and kp_joint[k])  # Angular(Time)Keypoint of a joint-space sub-system (hybrid sequence): n_x x n_x precision
        d.kp_joint[k] = int(jk)
        nqk = dims.n_x if jk else nq
        Q = _f64(Q, (nqk, nqk))
        for a in range(nqk):
            for b in range(nqk):
                d.kp_Q[k][a * nqk + b] = Q[a, b]
    d.limit_multiplicity = int(limit_multiplicity)  # SequentialSystem: number of sub-systems
    if limits2 is not None:  # second group of sub-systems with other bounds: dict(state_max, state_min, limit_weight[, penalty, multiplicity])
        d.is_sequence, d.limits2_set, d.penalty2 = 1, 1, float(limits2.get("penalty", 1.0))
        d.limit_multiplicity2 = int(limits2.get("multiplicity", 1))
        for i in range(len(limits2["state_max"])):
            d.state_max2[i] = float(limits2["state_max"][i])
            d.state_min2[i] = float(limits2["state_min"][i])
            d.limit_weight2[i] = int(limits2["limit_weight"][i])
    for k, fr in enumerate(kp_frames or []):  # 4x4 pose of the object frame of keypoint k's sub-system (TransformedSimulationInterface) or None
        if fr is not None:
            Tm = _f64(fr, (4, 4))
            d.kp_has_frame[k] = 1
            for a in range(3):
                d.kp_frame_p[k][a] = Tm[a, 3]
                for b in range(3):
                    d.kp_frame_R[k][a * 3 + b] = Tm[a, b]
    for k, ru in enumerate(kp_Ru or []):  # control penalty of keypoint k's own sub-system (SequentialSystem) or None
        if ru is not None:
            d.kp_has_Ru[k] = 1
            for a, v in enumerate(ru):
                d.kp_Ru[k][a] = float(v)
    for k, kd in enumerate(kp_dist or []):  # PosOrnKeypointDistFunct: None or dict(pos_radius=..., orn_thresh=[3])
        if kd is not None:
            d.kp_dist[k] = 1
            d.kp_pos_radius[k] = float(kd["pos_radius"])
            for i in range(3):
                d.kp_orn_thresh[k][i] = float(kd["orn_thresh"][i])
    return d


class Context:
    def __init__(self, device_id: int = 0):
        import weakref

        self.L = load()
        self.h = C.c_void_p()
        self._problems = weakref.WeakSet()
        rc = self.L.ilqr_ctx_create(device_id, C.byref(self.h))
        if rc:
            raise RuntimeError(f"ilqr_ctx_create failed (code {rc}): no usable HIP device {device_id}; there is no CPU fallback")

    def check(self, rc):
        if rc:
            raise RuntimeError(self.L.ilqr_last_error(self.h).decode())

    def set_stream(self, stream_ptr):
        self.check(self.L.ilqr_ctx_set_stream(self.h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        self.check(self.L.ilqr_ctx_synchronize(self.h))

    def set_crosscheck(self, generic_kernels=False, cp_lane_solve=False, cp_general=False, mfma_sweep=0):
        """Cross-check kernel variants (ilqr_ctx_set_crosscheck); context state, in force until changed."""
        self.check(self.L.ilqr_ctx_set_crosscheck(self.h, int(bool(generic_kernels)), int(bool(cp_lane_solve)), int(bool(cp_general)), int(mfma_sweep)))

    def crosscheck_from_env(self):
        """TEST PLUMBING of this Python wrapper (the library itself reads no environment variable): the parity tests select the cross-check
        variants per test case through ILQR_HIP_PATH=v1, ILQR_CP_SOLVE=lane, ILQR_CP=general, ILQR_SWEEP=mfma|rows, ILQR_FWD=wg|dpp, ILQR_APPLY=rows|dpp; every solve of BatchProblem passes them on."""
        self.set_crosscheck(os.environ.get("ILQR_HIP_PATH") == "v1", os.environ.get("ILQR_CP_SOLVE") == "lane", os.environ.get("ILQR_CP") == "general",
                            {"mfma": 1, "rows": 2}.get(os.environ.get("ILQR_SWEEP"), 0) + 4 * {"wg": 1, "dpp": 2}.get(os.environ.get("ILQR_FWD"), 0)
                            + 16 * {"rows": 1, "dpp": 2}.get(os.environ.get("ILQR_APPLY"), 0))

    def set_split(self, on: bool):
        """Two-stream solve of large batches on / off (ilqr_ctx_set_split); off = one kernel at a time, for profiler runs."""
        self.check(self.L.ilqr_ctx_set_split(self.h, int(on)))

    def profile(self, on: bool):
        self.check(self.L.ilqr_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self.check(self.L.ilqr_profile_reset(self.h))

    def profile_get(self, which):
        ms, n = C.c_double(), C.c_int()
        self.check(self.L.ilqr_profile_get(self.h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def fk_batch(self, desc: ProblemDesc, q):
        q = _f64(q)
        n, dof = q.shape
        pos, quat, jac = np.zeros((n, 3)), np.zeros((n, 4)), np.zeros((n, 6, dof))
        self.check(self.L.ilqr_fk_batch(self.h, C.byref(desc), n, _dp(q), _dp(pos), _dp(quat), _dp(jac)))
        return pos, quat, jac

    def close(self):
        if self.h:
            for p in list(self._problems):
                p.close()
            self.L.ilqr_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchProblem:
    """B instances of one lowered System (ilqr_problem)."""

    def __init__(self, ctx: Context, desc: ProblemDesc, batch: int):
        self.ctx, self.L, self.desc, self.B = ctx, ctx.L, desc, int(batch)
        self.dims = Dims()
        self.L.ilqr_dims_of(C.byref(desc), C.byref(self.dims))
        self.T = desc.horizon
        self.h = C.c_void_p()
        ctx.check(self.L.ilqr_problem_create(ctx.h, C.byref(desc), self.B, C.byref(self.h)))
        ctx._problems.add(self)
        self.m = 0

    # ---- inputs (host arrays)
    def set_init_state(self, q0, dq0=None):
        q0 = _f64(q0, (self.B, self.desc.dof))
        dq0 = _f64(dq0, (self.B, self.desc.dof)) if dq0 is not None else None
        self.ctx.check(self.L.ilqr_problem_set_init_state(self.h, q0.ctypes.data, dq0.ctypes.data if dq0 is not None else None))

    def set_keypoint_targets(self, k, target):
        t = _f64(target, (self.B, self.dims.n_f))
        self.ctx.check(self.L.ilqr_problem_set_keypoint_targets(self.h, k, t.ctypes.data))

    def set_controls(self, U0):
        U0 = _f64(U0, (self.B, self.T - 1, self.dims.n_u))
        self.ctx.check(self.L.ilqr_problem_set_controls(self.h, U0.ctypes.data))

    def set_constraints(self, A, b, lambda0=None):
        A, b = _f64(A), _f64(b)
        per_step = 1 if A.ndim == 3 else 0
        self.m = A.shape[-2]
        lam = _f64(lambda0, (self.B, self.T - 1, self.m)) if lambda0 is not None else None
        self.ctx.check(self.L.ilqr_problem_set_constraints(self.h, self.m, per_step, _dp(A), _dp(b), _dp(lam)))

    def reset_multipliers(self):
        self.ctx.check(self.L.ilqr_problem_reset_multipliers(self.h))

    # ---- inputs (device pointers, e.g. tensor.data_ptr())
    def set_init_state_dev(self, q0_ptr, dq0_ptr=None):
        self.ctx.check(self.L.ilqr_problem_set_init_state_dev(self.h, q0_ptr, dq0_ptr))

    def set_keypoint_targets_dev(self, k, ptr):
        self.ctx.check(self.L.ilqr_problem_set_keypoint_targets_dev(self.h, k, ptr))

    def set_controls_dev(self, ptr):
        self.ctx.check(self.L.ilqr_problem_set_controls_dev(self.h, ptr))

    # ---- solvers (asynchronous on the context's stream)
    def solve_recursive(self, nb_iter, line_search=True, early_stop=True):
        self.ctx.crosscheck_from_env()
        self.ctx.check(self.L.ilqr_solve_recursive(self.h, nb_iter, int(line_search), int(early_stop)))

    def solve_al(self, nb_iter, lag_update_step, penalty, scaling_factor, line_search=True, early_stop=True):
        self.ctx.crosscheck_from_env()
        self.ctx.check(self.L.ilqr_solve_al(self.h, nb_iter, lag_update_step, penalty, scaling_factor, int(line_search), int(early_stop)))

    def solve_batch_cp(self, psi, nb_iter, early_stop=True):
        psi = _f64(psi)
        assert psi.shape[0] == (self.T - 1) * self.dims.n_u
        self.ctx.crosscheck_from_env()
        self.ctx.check(self.L.ilqr_solve_batch_cp(self.h, _dp(psi), psi.shape[1], nb_iter, int(early_stop)))

    def solve_batch(self, nb_iter, early_stop=True):
        """BatchILQR::solve: the batch solver on the full control sequence (identity basis)."""
        self.ctx.check(self.L.ilqr_solve_batch(self.h, nb_iter, int(early_stop)))

    # ---- results
    def _get(self, name, shape):
        o = np.zeros(shape)
        self.ctx.check(getattr(self.L, "ilqr_problem_get_" + name)(self.h, o.ctypes.data))
        return o

    def X(self):
        return self._get("X", (self.B, self.T, self.dims.n_x))

    def fX(self):
        return self._get("fX", (self.B, self.T, self.dims.n_f))

    def U(self):
        return self._get("U", (self.B, self.T - 1, self.dims.n_u))

    def K(self):
        return self._get("K", (self.B, self.T - 1, self.dims.n_u, self.dims.n_x))

    def d(self):
        return self._get("d", (self.B, self.T - 1, self.dims.n_u))

    def cost(self):
        return self._get("cost", (self.B,))

    def alpha(self):
        return self._get("alpha", (self.B,))

    def lam(self):
        return self._get("lambda", (self.B, self.T - 1, self.m))

    def iters(self):
        o = np.zeros(self.B, dtype=np.int32)
        self.ctx.check(self.L.ilqr_problem_get_iters(self.h, o.ctypes.data_as(C.POINTER(C.c_int))))
        return o

    def status(self):
        o = np.zeros(self.B, dtype=np.int32)
        self.ctx.check(self.L.ilqr_problem_get_status(self.h, o.ctypes.data_as(C.POINTER(C.c_int))))
        return o

    def trace(self, nb_iter):
        ct, at = np.zeros((self.B, nb_iter)), np.zeros((self.B, nb_iter))
        self.ctx.check(self.L.ilqr_problem_get_trace(self.h, _dp(ct), _dp(at), nb_iter))
        return ct, at

    def get_X_dev(self, ptr):
        self.ctx.check(self.L.ilqr_problem_get_X_dev(self.h, ptr))

    def get_U_dev(self, ptr):
        self.ctx.check(self.L.ilqr_problem_get_U_dev(self.h, ptr))

    def get_cost_dev(self, ptr):
        self.ctx.check(self.L.ilqr_problem_get_cost_dev(self.h, ptr))

    # ---- receding horizon / tracking
    def warm_start(self, shift: int = 0):
        self.ctx.check(self.L.ilqr_problem_warm_start(self.h, int(shift)))

    def track(self, k: int, x_meas, with_feedforward: bool = False):
        x = _f64(x_meas, (self.B, self.dims.n_x))
        u = np.empty((self.B, self.dims.n_u))
        self.ctx.check(self.L.ilqr_problem_track(self.h, int(k), _dp(x), int(bool(with_feedforward)), _dp(u)))
        return u

    def close(self):
        if self.h:
            if self.ctx.h:  # a destroyed context has already destroyed its problems
                self.L.ilqr_problem_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
