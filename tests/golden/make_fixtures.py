#!/usr/bin/env python3
"""Generate the golden fixtures from the reference's tutorial notebooks (run in the dev container only;
/root/reference does not exist on the GPU box -- the generated files are committed).

Outputs (data only -- inputs and expected outputs, no reference source text):
  tests/golden/traces.json       problem definitions (literals of the notebooks' code cells, transcribed below)
                                 + the per-iteration "Iteration i, Cost: c, alpha= a" stream stored in the
                                 notebooks' outputs (the reference's only known-answer data)
  tests/golden/panda_chain.urdf  the kinematic skeleton (links, joints, origins, axes, limits) of
                                 pylqr_planner/Tutorials/model.urdf -- visuals, collisions, inertias, meshes dropped
"""
import json
import os
import re
import xml.etree.ElementTree as ET

REF = "/root/reference/pylqr_planner/Tutorials"
OUT = os.path.dirname(os.path.abspath(__file__))

LINE = re.compile(r"Iteration (\d+), Cost: (\S+), alpha= ([^,\s]+)")


def traces_of(nb_name):
    """Return, in notebook order, one list of (cost, alpha) per solve cell."""
    nb = json.load(open(os.path.join(REF, nb_name)))
    out = []
    for c in nb["cells"]:
        if c["cell_type"] != "code" or ".solve(" not in "".join(c["source"]):
            continue
        txt = "".join("".join(o.get("text", "")) for o in c.get("outputs", []) if o.get("output_type") == "stream")
        rows = []
        for m in LINE.finditer(txt):
            cost = m.group(2)
            rows.append([None if "nan" in cost else float(cost), float(m.group(3))])
        out.append({"call": [l for l in "".join(c["source"]).splitlines() if ".solve(" in l][0].strip(), "trace": rows})
    return out


# ---- literals common to the PosOrn tutorials (code cells 4 and 6 of each notebook)
Q0_TUT = [0.62991112, -0.2329776, -0.01423721, -1.70254115, 0.06251303, 1.50592777, 0.71771416]
KP1 = dict(pos=[0.554121212377707, -0.01575049935289518, 0.38295604872511507],
           orn=[0.014042440828406944, 0.915047647731553, 0.4024820607528928, 0.022333898196169735])
KP2 = dict(pos=[0.254121212377707, -0.07575049935289518, 0.13170744424127526],
           orn=[0.029927010072216945, 0.9121514607332729, 0.4087591864532181, 0.00011933313484481926])
PI10 = 31.41592653589793  # np.pi*10


def problem(kind, nb_deriv, T, dt, q0, q1diag, q2diag, ctimes=None, dq_limits=False, u0_last=0.0):
    kps = []
    for kp, qd, ts, i in ((KP1, q1diag, T // 2 - 1, 0), (KP2, q2diag, T - 1, 1)):
        d = dict(timestep=ts, pos=kp["pos"], orn=kp["orn"], Qdiag=qd)
        if nb_deriv == 2:
            d.update(dpos=[0, 0, 0], dorn=[0, 0, 0, 0])
        if ctimes:
            d["ctime"] = ctimes[i]
        kps.append(d)
    dof = 7
    nu = dof + (1 if kind == "POS_ORN_TIME" else 0)
    return dict(kind=kind, nb_deriv=nb_deriv, T=T, dt=dt, q0=q0, dq0=[0] * dof, R_diag=[1e-5] * nu,
                qMax=[PI10] * dof, qMin=[-PI10] * dof,
                dqMax=[10.0] * dof if dq_limits else None, dqMin=[-10.0] * dof if dq_limits else None,
                keypoints=kps, u0_step=[0.0] * (nu - 1) + [u0_last],
                base="panda_link0", tip="panda_tip")


P = [1, 1, 1, .1, .1, .1]
cases = {}

t = traces_of("POS_ORN_SYS.ipynb")
cases["POS_ORN_SYS"] = dict(
    problem=problem("POS_ORN", 1, 100, 0.1, Q0_TUT, P, P),
    solves=[dict(solver="BatchILQRCP", psi=dict(kind="unitstep", K=2), nb_iter=10, early_stop=True, **t[0]),
            dict(solver="ILQRRecursive", nb_iter=10, line_search=True, early_stop=True, **t[1]),
            dict(solver="BatchILQR", nb_iter=10, early_stop=True, **t[2])])

t = traces_of("POS_ORN_SYS_AL_ILQR.ipynb")
cases["POS_ORN_SYS_AL_ILQR"] = dict(
    problem=problem("POS_ORN", 1, 400, 0.01, Q0_TUT, P, P, dq_limits=True),
    solves=[dict(solver="ILQRRecursive", nb_iter=10, line_search=True, early_stop=True, **t[0]),
            # Constraint A 14x14 zero except A[5,5]=1 ; b zero except b[5]=2.0 ; init multipliers = b (cell 12)
            dict(solver="AL_ILQR", A_nonzero=[[5, 5, 1.0]], b_nonzero=[[5, 2.0]], m=14, lambda0="b",
                 nb_iter=100, lag_update_step=5, penalty=0.25, scaling_factor=1.1, line_search=True, early_stop=True, **t[1])])

t = traces_of("POS_ORN_SYS_2ND.ipynb")
cases["POS_ORN_SYS_2ND"] = dict(
    problem=problem("POS_ORN", 2, 400, 0.01, Q0_TUT, P + [1, 1, 1, 0, 0, 0], P + [1, 1, 1, .1, .1, .1], dq_limits=True),
    solves=[dict(solver="BatchILQRCP", psi=dict(kind="sawtooth", K=2), nb_iter=10, early_stop=True, **t[0]),
            dict(solver="ILQRRecursive", nb_iter=15, line_search=True, early_stop=True, **t[1])])

t = traces_of("POS_ORN_TIME_SYS.ipynb")
cases["POS_ORN_TIME_SYS"] = dict(
    problem=problem("POS_ORN_TIME", 1, 100, None, [0.0] * 7, P + [0], P + [.1], ctimes=[2, 5], dq_limits=True, u0_last=0.01),
    solves=[dict(solver="BatchILQRCP", psi=dict(kind="unitstep", K=2), nb_iter=20, early_stop=True, **t[0]),
            dict(solver="ILQRRecursive", nb_iter=20, line_search=True, early_stop=True, **t[1]),
            dict(solver="BatchILQR", nb_iter=40, early_stop=True, **t[2])])

t = traces_of("POS_ORN_TIME_SYS_2ND.ipynb")
cases["POS_ORN_TIME_SYS_2ND"] = dict(
    problem=problem("POS_ORN_TIME", 2, 50, None, [0.0] * 7, P + [1, 1, 1, 0, 0, 0, .1], P + [1, 1, 1, .1, .1, .1, .1],
                    ctimes=[2.5, 5], dq_limits=True, u0_last=0.01),
    # PSI = kron(sawtooth, diag(1..1,0)) + kron(unitstep, diag(0..0,1))  (cell 8)
    solves=[dict(solver="BatchILQRCP", psi=dict(kind="sawtooth+unitstep_dt", K=2), nb_iter=20, early_stop=True, **t[0]),
            dict(solver="ILQRRecursive", nb_iter=20, line_search=True, early_stop=True, **t[1]),
            dict(solver="BatchILQR", nb_iter=20, early_stop=True, **t[2])])

# ---- object-frame tutorials: TransformedSimulationInterface / SequentialSystem (literals of cells 8-16 of the two notebooks)
def frame_of(quat_wxyz, pos):
    """4x4 pose as the notebooks build it: scipy Rotation.from_quat(xyzw).as_matrix() (normalised quaternion) + position."""
    w, x, y, z = quat_wxyz
    n = (w * w + x * x + y * y + z * z) ** 0.5
    w, x, y, z = w / n, x / n, y / n, z / n
    R = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
    return [R[0] + [pos[0]], R[1] + [pos[1]], R[2] + [pos[2]], [0, 0, 0, 1]]


OBJ1 = frame_of([0.63758403393523, 0.2994657314658187, 0.6042309402208079, -0.37244039285286973], [0.62, 0.05, 0.34])
OBJ2 = frame_of([-0.03647984, 0.94060485, 0.33742794, 0.00860923], [0.32, 0.05, 0.54])


def frame_problem(T, dt, kps, n_sub, kind="POS_ORN", nb_deriv=1, u0_step=None):
    dof = 7
    nu = dof + (1 if kind == "POS_ORN_TIME" else 0)
    return dict(kind=kind, nb_deriv=nb_deriv, T=T, dt=dt, q0=Q0_TUT, dq0=[0] * dof, R_diag=[1e-5] * nu,
                qMax=[PI10] * dof, qMin=[-PI10] * dof, dqMax=[10.0] * dof, dqMin=[-10.0] * dof,
                keypoints=kps, u0_step=u0_step if u0_step is not None else [0.0] * nu, base="panda_link0", tip="panda_tip", lim_mult=n_sub)


t = traces_of("POS_ORN_SYS_OBJ_FRAME.ipynb")
cases["POS_ORN_SYS_OBJ_FRAME"] = dict(  # one PosOrnPlannerSys on TransformedSimulationInterface(rbt, obj1_frame)
    problem=frame_problem(400, 0.01, [
        dict(timestep=199, pos=[-0.30, 0.10, -0.15], orn=[1, 0, 0, 0], Qdiag=P, frame=OBJ1),
        dict(timestep=399, pos=[0, 0, -0.15], orn=[1, 0, 0, 0], Qdiag=P, frame=OBJ1)], 1),
    solves=[dict(solver="BatchILQRCP", psi=dict(kind="unitstep", K=2), nb_iter=25, early_stop=True, **t[0]),
            dict(solver="ILQRRecursive", nb_iter=10, line_search=True, early_stop=True, **t[1])])

t = traces_of("POS_ORN_MULTI_SYS.ipynb")
cases["POS_ORN_MULTI_SYS"] = dict(  # SequentialSystem(rbt, [sys1 in obj1_frame (keypoint at T/2), sys2 in obj2_frame (keypoint at T-1)])
    problem=frame_problem(600, 0.01, [
        dict(timestep=300, pos=[0, 0, -0.15], orn=[1, 0, 0, 0], Qdiag=[1, 1, 1, 0, 0, 0], frame=OBJ1, Ru=[1e-5] * 7),
        dict(timestep=599, pos=[0.1, 0.1, -0.1], orn=[1, 0, 0, 0], Qdiag=[1, 1, 1, 0, 0, 0], frame=OBJ2, Ru=[1e-5] * 7)], 2),
    solves=[dict(solver="BatchILQRCP", psi=dict(kind="unitstep", K=2), nb_iter=25, early_stop=True, **t[0]),
            dict(solver="ILQRRecursive", nb_iter=10, line_search=True, early_stop=True, **t[1])])

PV0 = [1, 1, 1, 0, 0, 0]
t = traces_of("POS_ORN_MULTI_SYS_2ND.ipynb")
cases["POS_ORN_MULTI_SYS_2ND"] = dict(  # the same two object-frame sub-systems, 2nd order (velocities in the object frames too; cells 12-18)
    problem=frame_problem(600, 0.01, [
        dict(timestep=300, pos=[0, 0, -0.15], orn=[1, 0, 0, 0], dpos=[0, 0, 0], dorn=[0, 0, 0, 0], Qdiag=PV0 + PV0, frame=OBJ1, Ru=[1e-5] * 7),
        dict(timestep=599, pos=[0.1, 0.1, -0.1], orn=[1, 0, 0, 0], dpos=[0, 0, 0], dorn=[0, 0, 0, 0], Qdiag=PV0 + PV0, frame=OBJ2, Ru=[1e-5] * 7)],
        2, nb_deriv=2),
    solves=[dict(solver="BatchILQRCP", psi=dict(kind="sawtooth", K=2), nb_iter=25, early_stop=True, **t[0]),
            dict(solver="ILQRRecursive", nb_iter=10, line_search=True, early_stop=True, **t[1])])

t = traces_of("POS_ORN_MULTI_SYS_TIME.ipynb")
cases["POS_ORN_MULTI_SYS_TIME"] = dict(  # two PosOrnTimePlannerSys in object frames, SpacetimeKeypoints at 2.5 s / 5 s, u0 = 0.1 everywhere (cell 18)
    problem=frame_problem(600, None, [
        dict(timestep=300, pos=[0, 0, -0.15], orn=[1, 0, 0, 0], Qdiag=PV0 + [.1], ctime=2.5, frame=OBJ1, Ru=[1e-5] * 8),
        dict(timestep=599, pos=[0.1, 0.1, -0.1], orn=[1, 0, 0, 0], Qdiag=PV0 + [.1], ctime=5, frame=OBJ2, Ru=[1e-5] * 8)],
        2, kind="POS_ORN_TIME", u0_step=[0.1] * 8),
    solves=[dict(solver="BatchILQRCP", psi=dict(kind="unitstep", K=2), nb_iter=25, early_stop=True, **t[0]),
            dict(solver="ILQRRecursive", nb_iter=10, line_search=True, early_stop=True, **t[1])])

# FK literals stored in the notebooks (POS_ORN_MULTI_SYS.ipynb cell 8: pose of the tutorial q0, incl. negative w)
nb = json.load(open(os.path.join(REF, "POS_ORN_MULTI_SYS.ipynb")))
src = "".join("".join(c["source"]) for c in nb["cells"] if c["cell_type"] == "code")
m = re.search(r"obj2_rot_base_quat\s*=\s*[^\[\n]*\[([^\]]+)\]", src)
m2 = re.search(r"obj2_pos_base\s*=\s*[^\[\n]*\[([^\]]+)\]", src)
kat = dict(q0=Q0_TUT, quat_from_notebook=[float(v) for v in m.group(1).split(",")],
           pos_from_notebook=[float(v) for v in m2.group(1).split(",")] if m2 else None)

json.dump(dict(cases=cases, fk_kat=kat), open(os.path.join(OUT, "traces.json"), "w"), indent=1)

# ---- kinematic skeleton of the URDF
root = ET.parse(os.path.join(REF, "model.urdf")).getroot()
lines = ['<?xml version="1.0"?>', "<!-- kinematic skeleton (joints/links only) of the Panda model used by the reference tutorials -->",
         '<robot name="%s">' % root.get("name")]
for l in root.findall("link"):
    lines.append('  <link name="%s"/>' % l.get("name"))
for j in root.findall("joint"):
    lines.append('  <joint name="%s" type="%s">' % (j.get("name"), j.get("type")))
    o = j.find("origin")
    if o is not None:
        lines.append('    <origin rpy="%s" xyz="%s"/>' % (o.get("rpy", "0 0 0"), o.get("xyz", "0 0 0")))
    lines.append('    <parent link="%s"/>' % j.find("parent").get("link"))
    lines.append('    <child link="%s"/>' % j.find("child").get("link"))
    a = j.find("axis")
    if a is not None:
        lines.append('    <axis xyz="%s"/>' % a.get("xyz"))
    lim = j.find("limit")
    if lim is not None:
        lines.append("    <limit %s/>" % " ".join('%s="%s"' % kv for kv in lim.attrib.items()))
    lines.append("  </joint>")
lines.append("</robot>")
open(os.path.join(OUT, "panda_chain.urdf"), "w").write("\n".join(lines) + "\n")
print("wrote traces.json (%d cases) and panda_chain.urdf" % len(cases))
# the same skeleton ships with the package so bench.py / smoke() need nothing outside the repo
import shutil
shutil.copyfile(os.path.join(OUT, "panda_chain.urdf"), os.path.join(os.path.dirname(os.path.dirname(OUT)), "ilqr_planner_amd", "data", "panda_chain.urdf"))
