"""a user-defined C++ System subclass is solved by solver::ILQRRecursive over its virtuals (csrc/host/ilqr_host_loop.cpp).
The reference exposes no Python trampolines, so the case is a C++ program (tests/cpp/user_system_main.cpp) built against the host mirror."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_user_defined_system_runs_the_host_loop(tmp_path):
    lib_dir = os.path.join(ROOT, "ilqr_planner_amd")
    assert os.path.exists(os.path.join(lib_dir, "libilqr_hip.so")), "build the library first (__graft_entry__.build())"
    exe = str(tmp_path / "user_system")
    host = os.path.join(lib_dir, "csrc", "host")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "user_system_main.cpp"), os.path.join(host, "ilqr_host.cpp"),
                           os.path.join(host, "ilqr_host_loop.cpp"), "-o", exe, "-L" + lib_dir, "-lilqr_hip", "-Wl,-rpath," + lib_dir])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().splitlines()[-1].startswith("ok")


import pytest


@pytest.mark.gpu
def test_user_subclass_of_kdlrobot_is_not_lowered(tmp_path):
    """A user subclass of sim::KDLRobot with its own kinematics must be solved over its virtuals, not with the base chain on the device
    (System::builtin() requires the exact simulator types of the mirror); the plain KDLRobot system goes to the GPU."""
    lib_dir = os.path.join(ROOT, "ilqr_planner_amd")
    exe = str(tmp_path / "user_robot")
    host = os.path.join(lib_dir, "csrc", "host")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "user_robot_main.cpp"), os.path.join(host, "ilqr_host.cpp"),
                           os.path.join(host, "ilqr_host_loop.cpp"), "-o", exe, "-L" + lib_dir, "-lilqr_hip", "-Wl,-rpath," + lib_dir])
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "panda_chain.urdf")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().splitlines()[-1].startswith("ok")
