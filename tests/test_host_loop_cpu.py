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
    assert r.stdout.strip().startswith("ok")
