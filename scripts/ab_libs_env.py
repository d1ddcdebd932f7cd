"""As scripts/ab_libs.py, with per-category averages of scripts/ab_env.py: `CFG=C2 B=256 CATS=1,2,4 python scripts/ab_libs_env.py lib1.so lib2.so ...`
runs scripts/ab_env.py once per library build (own process, capi.LIB_PATH set first).  Experiment tooling."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys, runpy; sys.path.insert(0, %r); import ilqr_planner_amd.capi as c; c.LIB_PATH = sys.argv[1]; "
        "runpy.run_path(%r, run_name='__main__')") % (ROOT, os.path.join(ROOT, "scripts", "ab_env.py"))
for lib in sys.argv[1:]:
    out = subprocess.run([sys.executable, "-c", code, os.path.abspath(lib)], capture_output=True, text=True)
    print(os.path.basename(lib), out.stdout.strip() or out.stderr[-800:], flush=True)
