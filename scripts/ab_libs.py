"""A/B of library builds on the GPU box: `python scripts/ab_libs.py [--bench-args ...] -- lib1.so lib2.so ...` runs bench.py once per
library (its own process, `capi.LIB_PATH` pointed at the build before anything loads it) and prints one digest line each: solve time with
profiling off and the per-category kernel averages of the profiled pass.  Experiment tooling; nothing in the product reads a library path
from the environment."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
cut = args.index("--") if "--" in args else 0
bench_args, libs = (args[:cut], args[cut + 1:]) if "--" in args else ([], args)
bench_args = bench_args or ["--no-cpu-baseline", "--steps", "5", "--warmup", "1"]
code = ("import sys, runpy; sys.path.insert(0, %r); import ilqr_planner_amd.capi as c; c.LIB_PATH = sys.argv[1]; "
        "sys.argv = ['bench.py'] + sys.argv[2:]; runpy.run_path(%r, run_name='__main__')") % (ROOT, os.path.join(ROOT, "bench.py"))
for lib in libs:
    out = subprocess.run([sys.executable, "-c", code, os.path.abspath(lib)] + bench_args, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(os.path.basename(lib), "FAILED", out.stderr[-1500:], flush=True)
        continue
    d = json.loads(line[-1])
    r = d["roofline"]
    ks = {r["kernel"]: r["avg_launch_ms"]}
    ks.update({k: v["avg_launch_ms"] for k, v in r["other"].items()})
    print(f"{os.path.basename(lib):40s} ms/step {d['ms_per_step']:.3f}  kernels(ms) {ks}", flush=True)
