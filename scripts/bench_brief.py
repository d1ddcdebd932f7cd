"""Run bench.py with the given args and print a one-line digest (used for quick A/B experiments on the GPU box)."""
import json, subprocess, sys, os
out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")] + sys.argv[1:],
                     capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print("bench failed:", out.stderr[-2000:])
    sys.exit(1)
d = json.loads(line[-1])
r = d["roofline"]
ks = {r["kernel"]: r["avg_launch_ms"]}
ks.update({k: v["avg_launch_ms"] for k, v in r["other"].items()})
print(os.environ.get("TAG", ""), "ms/step", round(d["ms_per_step"], 2), "value", round(d["value"]), "kernels(ms)", ks, "frac", r["frac"],
      "relerr", d.get("final_cost_rel_err_vs_oracle", {}).get("median"))
