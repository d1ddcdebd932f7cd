#!/bin/bash
# usage: scripts/pmc_ab.sh "COUNTER1 COUNTER2 ..." lib1.so lib2.so ...  -- one rocprofv3 --pmc pass (counters only) of a short C3 bench run
# per library build (capi.LIB_PATH pointed at it inside the python process rocprofv3 starts: no launcher in between); per-kernel averages.
set -e
CNT=$1; shift
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  OUT=$ROOT/gpurun_out/pmcab_$tag
  rm -rf "$OUT"
  rocprofv3 --pmc $CNT --output-format csv -d "$OUT" -o run -- python3 -c "import sys, runpy; sys.path.insert(0, '$ROOT'); import ilqr_planner_amd.capi as c; c.LIB_PATH = '$ROOT/$lib'; sys.argv = ['bench.py', '--steps', '1', '--warmup', '1', '--no-cpu-baseline', '--no-split']; runpy.run_path('$ROOT/bench.py', run_name='__main__')" > "$OUT.log" 2>&1
  echo "== $tag"
  python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"][:44]][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in sorted(agg.items()):
    if not k.startswith("void ilqr::k_"): continue
    print(f"{k:46s}", " ".join(f"{c}={v[1]/v[0]:.4g}" for c, v in sorted(cs.items())), f"n={list(cs.values())[0][0]}")
PY
done
