#!/bin/bash
# usage: scripts/pmc_brief.sh TAG "COUNTER1 COUNTER2 ..."   -- one rocprofv3 --pmc pass (counters only, no tracing) of a short bench run
set -e
TAG=${1:-p}; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf "$OUT"
rocprofv3 --pmc $1 --output-format csv -d "$OUT" -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-split $PROF_ARGS > "$OUT.log" 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"][:48]][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    if not k.startswith("void ilqr::k_") : continue
    print(k, " ".join(f"{c}={v[1]/v[0]:.4g}" for c, v in sorted(cs.items())), f"n={list(cs.values())[0][0]}")
PY
