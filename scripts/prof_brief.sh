#!/bin/bash
# usage: scripts/prof_brief.sh TAG   -- rocprofv3 kernel stats of a short bench run, digest printed and kept under gpurun_out/
set -e
TAG=${1:-p}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-split $PROF_ARGS > "$OUT.log" 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_profile.py gpurun_out/prof_$TAG.txt "$OUT" | head -14
