#!/bin/bash
# rocprofv3 kernel stats of scripts/time_batchwide.py (wide-basis batch solvers)
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_wide
rm -rf "$OUT"
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python3 scripts/time_batchwide.py > "$OUT.log" 2>&1
python3 scripts/summarize_profile.py gpurun_out/prof_wide.txt "$OUT" | head -24
cat "$OUT.log"
