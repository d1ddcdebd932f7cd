#!/bin/bash
# usage: scripts/collect_profiles.sh TAG [CONFIG]  -- everything the judged numbers come from, in one GPU call:
#   the bench line (default run: headline timed with profiling off and, for the MFMA-sweep systems, the two-stream schedule),
#   rocprofv3 --kernel-trace --stats and two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --no-split`, i.e. one kernel
#   at a time on one stream, so that the per-kernel durations are the ones the bench line's roofline block is computed from.
# Writes gpurun_out/<TAG>_{bench.json,kernel_stats.csv,kernels.txt}; copy what should be judged into profiles/.
set -e
TAG=${1:-r02_c3}; CFG=${2:-C3}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
cd $ROOT
python3 bench.py --config $CFG > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 300 $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_pmcf $OUT/${TAG}_pmcw
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o run -- python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-split > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmcf -o run -- python3 $ROOT/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-split > $OUT/${TAG}_pmcf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmcw -o run -- python3 $ROOT/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-split > $OUT/${TAG}_pmcw.log 2>&1
cd $ROOT
cp $OUT/${TAG}_stats/run_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
python3 scripts/summarize_profile.py $OUT/${TAG}_kernels.txt $OUT/${TAG}_stats $OUT/${TAG}_pmcf $OUT/${TAG}_pmcw | head -12
