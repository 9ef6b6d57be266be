#!/bin/bash
# rocprofv3 passes whose summaries go to profiles/ (tools/profile_summary.py): kernel trace + stats of the EXACT driver
# command (python3 bench.py --gpus 1 --steps 20 --warmup 5), HBM traffic counters (separate --pmc passes, kernel-trace
# only), and the counter calibration copy.  Usage: profile_run.sh [level]   (two_agent | four_agent)
LEVEL=${1:-two_agent}
ROUND=${ROUND:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${ROUND}_$LEVEL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/calib_copy $GRAFT_REPO_ROOT/tools/calib_copy.hip || exit 1
ARGS="--gpus 1 --steps 20 --warmup 5"
[ "$LEVEL" != two_agent ] && ARGS="$ARGS --level $LEVEL"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/stats.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS --no-cpu-baseline > $OUT/pmc_$c.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/calib_$c -- /tmp/calib_copy > $OUT/calib_$c.log 2>&1 || exit 1
done
echo profile_run $LEVEL done
