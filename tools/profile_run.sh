#!/bin/bash
# rocprofv3 passes whose summaries go to profiles/: kernel stats of the default bench run, HBM traffic counters
# (separate --pmc passes), and the counter calibration copy.
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r01
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/calib_copy $GRAFT_REPO_ROOT/tools/calib_copy.hip || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-double-buffer > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-double-buffer --steps 256 > $OUT/pmc_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/calib_$c -- /tmp/calib_copy > $OUT/calib_$c.log 2>&1
done
echo profile_run done
