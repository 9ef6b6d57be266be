#!/bin/bash
# The rest of a round's measurements in one gpurun call (run from the repo root on the GPU box): the GPU test suite, the
# driver's bench command, the vector-env adapter's rates, the render kernel's counters and the config-5 kernel stats.
export ROUND=${ROUND:-r04}
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/${ROUND}_gputest_final.log 2>&1; tail -3 $O/${ROUND}_gputest_final.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${ROUND}_bench_final.log 2>&1
python tools/vec_env_rate.py > $O/${ROUND}_vec_env_rate.txt 2>&1
rm -rf $O/pmc_render; bash tools/pmc_render.sh > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_render mjrl_render_kernel > $O/${ROUND}_pmc_render.txt 2>&1; rm -rf $O/pmc_render
( cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/$O/c5 &&
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/c5 -- python3 $GRAFT_REPO_ROOT/bench.py --level camera_latents --steps 200 --warmup 20 --no-cpu-baseline --no-extra-configs > $GRAFT_REPO_ROOT/$O/c5.log 2>&1 )
cp $(find $O/c5 -name "*kernel_stats.csv" | head -1) $O/${ROUND}_kernel_stats_config5_encoder.csv; rm -rf $O/c5
echo round_extras done
