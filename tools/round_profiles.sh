#!/bin/bash
# Everything the round's profiles/ summaries are made from, in one gpurun call (run from the repo root on the GPU box):
# rocprofv3 stats + HBM counters of the exact driver command (two_agent, four_agent), per-stage instruction mix, wave
# timelines, stage cycles, the PMC passes of the step and render kernels.  Summaries: tools/profile_summary.py etc.
export ROUND=${ROUND:-r04}
O=gpurun_out
# (raw traces are summarised here and deleted: gpurun copies back at most 64 MiB)
export PROFILE_OUT=$PWD/$O/profiles_$ROUND
mkdir -p $PROFILE_OUT
for lv in two_agent four_agent; do
  bash tools/profile_run.sh $lv > $O/${ROUND}_profile_run_$lv.log 2>&1
  python tools/profile_summary.py $lv 20 > $O/${ROUND}_profile_summary_$lv.log 2>&1
  grep "^{" $O/prof_${ROUND}_$lv/stats.log > $PROFILE_OUT/${ROUND}_bench_driver_command_$lv.log
  rm -rf $O/prof_${ROUND}_$lv
done
python tools/timeline_probe.py 4096 two_agent.xml > $O/${ROUND}_wave_timeline.txt 2>&1
python tools/timeline_probe.py 4096 four_agent.xml > $O/${ROUND}_wave_timeline_four_agent.txt 2>&1
python tools/stage_profile.py two_agent.xml > $O/${ROUND}_stage_cycles.txt 2>&1
python tools/stage_profile.py four_agent.xml >> $O/${ROUND}_stage_cycles.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && for lv in two_agent four_agent; do rm -rf $GRAFT_REPO_ROOT/$O/stage_mix_$lv;
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $GRAFT_REPO_ROOT/$O/stage_mix_$lv -- python3 $GRAFT_REPO_ROOT/tools/stage_mix.py $lv.xml > $GRAFT_REPO_ROOT/$O/stage_mix_$lv.log 2>&1; done )
for lv in two_agent four_agent; do echo "# $lv.xml"; python tools/stage_mix.py --summary $O/stage_mix_$lv; done > $O/${ROUND}_stage_instruction_mix.txt 2>&1
rm -rf $O/stage_mix_two_agent $O/stage_mix_four_agent
rm -rf $O/pmc; bash tools/pmc_run.sh > /dev/null 2>&1; python tools/pmc_summary.py $O/pmc mjrl_step_kernel_spec > $O/${ROUND}_pmc_instruction_mix.txt 2>&1
rm -rf $O/pmc
echo round_profiles done
