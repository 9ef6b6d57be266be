#!/bin/bash
# A/B of two specialised step kernels on the same box: tools/ab_bench.sh <level> <base.hsaco> [reps] [steps]
# (MJRL_SPEC_OBJECT makes kernel_cache hand out a saved code object of the same model shape.)  Alternates the two builds.
level=${1:-two_agent}; base=$2; reps=${3:-3}; steps=${4:-400}
for r in $(seq $reps); do
  for which in base new; do
    if [ $which = base ]; then export MJRL_SPEC_OBJECT=$base; else unset MJRL_SPEC_OBJECT; fi
    python bench.py --level $level --steps $steps --warmup 20 --no-cpu-baseline --no-extra-configs 2>/dev/null |
      python -c "import json,sys; l=json.loads(sys.stdin.readline()); print('$which', '$level', round(l['value']/1e6,3), 'M env-steps/s', round(l['ms_per_step'],4), 'ms  kernel', round(l['roofline']['kernel_ms'],4))"
  done
done
