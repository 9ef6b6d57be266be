#!/bin/bash
# A/B of the committed tree (tools/ab/base, built by tools/ab_base.sh) and the working tree on the same box:
# tools/ab_bench.sh <level> [reps] [steps] [extra bench.py flags].  Alternates the two trees' bench.py.
level=${1:-two_agent}; reps=${2:-3}; steps=${3:-400}; shift 3
root="$(cd "$(dirname "$0")/.." && pwd)"
for r in $(seq $reps); do
  for which in base new; do
    if [ $which = base ]; then dir=$root/tools/ab/base; else dir=$root; fi
    (cd $dir && python bench.py --level $level --steps $steps --warmup 20 --no-cpu-baseline --no-extra-configs "$@" 2>/dev/null) |
      python -c "import json,sys; l=json.loads(sys.stdin.readline()); print('$which', '$level', round(l['value']/1e6,3), 'M env-steps/s', round(l['ms_per_step'],4), 'ms  kernel', round(l['roofline']['kernel_ms'],4))"
  done
done
