#!/bin/bash
# A/B of the committed base tree (tools/ab/base) and the working tree on the DRIVER's command (20 timed steps, default
# flags except the extra configs), alternating, on one box.
root="$(cd "$(dirname "$0")/.." && pwd)"
for r in 1 2 3 4; do
  for which in base new; do
    if [ $which = base ]; then dir=$root/tools/ab/base; else dir=$root; fi
    (cd $dir && python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-configs "$@" 2>/dev/null) |
      python -c "import json,sys; l=json.loads(sys.stdin.readline()); print('$which', round(l['value']/1e6,3), 'M env-steps/s', round(l['ms_per_step'],4), 'ms  kernel', round(l['roofline']['kernel_ms'],4))"
  done
done
