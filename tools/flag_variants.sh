#!/bin/bash
# The headline under other instruction-scheduling strategies of the specialised kernel (objects prebuilt with the same
# MJRL_SPEC_FLAGS, which is part of the object's name).  One box, one after the other, the default first and last.
run() { echo "== flags: [$1]"; MJRL_SPEC_FLAGS="$1" python bench.py --steps 400 --warmup 30 --no-cpu-baseline --no-extra-configs 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(round(d['value'] / 1e6, 3), 'M env-steps/s, kernel', d.get('roofline', {}).get('kernel_ms'), 'ms')
"; }
run ""
run "-mllvm -amdgpu-sched-strategy=max-ilp"
run "-mllvm -amdgpu-sched-strategy=max-memory-clause"
run "-mllvm -amdgpu-schedule-metric-bias=0"
run "-mllvm -amdgpu-use-amdgpu-trackers"
run "-mllvm -amdgpu-sched-strategy=iterative-minreg"
run ""
