#!/bin/bash
# Config 2 (1024 copies: one wave per SIMD, the few-copies build) under other scheduling strategies (objects prebuilt).
run() { echo "== flags: [$1]"; MJRL_SPEC_FLAGS="$1" python bench.py --envs-per-gpu 1024 --steps 400 --warmup 30 --no-cpu-baseline --no-extra-configs 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(round(d['value'] / 1e6, 3), 'M env-steps/s, kernel', d.get('roofline', {}).get('kernel_ms'), 'ms')
"; }
run ""
run "-mllvm -amdgpu-sched-strategy=max-ilp"
run "-mllvm -amdgpu-schedule-metric-bias=0"
run ""
