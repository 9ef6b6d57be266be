#!/bin/bash
# Does the driver's 20-step window depend on how long the GPU has been busy before it?  (clock ramp)
for p in 1024 8192 1024 8192 1024 8192; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --preroll $p --no-extra-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('preroll', $p, round(d['value']/1e6,3), 'M', d['roofline']['kernel_ms'])"
done
