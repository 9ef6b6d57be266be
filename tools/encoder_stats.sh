#!/bin/bash
# per-kernel average duration of the encoder kernels on 1024 images (rocprofv3 --stats of tools/encoder_probe.py)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ep && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ep -- python3 $GRAFT_REPO_ROOT/tools/encoder_probe.py "$@" 2>/dev/null | grep encode
python3 -c "
import csv,glob
for r in csv.DictReader(open(glob.glob('/tmp/ep/*/*kernel_stats.csv')[0])):
    if 'enc' in r['Name']: print(r['Name'][:44], r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,2), 'min', round(float(r['MinNs'])/1e3,2))
"
