#!/bin/bash
# A whole built copy of the tree as COMMITTED (HEAD by default, or the revision given) under tools/ab/base/, for an A/B
# against the working tree on the same box (tools/ab_bench.sh): its own libmjrl_hip.so, oracle and specialised step
# kernels of the two headline levels.  The working tree is left alone (no stash: `git archive` of the revision).
# tools/ab/ is git-ignored and travels to the GPU box with the snapshot.
set -e
cd "$(dirname "$0")/.."
rev=${1:-HEAD}
rm -rf tools/ab/base
mkdir -p tools/ab/base
git archive "$rev" | tar -x -C tools/ab/base
git rev-parse "$rev" > tools/ab/base/REVISION
make -C tools/ab/base/mujoco-rl-environment-wrapper_amd/csrc -s
make -C tools/ab/base/oracle -s
(cd tools/ab/base && python - <<'PY'
import sys
sys.path.insert(0, ".")
import __graft_entry__ as e
e.load_package()
from mjrl_amd import blob, kernel_cache, levels, mjcf
for name in ("two_agent.xml", "four_agent.xml"):
    for few in (False, True):
        print(kernel_cache.code_object(blob.pack(mjcf.compile_mjcf(levels.level_path(name))), few=few))
PY
)
