#!/bin/bash
# Code objects of the specialised step kernel as COMMITTED (HEAD), for an A/B against the working tree on the same box:
# tools/ab/base_<level>.hsaco (tools/ab_bench.sh <level> tools/ab/base_<level>.hsaco).  Only meaningful while the working
# tree has not changed StepArgs or the blob layout.
set -e
cd "$(dirname "$0")/.."
git stash -q
trap 'git stash pop -q' EXIT
mkdir -p tools/ab
python - <<'PY'
import shutil, sys
sys.path.insert(0, ".")
import __graft_entry__ as e
e.load_package()
from mjrl_amd import blob, kernel_cache, levels, mjcf
for name in ("two_agent.xml", "four_agent.xml"):
    p = kernel_cache.code_object(blob.pack(mjcf.compile_mjcf(levels.level_path(name))))
    shutil.copy(p, f"tools/ab/base_{name[:-4]}.hsaco")
    print(p)
PY
