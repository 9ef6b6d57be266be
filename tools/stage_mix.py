"""Instruction mix of every stage of the step kernel from hardware counters: after `settle` steps of the bench regime
the batch is launched REPS times cut off after each stage in turn (mjrl_step_truncated: nothing is written back, so every
launch sees the same state) and once in full.  Run under
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d DIR -- python3 tools/stage_mix.py
and summarise with  tools/stage_mix.py --summary DIR  (per-wave counts of each stage = difference of successive cuts)."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CUTS = ["load", "kin", "com", "crb", "factor", "geom", "collide", "vel", "smooth", "rows", "pgs", "sensors", "euler", "store"]
REPS = 3
N_ENV = 1024

if len(sys.argv) > 2 and sys.argv[1] == "--summary":
    rows = {}
    for path in glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "mjrl_step_kernel" in r.get("Kernel_Name", ""):
                rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)
    tail = ids[-REPS * (len(CUTS) + 1):]                  # the diagnostic launches are the last ones of the run
    names = sorted(rows[tail[0]])
    prev = {n: 0.0 for n in names}
    print(f"{'stage':10s}" + "".join(f"{n.replace('SQ_', ''):>16s}" for n in names) + "   (per wave)")
    for k, cut in enumerate(CUTS + ["tail(full)"]):
        group = tail[REPS * k: REPS * (k + 1)]
        mean = {n: sum(rows[i][n] for i in group) / len(group) / N_ENV for n in names}
        print(f"{cut:10s}" + "".join(f"{mean[n] - prev[n]:16.1f}" for n in names))
        prev = mean
    print(f"{'total':10s}" + "".join(f"{prev[n]:16.1f}" for n in names))
    sys.exit(0)

os.environ["MJRL_SPEC_FLAGS"] = (os.environ.get("MJRL_SPEC_FLAGS", "") + " -DMJRL_DIAG -DMJRL_STAGE_CUT").strip()   # the diagnostic build
import numpy as np
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi
name = sys.argv[1] if len(sys.argv) > 1 else "two_agent.xml"
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 400
m = mjcf.compile_mjcf(levels.level_path(name))
h = _capi.Handle(blob.pack(m), N_ENV)
h.reset()
rng = np.random.default_rng(0)
for t in range(settle):
    h.set_field("ctrl", rng.uniform(-1, 1, (N_ENV, m.nu)))
    h.step_device(None, 0, 1)
h.set_field("ctrl", rng.uniform(-1, 1, (N_ENV, m.nu)))
h.sync()
for cut in CUTS:
    for _ in range(REPS):
        h.step_truncated(cut)
    h.sync()
for _ in range(REPS):                 # the whole step (these launches do move the state on)
    h.step_device(None, 0, 1)
h.sync()
