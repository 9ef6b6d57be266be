"""What a split of the step at the solver's door would have to beat, MEASURED: a full batch in the bench's regime (4096
copies of the 2-agent level, settled under random controls) launched cut off after the row build ("rows": everything in
front of the solver), after the solver ("pgs") and in full, the diagnostic build of the specialised kernel
(mjrl_step_truncated writes nothing back, so every launch of a cut sees the same state).  Wall time per launch over REPS
back-to-back launches.  A two-kernel step costs at least  t(rows) + t(solver on its own) + the rows' round trip through
HBM;  t(solver on its own) is bounded below by the longest solve of the batch.  Usage: split_probe.py [level] [n_env]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MJRL_SPEC_FLAGS"] = (os.environ.get("MJRL_SPEC_FLAGS", "") + " -DMJRL_DIAG -DMJRL_STAGE_CUT").strip()   # the diagnostic build
import numpy as np
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi
name = sys.argv[1] if len(sys.argv) > 1 else "two_agent.xml"
n_env = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
REPS = 40
m = mjcf.compile_mjcf(levels.level_path(name))
h = _capi.Handle(blob.pack(m), n_env)
h.reset()
rng = np.random.default_rng(0)
for t in range(600):
    if t % 20 == 0:
        h.set_field("ctrl", rng.uniform(-1, 1, (n_env, m.nu)))
    h.step_device(None, 0, 1)
h.sync()
stats = h.get_field("solver_stats")
print(f"{name} {n_env} copies after 600 steps: rows mean {stats[:, 1].mean():.1f} max {stats[:, 1].max()}, sweeps mean {stats[:, 2].mean():.1f} "
      f"p99 {np.percentile(stats[:, 2], 99):.0f} max {stats[:, 2].max()}")
print("kernel attached:", h.kernel, "| spec flags:", os.environ["MJRL_SPEC_FLAGS"])
times = {}
for cut in ("collide", "rows", "pgs", "sensors", "euler"):
    for _ in range(5):
        h.step_truncated(cut)
    h.sync()
    t0 = time.perf_counter()
    for _ in range(REPS):
        h.step_truncated(cut)
    h.sync()
    times[cut] = (time.perf_counter() - t0) / REPS * 1e6
    print(f"cut after {cut:8s}: {times[cut]:7.1f} us per launch")
print(f"solver stage inside the one kernel: {times['pgs'] - times['rows']:.1f} us of the launch; everything in front of it {times['rows']:.1f} us; "
      f"everything behind it {times['euler'] - times['pgs']:.1f} us")
t0 = time.perf_counter()
for _ in range(REPS):
    h.step_device(None, 0, 1)
h.sync()
print(f"whole steps through the production entry (same handle): {(time.perf_counter() - t0) / REPS * 1e6:.1f} us per launch")
# the solver's own lower bound as a kernel: the longest solve (sweeps x rows) -- from the wave timeline of one full step
tl = h.step_timeline()
dur = (tl[:, 1] - tl[:, 0]).astype(float) / 100.0
print(f"full diagnostic step: waves mean {dur.mean():.1f} us p99 {np.percentile(dur, 99):.1f} max {dur.max():.1f}; first start to last end "
      f"{(tl[:, 1].max() - tl[:, 0].min()) / 100.0:.1f} us")
