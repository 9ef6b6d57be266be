"""Per-stage cycle shares of the step kernel in the bench regime (a new random action every step), after `settle`
steps.  Use a small batch (64 or 256 copies): every stamp is an atomic add on one address, and in a full launch the
waves queue on it, which inflates the stages that wait on global loads (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
os.environ["MJRL_SPEC_FLAGS"] = (os.environ.get("MJRL_SPEC_FLAGS", "") + " -DMJRL_DIAG").strip()   # the diagnostic build of the specialised kernel
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi
name = sys.argv[1] if len(sys.argv) > 1 else "two_agent.xml"
n_env = int(sys.argv[2]) if len(sys.argv) > 2 else 256
settle = int(sys.argv[3]) if len(sys.argv) > 3 else 400
m = mjcf.compile_mjcf(levels.level_path(name))
h = _capi.Handle(blob.pack(m), n_env)
h.reset()
rng = np.random.default_rng(0)
for t in range(settle):
    h.set_field("ctrl", rng.uniform(-1, 1, (n_env, m.nu)))
    h.step_device(None, 0, 1)
h.sync()
tot = {}
for _ in range(5):
    h.set_field("ctrl", rng.uniform(-1, 1, (n_env, m.nu)))
    prof = h.step_profile()
    for k, v in prof.items():
        tot[k] = tot.get(k, 0) + v
s = sum(tot.values())
print("lds doubles", h.size("lds_doubles"))
for k, v in tot.items():
    print(f"{k:10s} {v / 5 / n_env:12.0f} cycles/env-step  {100 * v / s:5.1f} %")
print(f"total {s / 5 / n_env:.0f} cycles per env-step (wave clock, 100 MHz ticks if s_memtime is the constant clock)")
print("ncon mean", h.query("ncon").mean())
