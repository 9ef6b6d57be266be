"""Device-resident step rate of any shipped level (random ctrl every step, steps 300..400 of an episode).
Usage: level_rate.py [level.xml] [n_env]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi
name = sys.argv[1] if len(sys.argv) > 1 else "four_agent.xml"
n_env = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
m = mjcf.compile_mjcf(levels.level_path(name))
h = _capi.Handle(blob.pack(m), n_env)
h.reset()
rng = np.random.default_rng(0)
for t in range(400):
    if t % 5 == 0 or t >= 300:
        h.set_field("ctrl", rng.uniform(-1, 1, (n_env, max(m.nu, 1)))[:, :m.nu]) if m.nu else None
    if t == 300:
        h.sync(); t0 = time.perf_counter()
    h.step_device(None, 0, 1)
h.sync()
dt = (time.perf_counter() - t0) / 100
lds = h.size("lds_doubles") * 8 / 1024
print(f"{name}: {n_env} copies, LDS {lds:.1f} KiB per copy ({int(160 // lds)} per CU), kernel {h.kernel}: {dt * 1e6:.1f} us/step, "
      f"{n_env / dt / 1e6:.2f} M env-steps/s; contacts per copy {h.query('ncon').mean():.2f}, warnings {int(h.query('warn').max())}")
