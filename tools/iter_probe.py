import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi
for iters in (100, 30, 10):
    m = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    m.iterations = iters
    packed = blob.pack(m)
    for n_env in (256, 1536, 4096, 8192):
        h = _capi.Handle(packed, n_env)
        h.reset()
        rng = np.random.default_rng(0)
        for t in range(500):
            if t >= 380:
                h.set_field("ctrl", rng.uniform(-1, 1, (n_env, m.nu)))
            elif t % 10 == 0:
                h.set_field("ctrl", rng.uniform(-1, 1, (n_env, m.nu)))
            if t == 400:
                h.sync(); t0 = time.perf_counter()
            h.step_device(None, 0, 1)
        h.sync()
        dt = (time.perf_counter() - t0) / 100
        print(f"iterations {iters:3d} n_env {n_env:5d}: {dt*1e6:7.1f} us/step  {n_env/dt/1e6:6.2f} M/s", flush=True)
        h.close()
