"""Kernel time of one step against the number of env copies (no stamps): does a wave slow down when its CU fills up?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi
m = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
packed = blob.pack(m)
for n_env in [64, 256, 512, 1024, 1280, 2048, 4096, 8192]:
    h = _capi.Handle(packed, n_env)
    h.reset()
    rng = np.random.default_rng(0)
    # same physical regime for every size: the first 150 steps of an episode, timed over steps 50..150
    for t in range(150):
        if t % 10 == 0:
            h.set_field("ctrl", rng.uniform(-1, 1, (n_env, m.nu)))
        if t == 50:
            h.sync(); t0 = time.perf_counter()
        h.step_device(None, 0, 1)
    h.sync()
    dt = (time.perf_counter() - t0) / 100
    print(f"n_env {n_env:5d}: {dt * 1e6:8.1f} us/step   {n_env / dt / 1e6:6.2f} M env-steps/s   waves per CU {n_env / 256:.2f}", flush=True)
    h.close()
