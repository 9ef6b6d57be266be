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
start = int(sys.argv[1]) if len(sys.argv) > 1 else 50       # first timed step of the episode (50: free fall, 400: contacts)
for n_env in [64, 256, 512, 1024, 1536, 2048, 3072, 4096, 6144, 8192]:
    h = _capi.Handle(packed, n_env)
    h.reset()
    rng = np.random.default_rng(0)
    # same physical regime for every size: 100 steps of an episode from step `start`
    for t in range(start + 100):
        if t % 10 == 0 or t >= start - 20:
            h.set_field("ctrl", rng.uniform(-1, 1, (n_env, m.nu)))
        if t == start:
            h.sync(); t0 = time.perf_counter()
        h.step_device(None, 0, 1)
    h.sync()
    dt = (time.perf_counter() - t0) / 100
    print(f"n_env {n_env:5d}: {dt * 1e6:8.1f} us/step   {n_env / dt / 1e6:6.2f} M env-steps/s   waves per CU {n_env / 256:.2f}", flush=True)
    h.close()
