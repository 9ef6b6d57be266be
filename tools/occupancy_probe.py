"""Does the step kernel speed up with more waves per CU?  Same workload, smaller row caps -> smaller LDS image."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi
n_env = 4096
for nconmax, njmax in [(16, 80), (6, 40), (2, 24)]:
    m = mjcf.compile_mjcf(levels.level_path("two_agent.xml"), nconmax=nconmax, njmax=njmax)
    h = _capi.Handle(blob.pack(m), n_env)
    h.reset()
    rng = np.random.default_rng(0)
    for t in range(150):                       # the free-fall phase: no contacts, few limit rows, caps never bind
        if t % 10 == 0:
            h.set_field("ctrl", rng.uniform(-1, 1, (n_env, m.nu)))
        if t == 50:
            h.sync(); t0 = time.perf_counter()
        h.step_device(None, 0, 1)
    h.sync()
    dt = (time.perf_counter() - t0) / 100
    lds = h.size("lds_doubles") * 8 / 1024
    print(f"nconmax {nconmax:2d} njmax {njmax:2d}: LDS {lds:5.1f} KiB -> {int(160 // lds)} waves/CU, {dt * 1e3:.3f} ms/step, warn {h.query('warn').max()}")
    h.close()

print("per-wave cycles (stamps) at each cap:")
for nconmax, njmax in [(16, 80), (6, 40), (2, 24)]:
    m = mjcf.compile_mjcf(levels.level_path("two_agent.xml"), nconmax=nconmax, njmax=njmax)
    h = _capi.Handle(blob.pack(m), n_env)
    h.reset()
    for t in range(60):
        h.step_device(None, 0, 1)
    h.sync()
    tot = 0
    for _ in range(5):
        tot += sum(h.step_profile().values())
    print(f"  njmax {njmax}: {tot / 5 / n_env:.0f} cycles per wave-step")
    h.close()
