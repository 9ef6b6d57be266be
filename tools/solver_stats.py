"""Distribution of contacts / rows / PGS iterations over the batch in the bench regime (random action every step)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi
n_env = 4096
m = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
h = _capi.Handle(blob.pack(m), n_env)
h.reset()
rng = np.random.default_rng(0)
ioff = h.lds_offset("ints")
for t in range(1001):
    h.set_field("ctrl", rng.uniform(-1, 1, (n_env, m.nu)))
    if t % 200 == 0 and t > 0:
        img = h.step_debug(None, 0, 1, 0)
        ints = img[:, ioff:ioff + 4].copy().view(np.int32)
        ncon, nefc, niter = ints[:, 0], ints[:, 1], ints[:, 3]
        work = nefc * niter
        print(f"step {t}: ncon mean {ncon.mean():.2f} max {ncon.max()} | nefc mean {nefc.mean():.1f} max {nefc.max()} | niter mean {niter.mean():.1f} "
              f"p50 {np.percentile(niter,50):.0f} p90 {np.percentile(niter,90):.0f} p99 {np.percentile(niter,99):.0f} max {niter.max()} | "
              f"rows*iters mean {work.mean():.0f} p99 {np.percentile(work,99):.0f} max {work.max()}", flush=True)
    else:
        h.step_device(None, 0, 1)
